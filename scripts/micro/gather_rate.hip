// dev microbenchmark: issue rate of per-lane gathers through the vector L1 (texture addresser) of one CU.
// Every wave runs a stream of independent loads from a small (cache-resident) region:
//   mode 0: dwordx4, random 16-byte slots per lane        (the tap loads of the photometric kernels)
//   mode 1: dwordx4, 64 consecutive slots per wave (1 KiB) (coalesced reference)
//   mode 2: dword,   random per lane
//   mode 3: dwordx4, lanes in pairs on adjacent slots (two horizontal taps of one pixel)
//   mode 4: dwordx2, random per lane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float *__restrict__ src, const int *__restrict__ offs, float *out, int iters,
                                         int slots)
{
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  f32x4 acc = {0, 0, 0, 0};
  unsigned o = (unsigned)offs[(wave * 64 + lane) & 0xffff];
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int u = 0; u < 8; ++u)
    {
      unsigned slot; // slots is a power of two
      if (MODE == 1)
        slot = ((o & ~63u) + lane + u * 64u) & (unsigned)(slots - 1);
      else if (MODE == 3)
        slot = (((o + u * 977u) & ~1u) & (unsigned)(slots - 1)) + (lane & 1);
      else
        slot = (o + u * 977u) & (unsigned)(slots - 1);
      if (MODE == 2)
        acc[0] += src[(size_t)slot * 4];
      else if (MODE == 4)
      {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(src + (size_t)slot * 4);
        acc[0] += v[0]; acc[1] += v[1];
      }
      else
      {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (size_t)slot * 4);
        acc += v;
      }
    }
    o = o * 1664525u + 1013904223u;
    if (MODE == 1)
      o = __builtin_amdgcn_readfirstlane(o);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE>
void run(int blocks_per_cu, int region_kb, const char *name)
{
  const int slots = region_kb * 1024 / 16;
  float *src, *out; int *offs;
  hipMalloc(&src, (size_t)slots * 16 + 64);
  hipMemset(src, 0, (size_t)slots * 16 + 64);
  hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
  std::vector<int> h(65536);
  for (auto &v : h) v = rand();
  hipMalloc(&offs, h.size() * 4);
  hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 400, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, 256>>>(src, offs, out, 10, slots);
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(src, offs, out, iters, slots);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double loads_per_cu = (double)blocks_per_cu * 4 * iters * 8;
  printf("%-22s region %6d KB  waves/CU %2d : %.3f ms  %.1f cycles per wave-load per CU @2.4GHz\n", name, region_kb,
         blocks_per_cu * 4, ms, ms * 1e-3 * 2.4e9 / loads_per_cu);
  hipFree(src); hipFree(out); hipFree(offs);
}
int main()
{
  for (int kb : {16, 512, 16384})
  {
    run<0>(3, kb, "x4 random");
    run<0>(6, kb, "x4 random");
    run<3>(3, kb, "x4 lane pairs");
    run<1>(3, kb, "x4 coalesced");
    run<4>(3, kb, "x2 random");
    run<2>(3, kb, "x1 random");
  }
  return 0;
}
