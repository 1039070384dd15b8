// dev microbenchmark (r02): what does one dwordx4 wave-load cost the vector L1 of a CU when
//   (a) only some lanes are active (exec-masked),
//   (b) lanes read duplicate addresses (coarse pyramid levels: 2 / 4 / 8 lanes per texel),
// and what does the same access cost from LDS (ds_read_b128) -- the numbers behind the "bytes to VGPRs" cost model
// of the photometric sampler (DESIGN s3).  L1-resident region, 12 waves per CU, independent loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE: 0 all lanes coalesced; 1 even lanes only; 2 lanes with (lane & 3) == 0; 3 first 32 lanes; 4 first 16 lanes;
//       5 all lanes, 2 lanes per slot; 6 all lanes, 4 per slot; 7 all lanes, 8 per slot
template <int MODE>
__global__ __launch_bounds__(256) void kl1(const float *__restrict__ src, float *out, int iters, int slots)
{
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  f32x4 acc = {0, 0, 0, 0};
  unsigned o = (unsigned)wave * 7919u;
  bool active = true;
  if (MODE == 1) active = (lane & 1) == 0;
  if (MODE == 2) active = (lane & 3) == 0;
  if (MODE == 3) active = lane < 32;
  if (MODE == 4) active = lane < 16;
  const int div = MODE == 5 ? 2 : (MODE == 6 ? 4 : (MODE == 7 ? 8 : 1));
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int u = 0; u < 8; ++u)
    {
      const unsigned slot = ((o & ~63u) + (unsigned)(lane / div) + u * 64u) & (unsigned)(slots - 1);
      if (active)
      {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (size_t)slot * 4);
        acc += v;
      }
    }
    o = __builtin_amdgcn_readfirstlane(o * 1664525u + 1013904223u);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// LDS: every wave reads f32x4 items of a 32 KB region; MODE 0 consecutive per lane, 1: 4 lanes per item, 2: random item
template <int MODE>
__global__ __launch_bounds__(256) void klds(float *out, int iters)
{
  __shared__ __attribute__((aligned(16))) float s[8192];
  for (int i = threadIdx.x; i < 8192; i += 256)
    s[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc = {0, 0, 0, 0};
  unsigned o = threadIdx.x * 2654435761u;
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int u = 0; u < 8; ++u)
    {
      unsigned item;
      if (MODE == 0) item = (unsigned)(lane + u * 64 + it) & 2047u;
      else if (MODE == 1) item = (unsigned)(lane / 4 + u * 16 + it) & 2047u;
      else item = (o + u * 977u) & 2047u;
      acc += *reinterpret_cast<const f32x4 *>(s + item * 4);
    }
    o = o * 1664525u + 1013904223u;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <class F>
static double time_ms(F launch)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(10);
  hipEventRecord(e0);
  launch(400);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main()
{
  const int region_kb = 16, slots = region_kb * 1024 / 16, bpc = 3, grid = 256 * bpc;
  float *src, *out;
  hipMalloc(&src, (size_t)slots * 16 + 64); hipMemset(src, 0, (size_t)slots * 16 + 64);
  hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
  const double loads_per_cu = (double)bpc * 4 * 400 * 8;
  const char *names[] = {"all 64 lanes", "even lanes (32)", "every 4th lane (16)", "first 32 lanes", "first 16 lanes",
                         "64 lanes, 2 per slot", "64 lanes, 4 per slot", "64 lanes, 8 per slot"};
#define RUN(M) { double ms = time_ms([&](int it) { kl1<M><<<grid, 256>>>(src, out, it, slots); }); \
  printf("L1  dwordx4 %-24s: %.1f cycles per wave-load per CU @2.4GHz\n", names[M], ms * 1e-3 * 2.4e9 / loads_per_cu); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
  const char *ln[] = {"consecutive items", "4 lanes per item", "random items"};
#define RUNL(M) { double ms = time_ms([&](int it) { klds<M><<<grid, 256>>>(out, it); }); \
  printf("LDS ds_read_b128 %-20s: %.1f cycles per wave-read per CU @2.4GHz\n", ln[M], ms * 1e-3 * 2.4e9 / loads_per_cu); }
  RUNL(0) RUNL(1) RUNL(2)
  return 0;
}
