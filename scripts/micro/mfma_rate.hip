// dev microbenchmark: attainable rate of v_mfma_f32_16x16x4_f32 streams (NACC independent accumulators per wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int VALU>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0)
{
  f32x4 acc[NACC];
  for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = a0 * j;
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int t = 0; t < NACC; ++t)
    {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < VALU; ++j) v[j & 7] = v[j & 7] * a + b;
    }
  }
  float s = 0;
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int VALU>
void run(int blocks_per_cu, const char *name)
{
  float *out; hipMalloc(&out, 256 * 256 * 16 * 4 * sizeof(float));
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, VALU><<<grid, 256>>>(out, 10, 1.f, 2.f);
  hipEventRecord(e0);
  k<NACC, VALU><<<grid, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)grid * 4 * iters * NACC;
  printf("%s: waves/SIMD %d  NACC %d VALU/mfma %d : %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD @2.4GHz\n", name, blocks_per_cu, NACC,
         VALU, ms, mf * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024));
  hipFree(out);
}
int main()
{
  run<15, 0>(1, "pure"); run<15, 0>(2, "pure"); run<15, 0>(3, "pure");
  run<15, 1>(1, "valu1"); run<15, 2>(1, "valu2"); run<15, 2>(2, "valu2"); run<15, 4>(1, "valu4"); run<15, 4>(2, "valu4");
  run<4, 0>(1, "nacc4"); run<4, 0>(2, "nacc4");
  return 0;
}
