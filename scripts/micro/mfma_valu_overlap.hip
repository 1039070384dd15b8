// dev microbenchmark: do f32 MFMA (v_mfma_f32_16x16x4_f32) and VALU (v_fma_f32) overlap on gfx950?
//  (a) one SIMD, two waves: wave A issues only MFMAs, wave B only VALU  -> time vs each alone (max = overlap, sum = not)
//  (b) one wave: MFMAs with independent VALU between them               -> time vs the two streams alone
// 512-thread workgroups: waves 0-3 and 4-7 land on SIMDs 0-3 twice (two waves per SIMD), one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// role 0: every wave MFMA+VALU interleaved (nm MFMA then nv VALU per iteration); role 1: waves 0-3 MFMA only, 4-7 VALU only
__global__ __launch_bounds__(512) void k(float *out, int iters, int role, int nm, int nv, float a0, float b0)
{
  f32x4 acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = a0 * j;
  const int wave = threadIdx.x >> 6;
  const bool do_m = role == 0 || wave < 4, do_v = role == 0 || wave >= 4;
  for (int it = 0; it < iters; ++it)
  {
    if (do_m)
      for (int r = 0; r < nm; r += 8)
      {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
      }
    if (do_v)
      for (int r = 0; r < nv; r += 8)
      {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], a, b);
      }
  }
  float s = 0;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same wave, finely interleaved: 1 MFMA then VPM independent VALU, 8 accumulators / 8 VALU chains
template <int VPM>
__global__ __launch_bounds__(512) void kmix(float *out, int iters, float a0, float b0)
{
  f32x4 acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = a0 * j;
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int t = 0; t < 8; ++t)
    {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < VPM; ++j) v[(t + j) & 7] = __builtin_fmaf(v[(t + j) & 7], a, b);
    }
  }
  float s = 0;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float *g_out;
static float time_k(int role, int nm, int nv, int iters)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<256, 512>>>(g_out, 10, role, nm, nv, 1.f, 2.f);
  hipEventRecord(e0);
  k<<<256, 512>>>(g_out, iters, role, nm, nv, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
template <int VPM> static float time_mix(int iters)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kmix<VPM><<<256, 512>>>(g_out, 10, 1.f, 2.f);
  hipEventRecord(e0);
  kmix<VPM><<<256, 512>>>(g_out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main()
{
  hipMalloc(&g_out, 256 * 512 * sizeof(float));
  const int iters = 4000;
  const double clk = 2.4e9;
  // per iteration and wave: nm MFMA / nv VALU
  printf("two waves per SIMD (512-thread workgroup, 1 per CU), %d iterations; cycles per iteration @2.4 GHz\n", iters);
  for (int nv : {32, 64, 128})
  {
    const int nm = 8;
    const float tm = time_k(1, nm, 0, iters), tv = time_k(1, 0, nv, iters), tb = time_k(1, nm, nv, iters);
    printf("  split roles: wave A %d MFMA alone %.0f cyc | wave B %d VALU alone %.0f cyc | both waves together %.0f cyc  (sum %.0f, max %.0f)\n",
           nm, tm * 1e-3 * clk / iters, nv, tv * 1e-3 * clk / iters, tb * 1e-3 * clk / iters, (tm + tv) * 1e-3 * clk / iters,
           (tm > tv ? tm : tv) * 1e-3 * clk / iters);
  }
  for (int nv : {32, 64})
  {
    const int nm = 8;
    const float tm = time_k(0, nm, 0, iters), tv = time_k(0, 0, nv, iters), tb = time_k(0, nm, nv, iters);
    printf("  same wave, blocks (8 MFMA then %d VALU), 2 waves/SIMD: MFMA alone %.0f | VALU alone %.0f | both %.0f  (sum %.0f)\n", nv,
           tm * 1e-3 * clk / iters, tv * 1e-3 * clk / iters, tb * 1e-3 * clk / iters, (tm + tv) * 1e-3 * clk / iters);
  }
  printf("  same wave, fine interleave (1 MFMA + n VALU) x 8 per iteration, 2 waves/SIMD: n=0 %.0f | n=2 %.0f | n=4 %.0f | n=8 %.0f cycles per iteration\n",
         time_mix<0>(iters) * 1e-3 * clk / iters, time_mix<2>(iters) * 1e-3 * clk / iters, time_mix<4>(iters) * 1e-3 * clk / iters,
         time_mix<8>(iters) * 1e-3 * clk / iters);
  return 0;
}
