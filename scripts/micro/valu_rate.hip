// dev microbenchmark (r06): issue rate of the VALU forms the photometric linearize is made of, per SIMD, at 1 / 2 / 3 waves per SIMD:
//   v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32, v_mov_b32, ds_read_b128, and the tap step's mix (12 ds_read_b128 + 38 v_pk_fma_f32)
// 16 independent chains per wave (no dependent-issue stalls), s_memtime cycles per instruction per wave and per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_rate.hip -o scripts/micro/valu_rate && scripts/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int iters, float a0)
{
  __shared__ f32x4 lds[1024];
  f32x2 v[16];
  for (int j = 0; j < 16; ++j)
    v[j] = f32x2{a0 * j, a0 + j};
  f32x2 a = {a0, a0 + 1.f}, b = {a0 + 2.f, a0 + 3.f};
  f32x4 t[12];
  for (int j = 0; j < 12; ++j)
    t[j] = f32x4{0, 0, 0, 0};
  lds[threadIdx.x] = f32x4{a0, a0, a0, a0};
  lds[threadIdx.x + 256] = f32x4{a0, a0, a0, a0};
  __syncthreads();
  const unsigned addr = (unsigned)(uintptr_t)(lds + (threadIdx.x & 63)) ;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it)
  {
    if (OP == 0)
    {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j][0]) : "v"(a[0]), "v"(b[0]));
    }
    else if (OP == 1)
    {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(a), "v"(b));
    }
    else if (OP == 2)
    {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(a));
    }
    else if (OP == 3)
    {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(a));
    }
    else if (OP == 4)
    {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          asm volatile("v_mov_b32 %0, %1" : "+v"(v[j][0]) : "v"(a[0]));
    }
    else if (OP == 5)
    {
      // 64 ds_read_b128 per iteration, 12 in flight
#pragma unroll
      for (int r = 0; r < 64; ++r)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[r % 12]) : "v"(addr), "n"((r % 8) * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    else if (OP == 6)
    {
      // the tap step's mix, 64 VALU-port instructions + ~20 LDS reads per iteration: [8 reads, wait, 16 pk_fma, 4 reads, wait, 22 pk_fma] x ...
#pragma unroll
      for (int s = 0; s < 2; ++s)
      {
#pragma unroll
        for (int r = 0; r < 8; ++r)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[r]) : "v"(addr), "n"(r * 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#pragma unroll
        for (int j = 0; j < 16; ++j)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[j]) : "v"(a), "v"(f32x2{t[j % 8][0], t[j % 8][1]}));
#pragma unroll
        for (int r = 8; r < 12; ++r)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[r]) : "v"(addr), "n"((r - 8) * 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]));
#pragma unroll
        for (int j = 0; j < 16; ++j)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[j]) : "v"(a), "v"(f32x2{t[8 + j % 4][0], t[8 + j % 4][1]}));
      }
    }
    else if (OP == 7)
    {
      // v_fma_f32 with 2x the count of OP 1 (the same FLOPs as v_pk_fma_f32): 128 per iteration
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j][r & 1]) : "v"(a[0]), "v"(b[0]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int j = 0; j < 16; ++j)
    s += v[j][0] + v[j][1];
  for (int j = 0; j < 12; ++j)
    s += t[j][0];
  if ((threadIdx.x & 63) == 0)
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
  if (s == 12345.678f)
    out[1] = 1;
}

template <int OP>
void run(const char *name, int per_iter, unsigned long long *d, int iters)
{
  for (int wps = 1; wps <= 4; ++wps)
  {
    // wps workgroups of 256 threads per CU (launch_bounds 256, tiny LDS): grid = 256 CUs x wps; dispatch fills CUs round-robin
    const int grid = 256 * wps;
    hipMemset(d, 0, sizeof(unsigned long long) * grid * 8);
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 8);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * grid * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < grid * 4; ++i)
      sum += (double)h[i * 2];
    const double cyc_wave = sum / (grid * 4) / ((double)iters * per_iter);
    printf("%-28s waves/SIMD %d: %.2f cycles per instruction per wave, %.2f per SIMD (%.3f ms)\n", name, wps, cyc_wave, cyc_wave / wps, ms);
  }
}

int main()
{
  unsigned long long *d;
  hipMalloc(&d, sizeof(unsigned long long) * 256 * 4 * 8 * 2);
  const int iters = 2000;
  run<0>("v_fma_f32", 64, d, iters);
  run<7>("v_fma_f32 (2 lanes' worth)", 128, d, iters);
  run<1>("v_pk_fma_f32", 64, d, iters);
  run<2>("v_pk_mul_f32", 64, d, iters);
  run<3>("v_pk_add_f32", 64, d, iters);
  run<4>("v_mov_b32", 64, d, iters);
  run<5>("ds_read_b128 (12 in flight)", 64, d, iters);
  run<6>("tap mix: 24 ds_read + 64 pk_fma", 64, d, iters);
  return 0;
}
