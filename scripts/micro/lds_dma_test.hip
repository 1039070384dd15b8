// dev micro-test: semantics of the LDS-direct buffer loads the staged sampler of the photometric linearize relies on
// (photo_kernels.hip: dma16): LDS[m0 + 16 * lane] <- buffer[voff(lane) + soff], disabled lanes write nothing, the issuing
// wave's vmcnt covers the LDS write, large LDS offsets (> 32 KiB) work, 3 workgroups per CU.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/lds_dma_test.hip -o scripts/micro/lds_dma_test && scripts/micro/lds_dma_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, uint32_t lds_byte, uint32_t voff, uint32_t soff)
{
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_byte), "v"(voff), "s"(r), "s"(soff) : "memory");
}
__device__ __forceinline__ f32x4 lds_read16(uint32_t addr)
{
  typedef const f32x4 __attribute__((address_space(3))) *LdsF4;
  return *(LdsF4)(addr);
}
__global__ __launch_bounds__(256, 3) void k(const float *src, int n_texels, int *bad, float *out)
{
  __shared__ __attribute__((aligned(16))) float s_mem[4 * 64 * 40];
  __shared__ float s_pad[3072 + 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  s_pad[tid] = 1.f; // (keeps the second array allocated)
  // poison the wave's region
  for (int i = lane; i < 640 * 4; i += 64)
    s_mem[wave * 2560 + i] = -777.f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, n_texels * 16, 0x00020000);
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(s_mem + wave * 2560));
  // round 0: every lane gathers texel (blockIdx * 97 + lane * 13 + wave * 5) % n, round 1: lanes < 36 only -> slots 64..99
  const int cnt = 100;
  const uint32_t t0 = (uint32_t)((blockIdx.x * 97 + lane * 13 + wave * 5) % n_texels);
  const uint32_t t1 = (uint32_t)((blockIdx.x * 31 + (lane + 64) * 7 + wave * 3) % n_texels);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  dma16(r, lds0, t0 * 16u, 0u);
  if (lane + 64 < cnt)
    dma16(r, lds0 + 1024u, t1 * 16u, 0u);
  // soffset path: third load into slots 128.., plane offset 16 bytes (texel + 1)
  dma16(r, lds0 + 2048u, (t0 % (uint32_t)(n_texels - 1)) * 16u, 16u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int nbad = 0;
  {
    const f32x4 a = lds_read16(lds0 + lane * 16u);
    const f32x4 e = reinterpret_cast<const f32x4 *>(src)[t0];
    nbad += (a[0] != e[0]) + (a[1] != e[1]) + (a[2] != e[2]) + (a[3] != e[3]);
    const f32x4 b = lds_read16(lds0 + 1024u + lane * 16u);
    const f32x4 e1 = reinterpret_cast<const f32x4 *>(src)[t1];
    if (lane + 64 < cnt)
      nbad += (b[0] != e1[0]) + (b[1] != e1[1]) + (b[2] != e1[2]) + (b[3] != e1[3]);
    else
      nbad += (b[0] != -777.f) + (b[3] != -777.f); // disabled lanes: untouched
    const f32x4 c = lds_read16(lds0 + 2048u + lane * 16u);
    const f32x4 e2 = reinterpret_cast<const f32x4 *>(src)[t0 % (uint32_t)(n_texels - 1) + 1];
    nbad += (c[0] != e2[0]) + (c[1] != e2[1]) + (c[2] != e2[2]) + (c[3] != e2[3]);
  }
  if (nbad)
    atomicAdd(bad, nbad);
  if (blockIdx.x == 0 && tid == 70)
    out[0] = lds_read16(lds0 + 16u * 6)[1] + s_pad[3];
}
int main()
{
  const int n = 20000;
  std::vector<float> h(n * 4);
  for (int i = 0; i < n * 4; ++i)
    h[i] = (float)(i % 9973) * 0.25f + 1.f;
  float *d, *out;
  int *bad, hb = -1;
  hipMalloc(&d, h.size() * 4);
  hipMalloc(&out, 16);
  hipMalloc(&bad, 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(bad, 0, 4);
  k<<<4096, 256>>>(d, n, bad, out);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("lds_dma_test: %s, mismatches %d\n", hipGetErrorString(e), hb);
  return (e != hipSuccess || hb != 0) ? 1 : 0;
}
