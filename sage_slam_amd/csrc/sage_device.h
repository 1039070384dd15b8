// sage_device.h -- shared device-side helpers for the gfx950 kernels.
// CDNA4 only: wave64, DPP row ops, raw buffer loads, f32 MFMA.  No portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sage_ba.h"
#include "sage_internal.h"

namespace sage
{

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- wave-uniform values
// Everything derived from blockIdx (work item, edge descriptor, base pointers) is wave-uniform, but hipcc cannot
// prove it once it went through memory; an unproven-uniform buffer descriptor makes it wrap EVERY buffer_load in
// a waterfall loop (CDNA guide T20).  Passing the scalars through readfirstlane makes the uniformity provable:
// descriptors and base pointers then live in SGPRs and loads use the saddr / s_load forms.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
template <class T>
__device__ __forceinline__ T *uni(T *p)
{
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}

// ---------------------------------------------------------------- buffer loads
// Raw buffer descriptors (T8 in the CDNA guide): one SGPR quad per array, the per-lane
// tap offset in a VGPR, the per-channel plane offset in an SGPR (soffset).  Out-of-image
// taps are given weight 0 and offset 0 rather than relying on the range check.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 buf_load2(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
  return __builtin_bit_cast(f32x2, v);
}

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return __builtin_bit_cast(f32x4, v);
}

// ---------------------------------------------------------------- wave64 sum (DPP)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v)
{
  const int x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
  return v + __builtin_bit_cast(float, x);
}

// total over the 64 lanes; valid in lanes 48..63 (read it from lane 63).
__device__ __forceinline__ float wave_sum(float v)
{
  v = dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);       // row_half_mirror
  v = dpp_add<0x140>(v);       // row_mirror        -> every lane holds its row (16) sum
  v = dpp_add<0x142, 0xA>(v);  // row_bcast15 into rows 1,3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast31 into rows 2,3 -> row 3 holds the wave total
  return v;
}

// wave64 integer min / max (same DPP ladder; lanes a masked step does not write keep `identity`)
template <int CTRL, int ROW_MASK, bool IS_MIN>
__device__ __forceinline__ int dpp_minmax(int v, int identity)
{
  const int x = __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xF, false);
  return IS_MIN ? min(v, x) : max(v, x);
}
template <bool IS_MIN>
__device__ __forceinline__ int wave_minmax(int v) // result valid in lane 63
{
  constexpr int id = IS_MIN ? 0x7fffffff : (int)0x80000000;
  v = dpp_minmax<0xB1, 0xF, IS_MIN>(v, id);
  v = dpp_minmax<0x4E, 0xF, IS_MIN>(v, id);
  v = dpp_minmax<0x141, 0xF, IS_MIN>(v, id);
  v = dpp_minmax<0x140, 0xF, IS_MIN>(v, id);
  v = dpp_minmax<0x142, 0xA, IS_MIN>(v, id);
  v = dpp_minmax<0x143, 0xC, IS_MIN>(v, id);
  return v;
}

// ---------------------------------------------------------------- bilinear taps
// 4-tap zero-padded bilinear sampler of the reference (photometric_factor_kernels.cpp:106-139,
// geometric_factor_kernels.cpp:546-571): taps (xf,yf) (xc,yc) (xf,yc) (xc,yf), each contributing only
// when inside the level.  off[] are ELEMENT offsets inside the level image (0 when the tap is outside).
struct Taps
{
  int off[4];
  float w[4];
  int xf, yf; // floor coordinates (clamped to a band around the image) -- used by the LDS-staged sampler
};

__device__ __forceinline__ void make_taps(Taps &t, float u, float v, int W, int H)
{
  // keep the float->int conversion defined for wild coordinates; anything clamped is far outside.
  // (a constant upper clamp: `(float)W + 8` per pyramid level is a loop-invariant the compiler hoists and spills)
  const float fu = fminf(fmaxf(floorf(u), -8.0f), 32768.0f);
  const float fv = fminf(fmaxf(floorf(v), -8.0f), 32768.0f);
  const int xf = (int)fu, yf = (int)fv;
  const int xc = xf + 1, yc = yf + 1;
  const float lx = (float)xc - u, ly = (float)yc - v;
  const float ux = 1.0f - lx, uy = 1.0f - ly;
  const bool xf_ok = (xf >= 0) & (xf < W), xc_ok = (xc >= 0) & (xc < W);
  const bool yf_ok = (yf >= 0) & (yf < H), yc_ok = (yc >= 0) & (yc < H);
  const bool ok0 = xf_ok & yf_ok, ok1 = xc_ok & yc_ok, ok2 = xf_ok & yc_ok, ok3 = xc_ok & yf_ok;
  t.w[0] = ok0 ? lx * ly : 0.0f;
  t.w[1] = ok1 ? ux * uy : 0.0f;
  t.w[2] = ok2 ? lx * uy : 0.0f;
  t.w[3] = ok3 ? ux * ly : 0.0f;
  t.off[0] = ok0 ? yf * W + xf : 0;
  t.off[1] = ok1 ? yc * W + xc : 0;
  t.off[2] = ok2 ? yc * W + xf : 0;
  t.off[3] = ok3 ? yf * W + xc : 0;
  t.xf = xf;
  t.yf = yf;
}

// nearest full-resolution mask lookup with C round() (half away from zero):
// photometric_factor_kernels.cpp:159-166, geometric_factor_kernels.cpp:585-598
__device__ __forceinline__ float mask_lookup(const float *__restrict__ mask, float p, float q, int W, int H)
{
  const float rp = fminf(fmaxf(roundf(p), -8.0f), 32768.0f); // (keeps the float -> int conversion defined)
  const float rq = fminf(fmaxf(roundf(q), -8.0f), 32768.0f);
  const int xr = (int)rp, yr = (int)rq;
  const bool ok = (xr >= 0) & (xr < W) & (yr >= 0) & (yr < H);
  return ok ? mask[yr * W + xr] : 0.0f;
}

// ---------------------------------------------------------------- small pose helpers (uniform data)
struct Pose
{
  float R[9];
  float t[3];
};

__device__ __forceinline__ Pose load_pose(const float *__restrict__ p)
{
  Pose o;
#pragma unroll
  for (int i = 0; i < 9; ++i)
    o.R[i] = p[i];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    o.t[i] = p[9 + i];
  return o;
}

// the 12 floats are wave-uniform: pin them to SGPRs so they do not occupy VGPRs across the sampling loop
__device__ __forceinline__ Pose load_pose2(const float *__restrict__ R, const float *__restrict__ t)
{
  Pose o;
#pragma unroll
  for (int i = 0; i < 9; ++i)
    o.R[i] = uni(R[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i)
    o.t[i] = uni(t[i]);
  return o;
}

// T10 = T1^-1 T0 : R10 = R1^T R0, t10 = R1^T (t0 - t1)  (core/gtsam/photometric_factor.cpp:280-281)
__device__ __forceinline__ Pose relative_pose(const Pose &p0, const Pose &p1)
{
  Pose o;
#pragma unroll
  for (int i = 0; i < 3; ++i)
  {
#pragma unroll
    for (int j = 0; j < 3; ++j)
      o.R[i * 3 + j] = p1.R[0 * 3 + i] * p0.R[0 * 3 + j] + p1.R[1 * 3 + i] * p0.R[1 * 3 + j] +
                       p1.R[2 * 3 + i] * p0.R[2 * 3 + j];
    o.t[i] = p1.R[0 * 3 + i] * (p0.t[0] - p1.t[0]) + p1.R[1 * 3 + i] * (p0.t[1] - p1.t[1]) +
             p1.R[2 * 3 + i] * (p0.t[2] - p1.t[2]);
  }
  return o;
}

// world-frame left-perturbation Jacobian of X = T1^-1 T0 (d x~):  dX/dT0 = R1^T [ I | -[Xw]x ]
// (photometric_factor_kernels.cpp:283-297); dX/dT1 = -dX/dT0 (:258-268).  Rows i of the 3x6.
// Written out term by term: as a dense 3x3 * 3x6 product the zeros of [I | -[Xw]x] survive as `R * 0` products (not
// foldable under IEEE rules), which the compiler hoists out of the sub-tile loop as loop invariants and then spills.
__device__ __forceinline__ void dX_dT0(const Pose &p1, const float Xw[3], float out[3][6])
{
#pragma unroll
  for (int i = 0; i < 3; ++i)
  {
    const float r0 = p1.R[0 * 3 + i], r1 = p1.R[1 * 3 + i], r2 = p1.R[2 * 3 + i]; // row i of R1^T
    out[i][0] = r0;
    out[i][1] = r1;
    out[i][2] = r2;
    out[i][3] = r2 * Xw[1] - r1 * Xw[2];
    out[i][4] = r0 * Xw[2] - r2 * Xw[0];
    out[i][5] = r1 * Xw[0] - r0 * Xw[1];
  }
}

// ---------------------------------------------------------------- basis tile -> LDS, sampled depth
// Gathers the CS-float basis rows of the tile's kTile source pixels into LDS (row stride CS+1, conflict-free
// for both the per-pixel dot product and the MFMA operand reads) with 128-byte coalesced segments, then
// returns this thread's depth  s0 * (bias0[i] + basis0[i,:] . code0)
// (photometric_factor_kernels.cpp:1094-1095, geometric_factor_kernels.cpp:514-521).
// Rows past N are zero-filled so they are inert in the MFMA contractions.
template <int CS, int LD = CS + 1>
__device__ __forceinline__ float stage_basis_and_depth(float *s_basis, int *s_loc, const float *__restrict__ basis0,
                                                       const float *__restrict__ bias0,
                                                       const float *__restrict__ code0, float scale0,
                                                       int my_loc, bool in_range, int tile_rows)
{
  constexpr int F4 = CS / 4;          // float4 per row
  constexpr int RPP = kBlock / F4;    // rows per pass
  const int tid = threadIdx.x;
  s_loc[tid] = my_loc;
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < kTile / RPP; ++pass)
  {
    const int row = pass * RPP + tid / F4;
    const int c4 = tid % F4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < tile_rows)
      v = *reinterpret_cast<const f32x4 *>(basis0 + (size_t)s_loc[row] * CS + c4 * 4);
    float *dst = s_basis + row * LD + c4 * 4;
    dst[0] = v[0];
    dst[1] = v[1];
    dst[2] = v[2];
    dst[3] = v[3];
  }
  __syncthreads();
  float dot = 0.f;
  const float *row = s_basis + tid * LD;
#pragma unroll
  for (int j = 0; j < CS; ++j)
    dot += row[j] * code0[j];
  const float b = in_range ? bias0[my_loc] : 0.f;
  return scale0 * (b + dot);
}

} // namespace sage
