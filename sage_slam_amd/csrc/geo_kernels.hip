// geo_kernels.hip -- fused geometric (depth-consistency, Cauchy-robust) linearize / error kernels.
//
// Replaces cuda/geometric_factor_kernels.cpp:472-720 (+ host :882-950) and :127-218 (+ :837-880).
// One residual per source pixel, D = 14+2CS columns [pose0 pose1 code0 code1 scale0 scale1].
// Every column of the weighted row is  coef * (an entry of y or of t):
//     y = [a(6), kappa*d0, Ds, rho]      a = dX_z/dT0 - gradD^T Jpi dX/dT0,  kappa = r_z - gradD^T dpi/dd
//     t = [kappa*b0 (CS) ; beta (CS)]    b0 = own basis row, beta = bilinear sample of basis1
//   pose0 = +a, pose1 = -a, code0 = s0*t[0:CS], code1 = -s1*t[CS:2CS], scale0 = y6/s0, scale1 = -y7/s1
// so  J^T W J  is assembled from  sum w y y^T,  sum w t t^T  and  sum w y t^T, all f32 MFMA 16x16x4 contractions
// with K = 4 pixels.  b0 and beta are loaded from global memory directly in the MFMA operand layout (lane = (pixel k,
// channel i)), so they never exist per pixel in registers or LDS.
//
// Algorithmic bytes per source pixel: 4*(2CS + 9).
#include "sage_device.h"
#include "sage_internal.h"
#include "finalize_bodies.h"

#ifndef SAGE_GEO_WAVES
#define SAGE_GEO_WAVES 2 // workgroups per CU the linearize kernel is register-budgeted for (x4 waves)
#endif
#ifndef SAGE_GEO_AHEAD
#define SAGE_GEO_AHEAD 2 // pixel groups whose operand loads are in flight ahead of the MFMAs
#endif

namespace sage
{

struct GeoParams
{
  GeoEdge single;
  const GeoEdge *table;
  const WorkItem *work;
  float *partials;
  SageCamera cam;
  float eps, loss_param;
  int tiles_per_block;
  int width, height; // cam.w / cam.h as integers: scalar (SGPR) values for the buffer descriptors
  int n_work;
};
// MERGE (linearize only; LaunchCommon::merge_geo_weight): the blocks that involve code0 only through t0 = kappa*b0 -- the
// 3 t0 t0^T tiles and the 2 y t0^T tiles of 15 at CS = 32 -- are contracted by the photometric kernel of the same pair,
// which gets {omega, D, grad D} of every pixel through GeoEdge::px_out; their accumulators stay zero here

__device__ __forceinline__ int gload_loc(const void *loc, int is64, int n)
{
  return is64 ? (int)reinterpret_cast<const long long *>(loc)[n] : reinterpret_cast<const int *>(loc)[n];
}

// per-pixel hand-over from the geometry phase (lane = pixel) to the contraction phase (lane = (channel i, pixel k)):
//   [0..3] tap byte offsets (int bits)  [4..7] sqrt(omega)*tap weights  [8] sqrt(omega)*kappa  [9] loc*CS*4 (int bits)
//   [12..27] sqrt(omega)*y, zero-padded to the 16 operand rows
constexpr int kGeoStashLD = 28; // floats per pixel, 16-byte aligned rows

// which entry (row, col) of the  sum omega y y^T  tile carries scalar slot k of the partial layout
__device__ __forceinline__ void geo_scalar_rc(int k, int &r, int &c)
{
  r = 0;
  c = 0;
  if (k < 21)
  {
    int i = 0, base = 0;
    while (k >= base + (6 - i))
    {
      base += 6 - i;
      ++i;
    }
    r = i;
    c = i + (k - base);
  }
  else if (k < 27) { r = 6; c = k - 21; }
  else if (k < 33) { r = 7; c = k - 27; }
  else if (k == 33) { r = 6; c = 6; }
  else if (k == 34) { r = 6; c = 7; }
  else if (k == 35) { r = 7; c = 7; }
  else if (k < 42) { r = 8; c = k - 36; }
  else if (k == 42) { r = 8; c = 6; }
  else { r = 8; c = 7; }
}

// Geometry phase: lane = source pixel (depth from the keyframe's depth map, warp, 4-tap samples of D1 / grad D1,
// Cauchy weight) -> 20 floats per pixel in a wave-private LDS stash.  Contraction phase: lane = (channel i, pixel k);
// the operands t = [kappa*b0 ; beta] are loaded from global memory DIRECTLY in the MFMA operand layout (16 lanes read
// 64 contiguous bytes of one basis row / one basis1 texel), so no per-pixel basis tile ever sits in LDS: the kernel's
// LDS footprint is the 6 KiB stash per wave and the occupancy is set by the accumulators alone.  The 44 scalar sums
// sum omega y_i y_j ride on one more MFMA tile instead of 44 wave reductions.
//
// Cost model (measured, scripts/micro/mfma_rate.hip): on gfx950 an f32 MFMA 16x16x4 issues every ~35 cycles per SIMD
// and VALU instructions do NOT overlap with it -- every VALU op between two MFMAs adds its ~5 cycles, from the same
// wave or from another wave of the SIMD (f32 matrix and f32 vector peak are the same 157 TFLOP/s: same datapath).
// So the kernel time is (MFMA count * 35 + VALU count * 5) cycles per SIMD once memory latency is covered; a
// two-group ping-pong (geometry of one sub-tile under the MFMAs of another) was tried and lost to its own barriers.
// The contraction phase therefore keeps its per-group VALU work minimal: sqrt(omega) is folded into the tap weights,
// kappa and y in the geometry phase (one lane per pixel there, 16 lanes per pixel here), channel pairs are processed
// with packed f32 math, and the y operand is zero-padded in the stash instead of being masked.
constexpr int kGeoLinBlock = kBlock;

template <int CS, bool JAC, bool MERGE = false>
__global__ __launch_bounds__(kBlock, JAC ? SAGE_GEO_WAVES : 4) void geo_kernel(const GeoParams prm)
{
  constexpr int NW = kWaves;
  constexpr int N16 = geo_n16(CS);
  constexpr int NB = CS / 16;
  constexpr int NTT = N16 * (N16 + 1) / 2;
  constexpr int NT = NTT + N16;
  constexpr int NACC = NT + 1; // + the y y^T tile
  constexpr int STASH = JAC ? NW * 64 * kGeoStashLD : 1;
  constexpr int SUMBUF = JAC ? NACC * 256 : 1;
  __shared__ __attribute__((aligned(16))) float s_mem[STASH > SUMBUF ? STASH : SUMBUF];
  __shared__ float s_red[NW * 2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq = wave; // quarter of the 256-pixel sub-tile this wave owns
  const int bid = (int)blockIdx.x;
  WorkItem wi = prm.work[bid];
  wi.edge = uni(wi.edge);
  wi.tile = uni(wi.tile);
  GeoEdge E = prm.table ? prm.table[wi.edge] : prm.single;
  E.dpt0 = uni(E.dpt0); E.basis0 = uni(E.basis0); E.dpt1 = uni(E.dpt1); E.dgrad1 = uni(E.dgrad1);
  E.basis1 = uni(E.basis1); E.mask1 = uni(E.mask1); E.homo = uni(E.homo); E.loc = uni(E.loc);
  E.R0 = uni(E.R0); E.t0 = uni(E.t0); E.R1 = uni(E.R1); E.t1 = uni(E.t1); E.R10 = uni(E.R10); E.t10 = uni(E.t10);
  E.N = uni(E.N); E.loc_is_i64 = uni(E.loc_is_i64);
  E.px_out = uni(E.px_out);
  const int N = E.N;
  const float loss_param = E.loss_param > 0.f ? E.loss_param : prm.loss_param; // per-link parameter (mapper.cpp:369)

  const Pose p0 = JAC ? load_pose2(E.R0, E.t0) : Pose{};
  const Pose p1 = JAC ? load_pose2(E.R1, E.t1) : Pose{};
  Pose p10;
  if (E.R10)
    p10 = load_pose2(E.R10, E.t10);
  else
    p10 = relative_pose(load_pose2(E.R0, E.t0), load_pose2(E.R1, E.t1));

  const float fx = prm.cam.fx, fy = prm.cam.fy, cx = prm.cam.cx, cy = prm.cam.cy;
  const int W = prm.width, H = prm.height;
  // descriptors of the two basis arrays, built once from provably uniform values (else every load is a waterfall loop)
  const uint32_t basis_bytes = (uint32_t)W * (uint32_t)H * (uint32_t)(CS * 4);
  const __amdgpu_buffer_rsrc_t r_b0 = make_rsrc(E.basis0, basis_bytes), r_b1 = make_rsrc(E.basis1, basis_bytes);

  f32x4 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float err_acc = 0.f, vm_acc = 0.f;
  float *st_w = s_mem + wave * 64 * kGeoStashLD; // this wave's stash

  // sub-tiles this workgroup owns: [wi.tile, wi.tile + nsub)
  const int nsub = min(prm.tiles_per_block, (N + kTile - 1) / kTile - wi.tile);
  for (int h = 0; h < nsub; ++h)
  {
    bool slice_live = true; // false: no pixel of this wave's 64 is an inlier -- its contraction would add exact zeros
    {
    // wave priority: the geometry phase (loads + per-pixel arithmetic) ahead of the other waves' contraction phases
    // (r03: 0.550 -> 0.534 ms)
    if (JAC)
      __builtin_amdgcn_s_setprio(3);
    const int tile = wi.tile + h;
    const int n = tile * kTile + wq * 64 + lane;
    bool in_range = n < N;
    int my_loc = in_range ? gload_loc(E.loc, E.loc_is_i64, n) : 0;
    in_range = in_range && (unsigned)my_loc < (unsigned)(W * H); // a location outside the image is dropped, not read
    my_loc = in_range ? my_loc : 0;
    // depth of the source pixel: s0*(bias + basis.code), read from the keyframe's depth map (:514-521)
    const float d0 = in_range ? E.dpt0[my_loc] : 1.0f;

    float hm[3] = {0.f, 0.f, 1.f};
    if (in_range)
    {
      hm[0] = E.homo[3 * n + 0];
      hm[1] = E.homo[3 * n + 1];
      hm[2] = E.homo[3 * n + 2];
    }
    float rh[3], X[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
      rh[i] = p10.R[i * 3 + 0] * hm[0] + p10.R[i * 3 + 1] * hm[1] + p10.R[i * 3 + 2] * hm[2];
      X[i] = d0 * rh[i] + p10.t[i];
    }
    const bool pos = X[2] > prm.eps; // geometric_factor_kernels.cpp:541
    const float inv_z = 1.0f / X[2];
    const float u = (X[0] / X[2]) * fx + cx; // :543-544 (no half-pixel shift at level 0); true divisions
    const float v = (X[1] / X[2]) * fy + cy;
    Taps tp;
    make_taps(tp, u, v, W, H);
    float Ds = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      Ds += tp.w[k] * E.dpt1[tp.off[k]];
    const float m = mask_lookup(E.mask1, u, v, W, H);
    const float vm = (pos && in_range) ? m : 0.f;
    const float rho = Ds - X[2];
    const float mr = m * rho;
    const float err = (pos && in_range) ? logf(1.0f + mr * mr / loss_param) : 0.f; // :600
    err_acc += err;
    vm_acc += vm;
    slice_live = __ballot(vm != 0.f) != 0ull;
    if (JAC && slice_live)
    {
    const bool live = vm != 0.f;
    float y[9];
    float kappa;
    float gD_out[2] = {0.f, 0.f};
    {
      float gD[2] = {0.f, 0.f};
      const float *gx = E.dgrad1, *gy = E.dgrad1 + (size_t)W * H;
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        gD[0] += tp.w[k] * gx[tp.off[k]];
        gD[1] += tp.w[k] * gy[tp.off[k]];
      }
      float Xw[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        Xw[i] = d0 * (p0.R[i * 3 + 0] * hm[0] + p0.R[i * 3 + 1] * hm[1] + p0.R[i * 3 + 2] * hm[2]) + p0.t[i];
      float dX[3][6];
      dX_dT0(p1, Xw, dX);
      const float jx = -fx * X[0] * inv_z * inv_z, jy = -fy * X[1] * inv_z * inv_z;
#pragma unroll
      for (int j = 0; j < 6; ++j) // :671-679
      {
        const float Px = fx * inv_z * dX[0][j] + jx * dX[2][j];
        const float Py = fy * inv_z * dX[1][j] + jy * dX[2][j];
        y[j] = dX[2][j] - (gD[0] * Px + gD[1] * Py);
      }
      const float qx = fx * (rh[0] * inv_z - X[0] * rh[2] * inv_z * inv_z); // :681-682
      const float qy = fy * (rh[1] * inv_z - X[1] * rh[2] * inv_z * inv_z);
      kappa = rh[2] - (gD[0] * qx + gD[1] * qy); // :684-685
      gD_out[0] = gD[0];
      gD_out[1] = gD[1];
      y[6] = kappa * d0;
      y[7] = Ds;
      y[8] = rho;
    }
    // sqrt_cauchy_weight = m / sqrt(rho^2 + c)  (:690);  omega = its square
    const float om = live ? (m * m) / (rho * rho + loss_param) : 0.f;
    if (MERGE && in_range)
      reinterpret_cast<f32x4 *>(E.px_out)[n] = f32x4{om, Ds, gD_out[0], gD_out[1]};
    if (!live)
    {
#pragma unroll
      for (int j = 0; j < 9; ++j)
        y[j] = 0.f;
      kappa = 0.f;
    }
    {
      // sqrt(omega) goes into everything the contraction multiplies: omega t t^T = t' t'^T with t' = sqrt(omega) t
      const float so = sqrtf(om);
      f32x4 *st = reinterpret_cast<f32x4 *>(st_w + lane * kGeoStashLD);
      // byte offsets of the four basis1 texels / of the basis0 row: the contraction phase adds its channel offset
      st[0] = f32x4{__int_as_float(tp.off[0] * (CS * 4)), __int_as_float(tp.off[1] * (CS * 4)),
                    __int_as_float(tp.off[2] * (CS * 4)), __int_as_float(tp.off[3] * (CS * 4))};
      st[1] = f32x4{so * tp.w[0], so * tp.w[1], so * tp.w[2], so * tp.w[3]};
      st[2] = f32x4{so * kappa, __int_as_float(my_loc * (CS * 4)), 0.f, 0.f};
      st[3] = f32x4{so * y[0], so * y[1], so * y[2], so * y[3]};
      st[4] = f32x4{so * y[4], so * y[5], so * y[6], so * y[7]};
      st[5] = f32x4{so * y[8], 0.f, 0.f, 0.f};
      st[6] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    } // JAC
    } // geometry phase
    if (JAC && slice_live)
    {
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_wave_barrier(); // same-wave LDS hand-over (in-order LDS pipe): no workgroup barrier needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // ---- contraction over this wave's 64 pixels (staged by this wave in the previous half-step), K = 4 pixels per MFMA ----
    // Software pipeline over the 16 groups of 4 pixels: the operand loads of group g+2 and the stash reads of group
    // g+3 are issued before the MFMAs of group g (a group's ten dword loads see ~1-2 us of L2/HBM latency, its 15 MFMAs
    // take 0.2 us).  Fully unrolled so every stage lives in registers.
    {
      const int i = lane & 15, k = lane >> 4;
      const uint32_t lane_off = (uint32_t)i * (NB == 2 ? 8u : 4u);
      constexpr int G = 16, AHEAD = SAGE_GEO_AHEAD;
      f32x4 o4[G + 1], w4[G + 1];
      float yi[G + 1], kap[G + 1];
      int locp[G + 1];
      float tb[G + 1][NB], tq[G + 1][4][NB]; // CS = 16: one channel per lane
      f32x2 tbp[G + 1], tqp[G + 1][4];         // CS = 32: a channel pair per lane, kept as register pairs
// (macros, not lambdas: a buffer descriptor captured by a closure loses its provable uniformity and every load
//  becomes a waterfall loop)
#define SAGE_GEO_READ_STASH(g)                                                         \
  {                                                                                    \
    const float *p_ = st_w + ((g) * 4 + k) * kGeoStashLD;                              \
    o4[g] = *reinterpret_cast<const f32x4 *>(p_);                                      \
    w4[g] = *reinterpret_cast<const f32x4 *>(p_ + 4);                                  \
    const f32x2 kl_ = *reinterpret_cast<const f32x2 *>(p_ + 8); /* sqrt(omega)*kappa, loc */ \
    yi[g] = p_[12 + i];                                                                \
    kap[g] = kl_[0];                                                                   \
    locp[g] = __float_as_int(kl_[1]);                                                  \
  }
// CS = 32: one dwordx2 per lane covers the whole 128-byte row / texel with 16 lanes (one L1 request per pixel and tap
// instead of two); lane i then holds channels (2i, 2i+1), i.e. operand block 0 = even channels, block 1 = odd channels
// (the finalize kernel indexes the tiles accordingly).  CS = 16: one dword, block 0 = the 16 channels.
#define SAGE_GEO_ISSUE_LOADS(g)                                                                              \
  {                                                                                                          \
    if (NB == 2)                                                                                             \
    {                                                                                                        \
      tbp[g] = buf_load2(r_b0, (uint32_t)locp[g] + lane_off, 0);                                             \
      _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                          \
          tqp[g][q] = buf_load2(r_b1, (uint32_t)__float_as_int(o4[g][q]) + lane_off, 0);                     \
    }                                                                                                        \
    else                                                                                                     \
    {                                                                                                        \
      tb[g][0] = buf_load(r_b0, (uint32_t)locp[g] + lane_off, 0);                                            \
      _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                          \
          tq[g][q][0] = buf_load(r_b1, (uint32_t)__float_as_int(o4[g][q]) + lane_off, 0);                    \
    }                                                                                                        \
  }
// operands of group g: t' = sqrt(omega) [kappa*b0 ; beta], both MFMA operands of the t t^T tiles
// (CS = 32: the two channels of a lane as one register pair -- v_pk_mul / v_pk_fma with the tap weight broadcast:
//  5 VALU instructions per group instead of 10; same products, same order of the sums)
#define SAGE_GEO_STAGE_OPERANDS(g)                                                                             \
  {                                                                                                            \
    if (NB == 2)                                                                                               \
    {                                                                                                          \
      const f32x2 kb_ = kap[g] * tbp[g];                                                                       \
      f32x2 be_ = w4[g][0] * tqp[g][0];                                                                        \
      be_ += w4[g][1] * tqp[g][1];                                                                             \
      be_ += w4[g][2] * tqp[g][2];                                                                             \
      be_ += w4[g][3] * tqp[g][3]; /* beta (:592-595) */                                                       \
      bop[(g) & 1][0] = kb_[0];                                                                                \
      bop[(g) & 1][NB - 1] = kb_[1];                                                                           \
      bop[(g) & 1][NB] = be_[0];                                                                               \
      bop[(g) & 1][2 * NB - 1] = be_[1];                                                                       \
    }                                                                                                          \
    else                                                                                                       \
    {                                                                                                          \
      bop[(g) & 1][0] = kap[g] * tb[g][0];                                                                     \
      bop[(g) & 1][NB] = __builtin_fmaf(w4[g][3], tq[g][3][0], __builtin_fmaf(w4[g][2], tq[g][2][0],           \
                         __builtin_fmaf(w4[g][1], tq[g][1][0], w4[g][0] * tq[g][0][0])));                      \
    }                                                                                                          \
    ygv[(g) & 1] = yi[g];                                                                                      \
  }
      float bop[2][N16], ygv[2];
      // pipeline depth: loads AHEAD+1 groups ahead, stash reads one further, operand staging one group ahead -- the
      // MFMA burst of a group never waits on memory or on its own VALU preparation
#pragma unroll
      for (int g = 0; g < (AHEAD + 2 < G ? AHEAD + 2 : G); ++g)
        SAGE_GEO_READ_STASH(g)
#pragma unroll
      for (int g = 0; g < (AHEAD + 1 < G ? AHEAD + 1 : G); ++g)
        SAGE_GEO_ISSUE_LOADS(g)
      if (G > 0)
        SAGE_GEO_STAGE_OPERANDS(0)
#pragma unroll
      for (int g = 0; g < G; ++g)
      {
        __builtin_amdgcn_sched_barrier(0);
        if (g + AHEAD + 1 < G)
          SAGE_GEO_ISSUE_LOADS(g + AHEAD + 1)
        if (g + AHEAD + 2 < G)
          SAGE_GEO_READ_STASH(g + AHEAD + 2)
        if (g + 1 < G)
          SAGE_GEO_STAGE_OPERANDS(g + 1)
#pragma unroll
        for (int bi = 0; bi < N16; ++bi)
#pragma unroll
          for (int bj = bi; bj < N16; ++bj)
          {
            if (MERGE && bj < NB)
              continue; // t0 t0^T: in the photometric kernel's code-code tiles
            const int t = bi * N16 - (bi * (bi - 1)) / 2 + (bj - bi);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bop[g & 1][bi], bop[g & 1][bj], acc[t], 0, 0, 0);
          }
#pragma unroll
        for (int bj = 0; bj < N16; ++bj)
        {
          if (MERGE && bj < NB)
            continue; // y t0^T: in the photometric kernel's cross tiles
          acc[NTT + bj] = __builtin_amdgcn_mfma_f32_16x16x4f32(ygv[g & 1], bop[g & 1][bj], acc[NTT + bj], 0, 0, 0);
        }
        acc[NT] = __builtin_amdgcn_mfma_f32_16x16x4f32(ygv[g & 1], ygv[g & 1], acc[NT], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier(); // the stash is rewritten by the next sub-tile
    } // contraction phase
  } // sub-tile loop

  {
    const float se = wave_sum(err_acc), sn = wave_sum(vm_acc);
    if (lane == 63)
    {
      s_red[wave * 2 + 0] = se;
      s_red[wave * 2 + 1] = sn;
    }
  }
  __syncthreads(); // every wave is done with its stash: s_mem becomes the cross-wave sum buffer
  if (!JAC)
  {
    if (tid < 2)
    {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        a += s_red[w * 2 + tid];
      prm.partials[(size_t)bid * 2 + tid] = a;
    }
    return;
  }

  // ---- cross-wave sum in a fixed order (deterministic), NACC*256 floats ----
  for (int w = 0; w < NW; ++w)
  {
    if (w > 0)
      __syncthreads();
    if (wave == w)
    {
#pragma unroll
      for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
          float *q = s_mem + t * 256 + r * 64 + lane;
          *q = (w == 0) ? acc[t][r] : *q + acc[t][r];
        }
    }
  }
  __syncthreads();
  float *out = prm.partials + (size_t)bid * geo_partial_floats(CS);
  if (tid < kGeoScalars)
  {
    float a = 0.f;
    if (tid < 44)
    {
      int r, c;
      geo_scalar_rc(tid, r, c);
      a = s_mem[NT * 256 + (r & 3) * 64 + ((r >> 2) * 16 + c)];
    }
    else if (tid < 46)
    {
#pragma unroll
      for (int w = 0; w < NW; ++w)
        a += s_red[w * 2 + (tid - 44)];
    }
    out[tid] = a;
  }
  for (int idx = tid; idx < NT * 256; idx += kGeoLinBlock)
    out[kGeoScalars + idx] = s_mem[idx];
}

// ------------------------------------------------------------------------------------------------
// per-edge finalize (finalize_bodies.h), one workgroup per edge
template <int CS>
__global__ __launch_bounds__(kFinalizeBlock) void geo_finalize_kernel(const GeoFinalizeParams prm)
{
  __shared__ double s[geo_finalize_lds_doubles(CS)];
  geo_finalize_body<CS>(prm, prm.edge_base + (int)blockIdx.x, s);
}

hipError_t launch_stats_finalize(hipStream_t s, const LaunchCommon &lc, float *stats, float fallback, float scale);

template <int CS>
static hipError_t geo_lin_impl(hipStream_t s, const GeoEdge *single, const GeoEdge *table, const LaunchCommon &lc,
                               const SageCamera &cam, float eps, float loss_param, float weight, const EdgeOut &out)
{
  GeoParams p{};
  if (single)
    p.single = *single;
  p.table = table;
  p.work = lc.work;
  p.partials = lc.partials;
  p.cam = cam;
  p.eps = eps;
  p.loss_param = loss_param;
  p.tiles_per_block = lc.tiles_per_block;
  p.width = (int)cam.w;
  p.height = (int)cam.h;
  p.n_work = lc.n_work;
  if (lc.stage != 2)
  {
    if (lc.ev_start)
      (void)hipEventRecord(lc.ev_start, s);
    if (lc.merge_geo_weight > 0.f)
      hipLaunchKernelGGL((geo_kernel<CS, true, true>), dim3(lc.n_work), dim3(kGeoLinBlock), 0, s, p);
    else
      hipLaunchKernelGGL((geo_kernel<CS, true, false>), dim3(lc.n_work), dim3(kGeoLinBlock), 0, s, p);
    if (lc.ev_stop)
      (void)hipEventRecord(lc.ev_stop, s);
  }
  if (lc.stage == 1)
    return hipGetLastError();
  GeoFinalizeParams f{};
  if (single)
    f.single = *single;
  f.table = table;
  f.edge_first = lc.edge_first;
  f.edge_tiles = lc.edge_tiles;
  f.partials = lc.partials;
  f.AtA = out.AtA;
  f.Atb = out.Atb;
  f.stats = out.stats;
  f.wide = out.wide;
  f.weight = weight;
  f.photo_partials = lc.merge_geo_weight > 0.f ? lc.merge_photo_partials : nullptr;
  f.photo_rec_first = lc.merge_photo_rec_first;
  f.photo_rec_count = lc.merge_photo_rec_count;
  f.edge_base = lc.stage == 2 ? lc.edge_base : 0;
  const int n_fin = lc.stage == 2 ? lc.edge_count : lc.n_edges;
  if (n_fin > 0)
    hipLaunchKernelGGL((geo_finalize_kernel<CS>), dim3(n_fin), dim3(lc.fin_block), 0, s, f);
  return hipGetLastError();
}

template <int CS>
static hipError_t geo_err_impl(hipStream_t s, const GeoEdge *single, const GeoEdge *table, const LaunchCommon &lc,
                               const SageCamera &cam, float eps, float loss_param, float weight, float *stats)
{
  GeoParams p{};
  if (single)
    p.single = *single;
  p.table = table;
  p.work = lc.work;
  p.partials = lc.partials;
  p.cam = cam;
  p.eps = eps;
  p.loss_param = loss_param;
  p.tiles_per_block = lc.tiles_per_block;
  p.width = (int)cam.w;
  p.height = (int)cam.h;
  p.n_work = lc.n_work;
  if (lc.ev_start)
    (void)hipEventRecord(lc.ev_start, s);
  hipLaunchKernelGGL((geo_kernel<CS, false>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  if (lc.ev_stop)
    (void)hipEventRecord(lc.ev_stop, s);
  if (lc.stage == 1) // the caller forms the per-edge statistics itself (window error pass)
    return hipGetLastError();
  return launch_stats_finalize(s, lc, stats, 10.0f * weight, weight);
}

hipError_t launch_geo_linearize(hipStream_t s, int CS, const GeoEdge *single, const GeoEdge *table,
                                const LaunchCommon &lc, const SageCamera &cam, float eps, float loss_param,
                                float weight, const EdgeOut &out)
{
  if (CS == 32)
    return geo_lin_impl<32>(s, single, table, lc, cam, eps, loss_param, weight, out);
  if (CS == 16)
    return geo_lin_impl<16>(s, single, table, lc, cam, eps, loss_param, weight, out);
  return hipErrorInvalidValue;
}

hipError_t launch_geo_error(hipStream_t s, int CS, const GeoEdge *single, const GeoEdge *table,
                            const LaunchCommon &lc, const SageCamera &cam, float eps, float loss_param,
                            float weight, float *stats)
{
  if (CS == 32)
    return geo_err_impl<32>(s, single, table, lc, cam, eps, loss_param, weight, stats);
  if (CS == 16)
    return geo_err_impl<16>(s, single, table, lc, cam, eps, loss_param, weight, stats);
  return hipErrorInvalidValue;
}

} // namespace sage
