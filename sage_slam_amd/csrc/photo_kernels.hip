// photo_kernels.hip -- fused feature-metric (photometric) linearize / error kernels for gfx950.
//
// Replaces cuda/photometric_factor_kernels.cpp:33-368 (+ host reduction :1061-1164) and :370-522
// (+ :990-1059) of the reference.  The reference writes every residual's Jacobian row to global memory
// ([L,N,FS,13+CS], 236 MB per dense 128x160 edge) and reduces it with two GEMMs; here nothing per-residual
// ever leaves the CU:
//
//   per source pixel n (one lane):   G = sum_l w_l sum_c h h^T (2x2),  v = sum_l w_l sum_c h r,  e
//        with h = (fx_l gx, fy_l gy) the level-scaled sampled gradient, r = m (f0 - f1)
//   J_c = h^T P_unit,  P_unit = Q M_n,  Q = [A | q] (2x7): A = Jpi_unit R1^T [I | -[Xw]x],  q = dpi/dd
//        M_n = [[I6, -I6, 0, 0], [0, 0, s0 b_n^T, d_n/s0]]            (P_pose1 = -P_pose0)
//   => per pixel S = Q^T G Q (7x7), u = Q^T v (7); 37 scalars are summed with wave64 DPP reductions,
//      the code blocks  sum_n S66 b_n b_n^T  and  sum_n [S(0:6,6); S66 d; u6] b_n^T  are f32 MFMA
//      (v_mfma_f32_16x16x4_f32, K = 4 pixels) contractions fed from the LDS basis tile.
//
// Algorithmic bytes per source pixel: 4*[4*FS*rho + CS + 6].
#include "sage_device.h"
#include "sage_internal.h"

namespace sage
{

struct PhotoParams
{
  PhotoEdge single;
  const PhotoEdge *table;
  const WorkItem *work;
  float *partials;
  SagePyramid pyr;
  float w[SAGE_MAX_LEVELS];
  float eps;
  int tiles_per_block; // consecutive kTile-pixel sub-tiles one workgroup accumulates before writing its partial
};

__device__ __forceinline__ int load_loc(const void *loc, int is64, int n)
{
  return is64 ? (int)reinterpret_cast<const long long *>(loc)[n] : reinterpret_cast<const int *>(loc)[n];
}

__device__ __forceinline__ int sidx6(int i, int j) // upper-triangular index, i <= j < 6
{
  return i * 6 - (i * (i - 1)) / 2 + (j - i);
}

constexpr int kStageCap = 1024; // float4 slots of the LDS patch buffer (16 KiB)
constexpr int kTileDim = 16;    // 16 x 16 source pixels per tile (= kTile lanes)

// MODE 0: reference layout, direct gathers (per-edge operator API, sparse samplings)
// MODE 1: channel-group layout, direct dwordx4 gathers
// MODE 2: channel-group layout + 2-D source tiles + destination patches staged in LDS (window engine, dense sampling)
template <int CS, int FS, bool JAC, int MODE>
__global__ __launch_bounds__(kBlock, 3) void photo_kernel(const PhotoParams prm)
{
  constexpr bool PACKED = MODE >= 1;
  constexpr bool TILED = MODE == 2;
  constexpr int LD = CS + 1;
  constexpr int NT = photo_tiles(CS);
  __shared__ float s_basis[kTile * LD];
  __shared__ int s_loc[kTile];
  __shared__ float s_stash_raw[(JAC && !TILED) ? kTile * 9 : 1];
  __shared__ f32x4 s_stage[TILED ? kStageCap : 1];
  __shared__ int s_bbox[kWaves * 4];
  __shared__ float s_red[kWaves * kPhotoScalars];
  // the per-pixel stash (phase C/D) reuses the patch buffer: sampling is over by then
  float *s_stash = TILED ? reinterpret_cast<float *>(s_stage) : s_stash_raw;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WorkItem wi = prm.work[blockIdx.x];
  wi.edge = uni(wi.edge);
  wi.tile = uni(wi.tile);
  PhotoEdge E = prm.table ? prm.table[wi.edge] : prm.single;
  E.feat0 = uni(E.feat0); E.feat1 = uni(E.feat1); E.grad1 = uni(E.grad1); E.bias0 = uni(E.bias0);
  E.feat0_pk = uni(E.feat0_pk); E.feat1_pk = uni(E.feat1_pk); E.gx1_pk = uni(E.gx1_pk); E.gy1_pk = uni(E.gy1_pk);
  E.basis0 = uni(E.basis0); E.mask1 = uni(E.mask1); E.homo = uni(E.homo); E.loc = uni(E.loc);
  E.R0 = uni(E.R0); E.t0 = uni(E.t0); E.R1 = uni(E.R1); E.t1 = uni(E.t1); E.R10 = uni(E.R10); E.t10 = uni(E.t10);
  E.code0 = uni(E.code0); E.scale0 = uni(E.scale0); E.scale0_val = uni(E.scale0_val);
  E.N = uni(E.N); E.loc_is_i64 = uni(E.loc_is_i64);
  E.index_map0 = uni(E.index_map0); E.tiles0 = uni(E.tiles0); E.n_tiles0 = uni(E.n_tiles0); E.f0s = uni(E.f0s);
  const int N = E.N;
  const float scale0 = E.scale0 ? *E.scale0 : E.scale0_val;

  // ---- poses (wave-uniform) ----
  const Pose p0 = JAC ? load_pose2(E.R0, E.t0) : Pose{};
  const Pose p1 = JAC ? load_pose2(E.R1, E.t1) : Pose{};
  Pose p10;
  if (E.R10)
    p10 = load_pose2(E.R10, E.t10);
  else
    p10 = relative_pose(load_pose2(E.R0, E.t0), load_pose2(E.R1, E.t1));

  const SagePyramid &pyr = prm.pyr;
  const float fx0 = pyr.cam[0].fx, fy0 = pyr.cam[0].fy, cx0 = pyr.cam[0].cx, cy0 = pyr.cam[0].cy;
  const int W0 = (int)pyr.cam[0].w, H0 = (int)pyr.cam[0].h;
  const uint32_t pyr_bytes = (uint32_t)FS * (uint32_t)pyr.P * 4u;
  const __amdgpu_buffer_rsrc_t r_f0 = make_rsrc(PACKED ? E.feat0_pk : E.feat0, pyr_bytes);
  const __amdgpu_buffer_rsrc_t r_f1 = make_rsrc(PACKED ? E.feat1_pk : E.feat1, pyr_bytes);
  const __amdgpu_buffer_rsrc_t r_g1 =
      make_rsrc(PACKED ? (JAC ? E.gx1_pk : E.feat1_pk) : (JAC ? E.grad1 : E.feat1), (JAC && !PACKED) ? 2u * pyr_bytes : pyr_bytes);
  const __amdgpu_buffer_rsrc_t r_g1y = make_rsrc((PACKED && JAC) ? E.gy1_pk : E.feat1, pyr_bytes);
  const uint32_t plane = (uint32_t)pyr.P * 4u;

  for (int k = tid; k < kWaves * kPhotoScalars; k += kBlock)
    s_red[k] = 0.f;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float err_acc = 0.f, vm_acc = 0.f; // error-only path: lane-local sums over the sub-tiles

  for (int sub = 0; sub < prm.tiles_per_block; ++sub)
  {
  const int tile = wi.tile + sub;
  int n, my_loc, tile_rows;
  bool in_range;
  if (TILED)
  {
    if (tile >= E.n_tiles0)
      break;
    const int t = uni(E.tiles0[tile]);
    const int px = (t & 0xffff) + (tid & (kTileDim - 1)), py = (t >> 16) + (tid >> 4);
    const bool in_img = px < W0 && py < H0;
    my_loc = in_img ? py * W0 + px : 0;
    n = in_img ? E.index_map0[my_loc] : -1;
    in_range = n >= 0;
    if (!in_range)
      n = 0;
    tile_rows = kTile;
  }
  else
  {
    if (tile * kTile >= N)
      break;
    n = tile * kTile + tid;
    in_range = n < N;
    tile_rows = min(kTile, N - tile * kTile);
    my_loc = in_range ? load_loc(E.loc, E.loc_is_i64, n) : 0;
  }
  const float d = stage_basis_and_depth<CS>(s_basis, s_loc, E.basis0, E.bias0, E.code0, scale0, my_loc, in_range,
                                            tile_rows);

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
    X[i] = d * rh[i] + p10.t[i];
  }
  const bool pos = X[2] > prm.eps; // photometric_factor_kernels.cpp:96
  const float inv_z = 1.0f / X[2];
  // true divisions, like the reference: the coordinate chain's fp32 rounding is the dominant noise term of
  // the whole linearisation (every residual of the pixel inherits it), so no reciprocal shortcut here.
  const float p = (X[0] / X[2]) * fx0 + cx0; // :142-144 (level-0 pixel coordinates)
  const float q = (X[1] / X[2]) * fy0 + cy0;
  const float m = mask_lookup(E.mask1, p, q, W0, H0);
  const float vm = (pos && in_range) ? m : 0.0f; // sampled_valid_mask_1 (:237)

  // source coordinates at level 0 (+0.5): from homo in the Jacobian kernel (:101-103), from loc1d in the
  // error-only kernel (:423-424, :1012-1014)
  float su, sv;
  if (JAC)
  {
    su = hm[0] * fx0 + cx0 + 0.5f;
    sv = hm[1] * fy0 + cy0 + 0.5f;
  }
  else
  {
    su = (float)(my_loc % W0) + 0.5f;
    sv = (float)(my_loc / W0) + 0.5f;
  }

  float G00 = 0.f, G01 = 0.f, G11 = 0.f, v0 = 0.f, v1 = 0.f, err = 0.f;
  for (int l = 0; l < pyr.levels; ++l)
  {
    const float fxl = pyr.cam[l].fx, fyl = pyr.cam[l].fy;
    const int Wl = (int)pyr.cam[l].w, Hl = (int)pyr.cam[l].h;
    const float rx = fxl / fx0, ry = fyl / fy0;
    Taps ts, td;
    make_taps(ts, su * rx - 0.5f, sv * ry - 0.5f, Wl, Hl);
    make_taps(td, (p + 0.5f) * rx - 0.5f, (q + 0.5f) * ry - 0.5f, Wl, Hl);
    if (TILED && !in_range)
    {
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        ts.w[k] = td.w[k] = 0.f;
        ts.off[k] = td.off[k] = 0;
      }
    }
    const uint32_t lo = (uint32_t)pyr.level_offsets[l];
    uint32_t so[4], dof[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      so[k] = (lo + (uint32_t)ts.off[k]) * 4u;
      dof[k] = (lo + (uint32_t)td.off[k]) * 4u;
    }
    float g00 = 0.f, g01 = 0.f, g11 = 0.f, a0 = 0.f, a1 = 0.f, ee = 0.f;
    bool staged = false;
    if (TILED)
    {
      // ---- bounding box (level texels) of the tile's destination taps ----
      const int BIG = 1 << 28;
      int bb[4];
      bb[0] = wave_minmax<true>(in_range ? td.xf : BIG);
      bb[1] = wave_minmax<false>(in_range ? td.xf + 1 : -BIG);
      bb[2] = wave_minmax<true>(in_range ? td.yf : BIG);
      bb[3] = wave_minmax<false>(in_range ? td.yf + 1 : -BIG);
      if (lane == 63)
      {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          s_bbox[wave * 4 + k] = bb[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        int v = s_bbox[k];
#pragma unroll
        for (int w = 1; w < kWaves; ++w)
          v = (k & 1) ? max(v, s_bbox[w * 4 + k]) : min(v, s_bbox[w * 4 + k]);
        bb[k] = uni(v);
      }
      const int dx0 = max(bb[0], 0), dx1 = min(bb[1], Wl - 1), dy0 = max(bb[2], 0), dy1 = min(bb[3], Hl - 1);
      const int dbw = dx1 - dx0 + 1, dbh = dy1 - dy0 + 1;
      constexpr int NA = JAC ? 3 : 1; // arrays staged for the destination: f1 (+ gx, gy)
      staged = dbw > 0 && dbh > 0 && dbw <= 32 && NA * dbw * dbh <= kStageCap;
      if (staged)
      {
        const int dsz = dbw * dbh;
        int di[4]; // LDS slots of this lane's taps (only meaningful where the tap weight is non-zero)
        {
          const int b0 = (td.yf - dy0) * dbw + (td.xf - dx0);
          di[0] = b0; di[1] = b0 + dbw + 1; di[2] = b0 + dbw; di[3] = b0 + 1;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            di[k] = (td.w[k] != 0.f) ? di[k] : 0;
        }
        const int rr = tid >> 5, cc = tid & 31; // 8 patch rows x 32 columns per pass
        // pre-sampled source features of this keyframe: [L][FS/4][N][4]
        const f32x4 *f0s = reinterpret_cast<const f32x4 *>(E.f0s) + (size_t)l * (FS / 4) * N + n;
        for (int g = 0; g < FS / 4; ++g)
        {
          const uint32_t soff = (uint32_t)g * plane * 4u;
          if (g > 0)
            __syncthreads(); // the previous group's taps have been read
          if (cc < dbw)
            for (int r = rr; r < dbh; r += 8)
            {
              const uint32_t go = (lo + (uint32_t)((dy0 + r) * Wl + dx0 + cc)) * 16u;
              s_stage[r * dbw + cc] = buf_load4(r_f1, go, soff);
              if (JAC)
              {
                s_stage[dsz + r * dbw + cc] = buf_load4(r_g1, go, soff);
                s_stage[2 * dsz + r * dbw + cc] = buf_load4(r_g1y, go, soff);
              }
            }
          const f32x4 f0 = f0s[(size_t)g * N];
          __syncthreads();
          f32x4 f1 = {0.f, 0.f, 0.f, 0.f}, gx = f1, gy = f1;
#pragma unroll
          for (int k = 0; k < 4; ++k)
          {
            f1 += td.w[k] * s_stage[di[k]];
            if (JAC)
            {
              gx += td.w[k] * s_stage[dsz + di[k]];
              gy += td.w[k] * s_stage[2 * dsz + di[k]];
            }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c)
          {
            const float diff = f0[c] - f1[c];
            ee += diff * diff;
            if (JAC)
            {
              const float hx = fxl * gx[c], hy = fyl * gy[c];
              g00 += hx * hx;
              g01 += hx * hy;
              g11 += hy * hy;
              a0 += hx * diff;
              a1 += hy * diff;
            }
          }
        }
        __syncthreads(); // patch buffer is reused by the next level (and by the stash afterwards)
      }
      else
      {
        // rare: the tile's footprint does not fit the patch buffer (extreme warps) -> plain gathers, one tap at a time
        const f32x4 *f0s = reinterpret_cast<const f32x4 *>(E.f0s) + (size_t)l * (FS / 4) * N + n;
        for (int g = 0; g < FS / 4; ++g)
        {
          const uint32_t soff = (uint32_t)g * plane * 4u;
          const f32x4 f0 = f0s[(size_t)g * N];
          f32x4 f1 = {0.f, 0.f, 0.f, 0.f}, gx = f1, gy = f1;
          for (int k = 0; k < 4; ++k)
          {
            f1 += td.w[k] * buf_load4(r_f1, dof[k] * 4u, soff);
            if (JAC)
            {
              gx += td.w[k] * buf_load4(r_g1, dof[k] * 4u, soff);
              gy += td.w[k] * buf_load4(r_g1y, dof[k] * 4u, soff);
            }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c)
          {
            const float diff = f0[c] - f1[c];
            ee += diff * diff;
            if (JAC)
            {
              const float hx = fxl * gx[c], hy = fyl * gy[c];
              g00 += hx * hx;
              g01 += hx * hy;
              g11 += hy * hy;
              a0 += hx * diff;
              a1 += hy * diff;
            }
          }
        }
        staged = true; // handled
      }
    }
    if (TILED)
    {
      (void)staged; // both tiled paths have accumulated this level above
    }
    else if (PACKED)
    {
      // channel-group layout: one dwordx4 per tap and group of 4 channels (16 B per lane, 1 KiB contiguous per wave)
      for (int g = 0; g < FS / 4; ++g)
      {
        const uint32_t soff = (uint32_t)g * plane * 4u;
        // issue every tap load of the group first (16 independent dwordx4 in flight), then consume: without the
        // scheduling barrier hipcc serialises load->wait->use through one register quad
        f32x4 t0[4], t1[4], tx[JAC ? 4 : 1], ty[JAC ? 4 : 1];
        // source features: pose-independent, pre-sampled once per keyframe by the window engine ([L][FS/4][N][4])
        (void)t0;
        const f32x4 f0pre = reinterpret_cast<const f32x4 *>(E.f0s)[((size_t)l * (FS / 4) + g) * N + (in_range ? n : 0)];
#ifdef SAGE_EXP_NO_LOADS
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          t1[k] = f32x4{td.w[k], p, q, (float)dof[k]};
          if (JAC)
          {
            tx[k] = f32x4{p, td.w[k], q, (float)soff};
            ty[k] = f32x4{q, p, td.w[k], 1.f};
          }
        }
#else
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          t1[k] = buf_load4(r_f1, dof[k] * 4u, soff);
          if (JAC)
          {
            tx[k] = buf_load4(r_g1, dof[k] * 4u, soff);
            ty[k] = buf_load4(r_g1y, dof[k] * 4u, soff);
          }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        f32x4 f0 = f0pre, f1 = {0.f, 0.f, 0.f, 0.f}, gx = f1, gy = f1;
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          f1 += td.w[k] * t1[k];
          if (JAC)
          {
            gx += td.w[k] * tx[k];
            gy += td.w[k] * ty[k];
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
        {
          const float diff = f0[c] - f1[c];
          ee += diff * diff;
          if (JAC)
          {
            const float hx = fxl * gx[c], hy = fyl * gy[c];
            g00 += hx * hx;
            g01 += hx * hy;
            g11 += hy * hy;
            a0 += hx * diff;
            a1 += hy * diff;
          }
        }
      }
    }
    else
    {
#pragma unroll 4
    for (int c = 0; c < FS; ++c)
    {
      const uint32_t soff = (uint32_t)c * plane;
      float f0 = 0.f, f1 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        f0 += ts.w[k] * buf_load(r_f0, so[k], soff);
        f1 += td.w[k] * buf_load(r_f1, dof[k], soff);
      }
      const float diff = f0 - f1;
      ee += diff * diff;
      if (JAC)
      {
        const uint32_t soff_y = (uint32_t)(FS + c) * plane;
        float gx = 0.f, gy = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          gx += td.w[k] * buf_load(r_g1, dof[k], soff);
          gy += td.w[k] * buf_load(r_g1, dof[k], soff_y);
        }
        const float hx = fxl * gx, hy = fyl * gy;
        g00 += hx * hx;
        g01 += hx * hy;
        g11 += hy * hy;
        a0 += hx * diff;
        a1 += hy * diff;
      }
    }
    }
    const float wl = prm.w[l];
    err += wl * ee;
    if (JAC)
    {
      G00 += wl * g00;
      G01 += wl * g01;
      G11 += wl * g11;
      v0 += wl * a0;
      v1 += wl * a1;
    }
  }
  err *= vm; // within_mask * pow(diff,2)  (:228)

  if (!JAC)
  {
    err_acc += err;
    vm_acc += vm;
    __syncthreads(); // s_basis / s_loc are restaged by the next sub-tile
    continue;
  }

  // ---- per-pixel 7x7 reduced system ----
  const bool live = vm != 0.0f;
  const float vm2 = vm * vm; // gradient and residual both carry m (:200, :234)
  G00 *= vm2;
  G01 *= vm2;
  G11 *= vm2;
  v0 *= vm2;
  v1 *= vm2;
  float Q[2][7];
  {
    float Xw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      Xw[i] = d * (p0.R[i * 3 + 0] * hm[0] + p0.R[i * 3 + 1] * hm[1] + p0.R[i * 3 + 2] * hm[2]) + p0.t[i];
    float dX[3][6];
    dX_dT0(p1, Xw, dX);
    const float jx = -X[0] * inv_z * inv_z, jy = -X[1] * inv_z * inv_z;
#pragma unroll
    for (int j = 0; j < 6; ++j)
    {
      Q[0][j] = inv_z * dX[0][j] + jx * dX[2][j];
      Q[1][j] = inv_z * dX[1][j] + jy * dX[2][j];
    }
    Q[0][6] = rh[0] * inv_z - X[0] * rh[2] * inv_z * inv_z; // :324-325 without fx, fy
    Q[1][6] = rh[1] * inv_z - X[1] * rh[2] * inv_z * inv_z;
  }
  float sc[kPhotoScalars];
  float S6[7], u6;
  {
    float GQ0[7], GQ1[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
    {
      GQ0[j] = G00 * Q[0][j] + G01 * Q[1][j];
      GQ1[j] = G01 * Q[0][j] + G11 * Q[1][j];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j)
        sc[sidx6(i, j)] = live ? Q[0][i] * GQ0[j] + Q[1][i] * GQ1[j] : 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i)
      S6[i] = live ? Q[0][i] * GQ0[6] + Q[1][i] * GQ1[6] : 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j)
    {
      sc[21 + j] = S6[j] * d;
      sc[28 + j] = live ? Q[0][j] * v0 + Q[1][j] * v1 : 0.f;
    }
    u6 = live ? Q[0][6] * v0 + Q[1][6] * v1 : 0.f;
    sc[27] = S6[6] * d * d;
    sc[34] = u6 * d;
    sc[35] = err;
    sc[36] = vm;
  }
  // stash the rows that multiply b_n: c (6), sigma*d, u6, and sigma itself
  {
    float *st = s_stash + tid * 9;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      st[j] = S6[j];
    st[6] = S6[6] * d;
    st[7] = u6;
    st[8] = S6[6];
  }
#pragma unroll
  for (int k = 0; k < 37; ++k)
  {
    const float s = wave_sum(sc[k]);
    if (lane == 63)
      s_red[wave * kPhotoScalars + k] += s; // only this lane ever touches this slot
  }
  __syncthreads(); // stash visible to the wave's other lanes

  // ---- MFMA contractions over this wave's 64 pixels, 4 pixels (K) per instruction ----
  {
    const int i = lane & 15, k = lane >> 4;
#pragma unroll 4
    for (int g = 0; g < 16; ++g)
    {
      const int px = wave * 64 + g * 4 + k;
      const float *br = s_basis + px * LD;
      const float *st = s_stash + px * 9;
      const float sg = st[8];
      const float ai = (i < 8) ? st[i & 7] : 0.f;
      const float bl = br[i];
      if (CS == 32)
      {
        const float bh = br[16 + i];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sg * bl, bl, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sg * bl, bh, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(sg * bh, bh, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, bl, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, bh, acc[4], 0, 0, 0);
      }
      else
      {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sg * bl, bl, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, bl, acc[1], 0, 0, 0);
      }
    }
  }
  __syncthreads(); // everyone is done reading s_basis / s_stash before they are restaged or reused
  } // sub-tile loop

  if (!JAC)
  {
    const float se = wave_sum(err_acc), sn = wave_sum(vm_acc);
    if (lane == 63)
    {
      s_red[wave * 2 + 0] = se;
      s_red[wave * 2 + 1] = sn;
    }
    __syncthreads();
    if (tid < 2)
    {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w)
        a += s_red[w * 2 + tid];
      prm.partials[(size_t)blockIdx.x * 2 + tid] = a;
    }
    return;
  }

  {
    float *sw = s_basis + wave * (NT * 256);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sw[t * 256 + r * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  float *out = prm.partials + (size_t)blockIdx.x * photo_partial_floats(CS);
  if (tid < kPhotoScalars)
  {
    float a = 0.f;
    if (tid < 37)
#pragma unroll
      for (int w = 0; w < kWaves; ++w)
        a += s_red[w * kPhotoScalars + tid];
    out[tid] = a;
  }
  for (int idx = tid; idx < NT * 256; idx += kBlock)
  {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w)
      a += s_basis[w * (NT * 256) + idx];
    out[kPhotoScalars + idx] = a;
  }
}

// ------------------------------------------------------------------------------------------------
// finalize: sum the workgroup partials of an edge in a fixed order (deterministic), expand the reduced
// blocks into the reference layout [pose0 pose1 code0 scale0] (photometric_factor_kernels.cpp:350-364),
// apply 1/num_inliers and the zero-overlap fallback (:1139-1161).
// ------------------------------------------------------------------------------------------------
struct PhotoFinalizeParams
{
  PhotoEdge single;
  const PhotoEdge *table;
  const int32_t *edge_first;
  const int32_t *edge_tiles;
  const float *partials;
  float *AtA, *Atb, *stats;
  float wsum;
};

__device__ __forceinline__ double tile_elem(const double *s, int base, int tile, int row, int col)
{
  return s[base + tile * 256 + (row & 3) * 64 + ((row >> 2) * 16 + col)];
}

template <int CS>
__global__ __launch_bounds__(kBlock) void photo_finalize_kernel(const PhotoFinalizeParams prm)
{
  constexpr int PP = photo_partial_floats(CS);
  constexpr int D = 13 + CS;
  __shared__ double s[PP]; // partial sums and every derived product stay in double until the single final rounding
  const int e = blockIdx.x, tid = threadIdx.x;
  const PhotoEdge &E = prm.table ? prm.table[e] : prm.single;
  const float s0 = E.scale0 ? *E.scale0 : E.scale0_val;
  const int first = prm.edge_first[e], nt = prm.edge_tiles[e];
  for (int idx = tid; idx < PP; idx += kBlock)
  {
    double a = 0.0; // the per-workgroup partials are summed in double: free (a few dozen adds), and it keeps the
                    // engine's accumulation noise below the reference's own fp32 floor
    for (int t = 0; t < nt; ++t)
      a += (double)prm.partials[(size_t)(first + t) * PP + idx];
    s[idx] = a;
  }
  __syncthreads();
  const double s0d = (double)s0;
  const double n_in = s[36];
  const bool ok = n_in > 0.0;
  const double inv_n = ok ? 1.0 / n_in : 0.0;
  float *AtA = prm.AtA + (size_t)e * D * D;
  float *Atb = prm.Atb + (size_t)e * D;
  if (tid == 0)
  {
    prm.stats[2 * e + 0] = ok ? (float)(s[35] * inv_n) : 10.0f * prm.wsum;
    prm.stats[2 * e + 1] = (float)n_in;
  }
  auto X = [&](int row, int col) -> double { // sum_n a_n[row] * b_n[col]
    if (CS == 32)
      return tile_elem(s, kPhotoScalars, col < 16 ? 3 : 4, row, col & 15);
    return tile_elem(s, kPhotoScalars, 1, row, col);
  };
  auto CC = [&](int i, int j) -> double { // sum_n sigma_n b_n[i] b_n[j]
    if (CS == 32)
    {
      const int ti = i >> 4, tj = j >> 4;
      if (ti <= tj)
        return tile_elem(s, kPhotoScalars, ti + tj, i & 15, j & 15); // (0,0)->0 (0,1)->1 (1,1)->2
      return tile_elem(s, kPhotoScalars, 1, j & 15, i & 15);
    }
    return tile_elem(s, kPhotoScalars, 0, i, j);
  };
  for (int idx = tid; idx < D * D + D; idx += kBlock)
  {
    double val = 0.0;
    if (ok)
    {
      if (idx < D * D)
      {
        int i = idx / D, j = idx % D;
        if (i > j)
        {
          const int t = i;
          i = j;
          j = t;
        }
        // classes: pose (0..11), code (12..12+CS-1), scale (12+CS)
        if (j < 12)
        {
          const double sg = ((i >= 6) != (j >= 6)) ? -1.0 : 1.0;
          const int a = i % 6, b = j % 6;
          val = sg * s[a <= b ? sidx6(a, b) : sidx6(b, a)];
        }
        else if (i < 12)
        {
          const double sg = (i >= 6) ? -1.0 : 1.0;
          if (j < 12 + CS)
            val = sg * s0d * X(i % 6, j - 12);
          else
            val = sg * s[21 + i % 6] / s0d;
        }
        else if (i < 12 + CS)
        {
          if (j < 12 + CS)
            val = s0d * s0d * CC(i - 12, j - 12);
          else
            val = X(6, i - 12);
        }
        else
          val = s[27] / (s0d * s0d);
        val *= inv_n;
      }
      else
      {
        const int i = idx - D * D;
        if (i < 12)
          val = ((i >= 6) ? -1.0 : 1.0) * s[28 + i % 6];
        else if (i < 12 + CS)
          val = s0d * X(7, i - 12);
        else
          val = s[34] / s0d;
        val *= inv_n;
      }
    }
    if (idx < D * D)
      AtA[idx] = (float)val;
    else
      Atb[idx - D * D] = (float)val;
  }
}

// error-only finalize: stats[e] = {sum(err)/n_in or 10*wsum, n_in}   (:1049-1058)
struct StatsFinalizeParams
{
  const int32_t *edge_first;
  const int32_t *edge_tiles;
  const float *partials; // [n_work][2]
  float *stats;
  float fallback; // 10*sum(w) or 10*weight
  float scale;    // 1 (photometric) or weight (geometric)
  int n_edges;
};

__global__ void stats_finalize_kernel(const StatsFinalizeParams prm)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= prm.n_edges)
    return;
  const int first = prm.edge_first[e], nt = prm.edge_tiles[e];
  float se = 0.f, sn = 0.f;
  for (int t = 0; t < nt; ++t)
  {
    se += prm.partials[(size_t)(first + t) * 2 + 0];
    sn += prm.partials[(size_t)(first + t) * 2 + 1];
  }
  prm.stats[2 * e + 0] = sn > 0.f ? prm.scale * se / sn : prm.fallback;
  prm.stats[2 * e + 1] = sn;
}

hipError_t launch_stats_finalize(hipStream_t s, const LaunchCommon &lc, float *stats, float fallback, float scale)
{
  StatsFinalizeParams fp{lc.edge_first, lc.edge_tiles, lc.partials, stats, fallback, scale, lc.n_edges};
  hipLaunchKernelGGL(stats_finalize_kernel, dim3((lc.n_edges + 63) / 64), dim3(64), 0, s, fp);
  return hipGetLastError();
}

static PhotoParams make_params(const PhotoEdge *single, const PhotoEdge *table, const LaunchCommon &lc,
                               const SagePyramid &pyr, const float *weights_host, float eps, float *wsum)
{
  PhotoParams p{};
  if (single)
    p.single = *single;
  p.table = table;
  p.work = lc.work;
  p.partials = lc.partials;
  p.pyr = pyr;
  float ws = 0.f;
  for (int l = 0; l < pyr.levels; ++l)
  {
    p.w[l] = weights_host[l];
    ws += weights_host[l];
  }
  p.eps = eps;
  p.tiles_per_block = lc.tiles_per_block;
  *wsum = ws;
  return p;
}

template <int CS, int FS>
static hipError_t photo_lin_impl(hipStream_t s, const PhotoEdge *single, const PhotoEdge *table,
                                 const LaunchCommon &lc, const SagePyramid &pyr, const float *wh, float eps,
                                 const EdgeOut &out)
{
  float wsum;
  PhotoParams p = make_params(single, table, lc, pyr, wh, eps, &wsum);
  if (lc.ev_start)
    (void)hipEventRecord(lc.ev_start, s);
  if (lc.tiled)
    hipLaunchKernelGGL((photo_kernel<CS, FS, true, 2>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  else if (lc.packed)
    hipLaunchKernelGGL((photo_kernel<CS, FS, true, 1>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  else
    hipLaunchKernelGGL((photo_kernel<CS, FS, true, 0>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  if (lc.ev_stop)
    (void)hipEventRecord(lc.ev_stop, s);
  PhotoFinalizeParams f{};
  if (single)
    f.single = *single;
  f.table = table;
  f.edge_first = lc.edge_first;
  f.edge_tiles = lc.edge_tiles;
  f.partials = lc.partials;
  f.AtA = out.AtA;
  f.Atb = out.Atb;
  f.stats = out.stats;
  f.wsum = wsum;
  hipLaunchKernelGGL((photo_finalize_kernel<CS>), dim3(lc.n_edges), dim3(kBlock), 0, s, f);
  return hipGetLastError();
}

template <int CS, int FS>
static hipError_t photo_err_impl(hipStream_t s, const PhotoEdge *single, const PhotoEdge *table,
                                 const LaunchCommon &lc, const SagePyramid &pyr, const float *wh, float eps,
                                 float *stats)
{
  float wsum;
  PhotoParams p = make_params(single, table, lc, pyr, wh, eps, &wsum);
  if (lc.ev_start)
    (void)hipEventRecord(lc.ev_start, s);
  if (lc.tiled)
    hipLaunchKernelGGL((photo_kernel<CS, FS, false, 2>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  else if (lc.packed)
    hipLaunchKernelGGL((photo_kernel<CS, FS, false, 1>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  else
    hipLaunchKernelGGL((photo_kernel<CS, FS, false, 0>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  if (lc.ev_stop)
    (void)hipEventRecord(lc.ev_stop, s);
  return launch_stats_finalize(s, lc, stats, 10.0f * wsum, 1.0f);
}

hipError_t launch_photo_linearize(hipStream_t s, int CS, int FS, const PhotoEdge *single, const PhotoEdge *table,
                                  const LaunchCommon &lc, const SagePyramid &pyr, const float *weights_host,
                                  float eps, const EdgeOut &out)
{
  if (CS == 32 && FS == 16)
    return photo_lin_impl<32, 16>(s, single, table, lc, pyr, weights_host, eps, out);
  if (CS == 16 && FS == 16)
    return photo_lin_impl<16, 16>(s, single, table, lc, pyr, weights_host, eps, out);
  if (CS == 32 && FS == 32)
    return photo_lin_impl<32, 32>(s, single, table, lc, pyr, weights_host, eps, out);
  if (CS == 16 && FS == 32)
    return photo_lin_impl<16, 32>(s, single, table, lc, pyr, weights_host, eps, out);
  return hipErrorInvalidValue;
}

hipError_t launch_photo_error(hipStream_t s, int CS, int FS, const PhotoEdge *single, const PhotoEdge *table,
                              const LaunchCommon &lc, const SagePyramid &pyr, const float *weights_host,
                              float eps, float *stats)
{
  if (CS == 32 && FS == 16)
    return photo_err_impl<32, 16>(s, single, table, lc, pyr, weights_host, eps, stats);
  if (CS == 16 && FS == 16)
    return photo_err_impl<16, 16>(s, single, table, lc, pyr, weights_host, eps, stats);
  if (CS == 32 && FS == 32)
    return photo_err_impl<32, 32>(s, single, table, lc, pyr, weights_host, eps, stats);
  if (CS == 16 && FS == 32)
    return photo_err_impl<16, 32>(s, single, table, lc, pyr, weights_host, eps, stats);
  return hipErrorInvalidValue;
}

} // namespace sage
