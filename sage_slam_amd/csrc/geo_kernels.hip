// geo_kernels.hip -- fused geometric (depth-consistency, Cauchy-robust) linearize / error kernels.
//
// Replaces cuda/geometric_factor_kernels.cpp:472-720 (+ host :882-950) and :127-218 (+ :837-880).
// One residual per source pixel, D = 14+2CS columns [pose0 pose1 code0 code1 scale0 scale1].
// Every column of the weighted row is  coef * (an entry of y or of t):
//     y = [a(6), kappa*d0, Ds, rho]      a = dX_z/dT0 - gradD^T Jpi dX/dT0,  kappa = r_z - gradD^T dpi/dd
//     t = [kappa*b0 (CS) ; beta (CS)]    b0 = own basis row, beta = bilinear sample of basis1
//   pose0 = +a, pose1 = -a, code0 = s0*t[0:CS], code1 = -s1*t[CS:2CS], scale0 = y6/s0, scale1 = -y7/s1
// so  J^T W J  is assembled from  sum w y y^T (45 scalars, wave64 DPP sums),  sum w t t^T  and  sum w y t^T
// (f32 MFMA 16x16x4, K = 4 pixels).  beta is sampled directly in the MFMA operand layout (lane = (pixel k,
// channel i)), so it never exists per pixel in registers or LDS.
//
// Algorithmic bytes per source pixel: 4*(2CS + 9).
#include "sage_device.h"
#include "sage_internal.h"

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
};

__device__ __forceinline__ int gload_loc(const void *loc, int is64, int n)
{
  return is64 ? (int)reinterpret_cast<const long long *>(loc)[n] : reinterpret_cast<const int *>(loc)[n];
}

__device__ __forceinline__ int gsidx6(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

constexpr int kGeoStash = 11; // y(9) omega kappa

template <int CS, bool JAC>
__global__ __launch_bounds__(kBlock) void geo_kernel(const GeoParams prm)
{
  // Jacobian kernel: the LDS tile holds t = [b0 (CS) | beta (CS)] per pixel (row stride 2CS+1, conflict free for the
  // per-pixel accesses and <= 2-way for the MFMA operand reads); error kernel: only b0
  constexpr int LD = JAC ? 2 * CS + 1 : CS + 1;
  constexpr int N16 = geo_n16(CS);
  constexpr int NTT = N16 * (N16 + 1) / 2;
  constexpr int NT = NTT + N16;
  __shared__ float s_basis[kTile * LD];
  __shared__ int s_loc[kTile];
  __shared__ float s_stash[JAC ? kTile * kGeoStash : 1];
  __shared__ float s_red[kWaves * kGeoScalars];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WorkItem wi = prm.work[blockIdx.x];
  wi.edge = uni(wi.edge);
  wi.tile = uni(wi.tile);
  GeoEdge E = prm.table ? prm.table[wi.edge] : prm.single;
  E.bias0 = uni(E.bias0); E.basis0 = uni(E.basis0); E.dpt1 = uni(E.dpt1); E.dgrad1 = uni(E.dgrad1);
  E.basis1 = uni(E.basis1); E.mask1 = uni(E.mask1); E.homo = uni(E.homo); E.loc = uni(E.loc);
  E.R0 = uni(E.R0); E.t0 = uni(E.t0); E.R1 = uni(E.R1); E.t1 = uni(E.t1); E.R10 = uni(E.R10); E.t10 = uni(E.t10);
  E.code0 = uni(E.code0); E.scale0 = uni(E.scale0); E.scale1 = uni(E.scale1);
  E.scale0_val = uni(E.scale0_val); E.scale1_val = uni(E.scale1_val);
  E.N = uni(E.N); E.loc_is_i64 = uni(E.loc_is_i64);
  const int N = E.N;
  const float s0 = E.scale0 ? *E.scale0 : E.scale0_val;

  const Pose p0 = JAC ? load_pose2(E.R0, E.t0) : Pose{};
  const Pose p1 = JAC ? load_pose2(E.R1, E.t1) : Pose{};
  Pose p10;
  if (E.R10)
    p10 = load_pose2(E.R10, E.t10);
  else
    p10 = relative_pose(load_pose2(E.R0, E.t0), load_pose2(E.R1, E.t1));

  const float fx = prm.cam.fx, fy = prm.cam.fy, cx = prm.cam.cx, cy = prm.cam.cy;
  const int W = (int)prm.cam.w, H = (int)prm.cam.h;

  for (int k = tid; k < kWaves * kGeoScalars; k += kBlock)
    s_red[k] = 0.f;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float err_acc = 0.f, vm_acc = 0.f;

  for (int sub = 0; sub < prm.tiles_per_block; ++sub)
  {
  const int tile = wi.tile + sub;
  if (tile * kTile >= N)
    break;
  const int n = tile * kTile + tid;
  const bool in_range = n < N;
  const int tile_rows = min(kTile, N - tile * kTile);
  const int my_loc = in_range ? gload_loc(E.loc, E.loc_is_i64, n) : 0;
  const float d0 = stage_basis_and_depth<CS, LD>(s_basis, s_loc, E.basis0, E.bias0, E.code0, s0, my_loc, in_range,
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
  const float err = (pos && in_range) ? logf(1.0f + mr * mr / prm.loss_param) : 0.f; // :600

  if (!JAC)
  {
    err_acc += err;
    vm_acc += vm;
    __syncthreads();
    continue;
  }

  const bool live = vm != 0.f;
  float y[9];
  float kappa;
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
    y[6] = kappa * d0;
    y[7] = Ds;
    y[8] = rho;
  }
  // beta = bilinear sample of basis1 [H,W,CS] at the projection (:592-595), lane = pixel: every tap is one 4*CS-byte
  // texel read with dwordx4 loads, 16 independent loads in flight per half; the result goes to the LDS tile so the
  // MFMA loop below touches LDS only
  {
    float *trow = s_basis + tid * LD + CS;
#pragma unroll
    for (int h = 0; h < CS / 16; ++h)
    {
      f32x4 tq[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(E.basis1 + (size_t)tp.off[k] * CS + h * 16);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          tq[k][qq] = src[qq];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
      {
        f32x4 b = tp.w[0] * tq[0][qq];
        b += tp.w[1] * tq[1][qq];
        b += tp.w[2] * tq[2][qq];
        b += tp.w[3] * tq[3][qq];
        trow[h * 16 + qq * 4 + 0] = b[0];
        trow[h * 16 + qq * 4 + 1] = b[1];
        trow[h * 16 + qq * 4 + 2] = b[2];
        trow[h * 16 + qq * 4 + 3] = b[3];
      }
    }
  }
  // sqrt_cauchy_weight = m / sqrt(rho^2 + c)  (:690);  omega = its square
  const float om = live ? (m * m) / (rho * rho + prm.loss_param) : 0.f;
  if (!live)
  {
#pragma unroll
    for (int j = 0; j < 9; ++j)
      y[j] = 0.f;
    kappa = 0.f;
  }
  float sc[kGeoScalars];
#pragma unroll
  for (int i = 0; i < 6; ++i)
  {
#pragma unroll
    for (int j = i; j < 6; ++j)
      sc[gsidx6(i, j)] = om * y[i] * y[j];
    sc[21 + i] = om * y[6] * y[i];
    sc[27 + i] = om * y[7] * y[i];
    sc[36 + i] = om * y[8] * y[i];
  }
  sc[33] = om * y[6] * y[6];
  sc[34] = om * y[6] * y[7];
  sc[35] = om * y[7] * y[7];
  sc[42] = om * y[8] * y[6];
  sc[43] = om * y[8] * y[7];
  sc[44] = err;
  sc[45] = vm;
  {
    float *st = s_stash + tid * kGeoStash;
#pragma unroll
    for (int j = 0; j < 9; ++j)
      st[j] = y[j];
    st[9] = om;
    st[10] = kappa;
  }
#pragma unroll
  for (int k = 0; k < 46; ++k)
  {
    const float s = wave_sum(sc[k]);
    if (lane == 63)
      s_red[wave * kGeoScalars + k] += s;
  }
  __syncthreads();

  // ---- MFMA: T += w t t^T (upper-triangular 16x16 tiles), Xc += w y t^T ----
  {
    const int i = lane & 15, k = lane >> 4;
#pragma unroll 2
    for (int g = 0; g < 16; ++g)
    {
      const int px = wave * 64 + g * 4 + k;
      const float *st = s_stash + px * kGeoStash;
      const float *tr = s_basis + px * LD;
      const float om_k = st[9], kap = st[10];
      const float yi = (i < 9) ? st[i < 9 ? i : 0] : 0.f;
      float tv[N16];
#pragma unroll
      for (int b = 0; b < CS / 16; ++b)
      {
        tv[b] = kap * tr[b * 16 + i];
        tv[CS / 16 + b] = tr[CS + b * 16 + i];
      }
#pragma unroll
      for (int bi = 0; bi < N16; ++bi)
#pragma unroll
        for (int bj = bi; bj < N16; ++bj)
        {
          const int t = bi * N16 - (bi * (bi - 1)) / 2 + (bj - bi);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(om_k * tv[bi], tv[bj], acc[t], 0, 0, 0);
        }
#pragma unroll
      for (int bj = 0; bj < N16; ++bj)
        acc[NTT + bj] = __builtin_amdgcn_mfma_f32_16x16x4f32(om_k * yi, tv[bj], acc[NTT + bj], 0, 0, 0);
    }
  }
  __syncthreads(); // s_basis / s_stash are restaged by the next sub-tile
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

  // ---- cross-wave sum in a fixed order (deterministic), NT*256 floats staged in s_basis ----
  for (int w = 0; w < kWaves; ++w)
  {
    __syncthreads();
    if (wave == w)
    {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
          float *p = s_basis + t * 256 + r * 64 + lane;
          *p = (w == 0) ? acc[t][r] : *p + acc[t][r];
        }
    }
  }
  __syncthreads();
  float *out = prm.partials + (size_t)blockIdx.x * geo_partial_floats(CS);
  if (tid < kGeoScalars)
  {
    float a = 0.f;
    if (tid < 46)
#pragma unroll
      for (int w = 0; w < kWaves; ++w)
        a += s_red[w * kGeoScalars + tid];
    out[tid] = a;
  }
  for (int idx = tid; idx < NT * 256; idx += kBlock)
    out[kGeoScalars + idx] = s_basis[idx];
}

// ------------------------------------------------------------------------------------------------
struct GeoFinalizeParams
{
  GeoEdge single;
  const GeoEdge *table;
  const int32_t *edge_first;
  const int32_t *edge_tiles;
  const float *partials;
  float *AtA, *Atb, *stats;
  float weight;
};

template <int CS>
__global__ __launch_bounds__(kBlock) void geo_finalize_kernel(const GeoFinalizeParams prm)
{
  constexpr int PP = geo_partial_floats(CS);
  constexpr int D = 14 + 2 * CS;
  constexpr int N16 = geo_n16(CS);
  constexpr int NTT = N16 * (N16 + 1) / 2;
  __shared__ double s[PP]; // partial sums and every derived product stay in double until the single final rounding
  const int e = blockIdx.x, tid = threadIdx.x;
  const GeoEdge &E = prm.table ? prm.table[e] : prm.single;
  const float s0 = E.scale0 ? *E.scale0 : E.scale0_val;
  const float s1 = E.scale1 ? *E.scale1 : E.scale1_val;
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
  const double n_in = s[45];
  const bool ok = n_in > 0.0;
  const double wn = ok ? (double)prm.weight / n_in : 0.0;
  if (tid == 0)
  {
    prm.stats[2 * e + 0] = ok ? (float)(wn * s[44]) : 10.0f * prm.weight; // geometric_factor_kernels.cpp:934,944
    prm.stats[2 * e + 1] = (float)n_in;
  }
  auto telem = [&](int tile, int row, int col) -> double {
    return s[kGeoScalars + tile * 256 + (row & 3) * 64 + ((row >> 2) * 16 + col)];
  };
  auto TT = [&](int a, int b) -> double { // sum w t_a t_b
    int bi = a >> 4, bj = b >> 4, ra = a & 15, rb = b & 15;
    if (bi > bj || (bi == bj && ra > rb)) // always read the upper triangle: (w t_a) t_b != (w t_b) t_a in fp32
    {
      int t = bi; bi = bj; bj = t;
      t = ra; ra = rb; rb = t;
    }
    const int tile = bi * N16 - (bi * (bi - 1)) / 2 + (bj - bi);
    return telem(tile, ra, rb);
  };
  auto YT = [&](int r, int col) -> double { return telem(NTT + (col >> 4), r, col & 15); }; // sum w y_r t_col
  auto YY = [&](int a, int b) -> double { // sum w y_a y_b, a,b in 0..8 (never both 8)
    if (a > b)
    {
      const int t = a; a = b; b = t;
    }
    if (b < 6)
      return s[gsidx6(a, b)];
    if (b == 6)
      return a < 6 ? s[21 + a] : s[33];
    if (b == 7)
      return a < 6 ? s[27 + a] : (a == 6 ? s[34] : s[35]);
    return a < 6 ? s[36 + a] : (a == 6 ? s[42] : s[43]); // b == 8 (rho)
  };
  // column j -> (kind, index, coef): kind 0 = y entry, kind 1 = t entry
  auto column = [&](int j, int &kind, int &idx, double &coef) {
    if (j < 6) { kind = 0; idx = j; coef = 1.0; }
    else if (j < 12) { kind = 0; idx = j - 6; coef = -1.0; }
    else if (j < 12 + CS) { kind = 1; idx = j - 12; coef = (double)s0; }
    else if (j < 12 + 2 * CS) { kind = 1; idx = j - 12; coef = -(double)s1; }
    else if (j == 12 + 2 * CS) { kind = 0; idx = 6; coef = 1.0 / (double)s0; }
    else { kind = 0; idx = 7; coef = -1.0 / (double)s1; }
  };
  float *AtA = prm.AtA + (size_t)e * D * D;
  float *Atb = prm.Atb + (size_t)e * D;
  for (int q = tid; q < D * D + D; q += kBlock)
  {
    double val = 0.0;
    if (ok)
    {
      if (q < D * D)
      {
        int ki, ii, kj, ij;
        double ci, cj;
        column(q / D, ki, ii, ci);
        column(q % D, kj, ij, cj);
        double mv;
        if (ki == 0 && kj == 0)
          mv = YY(ii, ij);
        else if (ki == 1 && kj == 1)
          mv = TT(ii, ij);
        else
          mv = ki == 0 ? YT(ii, ij) : YT(ij, ii);
        val = wn * (ci * cj) * mv; // (ci*cj) first: exactly symmetric in (i, j)
      }
      else
      {
        int k, ii;
        double c;
        column(q - D * D, k, ii, c);
        val = wn * c * (k == 0 ? YY(ii, 8) : YT(8, ii));
      }
    }
    if (q < D * D)
      AtA[q] = (float)val;
    else
      Atb[q - D * D] = (float)val;
  }
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
  if (lc.ev_start)
    (void)hipEventRecord(lc.ev_start, s);
  hipLaunchKernelGGL((geo_kernel<CS, true>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  if (lc.ev_stop)
    (void)hipEventRecord(lc.ev_stop, s);
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
  f.weight = weight;
  hipLaunchKernelGGL((geo_finalize_kernel<CS>), dim3(lc.n_edges), dim3(kBlock), 0, s, f);
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
  if (lc.ev_start)
    (void)hipEventRecord(lc.ev_start, s);
  hipLaunchKernelGGL((geo_kernel<CS, false>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  if (lc.ev_stop)
    (void)hipEventRecord(lc.ev_stop, s);
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
