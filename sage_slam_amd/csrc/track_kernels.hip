// track_kernels.hip -- front-end tracker kernels (relative pose only, pre-sampled source features).
//
// Replaces cuda/photometric_factor_kernels.cpp:524-695 (6-dof), :697-873 (7-dof, + scale column) and
// :875-988 (error only) with their host reductions (:1166-1384).  Same fused structure as the mapping
// kernel: per pixel G (2x2), v (2), e over levels/channels, then S = Q^T G Q with the closed-form
// Q = [dpi/dxi (2x6) | dpi/dd * d/s0] (:680-681, :854-856 without fx,fy), reduced with wave64 DPP sums.
#include "sage_device.h"
#include "sage_internal.h"

namespace sage
{

struct TrackParams
{
  TrackEdge E;
  const WorkItem *work;
  float *partials;
  SagePyramid pyr;
  float eps;
  int dof;
  int exact_coord; // non-dyadic pyramid (pyramid_is_dyadic): the reference's own coordinate expression per pixel
};

__device__ __forceinline__ int sidx7(int i, int j) { return i * 7 - (i * (i - 1)) / 2 + (j - i); } // i<=j<7

template <int FS, bool JAC>
__global__ __launch_bounds__(kBlock) void track_kernel(const TrackParams prm)
{
  __shared__ float s_red[kWaves * kTrackScalars];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WorkItem wi = prm.work[blockIdx.x];
  wi.tile = uni(wi.tile);
  const TrackEdge &E = prm.E; // kernel argument: already in SGPRs
  const int N = E.N;
  const int n = wi.tile * kTile + tid;
  const bool in_range = n < N;
  const int nn = in_range ? n : 0;
  const Pose p10 = load_pose2(E.R, E.t);
  const SagePyramid &pyr = prm.pyr;
  const int exact_coord = prm.exact_coord;
  const float fx0 = pyr.cam[0].fx, fy0 = pyr.cam[0].fy, cx0 = pyr.cam[0].cx, cy0 = pyr.cam[0].cy;
  const int W0 = (int)pyr.cam[0].w, H0 = (int)pyr.cam[0].h;

  const float d = E.dpts0[nn];
  const float hm[3] = {E.homo[3 * nn + 0], E.homo[3 * nn + 1], E.homo[3 * nn + 2]};
  float rh[3], X[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
  {
    rh[i] = p10.R[i * 3 + 0] * hm[0] + p10.R[i * 3 + 1] * hm[1] + p10.R[i * 3 + 2] * hm[2];
    X[i] = d * rh[i] + p10.t[i];
  }
  const bool pos = X[2] > prm.eps;
  const float inv_z = 1.0f / X[2];
  const float p = (X[0] / X[2]) * fx0 + cx0;
  const float q = (X[1] / X[2]) * fy0 + cy0;
  const float m = mask_lookup(E.mask1, p, q, W0, H0);
  const float vm = (pos && in_range) ? m : 0.f;

  const uint32_t pyr_bytes = (uint32_t)FS * (uint32_t)pyr.P * 4u;
  const __amdgpu_buffer_rsrc_t r_f1 = make_rsrc(E.feat1, pyr_bytes);
  const __amdgpu_buffer_rsrc_t r_g1 = make_rsrc(JAC ? E.grad1 : E.feat1, JAC ? 2u * pyr_bytes : pyr_bytes);
  const uint32_t plane = (uint32_t)pyr.P * 4u;

  float G00 = 0.f, G01 = 0.f, G11 = 0.f, v0 = 0.f, v1 = 0.f, err = 0.f;
  for (int l = 0; l < pyr.levels; ++l)
  {
    const float fxl = pyr.cam[l].fx, fyl = pyr.cam[l].fy;
    const int Wl = (int)pyr.cam[l].w, Hl = (int)pyr.cam[l].h;
    Taps td;
    if (exact_coord) // (uniform) non-dyadic pyramid: the reference's own expression, photometric_factor_kernels.cpp:142-144
      make_taps(td, ((p + 0.5f) * fxl) / fx0 - 0.5f, ((q + 0.5f) * fyl) / fy0 - 0.5f, Wl, Hl);
    else
      make_taps(td, (p + 0.5f) * (fxl / fx0) - 0.5f, (q + 0.5f) * (fyl / fy0) - 0.5f, Wl, Hl);
    const uint32_t lo = (uint32_t)pyr.level_offsets[l];
    uint32_t dof[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      dof[k] = (lo + (uint32_t)td.off[k]) * 4u;
    const f32x4 *f0p = reinterpret_cast<const f32x4 *>(E.feat0s + ((size_t)l * N + nn) * FS);
    float g00 = 0.f, g01 = 0.f, g11 = 0.f, a0 = 0.f, a1 = 0.f, ee = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < FS / 4; ++c4)
    {
      const f32x4 f0v = f0p[c4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
      {
        const int c = c4 * 4 + cc;
        const uint32_t soff = (uint32_t)c * plane;
        float f1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          f1 += td.w[k] * buf_load(r_f1, dof[k], soff);
        const float diff = f0v[cc] - f1;
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
    const float wl = E.weights[l];
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
  err *= vm;

  float sc[kTrackScalars];
  int nsc;
  if (JAC)
  {
    const bool live = vm != 0.f;
    const float vm2 = vm * vm;
    G00 *= vm2; G01 *= vm2; G11 *= vm2; v0 *= vm2; v1 *= vm2;
    const float x_z = X[0] * inv_z, y_z = X[1] * inv_z;
    float Q[2][7] = {{inv_z, 0.f, -x_z * inv_z, -x_z * y_z, 1.f + x_z * x_z, -y_z, 0.f},
                     {0.f, inv_z, -y_z * inv_z, -(1.f + y_z * y_z), x_z * y_z, x_z, 0.f}};
    if (prm.dof == 7)
    {
      Q[0][6] = (rh[0] * inv_z - X[0] * rh[2] * inv_z * inv_z) * d / E.scale0;
      Q[1][6] = (rh[1] * inv_z - X[1] * rh[2] * inv_z * inv_z) * d / E.scale0;
    }
    float GQ0[7], GQ1[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
    {
      GQ0[j] = G00 * Q[0][j] + G01 * Q[1][j];
      GQ1[j] = G01 * Q[0][j] + G11 * Q[1][j];
    }
#pragma unroll
    for (int i = 0; i < 7; ++i)
    {
#pragma unroll
      for (int j = i; j < 7; ++j)
        sc[sidx7(i, j)] = live ? Q[0][i] * GQ0[j] + Q[1][i] * GQ1[j] : 0.f;
      sc[28 + i] = live ? Q[0][i] * v0 + Q[1][i] * v1 : 0.f;
    }
    sc[35] = err;
    sc[36] = vm;
    nsc = 37;
  }
  else
  {
    sc[0] = err;
    sc[1] = vm;
    nsc = 2;
  }
#pragma unroll
  for (int k = 0; k < (JAC ? 37 : 2); ++k)
  {
    const float s = wave_sum(sc[k]);
    if (lane == 63)
      s_red[wave * kTrackScalars + k] = s;
  }
  __syncthreads();
  const int PP = JAC ? kTrackScalars : 2;
  if (tid < nsc)
  {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w)
      a += s_red[w * kTrackScalars + tid];
    prm.partials[(size_t)blockIdx.x * PP + tid] = a;
  }
}

struct TrackFinalizeParams
{
  const float *partials;
  const float *weights;
  float *AtA, *Atb, *stats;
  int n_tiles, dof, levels, jac;
};

__global__ void track_finalize_kernel(const TrackFinalizeParams prm)
{
  __shared__ float s[kTrackScalars];
  const int tid = threadIdx.x;
  const int PP = prm.jac ? kTrackScalars : 2;
  if (tid < PP)
  {
    float a = 0.f;
    for (int t = 0; t < prm.n_tiles; ++t)
      a += prm.partials[(size_t)t * PP + tid];
    s[tid] = a;
  }
  __syncthreads();
  const float n_in = prm.jac ? s[36] : s[1];
  const float e = prm.jac ? s[35] : s[0];
  const bool ok = n_in > 0.f;
  if (tid == 0)
  {
    float ws = 0.f;
    for (int l = 0; l < prm.levels; ++l)
      ws += prm.weights[l];
    prm.stats[0] = ok ? e / n_in : 10.0f * ws; // photometric_factor_kernels.cpp:1224,1239
    prm.stats[1] = n_in;
  }
  if (!prm.jac)
    return;
  const int D = prm.dof;
  if (tid < D * D)
  {
    int i = tid / D, j = tid % D;
    if (i > j)
    {
      const int t = i; i = j; j = t;
    }
    prm.AtA[tid] = ok ? s[sidx7(i, j)] / n_in : 0.f;
  }
  if (tid < D)
    prm.Atb[tid] = ok ? s[28 + tid] / n_in : 0.f;
}

template <int FS>
static hipError_t track_impl(hipStream_t s, bool jac, int dof, const TrackEdge &edge, const LaunchCommon &lc,
                             const SagePyramid &pyr, float eps, float *AtA, float *Atb, float *stats)
{
  TrackParams p{};
  p.E = edge;
  p.work = lc.work;
  p.partials = lc.partials;
  p.pyr = pyr;
  p.eps = eps;
  p.dof = dof;
  p.exact_coord = pyramid_is_dyadic(pyr) ? 0 : 1;
  if (jac)
    hipLaunchKernelGGL((track_kernel<FS, true>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  else
    hipLaunchKernelGGL((track_kernel<FS, false>), dim3(lc.n_work), dim3(kBlock), 0, s, p);
  TrackFinalizeParams f{lc.partials, edge.weights, AtA, Atb, stats, lc.n_work, dof, pyr.levels, jac ? 1 : 0};
  hipLaunchKernelGGL(track_finalize_kernel, dim3(1), dim3(64), 0, s, f);
  return hipGetLastError();
}

hipError_t launch_track_linearize(hipStream_t s, int dof, int FS, const TrackEdge &edge, const LaunchCommon &lc,
                                  const SagePyramid &pyr, float eps, const EdgeOut &out)
{
  if (dof != 6 && dof != 7)
    return hipErrorInvalidValue;
  if (FS == 16)
    return track_impl<16>(s, true, dof, edge, lc, pyr, eps, out.AtA, out.Atb, out.stats);
  if (FS == 32)
    return track_impl<32>(s, true, dof, edge, lc, pyr, eps, out.AtA, out.Atb, out.stats);
  return hipErrorInvalidValue;
}

hipError_t launch_track_error(hipStream_t s, int FS, const TrackEdge &edge, const LaunchCommon &lc,
                              const SagePyramid &pyr, float eps, float *stats)
{
  if (FS == 16)
    return track_impl<16>(s, false, 6, edge, lc, pyr, eps, nullptr, nullptr, stats);
  if (FS == 32)
    return track_impl<32>(s, false, 6, edge, lc, pyr, eps, nullptr, nullptr, stats);
  return hipErrorInvalidValue;
}

} // namespace sage
