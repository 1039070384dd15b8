// keypoint_kernels.hip -- the sparse matched-keypoint factors: reprojection (fair loss) and 3-D match geometry
// (fair / L2 / huber / "unbiased"), mapper, loop and tracker variants.
//
// Replaces cuda/reprojection_factor_kernels.cpp of the reference: kernels :27-213 / :215-286 (mapper factor,
// D = 13+CS, [pose0 pose1 code0 scale0]) and :288-366 / :367-415 (tracker, D = 6), hosts :417-628.
// N is a few hundred keypoints at most (launch-latency, not bandwidth).  Small systems (tracker D = 6 / 7,
// loop closure D = 14): ONE launch -- a single workgroup keeps the weighted rows in LDS and contracts them itself in
// double (r04: tracker frame with a reprojection term 0.28 -> 0.20 ms, with match geometry 0.38 -> 0.23 ms).  The mapper
// factors (D = 13 + CS, 14 + 2 CS): rows written once (2N x (D+1) floats, the residual is the last column) and contracted
// by D workgroups in double -- two launches per call.  Same conventions as the dense factors: world-frame left-perturbation Jacobians,
// P_pose1 = -P_pose0, weight/num_inliers normalisation, 10*weight fallback without inliers.
// Second half of the file: cuda/match_geometry_factor_kernels.cpp (13 kernels there, one templated kernel here).
#include "sage_device.h"
#include "sage_internal.h"

namespace sage
{

struct ReprojParams
{
  const float *R10, *t10, *R0, *t0, *R1, *t1; // mapper: all six; tracker: R10/t10 = the relative pose
  const float *bias0, *basis0, *code0;        // mapper
  const int32_t *loc;                         // mapper
  const float *dpts0;                         // tracker: sampled depths [N]
  const float *homo, *matched;                // [N,3], [N,2]
  float scale0;
  SageCamera cam;
  float eps, loss_param, weight;
  int N;
  float *rows; // [2N][D+1]
  float *serr; // [N]
  float *sval; // [N]
};

// MODE 0: mapper factor (D = 13+CS), MODE 1: tracker (D = 6)
template <int CS, int MODE, bool JAC>
__device__ __forceinline__ void reproj_rows_body(const ReprojParams &p, int idx)
{
  constexpr int D = MODE == 0 ? 13 + CS : 6;
  const float fx = p.cam.fx, fy = p.cam.fy, cx = p.cam.cx, cy = p.cam.cy;
  const float hm[3] = {p.homo[3 * idx + 0], p.homo[3 * idx + 1], p.homo[3 * idx + 2]};
  float d0;
  int loc = 0;
  if (MODE == 0)
  {
    loc = p.loc[idx];
    float acc = p.bias0[loc]; // reprojection_factor_kernels.cpp:57-64
    for (int i = 0; i < CS; ++i)
      acc += p.basis0[(size_t)loc * CS + i] * p.code0[i];
    d0 = acc * p.scale0;
  }
  else
    d0 = p.dpts0[idx];
  float rh[3], X[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
  {
    rh[i] = p.R10[i * 3 + 0] * hm[0] + p.R10[i * 3 + 1] * hm[1] + p.R10[i * 3 + 2] * hm[2];
    X[i] = d0 * rh[i] + p.t10[i];
  }
  const bool pos = X[2] > p.eps; // :74
  const float px = (X[0] / X[2]) * fx + cx, py = (X[1] / X[2]) * fy + cy;
  const float sl = sqrtf(p.loss_param);
  const float diff[2] = {p.matched[2 * idx + 0] - px, p.matched[2 * idx + 1] - py};
  const float nx = fabsf(diff[0]) / sl, ny = fabsf(diff[1]) / sl;
  const float sw[2] = {pos ? sqrtf(1.0f / (p.loss_param * (1.0f + nx))) : 0.f,  // :83-84
                       pos ? sqrtf(1.0f / (p.loss_param * (1.0f + ny))) : 0.f};
  p.serr[idx] = pos ? 2.0f * (nx + ny - logf(1.0f + nx) - logf(1.0f + ny)) : 0.f; // :87-90
  p.sval[idx] = pos ? 1.f : 0.f;
  if (!JAC)
    return;
  const float inv_z = 1.0f / X[2];
  const float x_z = inv_z * X[0], y_z = inv_z * X[1];
  float *row0 = p.rows + ((size_t)idx * 2 + 0) * (D + 1), *row1 = p.rows + ((size_t)idx * 2 + 1) * (D + 1);
  if (MODE == 1)
  {
    const float J0[6] = {fx * inv_z, 0.f, -fx * x_z * inv_z, -fx * x_z * y_z, fx * (1.0f + x_z * x_z), -fx * y_z}; // :348
    const float J1[6] = {0.f, fy * inv_z, -fy * y_z * inv_z, -fy * (1.0f + y_z * y_z), fy * x_z * y_z, fy * x_z};
#pragma unroll
    for (int j = 0; j < 6; ++j)
    {
      row0[j] = sw[0] * J0[j];
      row1[j] = sw[1] * J1[j];
    }
  }
  else
  {
    const float Jpi[2][3] = {{fx * inv_z, 0.f, -fx * x_z * inv_z}, {0.f, fy * inv_z, -fy * y_z * inv_z}}; // :108-109
    float Xw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      Xw[i] = d0 * (p.R0[i * 3 + 0] * hm[0] + p.R0[i * 3 + 1] * hm[1] + p.R0[i * 3 + 2] * hm[2]) + p.t0[i];
    Pose p1;
#pragma unroll
    for (int i = 0; i < 9; ++i)
      p1.R[i] = p.R1[i];
    float dX[3][6];
    dX_dT0(p1, Xw, dX); // R1^T [I | -[Xw]x]  (:148-161); dX/dT1 = -dX/dT0 (:124-133)
#pragma unroll
    for (int j = 0; j < 6; ++j)
    {
      const float a0 = Jpi[0][0] * dX[0][j] + Jpi[0][1] * dX[1][j] + Jpi[0][2] * dX[2][j];
      const float a1 = Jpi[1][0] * dX[0][j] + Jpi[1][1] * dX[1][j] + Jpi[1][2] * dX[2][j];
      row0[j] = sw[0] * a0;
      row1[j] = sw[1] * a1;
      row0[6 + j] = sw[0] * (-a0);
      row1[6 + j] = sw[1] * (-a1);
    }
    const float jd0 = fx * (rh[0] * inv_z - X[0] * rh[2] * inv_z * inv_z); // :175-176
    const float jd1 = fy * (rh[1] * inv_z - X[1] * rh[2] * inv_z * inv_z);
    for (int i = 0; i < CS; ++i)
    {
      const float b = p.basis0[(size_t)loc * CS + i];
      row0[12 + i] = sw[0] * (jd0 * p.scale0 * b); // :182-183
      row1[12 + i] = sw[1] * (jd1 * p.scale0 * b);
    }
    row0[12 + CS] = sw[0] * (jd0 * d0 / p.scale0); // :186
    row1[12 + CS] = sw[1] * (jd1 * d0 / p.scale0);
  }
  row0[D] = sw[0] * diff[0]; // :188-189
  row1[D] = sw[1] * diff[1];
}

template <int CS, int MODE, bool JAC>
__global__ __launch_bounds__(256) void reproj_rows_kernel(const ReprojParams p)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < p.N)
    reproj_rows_body<CS, MODE, JAC>(p, idx);
}

// ---- one launch for the small systems (r04; tracker D = 6 / 7, loop closure D = 14): ONE workgroup writes the weighted
// rows of all keypoints into LDS and contracts them itself -- entry (a, j) of [AtA | Atb] by 256 / (D (D + 1)) threads over
// interleaved rows in double, folded in a fixed order -- instead of rows to memory + a D-workgroup reduce launch.  A tracker
// evaluation is launch-latency bound: one launch less per keypoint term and evaluation.
//   dynamic LDS: rows [RPP * N][D + 1] | serr [N] | sval [N] floats, then the partial sums
template <int D>
__device__ __forceinline__ void small_system_reduce(const float *rows, const float *serr, const float *sval, int N, int rpp,
                                                    float weight, float *AtA, float *Atb, float *stats, double *s_part)
{
  constexpr int NENT = D * (D + 1);            // (a, j): a < D, j <= D (j == D: the residual column -> Atb)
  constexpr int NGRP = 256 / NENT;             // row groups (D = 6: 6, D = 7: 4, D = 14: 1)
  static_assert(NGRP >= 1, "system too large for the one-workgroup contraction");
  const int tid = threadIdx.x;
  __shared__ double s_n[256], s_e[256];
  double n_in = 0.0, se = 0.0;
  for (int i = tid; i < N; i += 256)
  {
    n_in += (double)sval[i];
    se += (double)serr[i];
  }
  s_n[tid] = n_in;
  s_e[tid] = se;
  const int ent = tid % NENT, grp = tid / NENT;
  if (grp < NGRP)
  {
    const int a = ent / (D + 1), j = ent % (D + 1);
    double acc = 0.0;
    for (int k = grp; k < rpp * N; k += NGRP)
      acc += (double)rows[(size_t)k * (D + 1) + a] * (double)rows[(size_t)k * (D + 1) + j];
    s_part[grp * NENT + ent] = acc;
  }
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1)
  {
    if (tid < off)
    {
      s_n[tid] += s_n[tid + off];
      s_e[tid] += s_e[tid + off];
    }
    __syncthreads();
  }
  const double ninl = s_n[0];
  const double sc = ninl > 0.0 ? (double)weight / ninl : 0.0;
  if (tid < NENT)
  {
    double v = 0.0;
    for (int g = 0; g < NGRP; ++g)
      v += s_part[g * NENT + tid];
    v *= sc;
    const int a = tid / (D + 1), j = tid % (D + 1);
    if (j < D)
      AtA[(size_t)a * D + j] = (float)v;
    else
      Atb[a] = (float)v;
  }
  if (tid == 0)
  {
    stats[0] = ninl > 0.0 ? (float)(sc * s_e[0]) : weight * 10.0f;
    stats[1] = (float)ninl;
  }
}

template <int CS, int MODE, bool JAC>
__global__ __launch_bounds__(256) void reproj_small_kernel(ReprojParams p, float *AtA, float *Atb, float *stats)
{
  constexpr int D = MODE == 0 ? 13 + CS : 6;
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  p.rows = s_dyn;
  p.serr = s_dyn + (size_t)2 * p.N * (D + 1);
  p.sval = p.serr + p.N;
  for (int idx = threadIdx.x; idx < p.N; idx += 256)
    reproj_rows_body<CS, MODE, JAC>(p, idx);
  __syncthreads();
  if (JAC)
  {
    if constexpr (D * (D + 1) <= 256)
    {
      const size_t nf = (size_t)2 * p.N * (D + 1) + (size_t)2 * p.N;
      double *s_part = reinterpret_cast<double *>(s_dyn + nf + (nf & 1)); // (8-byte aligned)
      small_system_reduce<D>(p.rows, p.serr, p.sval, p.N, 2, p.weight, AtA, Atb, stats, s_part);
    }
  }
  else
  {
    __shared__ double s_n[256], s_e[256];
    const int tid = threadIdx.x;
    double n_in = 0.0, se = 0.0;
    for (int i = tid; i < p.N; i += 256)
    {
      n_in += (double)p.sval[i];
      se += (double)p.serr[i];
    }
    s_n[tid] = n_in;
    s_e[tid] = se;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1)
    {
      if (tid < off)
      {
        s_n[tid] += s_n[tid + off];
        s_e[tid] += s_e[tid + off];
      }
      __syncthreads();
    }
    if (tid == 0)
    {
      stats[0] = s_n[0] > 0.0 ? (float)((double)p.weight * s_e[0] / s_n[0]) : p.weight * 10.0f;
      stats[1] = (float)s_n[0];
    }
  }
}

// workgroup a: AtA[a][:] and Atb[a] = (weight/n) sum_rows J[row][a] * [J[row][:] | r[row]] in double; workgroup 0 also
// writes stats = {error, num_inliers}.  D+1 <= 128 columns x 2 row groups per workgroup (D <= 14+2*32 = 78).
__global__ __launch_bounds__(256) void reproj_reduce_kernel(const float *__restrict__ rows, const float *__restrict__ serr,
                                                            const float *__restrict__ sval, int N, int D, float weight,
                                                            float *__restrict__ AtA, float *__restrict__ Atb,
                                                            float *__restrict__ stats, int rows_per_point)
{
  __shared__ double s_acc[2][128];
  __shared__ double s_n[256], s_e[256];
  const int a = blockIdx.x, tid = threadIdx.x, j = tid & 127, grp = tid >> 7;
  double n_in = 0.0, se = 0.0;
  for (int i = tid; i < N; i += 256)
  {
    n_in += (double)sval[i];
    se += (double)serr[i];
  }
  s_n[tid] = n_in;
  s_e[tid] = se;
  double acc = 0.0;
  if (j <= D)
    for (int k = grp; k < rows_per_point * N; k += 2)
      acc += (double)rows[(size_t)k * (D + 1) + a] * (double)rows[(size_t)k * (D + 1) + j];
  s_acc[grp][j] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1)
  {
    if (tid < off)
    {
      s_n[tid] += s_n[tid + off];
      s_e[tid] += s_e[tid + off];
    }
    __syncthreads();
  }
  const double ninl = s_n[0];
  const double sc = ninl > 0.0 ? (double)weight / ninl : 0.0;
  if (tid <= D)
  {
    const double v = sc * (s_acc[0][tid] + s_acc[1][tid]);
    if (tid < D)
      AtA[(size_t)a * D + tid] = (float)v;
    else
      Atb[a] = (float)v;
  }
  if (a == 0 && tid == 0)
  {
    stats[0] = ninl > 0.0 ? (float)(sc * s_e[0]) : weight * 10.0f; // :507, :523
    stats[1] = (float)ninl;
  }
}

__global__ __launch_bounds__(256) void reproj_stats_kernel(const float *__restrict__ serr, const float *__restrict__ sval,
                                                           int N, float weight, float *__restrict__ stats)
{
  __shared__ double s_n[256], s_e[256];
  const int tid = threadIdx.x;
  double n_in = 0.0, se = 0.0;
  for (int i = tid; i < N; i += 256)
  {
    n_in += (double)sval[i];
    se += (double)serr[i];
  }
  s_n[tid] = n_in;
  s_e[tid] = se;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1)
  {
    if (tid < off)
    {
      s_n[tid] += s_n[tid + off];
      s_e[tid] += s_e[tid + off];
    }
    __syncthreads();
  }
  if (tid == 0)
  {
    stats[0] = s_n[0] > 0.0 ? (float)((double)weight * s_e[0] / s_n[0]) : weight * 10.0f; // :457-464
    stats[1] = (float)s_n[0];
  }
}

// scratch: rows 2N*(D+1) floats, then serr N, sval N
size_t reproj_scratch_floats(int N, int D) { return (size_t)2 * N * (D + 1) + (size_t)2 * N + 4; }

template <int CS, int MODE>
static hipError_t reproj_impl(hipStream_t s, ReprojParams p, bool jac, float *scratch, float *AtA, float *Atb, float *stats)
{
  constexpr int D = MODE == 0 ? 13 + CS : 6;
  const int N = p.N;
  p.rows = scratch;
  p.serr = scratch + (size_t)2 * N * (D + 1);
  p.sval = p.serr + N;
  const int grid = (N + 255) / 256;
  // small systems: one launch (rows in LDS, contracted by the same workgroup)
  const size_t small_lds = ((size_t)2 * N * (D + 1) + (size_t)2 * N + 2) * sizeof(float) + (size_t)256 * sizeof(double);
  if (D * (D + 1) <= 256 && N > 0 && small_lds <= 60 * 1024)
  {
    if (jac)
      hipLaunchKernelGGL((reproj_small_kernel<CS, MODE, true>), dim3(1), dim3(256), small_lds, s, p, AtA, Atb, stats);
    else
      hipLaunchKernelGGL((reproj_small_kernel<CS, MODE, false>), dim3(1), dim3(256), small_lds, s, p, AtA, Atb, stats);
    return hipGetLastError();
  }
  if (jac)
  {
    if (grid > 0)
      hipLaunchKernelGGL((reproj_rows_kernel<CS, MODE, true>), dim3(grid), dim3(256), 0, s, p);
    hipLaunchKernelGGL(reproj_reduce_kernel, dim3(D), dim3(256), 0, s, p.rows, p.serr, p.sval, N, D, p.weight, AtA, Atb,
                       stats, 2);
  }
  else
  {
    if (grid > 0)
      hipLaunchKernelGGL((reproj_rows_kernel<CS, MODE, false>), dim3(grid), dim3(256), 0, s, p);
    hipLaunchKernelGGL(reproj_stats_kernel, dim3(1), dim3(256), 0, s, p.serr, p.sval, N, p.weight, stats);
  }
  return hipGetLastError();
}

hipError_t launch_reproj(hipStream_t s, int CS, bool tracker, bool jac, const float *R10, const float *t10, const float *R0,
                         const float *t0, const float *R1, const float *t1, const float *bias0, const float *basis0,
                         const float *code0, const int32_t *loc, const float *dpts0, const float *homo,
                         const float *matched, float scale0, const SageCamera &cam, float eps, float loss_param,
                         float weight, int N, float *scratch, float *AtA, float *Atb, float *stats)
{
  ReprojParams p{};
  p.R10 = R10; p.t10 = t10; p.R0 = R0; p.t0 = t0; p.R1 = R1; p.t1 = t1;
  p.bias0 = bias0; p.basis0 = basis0; p.code0 = code0; p.loc = loc; p.dpts0 = dpts0; p.homo = homo; p.matched = matched;
  p.scale0 = scale0; p.cam = cam; p.eps = eps; p.loss_param = loss_param; p.weight = weight; p.N = N;
  if (tracker)
    return reproj_impl<16, 1>(s, p, jac, scratch, AtA, Atb, stats);
  if (CS == 32)
    return reproj_impl<32, 0>(s, p, jac, scratch, AtA, Atb, stats);
  if (CS == 16)
    return reproj_impl<16, 0>(s, p, jac, scratch, AtA, Atb, stats);
  return hipErrorInvalidValue;
}


// ------------------------------------------------------------------------------------------------
// match geometry: 3 residuals per matched keypoint, W (X1_matched - X0_in_1)
//   MODE 0 mapper  D = 14+2CS [pose0 pose1 code0 code1 scale0 scale1]  (match_geometry_factor_kernels.cpp:421-1041)
//   MODE 1 loop    D = 14     [pose0 pose1 scale0 scale1], unscaled depths handed over          (:296-419)
//   MODE 2 tracker D = 6      relative pose                                                      (:136-213)
//   MODE 3 tracker D = 7      relative pose + scale0                                             (:215-294)
//   loss 0 fair, 1 L2, 2 huber, 3 unbiased (mapper only)
// Normalisation: weight * mean over the N keypoints (hosts :1352-1858); every keypoint counts (sval = 1).
// ------------------------------------------------------------------------------------------------
struct MgParams
{
  const float *R10, *t10, *R0, *t0, *R1, *t1;
  const float *bias0, *bias1, *basis0, *basis1, *code0, *code1; // MODE 0
  const float *dpts0, *dpts1;                                   // MODE 1: unscaled; MODE 2/3: scaled
  const float *homo0, *homo1;
  const int32_t *loc0, *loc1;
  float scale0, scale1, loss_param, weight;
  int loss, N;
  float *rows, *serr, *sval;
};

template <int CS, int MODE, bool JAC>
__device__ __forceinline__ void mg_rows_body(const MgParams &p, int idx)
{
  constexpr int D = MODE == 0 ? 14 + 2 * CS : (MODE == 1 ? 14 : (MODE == 2 ? 6 : 7));
  const float h0[3] = {p.homo0[3 * idx + 0], p.homo0[3 * idx + 1], p.homo0[3 * idx + 2]};
  const float h1[3] = {p.homo1[3 * idx + 0], p.homo1[3 * idx + 1], p.homo1[3 * idx + 2]};
  const float ss = p.scale0 + p.scale1;
  float d0, d1;
  int l0 = 0, l1 = 0;
  if (MODE == 0)
  {
    l0 = p.loc0[idx];
    l1 = p.loc1[idx];
    float a0 = p.bias0[l0], a1 = p.bias1[l1]; // :601-616
    for (int i = 0; i < CS; ++i)
      a0 += p.basis0[(size_t)l0 * CS + i] * p.code0[i];
    for (int i = 0; i < CS; ++i)
      a1 += p.basis1[(size_t)l1 * CS + i] * p.code1[i];
    if (p.loss == 3) // :429-447
    {
      d0 = a0 * p.scale0 / ss;
      d1 = a1 * p.scale1 / ss;
    }
    else
    {
      d0 = a0 * p.scale0;
      d1 = a1 * p.scale1;
    }
  }
  else if (MODE == 1)
  {
    d0 = p.dpts0[idx] * p.scale0; // :303-304
    d1 = p.dpts1[idx] * p.scale1;
  }
  else
  {
    d0 = p.dpts0[idx];
    d1 = p.dpts1[idx];
  }
  float rh[3], X[3], diff[3], sw[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
  {
    rh[i] = p.R10[i * 3 + 0] * h0[0] + p.R10[i * 3 + 1] * h0[1] + p.R10[i * 3 + 2] * h0[2];
    X[i] = d0 * rh[i] + p.t10[i];
    diff[i] = d1 * h1[i] - X[i];
  }
  float err = 0.f;
  if (p.loss == 1) // L2 (:792-797)
  {
    sw[0] = sw[1] = sw[2] = 1.f;
    err = diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
  }
  else if (p.loss == 2) // huber (:935-962)
  {
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
      const float sq = diff[i] * diff[i];
      err += (sq <= p.loss_param) ? sq : (2.0f * sqrtf(p.loss_param * sq) - p.loss_param);
      sw[i] = fminf(1.0f, sqrtf(p.loss_param / sq));
    }
  }
  else // fair (:640-656), also the "unbiased" variant
  {
    const float sl = sqrtf(p.loss_param);
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
      const float n = fabsf(diff[i]) / sl;
      err += n - logf(1.0f + n);
      sw[i] = sqrtf(1.0f / (p.loss_param * (1.0f + n)));
    }
    err *= 2.0f;
  }
  p.serr[idx] = err;
  p.sval[idx] = 1.f;
  if (!JAC)
    return;
  float dX[3][6];
  if (MODE <= 1)
  {
    float Xw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      Xw[i] = d0 * (p.R0[i * 3 + 0] * h0[0] + p.R0[i * 3 + 1] * h0[1] + p.R0[i * 3 + 2] * h0[2]) + p.t0[i];
    Pose p1;
#pragma unroll
    for (int i = 0; i < 9; ++i)
      p1.R[i] = p.R1[i];
    dX_dT0(p1, Xw, dX); // R1^T [I | -[Xw]x] (:694-705); pose 1 gets the negative (:673-683)
  }
  else
  {
    const float E[3][6] = {{1, 0, 0, 0, X[2], -X[1]}, {0, 1, 0, -X[2], 0, X[0]}, {0, 0, 1, X[1], -X[0], 0}}; // :197-199
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j)
        dX[i][j] = E[i][j];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
  {
    float *row = p.rows + ((size_t)idx * 3 + i) * (D + 1);
#pragma unroll
    for (int j = 0; j < 6; ++j)
    {
      row[j] = sw[i] * dX[i][j];
      if (MODE <= 1)
        row[6 + j] = sw[i] * (-dX[i][j]);
    }
    if (MODE == 0)
    {
      const float *b0 = p.basis0 + (size_t)l0 * CS, *b1 = p.basis1 + (size_t)l1 * CS;
      if (p.loss == 3)
      {
        for (int j = 0; j < CS; ++j)
        {
          row[12 + j] = sw[i] * (rh[i] * b0[j] * p.scale0 / ss); // :560-563
          row[12 + CS + j] = sw[i] * (-h1[i] * b1[j] * p.scale1 / ss);
        }
        row[12 + 2 * CS] = sw[i] * (rh[i] * d0 * p.scale1 / (p.scale0 * ss) + h1[i] * d1 / ss); // :566-569
        row[13 + 2 * CS] = sw[i] * (-rh[i] * d0 / ss - h1[i] * d1 * p.scale0 / (p.scale1 * ss));
      }
      else
      {
        for (int j = 0; j < CS; ++j)
        {
          row[12 + j] = sw[i] * (rh[i] * p.scale0 * b0[j]); // :716-719
          row[12 + CS + j] = sw[i] * (-h1[i] * p.scale1 * b1[j]);
        }
        row[12 + 2 * CS] = sw[i] * (rh[i] * d0 / p.scale0); // :722-723
        row[13 + 2 * CS] = sw[i] * (-h1[i] * d1 / p.scale1);
      }
    }
    else if (MODE == 1)
    {
      row[12] = sw[i] * (rh[i] * p.dpts0[idx]); // :398-399
      row[13] = sw[i] * (-h1[i] * p.dpts1[idx]);
    }
    else if (MODE == 3)
      row[6] = sw[i] * (rh[i] * p.dpts0[idx] / p.scale0); // :278
    row[D] = sw[i] * diff[i];
  }
}

template <int CS, int MODE, bool JAC>
__global__ __launch_bounds__(256) void mg_rows_kernel(const MgParams p)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < p.N)
    mg_rows_body<CS, MODE, JAC>(p, idx);
}

// one launch for the small systems (see reproj_small_kernel)
template <int CS, int MODE, bool JAC>
__global__ __launch_bounds__(256) void mg_small_kernel(MgParams p, float *AtA, float *Atb, float *stats)
{
  constexpr int D = MODE == 0 ? 14 + 2 * CS : (MODE == 1 ? 14 : (MODE == 2 ? 6 : 7));
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  p.rows = s_dyn;
  p.serr = s_dyn + (size_t)3 * p.N * (D + 1);
  p.sval = p.serr + p.N;
  for (int idx = threadIdx.x; idx < p.N; idx += 256)
    mg_rows_body<CS, MODE, JAC>(p, idx);
  __syncthreads();
  if (JAC)
  {
    if constexpr (D * (D + 1) <= 256)
    {
      const size_t nf = (size_t)3 * p.N * (D + 1) + (size_t)2 * p.N;
      double *s_part = reinterpret_cast<double *>(s_dyn + nf + (nf & 1));
      small_system_reduce<D>(p.rows, p.serr, p.sval, p.N, 3, p.weight, AtA, Atb, stats, s_part);
    }
  }
  else
  {
    __shared__ double s_n[256], s_e[256];
    const int tid = threadIdx.x;
    double n_in = 0.0, se = 0.0;
    for (int i = tid; i < p.N; i += 256)
    {
      n_in += (double)p.sval[i];
      se += (double)p.serr[i];
    }
    s_n[tid] = n_in;
    s_e[tid] = se;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1)
    {
      if (tid < off)
      {
        s_n[tid] += s_n[tid + off];
        s_e[tid] += s_e[tid + off];
      }
      __syncthreads();
    }
    if (tid == 0)
    {
      stats[0] = s_n[0] > 0.0 ? (float)((double)p.weight * s_e[0] / s_n[0]) : p.weight * 10.0f;
      stats[1] = (float)s_n[0];
    }
  }
}

template <int CS, int MODE>
static hipError_t mg_impl(hipStream_t s, MgParams p, bool jac, float *scratch, float *AtA, float *Atb, float *stats)
{
  constexpr int D = MODE == 0 ? 14 + 2 * CS : (MODE == 1 ? 14 : (MODE == 2 ? 6 : 7));
  const int N = p.N;
  p.rows = scratch;
  p.serr = scratch + (size_t)3 * N * (D + 1);
  p.sval = p.serr + N;
  const int grid = (N + 255) / 256;
  const size_t small_floats = (size_t)3 * N * (D + 1) + (size_t)2 * N;
  const size_t small_lds = (small_floats + 2) * sizeof(float) + (size_t)256 * sizeof(double);
  if (D * (D + 1) <= 256 && N > 0 && small_lds <= 60 * 1024)
  {
    if (jac)
      hipLaunchKernelGGL((mg_small_kernel<CS, MODE, true>), dim3(1), dim3(256), small_lds, s, p, AtA, Atb, stats);
    else
      hipLaunchKernelGGL((mg_small_kernel<CS, MODE, false>), dim3(1), dim3(256), small_lds, s, p, AtA, Atb, stats);
    return hipGetLastError();
  }
  if (jac)
  {
    hipLaunchKernelGGL((mg_rows_kernel<CS, MODE, true>), dim3(grid), dim3(256), 0, s, p);
    hipLaunchKernelGGL(reproj_reduce_kernel, dim3(D), dim3(256), 0, s, p.rows, p.serr, p.sval, N, D, p.weight, AtA, Atb,
                       stats, 3);
  }
  else
  {
    hipLaunchKernelGGL((mg_rows_kernel<CS, MODE, false>), dim3(grid), dim3(256), 0, s, p);
    hipLaunchKernelGGL(reproj_stats_kernel, dim3(1), dim3(256), 0, s, p.serr, p.sval, N, p.weight, stats);
  }
  return hipGetLastError();
}

size_t mg_scratch_floats(int N, int D) { return (size_t)3 * N * (D + 1) + (size_t)2 * N + 4; }

hipError_t launch_match_geom(hipStream_t s, int mode, int loss, int CS, bool jac, const float *R10, const float *t10,
                             const float *R0, const float *t0, const float *R1, const float *t1, const float *bias0,
                             const float *bias1, const float *basis0, const float *basis1, const float *code0,
                             const float *code1, const float *dpts0, const float *dpts1, const float *homo0,
                             const float *homo1, const int32_t *loc0, const int32_t *loc1, float scale0, float scale1,
                             float loss_param, float weight, int N, float *scratch, float *AtA, float *Atb, float *stats)
{
  MgParams p{};
  p.R10 = R10; p.t10 = t10; p.R0 = R0; p.t0 = t0; p.R1 = R1; p.t1 = t1;
  p.bias0 = bias0; p.bias1 = bias1; p.basis0 = basis0; p.basis1 = basis1; p.code0 = code0; p.code1 = code1;
  p.dpts0 = dpts0; p.dpts1 = dpts1; p.homo0 = homo0; p.homo1 = homo1; p.loc0 = loc0; p.loc1 = loc1;
  p.scale0 = scale0; p.scale1 = scale1; p.loss_param = loss_param; p.weight = weight; p.loss = loss; p.N = N;
  switch (mode)
  {
  case 0:
    if (CS == 32)
      return mg_impl<32, 0>(s, p, jac, scratch, AtA, Atb, stats);
    if (CS == 16)
      return mg_impl<16, 0>(s, p, jac, scratch, AtA, Atb, stats);
    return hipErrorInvalidValue;
  case 1:
    return mg_impl<16, 1>(s, p, jac, scratch, AtA, Atb, stats);
  case 2:
    return mg_impl<16, 2>(s, p, jac, scratch, AtA, Atb, stats);
  case 3:
    return mg_impl<16, 3>(s, p, jac, scratch, AtA, Atb, stats);
  default:
    return hipErrorInvalidValue;
  }
}


// ------------------------------------------------------------------------------------------------
// f4 matching core: response-map argmax with cycle consistency (core/gtsam/match_geometry_factor.cpp:62-97,
// core/system/camera_tracker.cpp:608-633).  response[k][p] = -sum_c (q[c][k] - target[c][p])^2 is never materialised
// (the reference builds two K x H*W tensors): one workgroup owns KPB query descriptors (LDS), streams the target
// descriptor map once (coalesced over pixels, channel sum sequential in fp32 like the restated reference) and keeps a
// running (response, index) pair per query and lane; first index wins ties.
// ------------------------------------------------------------------------------------------------
constexpr int kMatchKPB = 8; // queries per workgroup: the target map is read K/8 times instead of K times

__global__ __launch_bounds__(256) void best_match_kernel(const float *__restrict__ query_map, const long long *__restrict__ query_loc,
                                                         const float *__restrict__ target_map, int K, int C, int HW,
                                                         long long *__restrict__ best_out)
{
  extern __shared__ float s_dyn[]; // [KPB][C] query descriptors, then the reduction scratch
  float *s_q = s_dyn;
  const int tid = threadIdx.x, k0 = blockIdx.x * kMatchKPB;
  const int nk = min(kMatchKPB, K - k0);
  for (int i = tid; i < kMatchKPB * C; i += 256)
  {
    const int kk = i / C, c = i - kk * C;
    s_q[i] = kk < nk ? query_map[(size_t)c * HW + query_loc[k0 + kk]] : 0.f;
  }
  __syncthreads();
  float best_r[kMatchKPB];
  int best_p[kMatchKPB];
#pragma unroll
  for (int kk = 0; kk < kMatchKPB; ++kk)
  {
    best_r[kk] = -INFINITY;
    best_p[kk] = 0x7fffffff;
  }
  for (int p = tid; p < HW; p += 256)
  {
    float acc[kMatchKPB];
#pragma unroll
    for (int kk = 0; kk < kMatchKPB; ++kk)
      acc[kk] = 0.f;
    for (int c = 0; c < C; ++c)
    {
      const float t = target_map[(size_t)c * HW + p];
#pragma unroll
      for (int kk = 0; kk < kMatchKPB; ++kk)
      {
        const float d = s_q[kk * C + c] - t;
        acc[kk] = __fadd_rn(acc[kk], __fmul_rn(d, d)); // no FMA contraction: the argmax must not depend on it
      }
    }
#pragma unroll
    for (int kk = 0; kk < kMatchKPB; ++kk)
    {
      const float r = -acc[kk];
      if (r > best_r[kk]) // ascending p per lane: '>' keeps the first maximum
      {
        best_r[kk] = r;
        best_p[kk] = p;
      }
    }
  }
  // cross-lane: larger response wins, equal responses -> smaller index
  float *s_r = s_dyn + kMatchKPB * C;
  int *s_p = reinterpret_cast<int *>(s_r + 256);
  for (int kk = 0; kk < nk; ++kk)
  {
    __syncthreads();
    s_r[tid] = best_r[kk];
    s_p[tid] = best_p[kk];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1)
    {
      if (tid < off)
      {
        const float r2 = s_r[tid + off];
        const int p2 = s_p[tid + off];
        if (r2 > s_r[tid] || (r2 == s_r[tid] && p2 < s_p[tid]))
        {
          s_r[tid] = r2;
          s_p[tid] = p2;
        }
      }
      __syncthreads();
    }
    if (tid == 0)
      best_out[k0 + kk] = s_p[0];
  }
}

__global__ void cycle_flags_kernel(const long long *__restrict__ kp_loc0, const long long *__restrict__ cyc_loc0, int K,
                                   int W, float thresh, int32_t *__restrict__ inlier, int *__restrict__ n_out)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K)
    return;
  const float dx = (float)(kp_loc0[k] % W) - (float)(cyc_loc0[k] % W);
  const float dy = (float)(kp_loc0[k] / W) - (float)(cyc_loc0[k] / W);
  const int in = (dx * dx + dy * dy) <= thresh * thresh; // match_geometry_factor.cpp:91-94
  inlier[k] = in;
  if (in)
    atomicAdd(n_out, 1); // integer count: order independent
}

hipError_t launch_cycle_match(hipStream_t s, const float *desc0, const float *desc1, const long long *kp_loc0, int K,
                              int C, int H, int W, float cyc_thresh, long long *raw_matched1, long long *cyc_matched0,
                              int32_t *inlier, int *n_inliers_dev)
{
  if (K <= 0)
    return hipSuccess;
  const int HW = H * W;
  const size_t shm = ((size_t)kMatchKPB * C + 512) * sizeof(float);
  const int grid = (K + kMatchKPB - 1) / kMatchKPB;
  hipLaunchKernelGGL(best_match_kernel, dim3(grid), dim3(256), shm, s, desc0, kp_loc0, desc1, K, C, HW, raw_matched1);
  hipLaunchKernelGGL(best_match_kernel, dim3(grid), dim3(256), shm, s, desc1, raw_matched1, desc0, K, C, HW, cyc_matched0);
  hipLaunchKernelGGL(cycle_flags_kernel, dim3((K + 255) / 256), dim3(256), 0, s, kp_loc0, cyc_matched0, K, W, cyc_thresh,
                     inlier, n_inliers_dev);
  return hipGetLastError();
}

} // namespace sage
