// solve_kernels.hip -- damped Gauss-Newton/LM step of the keyframe window on the device (gfx950).
//
// Replaces, for the batched window engine, the host side of the reference's optimisation step: the normal
// equations the factors hand to the solver (core/gtsam/photometric_factor.cpp:106-219 -> gtsam HessianFactor,
// ISAM2 update, core/mapping/mapper.cpp:118-156) and the LM damping policy of camera_tracker.cpp:1182
// (H + damp*diag(H)), followed by the manifold retraction of gtsam_traits.h:45-70.  The window's block normal
// equations (keyframe blocks of B = 7+CS rows, coupled along factor-graph links) stay in HBM end to end:
//
//   scatter   packed [K diag blocks | link blocks | gradient] (double)  ->  block-envelope storage of the lower
//             triangle (+ priors, LM damping, identity padding to BP rows), streamed into pinned host memory in the
//             order the factorisation consumes it, a ticket per block
//   factor    fixed-block Cholesky + substitutions on host cores (host_math.cpp: block_chol_solve_tr -- two halves and
//             a separator, each half as two pipelined stages)
//   retract   candidate variables = retract(current, delta), read zero-copy from the host's solution
//
// Everything is double: cond(H_damped) ~ 1e9 on the headline window.
//
// Why the factorisation is on the host: it is a dependency chain of K*B = 2.5 k pivots with ~27 M fused multiply-adds
// in total.  A one-workgroup device factorisation took 4.2 ms on MI355X (K = 64, B = 39: 1.1 us per pivot, 16 waves
// re-issue the panel update between two barriers; removed in r04), AVX-512 host cores do the same work in 0.13 ms.  So:
// scatter on the device -> block storage over PCIe (pinned, 3.2 MB, overlapped with the factorisation) -> host
// Cholesky -> retract on the device.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host_math.h"
#include "sage_device.h"
#include "sage_internal.h"

namespace sage
{

struct SolvePlan
{
  int K, B, Bp, nblk, nlinks;
  const int32_t *row_first; // [K] first block column of block row i
  const int32_t *row_off;   // [K] index of block (i, row_first[i]) in the block storage
  const int32_t *blk_row, *blk_col, *blk_src; // [nblk]; src = link index (bit 30: stored block is the link block
                                               // itself rather than its transpose) or -1
  const int32_t *perm, *pos; // elimination order: perm[position] = keyframe, pos[keyframe] = position
};

struct SolvePriors
{
  double code_w, scale_w, pose_w;
  float scale_init0;
  float pose_init0[12];
};

// gtsam_traits.h:78-89 : [t1 - R1 R0^T t0, log(R1 R0^T)]
__device__ inline void pose_local_dev(const float *origin, const float *other, double out[6])
{
  double Rr[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      Rr[i * 3 + j] = (double)other[i * 3 + 0] * origin[j * 3 + 0] + (double)other[i * 3 + 1] * origin[j * 3 + 1] +
                      (double)other[i * 3 + 2] * origin[j * 3 + 2];
  for (int i = 0; i < 3; ++i)
    out[i] = other[9 + i] - (Rr[i * 3 + 0] * origin[9] + Rr[i * 3 + 1] * origin[10] + Rr[i * 3 + 2] * origin[11]);
  const double tr = Rr[0] + Rr[4] + Rr[8];
  const double cs = fmin(1.0, fmax(-1.0, 0.5 * (tr - 1.0)));
  const double th = acos(cs);
  const double k = th < 1e-8 ? 0.5 : th / (2.0 * sin(th));
  out[3] = k * (Rr[7] - Rr[5]);
  out[4] = k * (Rr[2] - Rr[6]);
  out[5] = k * (Rr[3] - Rr[1]);
}

// ------------------------------------------------------------------------------------------------
// scatter: a workgroup per envelope block (device factorisation), or a few workgroups walking the blocks in the order
// the host factorisation consumes them and writing straight into pinned host memory (hybrid path): every block is
// followed by a ticket in `flags`, so the host starts on row 0 while the later rows are still crossing PCIe.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void solve_scatter_kernel(const SolvePlan P, const double *__restrict__ packed,
                                                            const float *__restrict__ vars0, int VS, int CS,
                                                            const SolvePriors pri, double damp, int transposed,
                                                            double *__restrict__ L, double *__restrict__ y,
                                                            const int32_t *__restrict__ order, unsigned *flags,
                                                            unsigned epoch, int first, int count, int deliver_fill)
{
  const int tid = threadIdx.x;
  const int B = P.B, Bp = P.Bp, BB = B * B;
  __shared__ double s_dadd[64], s_gadd[64];
  for (int it = first + blockIdx.x; it < first + count; it += gridDim.x) // [first, first+count) of the order list
  {
    const int b = order ? order[it] : it;
    const int i = P.blk_row[b], j = P.blk_col[b], srcf = P.blk_src[b];
    const int src = srcf < 0 ? -1 : (srcf & 0x3fffffff);
    const bool flip = srcf >= 0 && (srcf & 0x40000000);
    if (flags && src < 0 && i != j && !deliver_fill)
      continue; // structural fill-in: the host zeroes it at its first touch (BlockEnvelope::fill) -- nothing to deliver
    const int kf = P.perm[i]; // keyframe of this block row
    const double *diag = packed + (size_t)kf * BB;
    const double *lnk = packed + (size_t)P.K * BB + (size_t)(src < 0 ? 0 : src) * BB;
    const double *g = packed + (size_t)P.K * BB + (size_t)P.nlinks * BB + (size_t)kf * B;
    double *out = L + (size_t)b * Bp * Bp;
    if (i == j)
    {
      // diagonal priors (a9): code prior on every keyframe (zero prior mean), scale / pose priors on keyframe 0
      const float *var = vars0 + (size_t)kf * VS; // pose 12, scale, code CS
      if (tid < B)
      {
        double da = 0.0, ga = 0.0;
        if (tid >= 6 && tid < 6 + CS)
        {
          da = pri.code_w;
          ga = pri.code_w * (0.0 - (double)var[13 + tid - 6]);
        }
        if (kf == 0 && tid == 6 + CS && pri.scale_w > 0)
        {
          const double s = (double)var[12];
          da = pri.scale_w / (s * s);
          ga = pri.scale_w / s * (log((double)pri.scale_init0) - log(s));
        }
        if (kf == 0 && tid < 6 && pri.pose_w > 0)
        {
          double loc[6];
          pose_local_dev(var, pri.pose_init0, loc);
          da = pri.pose_w;
          ga = pri.pose_w * loc[tid];
        }
        s_dadd[tid] = da;
        s_gadd[tid] = ga;
      }
      __syncthreads();
    }
    // consecutive threads write consecutive doubles (the stores may be crossing PCIe); (r, c) = element of the block
    for (int o = tid; o < Bp * Bp; o += blockDim.x)
    {
      const int hi = o / Bp, lo = o - hi * Bp;
      const int r = transposed ? lo : hi, c = transposed ? hi : lo; // transposed: block stored [c][r]
      double v = 0.0;
      if (i == j)
      {
        if (r < B && c < B)
        {
          v = 0.5 * (diag[r * B + c] + diag[c * B + r]);
          if (r == c)
            v = (v + s_dadd[r]) * (1.0 + damp); // LM damping H + damp*diag(H) (camera_tracker.cpp:1182)
        }
        else if (r == c)
          v = 1.0 + damp; // identity padding: delta 0 on the padding rows
      }
      else if (src >= 0 && r < B && c < B)
        v = flip ? lnk[r * B + c] : lnk[c * B + r]; // packed link block is (a,b), a < b; this block is (row kf, col kf)
      out[o] = v;
    }
    if (i == j)
      for (int r = tid; r < Bp; r += blockDim.x)
        y[(size_t)i * Bp + r] = r < B ? g[r] + s_gadd[r] : 0.0;
    if (flags)
    {
      // the block (and its rhs rows) are visible to the host before the ticket is: the workgroup barrier orders every
      // lane's stores before lane 0's system-scope release (one cache write-back per block instead of one per wave)
      __syncthreads();
      if (tid == 0)
      {
        __threadfence_system();
        *reinterpret_cast<volatile unsigned *>(flags + b) = epoch;
      }
    }
    else
      __syncthreads(); // s_dadd / s_gadd are reused by the next block
  }
}

// ------------------------------------------------------------------------------------------------
// retract: candidate = current (+) delta   (gtsam_traits.h:45-70; tangent order [trans, rot], left update)
// ------------------------------------------------------------------------------------------------
__device__ inline void se3_exp_dev(const float *omega, const float *v, float *R, float *t) // mapping_utils.h:316-346
{
  float theta = sqrtf(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  float n[3] = {1.f, 0.f, 0.f};
  if (theta > 0)
  {
    n[0] = omega[0] / theta;
    n[1] = omega[1] / theta;
    n[2] = omega[2] / theta;
  }
  theta = fmaxf(theta, 1.0e-14f);
  const float s = sinf(theta), c = cosf(theta);
  const float Km[3][3] = {{0, -n[2], n[1]}, {n[2], 0, -n[0]}, {-n[1], n[0], 0}};
  float K2[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      K2[i][j] = Km[i][0] * Km[0][j] + Km[i][1] * Km[1][j] + Km[i][2] * Km[2][j];
  const float a = (1.0f - c) / theta, b = (theta - s) / theta;
  for (int i = 0; i < 3; ++i)
  {
    float acc = 0.f;
    for (int j = 0; j < 3; ++j)
    {
      const float id = (i == j) ? 1.f : 0.f;
      R[i * 3 + j] = id + s * Km[i][j] + (1.0f - c) * K2[i][j];
      acc += (id + a * Km[i][j] + b * K2[i][j]) * v[j];
    }
    t[i] = acc;
  }
}

// One workgroup.  Besides the candidate variables (device) it writes the host mirror -- candidate variables, delta and
// |delta|^2 -- straight into pinned host memory (h_*: device-visible), so no copy follows the solve.
// (1024 threads: in the hybrid path x lives in pinned host memory -- every read is a PCIe round trip, so the kernel's
// time is the number of sequential reads per thread)
__global__ __launch_bounds__(1024) void solve_retract_kernel(const double *__restrict__ x, int K, int B, int Bp, int CS,
                                                            int VS, const int32_t *__restrict__ pos,
                                                            const float *__restrict__ vars0,
                                                            float *__restrict__ vars1, float *__restrict__ h_vars,
                                                            double *__restrict__ h_delta, double *__restrict__ h_tail,
                                                            const volatile unsigned *go, unsigned epoch)
{
  const int tid = threadIdx.x;
  // The kernel is enqueued BEFORE the host factorises (right behind the scatter) and waits here for the host's word in
  // pinned memory: the solution is there (go == epoch), or there is none (bit 31 set: non-positive pivot / abort).  The
  // launch latency of a kernel issued into an idle queue (tens of microseconds on the step's critical path) is hidden
  // behind the factorisation.  A watchdog (2 s of the 100 MHz wall clock) ends the wait if the host never answers.
  __shared__ int s_go;
  if (tid == 0)
  {
    const unsigned long long t0 = wall_clock64();
    unsigned v;
    bool timed_out = false;
    for (;;)
    {
      v = *go;
      // the word carries the epoch of the LAST solve the host answered: mine -> go / abort (bit 31); a NEWER one (the host
      // gave up on this solve -- its own word was overwritten before this kernel ran) -> abort; an older one -> keep waiting
      const unsigned d = ((v & 0x7fffffffu) - epoch) & 0x7fffffffu;
      if (d == 0u)
        break;
      if (d < 0x40000000u && (v & 0x7fffffffu) != 0u)
      {
        v = 0x80000000u;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > 200000000ull)
      {
        v = 0x80000000u;
        timed_out = true;
        break;
      }
    }
    __threadfence_system();
    s_go = (v & 0x80000000u) ? 0 : 1;
    // status word of the host mirror (solver_host_status): 0 candidate written, 1 no candidate (the host aborted the solve:
    // non-positive pivot / error), 2 the host's word never came -- the caller then treats the evaluation as failed instead
    // of reading a stale candidate
    *reinterpret_cast<volatile int *>(h_tail + 1) = timed_out ? 2 : (s_go ? 0 : 1);
  }
  __syncthreads();
  if (!s_go)
    return;
  double nrm = 0.0;
  for (int idx = tid; idx < K * B; idx += blockDim.x)
  {
    const int k = idx / B, r = idx - k * B;
    const double d = x[(size_t)pos[k] * Bp + r]; // x is in elimination order
    h_delta[idx] = d;
    nrm += d * d;
    const float *v0 = vars0 + (size_t)k * VS;
    if (r >= 6)
    {
      const int slot = r < 6 + CS ? 13 + r - 6 : 12; // code entries, then the scale
      const float v = v0[slot] + (float)d;
      vars1[(size_t)k * VS + slot] = v;
      h_vars[(size_t)k * VS + slot] = v;
    }
  }
  for (int k = tid; k < K; k += blockDim.x)
  {
    const double *xk = x + (size_t)pos[k] * Bp;
    const float *v0 = vars0 + (size_t)k * VS;
    float d6[6], dR[9], dt[3], o[12];
    for (int i = 0; i < 6; ++i)
      d6[i] = (float)xk[i];
    se3_exp_dev(d6 + 3, d6, dR, dt);
    for (int i = 0; i < 3; ++i)
    {
      for (int j = 0; j < 3; ++j)
        o[i * 3 + j] = dR[i * 3 + 0] * v0[0 * 3 + j] + dR[i * 3 + 1] * v0[1 * 3 + j] + dR[i * 3 + 2] * v0[2 * 3 + j];
      o[9 + i] = dR[i * 3 + 0] * v0[9] + dR[i * 3 + 1] * v0[10] + dR[i * 3 + 2] * v0[11] + dt[i];
    }
    for (int i = 0; i < 12; ++i)
    {
      vars1[(size_t)k * VS + i] = o[i];
      h_vars[(size_t)k * VS + i] = o[i];
    }
  }
  __shared__ double s_n[16];
  for (int off = 32; off > 0; off >>= 1)
    nrm += __shfl_down(nrm, off);
  if ((tid & 63) == 0)
    s_n[tid >> 6] = nrm;
  __syncthreads();
  if (tid == 0)
  {
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) // fixed order
      tot += s_n[w];
    h_tail[0] = tot;
  }
}

// ------------------------------------------------------------------------------------------------
// host side: plan + launcher
// ------------------------------------------------------------------------------------------------
struct DeviceSolver
{
  int K = 0, B = 0, Bp = 0, nblk = 0, nlinks = 0;
  void *d_int = nullptr;   // all int tables in one allocation
  void *h_pinned = nullptr; // [K*VS floats | K*B doubles | tail double | status int | go word of the pre-launched retract]
  size_t h_go_off = 0;
  unsigned go_epoch = 0;
  size_t h_vars_off = 0, h_delta_off = 0, h_tail_off = 0, h_status_off = 0, h_bytes = 0;
  SolvePlan plan{};
  int VS = 0;
  void *h_T = nullptr, *h_y = nullptr;       // pinned, one allocation: block storage, then the right-hand side
  std::vector<double> h_X;                   // inverses of the diagonal factors
  std::vector<int32_t> h_row_first, h_row_off, h_a_first, h_a_cnt, h_a_off, h_col_ptr, h_col_rows;
  int n1 = 0, n2 = 0; // two independent leading row ranges [0,n1) and [n1,n1+n2) of the elimination order (0: none)
  // hybrid path: the scatter kernel writes blocks + rhs straight into h_T / h_y in consumption order and posts a
  // ticket (the epoch of this solve) per block in h_flags
  const int32_t *d_order = nullptr;
  unsigned *h_flags = nullptr;
  unsigned epoch = 0;
  int scatter_wgs = 32;
  std::vector<int32_t> h_pos, h_perm; // elimination order (host copies)
  std::vector<uint8_t> h_fill;        // per block: structural fill-in (not delivered by the scatter kernel: BlockEnvelope::fill)
  std::vector<int32_t> h_pair_off;    // order-list offset of "pair row" t (row t of the first half + row t of the second);
                                      // entry T = start of the separator rows, entry T+1 = nblk
};

static int solver_bp(int B) { return (B + 7) / 8 * 8; }

int solver_create(DeviceSolver **out, int K, int B, int VS, const std::vector<std::pair<int, int>> &links,
                  hipStream_t stream, bool allow_split)
{
  *out = nullptr;
  const int Bp = solver_bp(B);
  if ((Bp != 40 && Bp != 24) || K < 1)
    return SAGE_E_UNSUPPORTED;
  BlockPlan bp;
  {
    const int rcp = plan_blocks(K, links, allow_split && !sage::env_flag("SAGE_SOLVE_NO_SPLIT"), bp);
    if (rcp != SAGE_OK)
      return rcp;
  }
  const int n1 = bp.n1, n2 = bp.n2, nblk = bp.nblk;
  const std::vector<int32_t> &perm = bp.perm, &pos = bp.pos, &row_first = bp.row_first, &row_off = bp.row_off,
                             &a_first = bp.a_first, &a_cnt = bp.a_cnt, &a_off = bp.a_off, &blk_row = bp.blk_row,
                             &blk_col = bp.blk_col, &blk_src = bp.blk_src;
  DeviceSolver *S = new DeviceSolver;
  S->K = K; S->B = B; S->Bp = Bp; S->nblk = nblk; S->nlinks = (int)links.size(); S->VS = VS;
  // one allocation for the int tables
  std::vector<int32_t> all;
  auto put = [&](const std::vector<int32_t> &v) {
    const size_t off = all.size();
    all.insert(all.end(), v.begin(), v.end());
    while (all.size() % 4)
      all.push_back(0);
    return off;
  };
  const size_t o_rf = put(row_first), o_ro = put(row_off), o_br = put(blk_row), o_bc = put(blk_col),
               o_bs = put(blk_src), o_pm = put(perm), o_ps = put(pos);
  // consumption order of the host factorisation: the two halves row by row side by side, the separator last
  std::vector<int32_t> order, pair_off;
  {
    auto push_row = [&](int i) {
      for (int q = 0; q < a_cnt[i]; ++q)
        order.push_back(a_off[i] + q);
      for (int q = 0; q <= i - row_first[i]; ++q)
        order.push_back(row_off[i] + q);
    };
    if (n1 > 0)
    {
      for (int t = 0; t < std::max(n1, n2); ++t)
      {
        pair_off.push_back((int32_t)order.size());
        if (t < n1)
          push_row(t);
        if (t < n2)
          push_row(n1 + t);
      }
      pair_off.push_back((int32_t)order.size());
      for (int i = n1 + n2; i < K; ++i)
        push_row(i);
      pair_off.push_back((int32_t)order.size());
    }
    else
      for (int i = 0; i < K; ++i)
        push_row(i);
  }
  const size_t o_ord = put(order);
  auto fail = [&](int rc) {
    solver_destroy(S);
    return rc;
  };
  if (hipMalloc(&S->d_int, all.size() * sizeof(int32_t)) != hipSuccess)
    return fail((int)hipErrorOutOfMemory);
  if (hipMemcpyAsync(S->d_int, all.data(), all.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream) != hipSuccess)
    return fail((int)hipErrorUnknown);
  if (hipStreamSynchronize(stream) != hipSuccess)
    return fail((int)hipErrorUnknown);
  const size_t ty_doubles = (size_t)nblk * Bp * Bp + (size_t)K * Bp;
  S->h_row_first = row_first;
  S->h_row_off = row_off;
  S->h_a_first = a_first;
  S->h_a_cnt = a_cnt;
  S->h_a_off = a_off;
  S->h_col_ptr = bp.col_ptr;
  S->h_col_rows = bp.col_rows;
  {
    if (hipHostMalloc(&S->h_T, ty_doubles * sizeof(double) + (size_t)nblk * sizeof(unsigned), hipHostMallocDefault) !=
        hipSuccess)
      return fail((int)hipErrorOutOfMemory);
    S->h_y = reinterpret_cast<double *>(S->h_T) + (size_t)nblk * Bp * Bp;
    S->h_flags = reinterpret_cast<unsigned *>(reinterpret_cast<double *>(S->h_T) + ty_doubles);
    std::memset(S->h_flags, 0, (size_t)nblk * sizeof(unsigned));
    S->h_X.assign((size_t)K * Bp * Bp, 0.0);
  }
  S->h_vars_off = 0;
  S->h_delta_off = ((size_t)K * VS * sizeof(float) + 15) / 16 * 16;
  S->h_tail_off = S->h_delta_off + (size_t)K * B * sizeof(double);
  S->h_go_off = S->h_tail_off + 2 * sizeof(double);
  S->h_bytes = S->h_go_off + 2 * sizeof(double);
  if (hipHostMalloc(&S->h_pinned, S->h_bytes, hipHostMallocDefault) != hipSuccess)
    return fail((int)hipErrorOutOfMemory);
  const int32_t *base = reinterpret_cast<const int32_t *>(S->d_int);
  SolvePlan &P = S->plan;
  P.K = K; P.B = B; P.Bp = Bp; P.nblk = nblk; P.nlinks = (int)links.size();
  P.row_first = base + o_rf; P.row_off = base + o_ro;
  P.blk_row = base + o_br; P.blk_col = base + o_bc; P.blk_src = base + o_bs; P.perm = base + o_pm; P.pos = base + o_ps;
  S->n1 = n1;
  S->n2 = n2;
  S->d_order = base + o_ord;
  S->h_pos = pos;
  S->h_perm = perm;
  S->h_fill.assign((size_t)nblk, 0);
  for (int b = 0; b < nblk; ++b)
    S->h_fill[b] = (blk_src[b] < 0 && blk_row[b] != blk_col[b]) ? 1 : 0;
  S->h_pair_off = pair_off;
  if (const char *e = getenv("SAGE_SCATTER_WGS")) // (diagnostic: workgroups of the scatter kernel)
    S->scatter_wgs = std::max(1, atoi(e));
  std::memset(S->h_pinned, 0, S->h_bytes);
  *out = S;
  return SAGE_OK;
}

void solver_destroy(DeviceSolver *S)
{
  if (!S)
    return;
  void *bufs[] = {S->d_int};
  for (void *p : bufs)
    if (p)
      (void)hipFree(p);
  void *hbufs[] = {S->h_pinned, S->h_T};
  for (void *p : hbufs)
    if (p)
      (void)hipHostFree(p);
  delete S;
}

// enqueue scatter and retract, factorise on the host in between (the retract waits on the device for the host's word);
// the candidate's host mirror is valid once the stream has drained (or a later kernel's ticket has been seen).
int solver_run(DeviceSolver *S, hipStream_t stream, const double *packed_dev, const float *vars0, float *vars1,
               int CS, double damp, double code_w, double scale_w, double pose_w, float scale_init0,
               const float *pose_init0)
{
  SolvePriors pri{};
  pri.code_w = code_w; pri.scale_w = scale_w; pri.pose_w = pose_w; pri.scale_init0 = scale_init0;
  for (int i = 0; i < 12; ++i)
    pri.pose_init0[i] = pose_init0[i];
  {
    // hybrid: the dependency chain of the factorisation runs on host cores, everything around it stays on the device.
    // The scatter kernel streams the blocks into pinned host memory in the order the factorisation consumes them and
    // tickets each one, so the host works on row 0 while the rest is still crossing PCIe (no D2H copy, no stream sync).
    static const bool dbgt = sage::env_flag("SAGE_DEBUG_TIMING");
    hipError_t eh;
    bool no_lookahead = false;
    if (S->n1 > 0)
    {
      // the helper core (and, for loop-closure plans with long separator rows, the worker pool) wakes up while this
      // thread waits for the device
      BlockEnvelope pe;
      pe.K = S->K; pe.Bp = S->Bp;
      pe.row_first = S->h_row_first.data(); pe.row_off = S->h_row_off.data();
      pe.a_first = S->h_a_first.data(); pe.a_cnt = S->h_a_cnt.data(); pe.a_off = S->h_a_off.data();
      pe.n1 = S->n1; pe.n2 = S->n2;
      no_lookahead = block_chol_arm(block_plan_has_arrow_rows(pe), block_plan_long_arrow_chains(pe));
    }
    S->epoch += 1;
    if (S->epoch == 0) // wrapped: 0 is the "never written" value
      S->epoch = 1;
    hipLaunchKernelGGL(solve_scatter_kernel, dim3(std::min(S->scatter_wgs, S->nblk)), dim3(256), 0, stream, S->plan,
                       packed_dev, vars0, S->VS, CS, pri, damp, 1, reinterpret_cast<double *>(S->h_T),
                       reinterpret_cast<double *>(S->h_y), S->d_order, S->h_flags, S->epoch, 0, S->nblk,
                       sage::env_flag("SAGE_SCATTER_FILL") ? 1 : 0); // (diagnostic: ship the zero fill blocks as r04 did)
    if ((eh = hipGetLastError()) != hipSuccess)
      return (int)eh;
    // the retract right behind the scatter: it waits on the device for this thread's word (solve_retract_kernel)
    char *hp = reinterpret_cast<char *>(S->h_pinned);
    volatile unsigned *go = reinterpret_cast<volatile unsigned *>(hp + S->h_go_off);
    S->go_epoch = (S->go_epoch + 1) & 0x7fffffffu;
    if (S->go_epoch == 0)
      S->go_epoch = 1;
    hipLaunchKernelGGL(solve_retract_kernel, dim3(1), dim3(1024), 0, stream,
                       reinterpret_cast<const double *>(S->h_y), S->K, S->B, S->Bp, CS, S->VS, S->plan.pos, vars0, vars1,
                       reinterpret_cast<float *>(hp + S->h_vars_off), reinterpret_cast<double *>(hp + S->h_delta_off),
                       reinterpret_cast<double *>(hp + S->h_tail_off), go, S->go_epoch);
    if ((eh = hipGetLastError()) != hipSuccess)
      return (int)eh;
    struct GoGuard // whatever happens below, the waiting kernel gets its word
    {
      volatile unsigned *go;
      unsigned word;
      ~GoGuard()
      {
        std::atomic_thread_fence(std::memory_order_release);
        *go = word;
      }
    } guard{go, S->go_epoch | 0x80000000u};
    const auto t1 = std::chrono::steady_clock::now();
    BlockEnvelope env;
    env.K = S->K; env.Bp = S->Bp;
    env.row_first = S->h_row_first.data(); env.row_off = S->h_row_off.data();
    env.a_first = S->h_a_first.data(); env.a_cnt = S->h_a_cnt.data(); env.a_off = S->h_a_off.data();
    env.col_ptr = S->h_col_ptr.data(); env.col_rows = S->h_col_rows.data();
    env.n1 = S->n1; env.n2 = S->n2;
    env.ready = S->h_flags; env.epoch = S->epoch;
    env.fill = S->h_fill.data();
    env.no_lookahead = no_lookahead;
    const int bad = block_chol_solve_tr(env, reinterpret_cast<double *>(S->h_T), S->h_X.data(),
                                        reinterpret_cast<double *>(S->h_y));
    const auto t2 = std::chrono::steady_clock::now();
    if (dbgt)
      fprintf(stderr, "[sage hybrid solve] wait for the system + host cholesky %.3f ms\n",
              std::chrono::duration<double, std::milli>(t2 - t1).count());
    if (bad == -2)
      return SAGE_E_STATE; // the device never delivered a block (see block_chol_solve_tr)
    if (bad)
      return SAGE_E_NOT_PSD;
    guard.word = S->go_epoch; // the solution is in h_y: go
    (void)eh;
  }
  return SAGE_OK;
}

const float *solver_host_vars(const DeviceSolver *S) { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(S->h_pinned) + S->h_vars_off); }
const double *solver_host_delta(const DeviceSolver *S) { return reinterpret_cast<const double *>(reinterpret_cast<const char *>(S->h_pinned) + S->h_delta_off); }
double solver_host_step_norm2(const DeviceSolver *S) { return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(S->h_pinned) + S->h_tail_off); }
int solver_host_status(const DeviceSolver *S) { return *reinterpret_cast<const int *>(reinterpret_cast<const char *>(S->h_pinned) + S->h_tail_off + sizeof(double)); }

} // namespace sage
