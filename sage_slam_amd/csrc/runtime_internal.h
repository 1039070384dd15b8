// runtime_internal.h -- shared by the host-runtime translation units behind the C ABI (include/sage_ba.h):
//   operators.hip       workspaces, the per-edge operator API (df::*_calculate mirrors), the producer entry points
//   tracker.hip         tracker wiring of the LM callbacks (sage_track_frame)
//   window.hip          the batched window engine: tables, work lists, linearize / error / solve / LM iteration
//   window_dist.hip     sharded windows: NUMA placement, all-reduce hook, native RCCL binding
//   window_factors.hip  f2: per-Values factor cache behind the gtsam adapter (prepass, factor blocks, NearestPsd)
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <sched.h>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <dlfcn.h>
// RCCL: types only -- the library is bound at run time with dlopen (sage_rccl_*), hosts without it never load it, and a
// build host without the RCCL headers still compiles (the handful of types the binding needs are declared here then)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C"
{
  typedef struct ncclComm *ncclComm_t;
  typedef struct
  {
    char internal[128];
  } ncclUniqueId;
  typedef enum { ncclSuccess = 0 } ncclResult_t;
  typedef enum { ncclSum = 0 } ncclRedOp_t;
  typedef enum { ncclDouble = 8 } ncclDataType_t; // nccl.h: ncclFloat64 = ncclDouble = 8
}
#endif

#include "host_math.h"
#include "sage_ba.h"
#include "sage_internal.h"

using namespace sage;

#define SAGE_HIP(expr)                \
  do                                  \
  {                                   \
    hipError_t _e = (expr);           \
    if (_e != hipSuccess)             \
      return (int)_e;                 \
  } while (0)

namespace sage_rt
{

struct DevBuf
{
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes)
  {
    if (bytes <= cap)
      return 0;
    if (p)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
      return (int)e;
    cap = bytes;
    return 0;
  }
  void release()
  {
    if (p)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T *as() const { return reinterpret_cast<T *>(p); }
};

inline int pick_tiles_per_block(long long total_tiles)
{
  // keep >= ~4 workgroups per CU in flight while amortising the partial write (one per workgroup)
  if (total_tiles >= 8192)
    return 4;
  if (total_tiles >= 4096)
    return 2;
  return 1;
}

// host-built work list for a set of edges with per-edge pixel counts
struct WorkList
{
  std::vector<WorkItem> work;
  std::vector<int32_t> edge_first, edge_tiles; // per edge: first work item, number of work items
  std::vector<int32_t> rec_first, rec_count;   // per edge: first partial record, number of partial records
  int tiles_per_block = 1;
  int flush = 1; // sub-tiles per partial record (== tiles_per_block unless the photometric linearize asks for less)
  int n_records = 0;
  // `order`: optional sequence of the edges (a permutation of 0..N.size()-1) the work items are laid out in; every
  // edge's items stay contiguous
  void build(const std::vector<int> &N, int tpb_override = 0, const std::vector<int> *order = nullptr, int flush_ = 0)
  {
    long long total = 0;
    for (int n : N)
      total += (n + kTile - 1) / kTile;
    tiles_per_block = tpb_override > 0 ? tpb_override : pick_tiles_per_block(total);
    flush = (flush_ > 0 && flush_ < tiles_per_block && tiles_per_block % flush_ == 0) ? flush_ : tiles_per_block;
    work.clear();
    edge_first.assign(N.size(), 0);
    edge_tiles.assign(N.size(), 0);
    rec_first.assign(N.size(), 0);
    rec_count.assign(N.size(), 0);
    n_records = 0;
    for (size_t i = 0; i < N.size(); ++i)
    {
      const size_t e = order ? (size_t)(*order)[i] : i;
      const int tiles = (N[e] + kTile - 1) / kTile;
      edge_first[e] = (int32_t)work.size();
      for (int t = 0; t < tiles; t += tiles_per_block)
        work.push_back(WorkItem{(int32_t)e, t});
      edge_tiles[e] = (int32_t)work.size() - edge_first[e];
      rec_first[e] = n_records;
      rec_count[e] = (tiles + flush - 1) / flush;
      n_records += rec_count[e];
    }
  }
};

} // namespace sage_rt
using namespace sage_rt;

// =====================================================================================================
// workspace
// =====================================================================================================
struct SageWorkspace
{
  hipStream_t stream = nullptr;
  DevBuf work, edge_first, edge_tiles, partials, misc, dpt0;
  float *host_stats = nullptr; // pinned, 16 floats: [0, 4) the operator's {error, inliers}, written by its kernels; [8] the ticket
  unsigned ticket_epoch = 0;
  int cached_N = -1;
  int n_work = 0;
  int tiles_per_block = 1;
  // tracker wiring (sage_track_frame): one evaluation = several operator launches that leave their statistics on the
  // device (defer_fetch: no D2H + stream synchronise per operator; stats_ptr: where this operator's {error, inliers} go),
  // then ONE copy of everything into pinned memory and one synchronise.  Buffers persist across frames.
  bool defer_fetch = false;
  float *stats_ptr = nullptr;
  DevBuf trk_dpts, trk_kp_dpts;      // dof 7: depths scaled for the evaluation
  float *trk_host = nullptr;         // pinned: [pose 12 | pad 4 | photo AtA 49 Atb 7 | keypoint AtA 49 Atb 7 | stats 2 + 2]
};

// where an operator's kernels put {error, inliers}: the tracker's evaluation buffer, or the workspace's pinned mirror (the
// kernels write it over PCIe themselves: no device-to-host copy afterwards)
static inline float *ws_stats(SageWorkspace *ws) { return ws->stats_ptr ? ws->stats_ptr : ws->host_stats; }
// enqueue a one-lane kernel that posts a ticket behind everything in the workspace's stream and spin until it shows up
// (instead of a blocking hipStreamSynchronize: operators.hip)
int ws_ticket_wait(SageWorkspace *ws);


// instantiated (CS, FS) combinations of the factor kernels
static inline bool supported(int CS, int FS)
{
  return (CS == 16 || CS == 32) && (FS == 16 || FS == 32);
}

// ---- window engine ----
namespace sage
{
struct AdjEntry // one (edge, role) incidence of a keyframe
{
  int32_t type; // 0 photo, 1 geo
  int32_t edge; // local edge index
  int32_t role; // 0: keyframe is the edge's source ("0"), 1: destination ("1")
};

struct LinkEdges // local edge indices of a link, -1 if the link is not owned by this rank
{
  int32_t e_ab, e_ba; // same indices for photo and geo tables
};

struct AssembleParams
{
  const float *AtA_p, *Atb_p, *stats_p; // photo per-edge results
  const float *AtA_g, *Atb_g, *stats_g;
  const double *wide_p, *wide_g; // optional: per-edge [D*D + D] results before their fp32 rounding (EdgeOut::wide)
  const int32_t *adj_start; // [K+1]
  const AdjEntry *adj;
  const LinkEdges *links; // [nlinks]
  double *packed;
  double *tail_mirror; // pinned host copy of the 4-double tail (single-rank windows), or null
  int K, nlinks, CS, n_edges_p, n_edges_g;
  int split;               // > 1: every output block is shared by `split` consecutive workgroups (small workgroups)
  const int32_t *blocks;   // optional: the output blocks to assemble (ids 0..K-1 keyframes, K..K+nlinks-1 links, K+nlinks tail)
};

struct ErrorTotalsSide
{
  const int32_t *edge_first, *edge_tiles;
  const float *partials; // [n_work][2]
  float *stats;          // [n_edges][2]
  float fallback, scale;
  int n_edges;           // 0: factor type unused
  int stride, err_off, cnt_off; // record layout: floats per workgroup record, slots of the error sum / the inlier count
};

} // namespace sage

struct SageWindow
{
  SageWindowConfig cfg;
  hipStream_t stream = nullptr;
  bool finalized = false;
  int rank = 0, world = 1;
  int K = 0, B = 0, VS = 0; // VS: floats per keyframe in the device variable array
  std::vector<SageKeyframeView> views;
  // host variables: [set][kf] ; set 0 = current, 1 = candidate
  std::vector<float> pose[2], code[2], scale[2];
  std::vector<float> link_geo_loss; // per link: the geometric factors' Cauchy parameter, 0 = cfg.geo_loss_param
  std::vector<float> code_init, scale_init, pose_init;
  std::vector<float> code_added; // codes as added (code_init is the zero prior mean)
  std::vector<std::pair<int, int>> links; // (a, b) with a < b
  std::vector<int> local_links;           // indices into links (links with at least one local directed edge)
  std::vector<int> local_edges;           // this rank's directed edges, global ids 2 * link + direction, ascending (local edge
                                          // index = position in this list; a single-rank window: the identity)
  int n_edges = 0;                        // local directed edges per factor type (= 2 * local links)
  // device
  DevBuf vars[2];                       // [K][VS]: pose 12, scale 1, code CS
  DevBuf wide_p, wide_g;                // per-edge results before their fp32 rounding (EdgeOut::wide)
  DevBuf sorted_loc, sorted_homo;       // raster-ordered copies of the keyframes' sampled locations
  std::vector<std::pair<const int64_t *, const float *>> user_samples; // the caller's arrays
  std::vector<int> user_n; // ... and their lengths (views[k].N becomes the tile-padded slot count for keyframes relaid with holes)
  DevBuf dpt, dgrad, depth_items[2];    // per-keyframe depth maps of the set being evaluated
  int n_depth = 0;                      // keyframes this rank's edges touch (= entries of depth_items)
  int dpt_set = -1;                     // variable set the depth maps currently hold (-1: none) ...
  bool dgrad_valid = false;             // ... and whether their gradients are up to date as well
  DeviceSolver *last_solver = nullptr;   // the solver whose pinned mirror holds the pending candidate
  SageAllReduceFn allreduce = nullptr;  // sharded windows: caller-provided sum all-reduce (see sage_ba.h)
  void *allreduce_user = nullptr;
  void *rccl_hook = nullptr;            // sage_window_use_rccl: owned {comm, stream} record behind `allreduce`
  // sharded windows, domain-decomposed solve (shard_solve.cpp): the all-reduced payload is the separator system
  SageShardPlan *shard = nullptr;
  DevBuf sepbuf;                        // device copy of the separator buffer (what the collective sums)
  std::vector<double> h_sep;
  double *h_err = nullptr;              // pinned [16]: {linearize tail[4], error pass totals[4], tickets of error_totals_kernel[4],
                                        // tickets of mirror_totals_kernel[4]} written by the kernels
  uint64_t mirror_epoch = 0;            // ticket value of the last mirror_totals_kernel (its own slots: h_err[12..15])
  // development aid (sage_window_emulate_peers): after every all-reduce the contribution of the ranks that are not there
  // is added from a caller-provided table of packed systems (one per LM iterate since the last reset)
  const double *emu_rest = nullptr;
  int emu_n = 0, emu_cur = 0;           // emu_cur: index of the current iterate (reset -> 0, accept -> +1)
  uint64_t err_epoch = 0;               // ticket value of the last error pass (a host thread can spin on the mirror
                                        // instead of synchronising the stream: window_spin_totals)
  DevBuf geo_px;                        // merged linearize: per local edge and source pixel {omega, D, dD/dx, dD/dy} (geo -> photo)
  bool merge_ok = false;                // both factor types on, geometric weight > 0, not switched off (SAGE_NO_MERGE)
  DevBuf pk;                            // engine-internal channel-group pyramids [K][3 (f,gx,gy)][FS/4][P][4]
  DevBuf f0s;                           // per keyframe: pre-sampled source features, negated [L][FS/4][N][4]
  DevBuf ptab[2], gtab[2];              // edge tables per variable set
  DevBuf work_p, first_p, tiles_p, work_g, first_g, tiles_g;
  DevBuf rec_first_p, rec_count_p;      // photometric linearize: partial RECORDS per edge (flush_p sub-tiles each)
  int flush_p = 0, n_rec_p = 0;
  std::vector<int> Nedge;               // samples (slots) per local directed edge: what the photometric run plan is built from
  int tpb_heur = 1;                     // run length the static rule chose (sage_window_tune_runs measures alternatives)
  DevBuf part_p, part_g;
  DevBuf AtA_p, Atb_p, stats_p, AtA_g, Atb_g, stats_g;
  DevBuf adj_start, adj, link_edges, packed, errbuf;
  int n_work_p = 0, n_work_g = 0, tpb_p = 1, tpb_g = 1;
  std::vector<double> host_packed;
  std::vector<double> delta;
  // device solver (solve_kernels.hip); nullptr -> host envelope Cholesky (envelope wider than the LDS panel).  After a device solve the candidate's host mirrors are refreshed lazily (sync_candidate).
  sage::DeviceSolver *solver = nullptr;
  bool cand_pending = false;
  double residuals_per_lin = 0, bytes_per_lin = 0;
  bool have_lin = false;
  // linearize-at-candidate LM (SageLmConfig::linearize_at_candidate): which variables the packed system belongs to
  uint64_t vars_epoch = 1, lin_epoch = 0; // lin_epoch == vars_epoch: `packed` is the linearisation at the current variables
  bool spec_err_valid = false;
  bool packed_reduced = false; // sharded windows: `packed` has been summed over the ranks since it was last assembled
  double spec_error = 0.0;                // total error at that linearisation point (priors included)
  DevBuf packed_save;                     // linearize-at-candidate: the candidate's (reduced) system is formed here; an accepted
                                          // candidate swaps it with `packed` (which always is the current estimate's system)
  DevBuf packed_loc;                      // reduced windows: this rank's un-reduced share (only the blocks its edges touch are
                                          // ever written, the rest stays zero), the send buffer of the out-of-place all-reduce
  DevBuf asm_blocks;                      // ids of those blocks (keyframes, links, tail) for the assembly of packed_loc
  int n_asm_blocks = 0;
  // optional out-of-place form of the all-reduce hook (native RCCL: send != recv); without it: copy + in-place hook
  int (*allreduce2)(const double *send, double *recv, size_t n, void *user) = nullptr;
  // f2: per-Values factor cache (sage_window_prepass): host copies of every local edge's results and the values
  // (all K keyframes) they were evaluated at
  struct FactorCache
  {
    bool lin = false, err = false;
    std::vector<float> pose, code, scale;         // the key: [K][12], [K][CS], [K]
    std::vector<float> Ap, bp, sp, Ag, bg, sg;    // per local directed edge: AtA, Atb, (error, n_inliers)
    // sage_window_prepare_factors: the projected (NearestPsd) double matrices of every local edge, computed on several
    // host threads right after a prepass; psd_mode < 0: not prepared for the cached linearisation
    std::vector<double> Cp, Cg;
    int psd_mode = -1;
  } fc;
  // optional kernel timing (HIP events on `stream`)
  // phase marks of an LM iteration on the stream's timeline (profiling only): 0 start of the iteration, 1 system
  // assembled, 2 all-reduce of the system enqueued / done, 3 candidate written (scatter + host factorisation + retract),
  // 4 error pass done.  An iteration is the list of marks in the order they were recorded (the classic sequence and the
  // linearize-at-candidate one order them differently, a rejected evaluation repeats some): the time between two
  // consecutive marks is booked to the phase the LATER mark closes
  struct PhaseMarks
  {
    std::vector<std::pair<int, hipEvent_t>> ev; // (mark, event) in the order they were recorded
  };
  std::vector<PhaseMarks> phase_pending;
  PhaseMarks phase_cur;
  double phase_ms[4] = {0, 0, 0, 0}; // linearize, all-reduce, solve, error pass
  int phase_n = 0;
  bool profiling = false;
  int prof_level = 0; // 1: all hot kernels + phase marks, 2: the photometric linearize only
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[4];
  std::vector<hipEvent_t> ev_free; // recycled events (creating / destroying one per mark costs API time inside the region being profiled)
  double prof_ms[4] = {0, 0, 0, 0};
  int prof_n[4] = {0, 0, 0, 0};
};


// shared between the window translation units
template <class T>
static int upload(DevBuf &b, const std::vector<T> &v, hipStream_t s)
{
  int rc = b.reserve(std::max<size_t>(v.size(), 1) * sizeof(T));
  if (rc)
    return rc;
  if (!v.empty())
    SAGE_HIP(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return 0;
}
int window_upload_vars(SageWindow *w, int set);
int window_local_edge(const SageWindow *w, int global_edge); // local index of directed edge 2 * link + dir, or -1
int window_linearize_set(SageWindow *w, int set, double *dst = nullptr, bool local_blocks = false, bool merge = false);
int window_sync_candidate(SageWindow *w, bool stream_idle = false);
void window_phase_mark(SageWindow *w, int which); // profiling: record phase mark `which` on the window's stream
