// runtime.hip -- host runtime behind the C ABI (include/sage_ba.h): workspaces, the per-edge operator API
// that mirrors the reference's df::*_calculate free functions, the tracker wiring and the batched window
// engine (edge tables, work lists, deterministic assembly into block-sparse normal equations, host solve).
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

namespace
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

int pick_tiles_per_block(long long total_tiles)
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

} // namespace

// =====================================================================================================
// workspace
// =====================================================================================================
struct SageWorkspace
{
  hipStream_t stream = nullptr;
  DevBuf work, edge_first, edge_tiles, partials, stats, misc, dpt0;
  float *host_stats = nullptr; // pinned, 2 floats
  int cached_N = -1;
  int n_work = 0;
  int tiles_per_block = 1;
  // tracker wiring (sage_track_frame): one evaluation = several operator launches that leave their statistics on the
  // device (defer_fetch: no D2H + stream synchronise per operator; stats_ptr: where this operator's {error, inliers} go),
  // then ONE copy of everything into pinned memory and one synchronise.  Buffers persist across frames.
  bool defer_fetch = false;
  float *stats_ptr = nullptr;
  DevBuf trk, trk_dpts, trk_kp_dpts; // trk: [pose 12 | photo AtA 49 Atb 7 | keypoint AtA 49 Atb 7 | stats 2 + 2 | pad]
  float *trk_host = nullptr;         // pinned: [pose 12 | pad 4 | results 116]
};

static inline float *ws_stats(SageWorkspace *ws) { return ws->stats_ptr ? ws->stats_ptr : ws->stats.as<float>(); }

extern "C" int sage_workspace_create(void *hip_stream, SageWorkspace **out)
{
  if (!out)
    return SAGE_E_INVALID;
  int ndev = 0;
  SAGE_HIP(hipGetDeviceCount(&ndev));
  if (ndev < 1)
    return (int)hipErrorNoDevice;
  SageWorkspace *ws = new SageWorkspace();
  ws->stream = reinterpret_cast<hipStream_t>(hip_stream);
  hipError_t e = hipHostMalloc((void **)&ws->host_stats, 16 * sizeof(float), hipHostMallocDefault);
  if (e != hipSuccess)
  {
    delete ws;
    return (int)e;
  }
  *out = ws;
  return SAGE_OK;
}

extern "C" void sage_workspace_destroy(SageWorkspace *ws)
{
  if (!ws)
    return;
  ws->work.release();
  ws->edge_first.release();
  ws->edge_tiles.release();
  ws->partials.release();
  ws->stats.release();
  ws->misc.release();
  ws->dpt0.release();
  ws->trk.release();
  ws->trk_dpts.release();
  ws->trk_kp_dpts.release();
  if (ws->host_stats)
    (void)hipHostFree(ws->host_stats);
  if (ws->trk_host)
    (void)hipHostFree(ws->trk_host);
  delete ws;
}

static int ws_prepare(SageWorkspace *ws, int N, size_t partial_floats, LaunchCommon *lc)
{
  if (!ws || N < 0)
    return SAGE_E_INVALID;
  if (ws->cached_N != N)
  {
    WorkList wl;
    wl.build(std::vector<int>{N});
    if (wl.work.empty())
      wl.work.push_back(WorkItem{0, 0}); // N == 0: one empty workgroup so the finalize sees zeros
    if (wl.edge_tiles[0] == 0)
      wl.edge_tiles[0] = 1;
    int rc;
    if ((rc = ws->work.reserve(wl.work.size() * sizeof(WorkItem))))
      return rc;
    if ((rc = ws->edge_first.reserve(sizeof(int32_t))))
      return rc;
    if ((rc = ws->edge_tiles.reserve(sizeof(int32_t))))
      return rc;
    if ((rc = ws->stats.reserve(4 * sizeof(float))))
      return rc;
    SAGE_HIP(hipMemcpyAsync(ws->work.p, wl.work.data(), wl.work.size() * sizeof(WorkItem), hipMemcpyHostToDevice,
                            ws->stream));
    SAGE_HIP(hipMemcpyAsync(ws->edge_first.p, wl.edge_first.data(), sizeof(int32_t), hipMemcpyHostToDevice,
                            ws->stream));
    SAGE_HIP(hipMemcpyAsync(ws->edge_tiles.p, wl.edge_tiles.data(), sizeof(int32_t), hipMemcpyHostToDevice,
                            ws->stream));
    SAGE_HIP(hipStreamSynchronize(ws->stream)); // wl goes out of scope
    ws->cached_N = N;
    ws->n_work = (int)wl.work.size();
    ws->tiles_per_block = wl.tiles_per_block;
  }
  int rc;
  if ((rc = ws->partials.reserve((size_t)ws->n_work * partial_floats * sizeof(float))))
    return rc;
  lc->work = ws->work.as<WorkItem>();
  lc->edge_first = ws->edge_first.as<int32_t>();
  lc->edge_tiles = ws->edge_tiles.as<int32_t>();
  lc->n_work = ws->n_work;
  lc->n_edges = 1;
  lc->partials = ws->partials.as<float>();
  lc->tiles_per_block = ws->tiles_per_block;
  return SAGE_OK;
}

static int ws_fetch_stats(SageWorkspace *ws, float *error_host, float *num_inliers_host)
{
  SAGE_HIP(hipMemcpyAsync(ws->host_stats, ws->stats.p, 2 * sizeof(float), hipMemcpyDeviceToHost, ws->stream));
  SAGE_HIP(hipStreamSynchronize(ws->stream));
  if (error_host)
    *error_host = ws->host_stats[0];
  if (num_inliers_host)
    *num_inliers_host = ws->host_stats[1];
  return SAGE_OK;
}

// source depths of a per-edge operator call: the kernels read them from a map indexed by loc1d.  With few samples only
// those pixels are formed (N*CS reads instead of H*W*CS); the photometric kernels also range-check loc1d against H*W
// (PhotoEdge::HW), so a bad location reads nothing -- the reference's tensor index() would throw on it.
static int operator_depths(SageWorkspace *ws, int CS, const float *bias0, const float *basis0, const float *code0,
                           float scale0, const void *loc, int loc_is_i64, int N, int H, int W)
{
  int rc;
  if ((rc = ws->dpt0.reserve((size_t)H * W * sizeof(float))))
    return rc;
  if ((long long)N * 2 <= (long long)H * W)
    SAGE_HIP(launch_depth_samples(ws->stream, CS, ws->dpt0.as<float>(), bias0, basis0, code0, scale0, loc, loc_is_i64,
                                  N, H * W));
  else
    SAGE_HIP(launch_depth_and_grad(ws->stream, CS, ws->dpt0.as<float>(), nullptr, bias0, basis0, code0, nullptr,
                                   scale0, H, W));
  return SAGE_OK;
}

static bool supported(int CS, int FS)
{
  return (CS == 16 || CS == 32) && (FS == 16 || FS == 32);
}

// =====================================================================================================
// per-edge operator API
// =====================================================================================================
extern "C" int sage_photometric_jac_error_calculate(
    SageWorkspace *ws, float *AtA_dev, float *Atb_dev, float *error_host, float *num_inliers_host,
    const float *R10, const float *t10, const float *R0, const float *t0, const float *R1, const float *t1,
    const float *bias0, const float *basis0, const float *code0, const float *mask1, const int64_t *loc1d,
    const float *homo, const float *feat0, const float *feat1, const float *grad1, float scale0,
    const SagePyramid *pyr, float eps, const float *weights_host, int N, int FS, int CS)
{
  if (!ws || !AtA_dev || !Atb_dev || !pyr || !weights_host || !R0 || !t0 || !R1 || !t1 || !bias0 || !basis0 ||
      !code0 || !mask1 || !loc1d || !homo || !feat0 || !feat1 || !grad1)
    return SAGE_E_INVALID;
  if (!supported(CS, FS) || pyr->levels < 1 || pyr->levels > SAGE_MAX_LEVELS)
    return SAGE_E_UNSUPPORTED;
  LaunchCommon lc;
  int rc = ws_prepare(ws, N, photo_partial_floats(CS), &lc);
  if (rc)
    return rc;
  // depth map of the source keyframe at (code0, scale0): the kernel reads its sample depths from it
  if ((rc = operator_depths(ws, CS, bias0, basis0, code0, scale0, loc1d, 1, N, (int)pyr->cam[0].h, (int)pyr->cam[0].w)))
    return rc;
  PhotoEdge e{};
  e.dpt0 = ws->dpt0.as<float>();
  e.feat0 = feat0; e.feat1 = feat1; e.grad1 = grad1; e.bias0 = bias0; e.basis0 = basis0; e.mask1 = mask1;
  e.homo = homo; e.loc = loc1d; e.loc_is_i64 = 1;
  e.R0 = R0; e.t0 = t0; e.R1 = R1; e.t1 = t1; e.R10 = R10; e.t10 = R10 ? t10 : nullptr;
  e.code0 = code0; e.scale0 = nullptr; e.scale0_val = scale0; e.N = N;
  EdgeOut out{AtA_dev, Atb_dev, ws->stats.as<float>()};
  SAGE_HIP(launch_photo_linearize(ws->stream, CS, FS, &e, nullptr, lc, *pyr, weights_host, eps, out));
  return ws_fetch_stats(ws, error_host, num_inliers_host);
}

extern "C" int sage_photometric_error_calculate(
    SageWorkspace *ws, float *error_host, float *num_inliers_host, const float *R10, const float *t10,
    const float *bias0, const float *basis0, const float *code0, const float *mask1, const int64_t *loc1d,
    const float *homo, const float *feat0, const float *feat1, float scale0, const SagePyramid *pyr, float eps,
    const float *weights_host, int N, int FS, int CS)
{
  if (!ws || !pyr || !weights_host || !R10 || !t10 || !bias0 || !basis0 || !code0 || !mask1 || !loc1d || !homo ||
      !feat0 || !feat1)
    return SAGE_E_INVALID;
  if (!supported(CS, FS) || pyr->levels < 1 || pyr->levels > SAGE_MAX_LEVELS)
    return SAGE_E_UNSUPPORTED;
  LaunchCommon lc;
  int rc = ws_prepare(ws, N, 2, &lc);
  if (rc)
    return rc;
  // depth map of the source keyframe at (code0, scale0): the kernel reads its sample depths from it
  if ((rc = operator_depths(ws, CS, bias0, basis0, code0, scale0, loc1d, 1, N, (int)pyr->cam[0].h, (int)pyr->cam[0].w)))
    return rc;
  PhotoEdge e{};
  e.dpt0 = ws->dpt0.as<float>();
  e.feat0 = feat0; e.feat1 = feat1; e.grad1 = nullptr; e.bias0 = bias0; e.basis0 = basis0; e.mask1 = mask1;
  e.homo = homo; e.loc = loc1d; e.loc_is_i64 = 1;
  e.R10 = R10; e.t10 = t10; e.code0 = code0; e.scale0 = nullptr; e.scale0_val = scale0; e.N = N;
  SAGE_HIP(launch_photo_error(ws->stream, CS, FS, &e, nullptr, lc, *pyr, weights_host, eps, ws->stats.as<float>()));
  return ws_fetch_stats(ws, error_host, num_inliers_host);
}

static int track_common(SageWorkspace *ws, bool jac, int dof, float *AtA, float *Atb, float *error_host,
                        float *num_inliers_host, const float *R, const float *t, const float *mask1,
                        const float *dpts0, const float *homo, const float *feat0s, const float *feat1,
                        const float *grad1, const SagePyramid *pyr, float scale0, float eps,
                        const float *weights_dev, int N, int FS)
{
  if (!ws || !pyr || !R || !t || !mask1 || !dpts0 || !homo || !feat0s || !feat1 || !weights_dev ||
      (jac && (!grad1 || !AtA || !Atb)))
    return SAGE_E_INVALID;
  if ((FS != 16 && FS != 32) || pyr->levels < 1 || pyr->levels > SAGE_MAX_LEVELS)
    return SAGE_E_UNSUPPORTED;
  // tracker kernels process exactly one kTile per workgroup
  if (ws->cached_N != -(N + 2))
  {
    std::vector<WorkItem> work;
    for (int tl = 0; tl < std::max(1, (N + kTile - 1) / kTile); ++tl)
      work.push_back(WorkItem{0, tl});
    int rc;
    if ((rc = ws->work.reserve(work.size() * sizeof(WorkItem))))
      return rc;
    if ((rc = ws->stats.reserve(4 * sizeof(float))))
      return rc;
    SAGE_HIP(hipMemcpyAsync(ws->work.p, work.data(), work.size() * sizeof(WorkItem), hipMemcpyHostToDevice,
                            ws->stream));
    SAGE_HIP(hipStreamSynchronize(ws->stream));
    ws->cached_N = -(N + 2);
    ws->n_work = (int)work.size();
  }
  int rc;
  if ((rc = ws->partials.reserve((size_t)ws->n_work * kTrackScalars * sizeof(float))))
    return rc;
  LaunchCommon lc{};
  lc.work = ws->work.as<WorkItem>();
  lc.n_work = ws->n_work;
  lc.n_edges = 1;
  lc.partials = ws->partials.as<float>();
  lc.tiles_per_block = 1;
  TrackEdge e{};
  e.feat0s = feat0s; e.feat1 = feat1; e.grad1 = grad1; e.mask1 = mask1; e.homo = homo; e.dpts0 = dpts0;
  e.R = R; e.t = t; e.weights = weights_dev; e.scale0 = scale0; e.N = N;
  if (jac)
  {
    EdgeOut out{AtA, Atb, ws_stats(ws)};
    SAGE_HIP(launch_track_linearize(ws->stream, dof, FS, e, lc, *pyr, eps, out));
  }
  else
    SAGE_HIP(launch_track_error(ws->stream, FS, e, lc, *pyr, eps, ws_stats(ws)));
  if (ws->defer_fetch)
    return SAGE_OK;
  return ws_fetch_stats(ws, error_host, num_inliers_host);
}

extern "C" int sage_tracker_photo_jac_error_calculate(
    SageWorkspace *ws, int dof, float *AtA_dev, float *Atb_dev, float *error_host, float *num_inliers_host,
    const float *R, const float *t, const float *mask1, const float *dpts0, const float *homo,
    const float *feat0s, const float *feat1, const float *grad1, const SagePyramid *pyr, float scale0, float eps,
    const float *weights_dev, int N, int FS)
{
  if (dof != 6 && dof != 7)
    return SAGE_E_INVALID;
  return track_common(ws, true, dof, AtA_dev, Atb_dev, error_host, num_inliers_host, R, t, mask1, dpts0, homo,
                      feat0s, feat1, grad1, pyr, scale0, eps, weights_dev, N, FS);
}

extern "C" int sage_tracker_photo_error_calculate(
    SageWorkspace *ws, float *error_host, float *num_inliers_host, const float *R, const float *t,
    const float *mask1, const float *dpts0, const float *homo, const float *feat0s, const float *feat1,
    const SagePyramid *pyr, float eps, const float *weights_dev, int N, int FS)
{
  return track_common(ws, false, 6, nullptr, nullptr, error_host, num_inliers_host, R, t, mask1, dpts0, homo,
                      feat0s, feat1, nullptr, pyr, 1.0f, eps, weights_dev, N, FS);
}

extern "C" int sage_geometric_jac_error_calculate(
    SageWorkspace *ws, float *AtA_dev, float *Atb_dev, float *error_host, float *num_inliers_host,
    const float *R10, const float *t10, const float *R0, const float *t0, const float *R1, const float *t1,
    const float *bias0, const float *basis0, const float *code0, const float *dpt1, const float *dgrad1,
    const float *basis1, const float *mask1, const int32_t *loc1d, const float *homo, float scale0, float scale1,
    const SageCamera *cam, float eps, float loss_param, float weight, int N, int CS)
{
  if (!ws || !AtA_dev || !Atb_dev || !cam || !R0 || !t0 || !R1 || !t1 || !bias0 || !basis0 || !code0 || !dpt1 ||
      !dgrad1 || !basis1 || !mask1 || !loc1d || !homo)
    return SAGE_E_INVALID;
  if (CS != 16 && CS != 32)
    return SAGE_E_UNSUPPORTED;
  LaunchCommon lc;
  int rc = ws_prepare(ws, N, geo_partial_floats(CS), &lc);
  if (rc)
    return rc;
  // depth map of the source keyframe at (code0, scale0): the kernel reads its sample depths from it
  if ((rc = operator_depths(ws, CS, bias0, basis0, code0, scale0, loc1d, 0, N, (int)cam->h, (int)cam->w)))
    return rc;
  GeoEdge e{};
  e.dpt0 = ws->dpt0.as<float>();
  e.bias0 = bias0; e.basis0 = basis0; e.dpt1 = dpt1; e.dgrad1 = dgrad1; e.basis1 = basis1; e.mask1 = mask1;
  e.homo = homo; e.loc = loc1d; e.loc_is_i64 = 0;
  e.R0 = R0; e.t0 = t0; e.R1 = R1; e.t1 = t1; e.R10 = R10; e.t10 = R10 ? t10 : nullptr;
  e.code0 = code0; e.scale0 = nullptr; e.scale1 = nullptr; e.scale0_val = scale0; e.scale1_val = scale1; e.N = N;
  EdgeOut out{AtA_dev, Atb_dev, ws->stats.as<float>()};
  SAGE_HIP(launch_geo_linearize(ws->stream, CS, &e, nullptr, lc, *cam, eps, loss_param, weight, out));
  return ws_fetch_stats(ws, error_host, num_inliers_host);
}

extern "C" int sage_geometric_error_calculate(
    SageWorkspace *ws, float *error_host, float *num_inliers_host, const float *R10, const float *t10,
    const float *bias0, const float *basis0, const float *code0, const float *dpt1, const float *mask1,
    const int32_t *loc1d, const float *homo, float scale0, const SageCamera *cam, float eps, float loss_param,
    float weight, int N, int CS)
{
  if (!ws || !cam || !R10 || !t10 || !bias0 || !basis0 || !code0 || !dpt1 || !mask1 || !loc1d || !homo)
    return SAGE_E_INVALID;
  if (CS != 16 && CS != 32)
    return SAGE_E_UNSUPPORTED;
  LaunchCommon lc;
  int rc = ws_prepare(ws, N, 2, &lc);
  if (rc)
    return rc;
  if ((rc = operator_depths(ws, CS, bias0, basis0, code0, scale0, loc1d, 0, N, (int)cam->h, (int)cam->w)))
    return rc;
  GeoEdge e{};
  e.dpt0 = ws->dpt0.as<float>();
  e.bias0 = bias0; e.basis0 = basis0; e.dpt1 = dpt1; e.mask1 = mask1; e.homo = homo; e.loc = loc1d;
  e.loc_is_i64 = 0; e.R10 = R10; e.t10 = t10; e.code0 = code0; e.scale0_val = scale0; e.scale1_val = 1.f; e.N = N;
  SAGE_HIP(launch_geo_error(ws->stream, CS, &e, nullptr, lc, *cam, eps, loss_param, weight, ws->stats.as<float>()));
  return ws_fetch_stats(ws, error_host, num_inliers_host);
}

extern "C" int sage_depth_and_grad(SageWorkspace *ws, float *dpt, float *grad, const float *bias, const float *basis,
                                   const float *code, float scale, int H, int W, int CS)
{
  if (!ws || !dpt || !bias || !basis || !code)
    return SAGE_E_INVALID;
  if (CS != 16 && CS != 32)
    return SAGE_E_UNSUPPORTED;
  SAGE_HIP(launch_depth_and_grad(ws->stream, CS, dpt, grad, bias, basis, code, nullptr, scale, H, W));
  return SAGE_OK;
}

// ---- sparse reprojection factor (reproj_kernels.hip) ----
static int reproj_common(SageWorkspace *ws, bool tracker, bool jac, float *AtA, float *Atb, float *error_host,
                         float *num_inliers_host, const float *R10, const float *t10, const float *R0, const float *t0,
                         const float *R1, const float *t1, const float *bias0, const float *basis0, const float *code0,
                         const int32_t *loc, const float *dpts0, const float *homo, const float *matched, float scale0,
                         const SageCamera *cam, float eps, float loss_param, float weight, int N, int CS)
{
  if (!ws || !cam || N < 0 || !R10 || !t10 || (N > 0 && (!homo || !matched)) || (jac && (!AtA || !Atb)))
    return SAGE_E_INVALID;
  if (!tracker && (!bias0 || !basis0 || !code0 || (N > 0 && !loc) || (jac && (!R0 || !t0 || !R1 || !t1))))
    return SAGE_E_INVALID;
  if (tracker && N > 0 && !dpts0)
    return SAGE_E_INVALID;
  if (!tracker && CS != 16 && CS != 32)
    return SAGE_E_UNSUPPORTED;
  const int D = tracker ? 6 : 13 + CS;
  int rc;
  if ((rc = ws->misc.reserve(reproj_scratch_floats(N, D) * sizeof(float))) || (rc = ws->stats.reserve(4 * sizeof(float))))
    return rc;
  SAGE_HIP(launch_reproj(ws->stream, CS, tracker, jac, R10, t10, R0, t0, R1, t1, bias0, basis0, code0, loc, dpts0, homo,
                         matched, scale0, *cam, eps, loss_param, weight, N, ws->misc.as<float>(), AtA, Atb,
                         ws_stats(ws)));
  if (ws->defer_fetch)
    return SAGE_OK;
  return ws_fetch_stats(ws, error_host, num_inliers_host);
}

extern "C" int sage_reprojection_jac_error_calculate(SageWorkspace *ws, float *AtA_dev, float *Atb_dev, float *error_host,
                                                     float *num_inliers_host, const float *R10, const float *t10,
                                                     const float *R0, const float *t0, const float *R1, const float *t1,
                                                     const float *bias0, const float *basis0, const float *code0,
                                                     const int32_t *loc1d, const float *homo, const float *matched_2d,
                                                     float scale0, const SageCamera *cam, float eps, float loss_param,
                                                     float weight, int N, int CS)
{
  return reproj_common(ws, false, true, AtA_dev, Atb_dev, error_host, num_inliers_host, R10, t10, R0, t0, R1, t1, bias0,
                       basis0, code0, loc1d, nullptr, homo, matched_2d, scale0, cam, eps, loss_param, weight, N, CS);
}

extern "C" int sage_reprojection_error_calculate(SageWorkspace *ws, float *error_host, float *num_inliers_host,
                                                 const float *R10, const float *t10, const float *bias0,
                                                 const float *basis0, const float *code0, const int32_t *loc1d,
                                                 const float *homo, const float *matched_2d, float scale0,
                                                 const SageCamera *cam, float eps, float loss_param, float weight, int N,
                                                 int CS)
{
  return reproj_common(ws, false, false, nullptr, nullptr, error_host, num_inliers_host, R10, t10, nullptr, nullptr,
                       nullptr, nullptr, bias0, basis0, code0, loc1d, nullptr, homo, matched_2d, scale0, cam, eps,
                       loss_param, weight, N, CS);
}

extern "C" int sage_tracker_reproj_jac_error_calculate(SageWorkspace *ws, float *AtA_dev, float *Atb_dev,
                                                       float *error_host, float *num_inliers_host, const float *R,
                                                       const float *t, const float *sampled_dpts0, const float *homo,
                                                       const float *matched_2d, const SageCamera *cam, float eps,
                                                       float loss_param, float weight, int N)
{
  return reproj_common(ws, true, true, AtA_dev, Atb_dev, error_host, num_inliers_host, R, t, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, sampled_dpts0, homo, matched_2d, 1.f, cam, eps,
                       loss_param, weight, N, 16);
}

extern "C" int sage_tracker_reproj_error_calculate(SageWorkspace *ws, float *error_host, float *num_inliers_host,
                                                   const float *R, const float *t, const float *sampled_dpts0,
                                                   const float *homo, const float *matched_2d, const SageCamera *cam,
                                                   float eps, float loss_param, float weight, int N)
{
  return reproj_common(ws, true, false, nullptr, nullptr, error_host, num_inliers_host, R, t, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, sampled_dpts0, homo, matched_2d, 1.f, cam, eps,
                       loss_param, weight, N, 16);
}

// ---- match-geometry factors (keypoint_kernels.hip) ----
static int mg_common(SageWorkspace *ws, int mode, int loss, bool jac, float *AtA, float *Atb, float *error_host,
                     const float *R10, const float *t10, const float *R0, const float *t0, const float *R1,
                     const float *t1, const float *bias0, const float *bias1, const float *basis0, const float *basis1,
                     const float *code0, const float *code1, const float *dpts0, const float *dpts1, const float *homo0,
                     const float *homo1, const int32_t *loc0, const int32_t *loc1, float scale0, float scale1,
                     float loss_param, float weight, int N, int CS)
{
  if (!ws || N < 1 || !R10 || !t10 || !homo0 || !homo1 || (jac && (!AtA || !Atb)))
    return SAGE_E_INVALID;
  if (mode == 0 && (!bias0 || !bias1 || !basis0 || !basis1 || !code0 || !code1 || !loc0 || !loc1))
    return SAGE_E_INVALID;
  if (mode != 0 && (!dpts0 || !dpts1))
    return SAGE_E_INVALID;
  if (mode <= 1 && jac && (!R0 || !t0 || !R1 || !t1))
    return SAGE_E_INVALID;
  if (loss < 0 || loss > 3 || (loss == SAGE_LOSS_UNBIASED && mode != 0) || (mode != 0 && loss != SAGE_LOSS_FAIR))
    return SAGE_E_INVALID;
  if (mode == 0 && CS != 16 && CS != 32)
    return SAGE_E_UNSUPPORTED;
  const int D = mode == 0 ? 14 + 2 * CS : (mode == 1 ? 14 : (mode == 2 ? 6 : 7));
  int rc;
  if ((rc = ws->misc.reserve(mg_scratch_floats(N, D) * sizeof(float))) || (rc = ws->stats.reserve(4 * sizeof(float))))
    return rc;
  SAGE_HIP(launch_match_geom(ws->stream, mode, loss, CS, jac, R10, t10, R0, t0, R1, t1, bias0, bias1, basis0, basis1,
                             code0, code1, dpts0, dpts1, homo0, homo1, loc0, loc1, scale0, scale1, loss_param, weight, N,
                             ws->misc.as<float>(), AtA, Atb, ws_stats(ws)));
  if (ws->defer_fetch)
    return SAGE_OK;
  return ws_fetch_stats(ws, error_host, nullptr);
}

extern "C" int sage_match_geometry_jac_error_calculate(SageWorkspace *ws, float *AtA_dev, float *Atb_dev,
                                                       float *error_host, const float *R10, const float *t10,
                                                       const float *R0, const float *t0, const float *R1,
                                                       const float *t1, const float *bias0, const float *bias1,
                                                       const float *basis0, const float *basis1, const float *code0,
                                                       const float *code1, const float *homo0,
                                                       const float *matched_homo1, const int32_t *loc1d_0,
                                                       const int32_t *matched_loc1d_1, float scale0, float scale1,
                                                       float loss_param, float weight, int loss, int N, int CS)
{
  return mg_common(ws, 0, loss, true, AtA_dev, Atb_dev, error_host, R10, t10, R0, t0, R1, t1, bias0, bias1, basis0,
                   basis1, code0, code1, nullptr, nullptr, homo0, matched_homo1, loc1d_0, matched_loc1d_1, scale0, scale1,
                   loss_param, weight, N, CS);
}

extern "C" int sage_match_geometry_error_calculate(SageWorkspace *ws, float *error_host, const float *R10,
                                                   const float *t10, const float *bias0, const float *bias1,
                                                   const float *basis0, const float *basis1, const float *code0,
                                                   const float *code1, const float *homo0, const float *matched_homo1,
                                                   const int32_t *loc1d_0, const int32_t *matched_loc1d_1, float scale0,
                                                   float scale1, float loss_param, float weight, int loss, int N, int CS)
{
  return mg_common(ws, 0, loss, false, nullptr, nullptr, error_host, R10, t10, nullptr, nullptr, nullptr, nullptr, bias0,
                   bias1, basis0, basis1, code0, code1, nullptr, nullptr, homo0, matched_homo1, loc1d_0, matched_loc1d_1,
                   scale0, scale1, loss_param, weight, N, CS);
}

extern "C" int sage_loop_mg_jac_error_calculate(SageWorkspace *ws, float *AtA_dev, float *Atb_dev, float *error_host,
                                                const float *R10, const float *t10, const float *R0, const float *t0,
                                                const float *R1, const float *t1, const float *unscaled_dpts0,
                                                const float *matched_unscaled_dpts1, const float *homo0,
                                                const float *matched_homo1, float scale0, float scale1, float loss_param,
                                                float weight, int N)
{
  return mg_common(ws, 1, SAGE_LOSS_FAIR, true, AtA_dev, Atb_dev, error_host, R10, t10, R0, t0, R1, t1, nullptr, nullptr,
                   nullptr, nullptr, nullptr, nullptr, unscaled_dpts0, matched_unscaled_dpts1, homo0, matched_homo1,
                   nullptr, nullptr, scale0, scale1, loss_param, weight, N, 16);
}

extern "C" int sage_loop_mg_error_calculate(SageWorkspace *ws, float *error_host, const float *R10, const float *t10,
                                            const float *unscaled_dpts0, const float *matched_unscaled_dpts1,
                                            const float *homo0, const float *matched_homo1, float scale0, float scale1,
                                            float loss_param, float weight, int N)
{
  return mg_common(ws, 1, SAGE_LOSS_FAIR, false, nullptr, nullptr, error_host, R10, t10, nullptr, nullptr, nullptr,
                   nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, unscaled_dpts0, matched_unscaled_dpts1,
                   homo0, matched_homo1, nullptr, nullptr, scale0, scale1, loss_param, weight, N, 16);
}

extern "C" int sage_tracker_match_geom_jac_error_calculate(SageWorkspace *ws, float *AtA_dev, float *Atb_dev,
                                                           float *error_host, const float *R, const float *t,
                                                           const float *sampled_dpts0, const float *matched_dpts1,
                                                           const float *homo0, const float *matched_homo1, float scale0,
                                                           float loss_param, float weight, int with_scale, int N)
{
  return mg_common(ws, with_scale ? 3 : 2, SAGE_LOSS_FAIR, true, AtA_dev, Atb_dev, error_host, R, t, nullptr, nullptr,
                   nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sampled_dpts0, matched_dpts1,
                   homo0, matched_homo1, nullptr, nullptr, scale0, 1.f, loss_param, weight, N, 16);
}

extern "C" int sage_tracker_match_geom_error_calculate(SageWorkspace *ws, float *error_host, const float *R,
                                                       const float *t, const float *sampled_dpts0,
                                                       const float *matched_dpts1, const float *homo0,
                                                       const float *matched_homo1, float loss_param, float weight, int N)
{
  return mg_common(ws, 2, SAGE_LOSS_FAIR, false, nullptr, nullptr, error_host, R, t, nullptr, nullptr, nullptr, nullptr,
                   nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sampled_dpts0, matched_dpts1, homo0,
                   matched_homo1, nullptr, nullptr, 1.f, 1.f, loss_param, weight, N, 16);
}

extern "C" int sage_cycle_match(SageWorkspace *ws, const float *desc0, const float *desc1, const int64_t *kp_loc1d_0,
                                int K, int C, int H, int W, float cyc_thresh, int64_t *raw_matched_loc1d_1,
                                int64_t *cyc_matched_loc1d_0, int32_t *inlier_flags, int *n_inliers_host)
{
  if (!ws || K < 0 || C < 1 || C > 1024 || H < 1 || W < 1 || !n_inliers_host ||
      (K > 0 && (!desc0 || !desc1 || !kp_loc1d_0 || !raw_matched_loc1d_1 || !cyc_matched_loc1d_0 || !inlier_flags)))
    return SAGE_E_INVALID;
  int rc = ws->misc.reserve(sizeof(int));
  if (rc)
    return rc;
  SAGE_HIP(hipMemsetAsync(ws->misc.p, 0, sizeof(int), ws->stream));
  SAGE_HIP(launch_cycle_match(ws->stream, desc0, desc1, reinterpret_cast<const long long *>(kp_loc1d_0), K, C, H, W,
                              cyc_thresh, reinterpret_cast<long long *>(raw_matched_loc1d_1),
                              reinterpret_cast<long long *>(cyc_matched_loc1d_0), inlier_flags, ws->misc.as<int>()));
  SAGE_HIP(hipMemcpyAsync(n_inliers_host, ws->misc.p, sizeof(int), hipMemcpyDeviceToHost, ws->stream));
  SAGE_HIP(hipStreamSynchronize(ws->stream));
  return SAGE_OK;
}

extern "C" int sage_valid_locations(SageWorkspace *ws, const float *mask_dev, const SageCamera *cam,
                                    int64_t *loc1d_dev, float *homo_dev, int *n_valid_host)
{
  if (!ws || !mask_dev || !cam || !loc1d_dev || !homo_dev || !n_valid_host)
    return SAGE_E_INVALID;
  int rc = ws->misc.reserve(sizeof(int));
  if (rc)
    return rc;
  SAGE_HIP(launch_valid_locations(ws->stream, mask_dev, *cam, reinterpret_cast<long long *>(loc1d_dev), homo_dev,
                                  ws->misc.as<int>()));
  SAGE_HIP(hipMemcpyAsync(n_valid_host, ws->misc.p, sizeof(int), hipMemcpyDeviceToHost, ws->stream));
  SAGE_HIP(hipStreamSynchronize(ws->stream));
  return SAGE_OK;
}

extern "C" int sage_sample_locations(SageWorkspace *ws, const int64_t *valid_loc1d_dev, const float *valid_homo_dev,
                                     int n_valid, int64_t seed, int num_samples, int64_t *loc1d_dev, float *homo_dev,
                                     int *n_out_host)
{
  if (!ws || !valid_loc1d_dev || !valid_homo_dev || n_valid < 0 || num_samples < 0 || !loc1d_dev || !homo_dev ||
      !n_out_host)
    return SAGE_E_INVALID;
  std::vector<int64_t> idx((size_t)std::max(n_valid, 1));
  int rc = sage_shuffle_indices(seed, n_valid, idx.data());
  if (rc)
    return rc;
  const int n = std::min(num_samples, n_valid); // mapper.cpp:1336
  if ((rc = ws->misc.reserve((size_t)std::max(n, 1) * sizeof(int64_t))))
    return rc;
  if (n > 0)
  {
    SAGE_HIP(hipMemcpyAsync(ws->misc.p, idx.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, ws->stream));
    SAGE_HIP(launch_gather_locations(ws->stream, reinterpret_cast<const long long *>(valid_loc1d_dev), valid_homo_dev,
                                     ws->misc.as<long long>(), n, reinterpret_cast<long long *>(loc1d_dev), homo_dev));
    SAGE_HIP(hipStreamSynchronize(ws->stream)); // idx goes out of scope
  }
  *n_out_host = n;
  return SAGE_OK;
}

extern "C" int sage_sort_locations(SageWorkspace *ws, const int64_t *loc1d_dev, const float *homo_dev, int n, int H,
                                   int W, int64_t *loc1d_out_dev, float *homo_out_dev, int *sorted_host)
{
  if (!ws || !loc1d_dev || !homo_dev || n < 0 || H < 1 || W < 1 || !loc1d_out_dev || !homo_out_dev || !sorted_host ||
      loc1d_out_dev == loc1d_dev || homo_out_dev == homo_dev)
    return SAGE_E_INVALID;
  const int HW = H * W;
  // scratch: [item | status (2 ints) | mark plane]
  const size_t off_status = (sizeof(SortItem) + 15) / 16 * 16, off_mark = off_status + 16;
  int rc = ws->misc.reserve(off_mark + (size_t)HW * sizeof(int));
  if (rc)
    return rc;
  char *base = ws->misc.as<char>();
  const SortItem it{reinterpret_cast<const long long *>(loc1d_dev), homo_dev, reinterpret_cast<long long *>(loc1d_out_dev),
                    homo_out_dev, n};
  int status[2] = {0, 0};
  SAGE_HIP(hipMemcpyAsync(base, &it, sizeof(it), hipMemcpyHostToDevice, ws->stream));
  SAGE_HIP(launch_sort_locations(ws->stream, reinterpret_cast<const SortItem *>(base), 1, n, HW,
                                 reinterpret_cast<int *>(base + off_mark), reinterpret_cast<int *>(base + off_status)));
  SAGE_HIP(hipMemcpyAsync(status, base + off_status, sizeof(status), hipMemcpyDeviceToHost, ws->stream));
  SAGE_HIP(hipStreamSynchronize(ws->stream)); // `it` and `status` are locals
  if (status[0] > 0)
    return SAGE_E_INVALID;
  *sorted_host = status[1] == n;
  if (!*sorted_host && n > 0)
  {
    // a pixel listed twice: keep the caller's order (and every sample)
    SAGE_HIP(hipMemcpyAsync(loc1d_out_dev, loc1d_dev, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToDevice, ws->stream));
    SAGE_HIP(hipMemcpyAsync(homo_out_dev, homo_dev, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToDevice, ws->stream));
    SAGE_HIP(hipStreamSynchronize(ws->stream));
  }
  return SAGE_OK;
}

extern "C" int sage_gaussian_pyramid_with_grad(SageWorkspace *ws, float *pyr_dev, float *grad_dev,
                                               const float *feat, const float *mask, const SagePyramid *pyr, int FS)
{
  if (!ws || !pyr_dev || !grad_dev || !feat || !mask || !pyr)
    return SAGE_E_INVALID;
  const int H = (int)pyr->cam[0].h, W = (int)pyr->cam[0].w;
  for (int l = 0; l + 1 < pyr->levels; ++l) // the reference's conv / camera / mask pyramids only agree for even sizes
    if (((int)pyr->cam[l].h & 1) || ((int)pyr->cam[l].w & 1))
      return SAGE_E_UNSUPPORTED;
  const size_t scratch = ((size_t)FS * (H / 2) * (W / 2) + (size_t)(H / 2) * (W / 2)) * 2 * sizeof(float);
  int rc = ws->misc.reserve(scratch);
  if (rc)
    return rc;
  SAGE_HIP(launch_gaussian_pyramid_with_grad(ws->stream, pyr_dev, grad_dev, feat, mask, *pyr, FS,
                                             ws->misc.as<float>()));
  return SAGE_OK;
}

// =====================================================================================================
// tracker: product wiring of the LM callbacks to the HIP kernels
// =====================================================================================================
namespace
{
struct TrackCtx
{
  const SageTrackProblem *prob;
  int dof;
};

// layout of SageWorkspace::trk (floats) and of the pinned mirror trk_host
constexpr int kTrkPose = 0, kTrkOut = 12, kTrkStats = 12 + 112, kTrkFloats = 12 + 112 + 4 + 4;
constexpr int kTrkHostOut = 16; // results start here in trk_host ([0, 12) is the pose on its way to the device)

// depths the kernels of one evaluation read: dof 6 -> the caller's metric depths; dof 7 -> scale * unscaled
// (camera_tracker.cpp:264, :273 candidate error; :431, :453 Jacobian)
int track_depths(TrackCtx *c, float scale, const float **photo, const float **kp)
{
  const SageTrackProblem *p = c->prob;
  SageWorkspace *ws = p->ws;
  *photo = p->dpts0_dev;
  *kp = p->kp_dpts0_dev;
  if (c->dof != 7)
    return 0;
  hipStream_t s = ws->stream;
  if (p->use_photo)
  {
    SAGE_HIP(launch_scale_array(s, ws->trk_dpts.as<float>(), p->dpts0_dev, scale, p->N));
    *photo = ws->trk_dpts.as<float>();
  }
  if (p->use_keypoints)
  {
    SAGE_HIP(launch_scale_array(s, ws->trk_kp_dpts.as<float>(), p->kp_dpts0_dev, scale, p->NK));
    *kp = ws->trk_kp_dpts.as<float>();
  }
  return 0;
}

// the pose of the evaluation -> device (from pinned memory: a true asynchronous copy)
static int track_upload_pose(SageWorkspace *ws, const float *pose12)
{
  std::memcpy(ws->trk_host, pose12, 12 * sizeof(float));
  SAGE_HIP(hipMemcpyAsync(ws->trk.as<float>() + kTrkPose, ws->trk_host, 12 * sizeof(float), hipMemcpyHostToDevice,
                          ws->stream));
  return 0;
}

struct DeferGuard // operators called inside leave their statistics on the device (no per-operator synchronise)
{
  SageWorkspace *ws;
  explicit DeferGuard(SageWorkspace *w) : ws(w) { ws->defer_fetch = true; }
  ~DeferGuard()
  {
    ws->defer_fetch = false;
    ws->stats_ptr = nullptr;
  }
};

// CameraTracker::ComputeJacobianAndError (camera_tracker.cpp:282-328 dof 6, :330-374 dof 7): every term's kernels are
// enqueued, then ONE device-to-host copy of both terms' AtA / Atb / statistics and one stream synchronise (the reference
// pays a .item() synchronise per term and three more in each term's host reduction)
int track_lin_cb(void *vctx, const float *pose12, float scale, float *AtA, float *Atb, float *error)
{
  TrackCtx *c = static_cast<TrackCtx *>(vctx);
  const SageTrackProblem *p = c->prob;
  SageWorkspace *ws = p->ws;
  hipStream_t s = ws->stream;
  const int dof = c->dof;
  int rc = track_upload_pose(ws, pose12);
  if (rc)
    return rc;
  const float *R = ws->trk.as<float>() + kTrkPose, *t = R + 9;
  const float *dp, *kdp;
  if ((rc = track_depths(c, scale, &dp, &kdp)))
    return rc;
  float *dA = ws->trk.as<float>() + kTrkOut, *db = dA + 49, *dA2 = dA + 56, *db2 = dA2 + 49;
  float *st = ws->trk.as<float>() + kTrkStats;
  {
    DeferGuard guard(ws);
    ws->stats_ptr = st;
    if (p->use_photo &&
        (rc = sage_tracker_photo_jac_error_calculate(ws, dof, dA, db, nullptr, nullptr, R, t, p->mask1_dev, dp, p->homo_dev,
                                                     p->feat0s_dev, p->feat1_dev, p->grad1_dev, &p->pyr, scale, p->eps,
                                                     p->weights_dev, p->N, p->FS)))
      return rc;
    ws->stats_ptr = st + 2;
    if (p->use_keypoints)
    {
      if (dof == 6)
        rc = sage_tracker_reproj_jac_error_calculate(ws, dA2, db2, nullptr, nullptr, R, t, kdp, p->kp_homo0_dev,
                                                     p->kp_matched_2d_dev, &p->pyr.cam[0], p->eps, p->kp_loss_param,
                                                     p->kp_weight, p->NK);
      else
        rc = sage_tracker_match_geom_jac_error_calculate(ws, dA2, db2, nullptr, R, t, kdp, p->kp_matched_dpts1_dev,
                                                         p->kp_homo0_dev, p->kp_matched_homo1_dev, scale,
                                                         p->kp_loss_param, p->kp_weight, 1, p->NK);
      if (rc)
        return rc;
    }
  }
  float *host = ws->trk_host + kTrkHostOut;
  SAGE_HIP(hipMemcpyAsync(host, dA, (112 + 4) * sizeof(float), hipMemcpyDeviceToHost, s));
  SAGE_HIP(hipStreamSynchronize(s));
  const float e_photo = p->use_photo ? host[112] : 0.f, e_kp = p->use_keypoints ? host[114] : 0.f;
  // AtA = zeros; AtA += photo_AtA; AtA += keypoint_AtA  (fp32 tensor adds, :296-318 / :344-364)
  for (int i = 0; i < dof * dof; ++i)
    AtA[i] = (p->use_photo ? 0.f + host[i] : 0.f) + (p->use_keypoints ? host[56 + i] : 0.f);
  for (int i = 0; i < dof; ++i)
    Atb[i] = (p->use_photo ? 0.f + host[49 + i] : 0.f) + (p->use_keypoints ? host[56 + 49 + i] : 0.f);
  *error = e_photo + e_kp;
  return 0;
}

// CameraTracker::ComputeError (:220-248 dof 6, :250-280 dof 7)
int track_err_cb(void *vctx, const float *pose12, float scale, float *error)
{
  TrackCtx *c = static_cast<TrackCtx *>(vctx);
  const SageTrackProblem *p = c->prob;
  SageWorkspace *ws = p->ws;
  int rc = track_upload_pose(ws, pose12);
  if (rc)
    return rc;
  const float *R = ws->trk.as<float>() + kTrkPose, *t = R + 9;
  const float *dp, *kdp;
  if ((rc = track_depths(c, scale, &dp, &kdp)))
    return rc;
  float *st = ws->trk.as<float>() + kTrkStats;
  {
    DeferGuard guard(ws);
    ws->stats_ptr = st;
    if (p->use_photo &&
        (rc = sage_tracker_photo_error_calculate(ws, nullptr, nullptr, R, t, p->mask1_dev, dp, p->homo_dev, p->feat0s_dev,
                                                 p->feat1_dev, &p->pyr, p->eps, p->weights_dev, p->N, p->FS)))
      return rc;
    ws->stats_ptr = st + 2;
    if (p->use_keypoints)
    {
      if (c->dof == 6)
        rc = sage_tracker_reproj_error_calculate(ws, nullptr, nullptr, R, t, kdp, p->kp_homo0_dev, p->kp_matched_2d_dev,
                                                 &p->pyr.cam[0], p->eps, p->kp_loss_param, p->kp_weight, p->NK);
      else
        rc = sage_tracker_match_geom_error_calculate(ws, nullptr, R, t, kdp, p->kp_matched_dpts1_dev, p->kp_homo0_dev,
                                                     p->kp_matched_homo1_dev, p->kp_loss_param, p->kp_weight, p->NK);
      if (rc)
        return rc;
    }
  }
  float *host = ws->trk_host + kTrkHostOut;
  SAGE_HIP(hipMemcpyAsync(host + 112, st, 4 * sizeof(float), hipMemcpyDeviceToHost, ws->stream));
  SAGE_HIP(hipStreamSynchronize(ws->stream));
  *error = (p->use_photo ? host[112] : 0.f) + (p->use_keypoints ? host[114] : 0.f);
  return 0;
}
} // namespace

extern "C" int sage_track_frame(const SageLmConfig *cfg, int dof, const SageTrackProblem *prob, float *pose12,
                                float *scale, float *final_error, int *iters, SageLmTraceEntry *trace, int trace_cap,
                                int *trace_len)
{
  if (!cfg || !prob || !prob->ws || !pose12 || (dof != 6 && dof != 7) || (dof == 7 && !scale))
    return SAGE_E_INVALID;
  if (!prob->use_photo && !prob->use_keypoints) // "at least one factor should be enabled" (camera_tracker.cpp:1328)
    return SAGE_E_INVALID;
  if (prob->use_photo && (!prob->mask1_dev || !prob->dpts0_dev || !prob->homo_dev || !prob->feat0s_dev ||
                          !prob->feat1_dev || !prob->grad1_dev || !prob->weights_dev || prob->N < 1))
    return SAGE_E_INVALID;
  if (prob->use_keypoints &&
      (!prob->kp_dpts0_dev || !prob->kp_homo0_dev || prob->NK < 1 ||
       (dof == 6 ? !prob->kp_matched_2d_dev : (!prob->kp_matched_dpts1_dev || !prob->kp_matched_homo1_dev))))
    return SAGE_E_INVALID;
  TrackCtx ctx;
  ctx.prob = prob;
  ctx.dof = dof;
  SageWorkspace *ws = prob->ws;
  int rc;
  // evaluation buffers of the workspace: allocated once, reused by every frame tracked through it
  if ((rc = ws->trk.reserve(kTrkFloats * sizeof(float))))
    return rc;
  if (!ws->trk_host)
    SAGE_HIP(hipHostMalloc((void **)&ws->trk_host, (kTrkHostOut + 112 + 4 + 12) * sizeof(float), hipHostMallocDefault));
  if (dof == 7 && ((prob->use_photo && (rc = ws->trk_dpts.reserve((size_t)prob->N * sizeof(float)))) ||
                   (prob->use_keypoints && (rc = ws->trk_kp_dpts.reserve((size_t)prob->NK * sizeof(float))))))
    return rc;
  rc = sage_track_lm(cfg, dof, track_lin_cb, track_err_cb, &ctx, pose12, scale, final_error, iters, trace, trace_cap,
                     trace_len);
  ws->defer_fetch = false;
  ws->stats_ptr = nullptr;
  return rc;
}

// =====================================================================================================
// window engine
// =====================================================================================================
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
  const int32_t *blk_list; // optional: blockIdx.x -> output block (pipelined solve assembles row chunks), else identity
  int split;               // > 1: every output block is shared by `split` consecutive workgroups (small workgroups)
};

// B-index (0..6+CS: pose 6, code CS, scale) -> column of the per-edge system, or -1 if absent
__device__ __forceinline__ int edge_col(int type, int role, int bi, int CS)
{
  if (bi < 6)
    return role * 6 + bi;
  if (type == 0)
  {
    if (role == 1)
      return -1; // a photometric edge does not touch code1 / scale1
    return bi < 6 + CS ? 12 + (bi - 6) : 12 + CS;
  }
  if (bi < 6 + CS)
    return 12 + role * CS + (bi - 6);
  return 12 + 2 * CS + role;
}

// one workgroup per output block; thread per element; contributions summed in a fixed order (deterministic)
__global__ __launch_bounds__(1024) void assemble_kernel(const AssembleParams p)
{
  const int B = 7 + p.CS, BB = B * B;
  const int Dp = 13 + p.CS, Dg = 14 + 2 * p.CS;
  const int split = p.split > 1 ? p.split : 1;
  const int part = (int)blockIdx.x % split, bslot = (int)blockIdx.x / split;
  const int blk = p.blk_list ? p.blk_list[bslot] : bslot;
  const int tstride = (int)blockDim.x * split, tfirst = part * (int)blockDim.x + (int)threadIdx.x; // element striding
  double *diag = p.packed;
  double *lnk = diag + (size_t)p.K * BB;
  double *g = lnk + (size_t)p.nlinks * BB;
  double *tail = g + (size_t)p.K * B;
  if (blk < p.K)
  {
    // fp64 accumulation of the fp32 per-edge results (the reference widens to double before gtsam sums them:
    // photometric_factor.cpp:305-306).  Adjacency loop outside, the lane's (at most two) outputs inside: the gathers of
    // different adjacency entries are independent, so they overlap instead of forming one chain of ~150 dependent loads
    const int k = blk;
    const int a0 = p.adj_start[k], a1 = p.adj_start[k + 1];
    constexpr int S = 2;
    for (int base = 0; base < BB + B; base += S * tstride) // one pass with 1024 threads (or 4 x 256)
    {
    double acc[S] = {0.0, 0.0};
    int bi[S], bj[S];
    bool isg[S], valid[S];
#pragma unroll
    for (int s = 0; s < S; ++s)
    {
      const int idx = base + tfirst + s * tstride;
      valid[s] = idx < BB + B;
      isg[s] = idx >= BB;
      bi[s] = isg[s] ? idx - BB : idx / B;
      bj[s] = isg[s] ? 0 : idx % B;
    }
#pragma unroll 4
    for (int a = a0; a < a1; ++a)
    {
      const AdjEntry ae = p.adj[a];
      const int D = ae.type == 0 ? Dp : Dg;
      const float *A = ae.type == 0 ? p.AtA_p : p.AtA_g;
      const float *b = ae.type == 0 ? p.Atb_p : p.Atb_g;
#pragma unroll
      for (int s = 0; s < S; ++s)
      {
        if (!valid[s])
          continue;
        const int ci = edge_col(ae.type, ae.role, bi[s], p.CS);
        const int cj = isg[s] ? 0 : edge_col(ae.type, ae.role, bj[s], p.CS);
        if (ci < 0 || cj < 0)
          continue;
        const double *Wd = ae.type == 0 ? p.wide_p : p.wide_g;
        if (Wd)
          acc[s] += Wd[(size_t)ae.edge * (D * D + D) + (isg[s] ? (size_t)D * D + ci : (size_t)ci * D + cj)];
        else
          acc[s] += isg[s] ? (double)b[(size_t)ae.edge * D + ci] : (double)A[(size_t)ae.edge * D * D + (size_t)ci * D + cj];
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
    {
      const int idx = base + tfirst + s * tstride;
      if (!valid[s])
        continue;
      if (isg[s])
        g[(size_t)k * B + bi[s]] = acc[s];
      else
        diag[(size_t)k * BB + idx] = acc[s];
    }
    } // passes
  }
  else if (blk < p.K + p.nlinks)
  {
    const int l = blk - p.K;
    const LinkEdges le = p.links[l];
    for (int idx = tfirst; idx < BB; idx += tstride)
    {
      const int bi = idx / B, bj = idx % B; // bi indexes keyframe a (older), bj keyframe b
      double acc = 0.0;
      if (le.e_ab >= 0)
      {
        for (int type = 0; type < 2; ++type)
        {
          if ((type == 0 && !p.AtA_p) || (type == 1 && !p.AtA_g))
            continue;
          const int D = type == 0 ? Dp : Dg;
          const float *A = type == 0 ? p.AtA_p : p.AtA_g;
          const double *Wd = type == 0 ? p.wide_p : p.wide_g;
          const size_t ws = (size_t)D * D + D;
          // edge a->b : a has role 0, b has role 1
          int ci = edge_col(type, 0, bi, p.CS), cj = edge_col(type, 1, bj, p.CS);
          if (ci >= 0 && cj >= 0)
            acc += Wd ? Wd[(size_t)le.e_ab * ws + (size_t)ci * D + cj] : (double)A[(size_t)le.e_ab * D * D + (size_t)ci * D + cj];
          // edge b->a : b has role 0, a has role 1
          ci = edge_col(type, 1, bi, p.CS);
          cj = edge_col(type, 0, bj, p.CS);
          if (ci >= 0 && cj >= 0)
            acc += Wd ? Wd[(size_t)le.e_ba * ws + (size_t)ci * D + cj] : (double)A[(size_t)le.e_ba * D * D + (size_t)ci * D + cj];
        }
      }
      lnk[(size_t)l * BB + idx] = acc;
    }
  }
  else
  {
    // tail: total errors / inlier counts of the local edges; one wave per sum, fixed lane order (deterministic)
    if (part != 0)
      return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool photo = (wave & 1) == 0;
    const int which = wave >> 1; // 0: error, 1: inliers
    const float *st = photo ? p.stats_p : p.stats_g;
    const int n = photo ? p.n_edges_p : p.n_edges_g;
    double acc = 0.0;
    if (st && wave < 4)
      for (int e = lane; e < n; e += 64)
        acc += (double)st[2 * e + which];
    for (int off = 32; off > 0; off >>= 1)
      acc += __shfl_down(acc, off);
    if (lane == 0 && wave < 4)
    {
      tail[which * 2 + (photo ? 0 : 1)] = acc; // [err_photo err_geo n_photo n_geo]
      if (p.tail_mirror)
        p.tail_mirror[which * 2 + (photo ? 0 : 1)] = acc;
    }
  }
}

// error pass of a window in ONE tail kernel: per-edge statistics of both factor types from the workgroup partials
// (what stats_finalize_kernel does: photometric_factor_kernels.cpp:1049-1058, geometric :868-878) and their totals
// (a wave-parallel sum in a fixed lane order) -- same summation orders, three launches and their gaps less on the step's critical path.
struct ErrorTotalsSide
{
  const int32_t *edge_first, *edge_tiles;
  const float *partials; // [n_work][2]
  float *stats;          // [n_edges][2]
  float fallback, scale;
  int n_edges;           // 0: factor type unused
  int stride, err_off, cnt_off; // record layout: floats per workgroup record, slots of the error sum / the inlier count
};

__global__ __launch_bounds__(1024) void error_totals_kernel(const ErrorTotalsSide ph, const ErrorTotalsSide ge, double *out,
                                                            double *mirror)
{
  for (int idx = threadIdx.x; idx < ph.n_edges + ge.n_edges; idx += blockDim.x)
  {
    const bool photo = idx < ph.n_edges;
    const ErrorTotalsSide &sd = photo ? ph : ge;
    const int e = photo ? idx : idx - ph.n_edges;
    const int first = sd.edge_first[e], nt = sd.edge_tiles[e];
    float se = 0.f, sn = 0.f;
    for (int t = 0; t < nt; ++t)
    {
      se += sd.partials[(size_t)(first + t) * sd.stride + sd.err_off];
      sn += sd.partials[(size_t)(first + t) * sd.stride + sd.cnt_off];
    }
    sd.stats[2 * e + 0] = sn > 0.f ? sd.scale * se / sn : sd.fallback;
    sd.stats[2 * e + 1] = sn;
  }
  __threadfence_block();
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= 4)
    return;
  const bool photo = (wave & 1) == 0;
  const int which = wave >> 1;
  const ErrorTotalsSide &sd = photo ? ph : ge;
  double acc = 0.0;
  for (int e = lane; e < sd.n_edges; e += 64)
    acc += (double)sd.stats[2 * e + which];
  for (int off = 32; off > 0; off >>= 1)
    acc += __shfl_down(acc, off);
  if (lane == 0)
  {
    out[which * 2 + (photo ? 0 : 1)] = acc;
    if (mirror)
      mirror[which * 2 + (photo ? 0 : 1)] = acc;
  }
}

__global__ void copy_floats_kernel(const float *__restrict__ src, float *__restrict__ dst, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    dst[i] = src[i];
}

// sharded windows: the reduced totals (tail of the packed buffer, error buffer) -> pinned host mirror
__global__ void mirror_totals_kernel(const double *__restrict__ tail, const double *__restrict__ err,
                                     double *__restrict__ mirror)
{
  const int t = threadIdx.x;
  if (t < 4)
    mirror[t] = tail[t];
  else if (t < 8)
    mirror[t] = err[t - 4];
}

} // namespace sage

struct SageWindow
{
  SageWindowConfig cfg;
  hipStream_t stream = nullptr;
  // optional second stream (SAGE_TWO_STREAMS=1): the geometric and the photometric linearize kernels are independent
  // (both only need the depth maps).  Measured on MI355X: 2.75 vs 2.78 ms per step -- both kernels are bound by the
  // SIMDs' instruction issue, so running them side by side only stretches each of them; off by default.
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool two_streams = false;
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
  std::vector<int> local_links;           // indices into links
  int n_edges = 0;                        // local directed edges per factor type (= 2 * local links)
  // device
  DevBuf vars[2];                       // [K][VS]: pose 12, scale 1, code CS
  DevBuf wide_p, wide_g;                // per-edge results before their fp32 rounding (EdgeOut::wide)
  DevBuf sorted_loc, sorted_homo;       // raster-ordered copies of the keyframes' sampled locations
  std::vector<std::pair<const int64_t *, const float *>> user_samples; // the caller's arrays
  DevBuf dpt, dgrad, depth_items[2];    // per-keyframe depth maps of the set being evaluated
  int n_depth = 0;                      // keyframes this rank's edges touch (= entries of depth_items)
  int dpt_set = -1;                     // variable set the depth maps currently hold (-1: none) ...
  bool dgrad_valid = false;             // ... and whether their gradients are up to date as well
  // pipelined LM iteration (single-rank windows, sage_window_lm_step, opt-in): the photometric linearize works through
  // the links from both ends of the window inwards and reports finished links through pinned flags; the host launches
  // the post-processing of finished row chunks on `pstream` while both halves of the two-core factorisation consume
  // their rows (see pipe_* below)
  struct PipeChunk
  {
    int ord_first, ord_count; // slice of the solver's order list (rows of both halves, or the separator)
    int need_lo, need_hi;     // local links [0, need_lo] and [need_hi, n) must be linearised before the rows are final
    int blk_first, blk_count; // slice of blk_list (assemble blocks of the chunk)
  };
  bool pipe_enabled = false;
  int device = 0;
  hipStream_t pstream = nullptr;
  hipEvent_t ev_geo = nullptr;
  DevBuf pipe_group, pipe_cnt, pipe_total, pipe_blk_list;
  unsigned *pipe_flags = nullptr;      // pinned [local links]
  unsigned pipe_epoch = 0;
  std::vector<PipeChunk> pipe_chunks;
  std::vector<int> pipe_chunk_of_pos;  // elimination position -> chunk
  std::atomic<int> pipe_next{0};       // per solve: next chunk to launch (both factorisation threads read it)
  std::atomic_flag pipe_lock = ATOMIC_FLAG_INIT;
  int pipe_fin_lo = 0, pipe_fin_hi = 0; // links [0, fin_lo) and [fin_hi, n) have their photometric edges finalised
  std::chrono::steady_clock::time_point pipe_t0; // start of the current pipelined linearize (diagnostics)
  double pipe_t_wait = 0, pipe_t_launch = 0; // SAGE_DEBUG_TIMING: seconds in flag waits / chunk launches of a solve
  DeviceSolver *last_solver = nullptr;   // the solver whose pinned mirror holds the pending candidate
  SageAllReduceFn allreduce = nullptr;  // sharded windows: caller-provided sum all-reduce (see sage_ba.h)
  void *allreduce_user = nullptr;
  void *rccl_hook = nullptr;            // sage_window_use_rccl: owned {comm, stream} record behind `allreduce`
  DevBuf order_p, order_g;              // launch order of the photometric / geometric work lists (build_launch_order)
  // sharded windows, domain-decomposed solve (shard_solve.cpp): the all-reduced payload is the separator system
  SageShardPlan *shard = nullptr;
  DevBuf sepbuf;                        // device copy of the separator buffer (what the collective sums)
  std::vector<double> h_sep;
  double *h_err = nullptr;              // pinned [8]: {linearize tail[4], error pass totals[4]} written by the kernels
  DevBuf pk;                            // engine-internal channel-group pyramids [K][3 (f,gx,gy)][FS/4][P][4]
  DevBuf f0s;                           // per keyframe: pre-sampled source features [L][FS/4][N][4]
  DevBuf ptab[2], gtab[2];              // edge tables per variable set
  DevBuf work_p, first_p, tiles_p, work_g, first_g, tiles_g;
  DevBuf rec_first_p, rec_count_p;      // photometric linearize: partial RECORDS per edge (flush_p sub-tiles each)
  int flush_p = 0, n_rec_p = 0;
  DevBuf part_p, part_g;
  DevBuf AtA_p, Atb_p, stats_p, AtA_g, Atb_g, stats_g;
  DevBuf adj_start, adj, link_edges, packed, errbuf;
  int n_work_p = 0, n_work_g = 0, tpb_p = 1, tpb_g = 1;
  std::vector<double> host_packed;
  std::vector<double> delta;
  // device solver (solve_kernels.hip); nullptr -> host envelope Cholesky (envelope wider than the LDS panel, or
  // SAGE_HOST_SOLVE=1).  After a device solve the candidate's host mirrors are refreshed lazily (sync_candidate).
  sage::DeviceSolver *solver = nullptr;
  bool cand_pending = false;
  double residuals_per_lin = 0, bytes_per_lin = 0;
  bool have_lin = false;
  // linearize-at-candidate LM (SageLmConfig::linearize_at_candidate): which variables the packed system belongs to
  uint64_t vars_epoch = 1, lin_epoch = 0; // lin_epoch == vars_epoch: `packed` is the linearisation at the current variables
  bool spec_err_valid = false;
  bool packed_reduced = false; // sharded windows: `packed` has been summed over the ranks since it was last assembled
  double spec_error = 0.0;                // total error at that linearisation point (priors included)
  DevBuf packed_save;                     // the current system while the candidate's is being formed in `packed`
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
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[4];
  double prof_ms[4] = {0, 0, 0, 0};
  int prof_n[4] = {0, 0, 0, 0};
};

static void prof_attach(SageWindow *w, int which, LaunchCommon &lc)
{
  if (!w->profiling)
    return;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
    return;
  lc.ev_start = a;
  lc.ev_stop = b;
  w->pending[which].emplace_back(a, b);
}

extern "C" int sage_window_set_profiling(SageWindow *w, int on)
{
  if (!w)
    return SAGE_E_INVALID;
  w->profiling = on != 0;
  return SAGE_OK;
}

extern "C" int sage_window_get_kernel_time(SageWindow *w, int which, double *total_ms, int *launches)
{
  if (!w || which < 0 || which > 3)
    return SAGE_E_INVALID;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  for (auto &pr : w->pending[which])
  {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
    {
      w->prof_ms[which] += ms;
      w->prof_n[which] += 1;
    }
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  w->pending[which].clear();
  if (total_ms)
    *total_ms = w->prof_ms[which];
  if (launches)
    *launches = w->prof_n[which];
  w->prof_ms[which] = 0;
  w->prof_n[which] = 0;
  return SAGE_OK;
}

static void upload_vars_host(SageWindow *w, int set, std::vector<float> &buf)
{
  buf.assign((size_t)w->K * w->VS, 0.f);
  const int CS = w->cfg.CS;
  for (int k = 0; k < w->K; ++k)
  {
    float *d = &buf[(size_t)k * w->VS];
    std::memcpy(d, &w->pose[set][(size_t)k * 12], 12 * sizeof(float));
    d[12] = w->scale[set][k];
    std::memcpy(d + 13, &w->code[set][(size_t)k * CS], CS * sizeof(float));
  }
}

static int upload_vars(SageWindow *w, int set)
{
  if (w->dpt_set == set)
    w->dpt_set = -1;
  if (set == 0)
    ++w->vars_epoch; // whatever was linearised is no longer the system at the current variables
  std::vector<float> buf;
  upload_vars_host(w, set, buf);
  SAGE_HIP(hipMemcpyAsync(w->vars[set].p, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice, w->stream));
  SAGE_HIP(hipStreamSynchronize(w->stream)); // buf is a temporary
  return SAGE_OK;
}

extern "C" int sage_window_create(const SageWindowConfig *cfg, void *hip_stream, SageWindow **out)
{
  if (!cfg || !out || !cfg->mask_dev)
    return SAGE_E_INVALID;
  if (!supported(cfg->CS, cfg->FS) || cfg->pyr.levels < 1 || cfg->pyr.levels > SAGE_MAX_LEVELS)
    return SAGE_E_UNSUPPORTED;
  int ndev = 0;
  SAGE_HIP(hipGetDeviceCount(&ndev));
  if (ndev < 1)
    return (int)hipErrorNoDevice;
  SageWindow *w = new SageWindow();
  w->cfg = *cfg;
  w->stream = reinterpret_cast<hipStream_t>(hip_stream);
  w->B = 7 + cfg->CS;
  w->VS = ((13 + cfg->CS + 3) / 4) * 4;
  if (sage::env_flag("SAGE_TWO_STREAMS") && hipStreamCreateWithFlags(&w->stream2, hipStreamNonBlocking) == hipSuccess &&
      hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming) == hipSuccess &&
      hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming) == hipSuccess)
    w->two_streams = true;
  *out = w;
  return SAGE_OK;
}

extern "C" void sage_window_destroy(SageWindow *w)
{
  if (!w)
    return;
  DevBuf *bufs[] = {&w->packed_save, &w->rec_first_p, &w->rec_count_p, &w->order_p, &w->order_g, &w->wide_p, &w->wide_g, &w->sorted_loc, &w->sorted_homo, &w->vars[0], &w->vars[1], &w->dpt, &w->dgrad, &w->depth_items[0], &w->depth_items[1],
                    &w->pk, &w->f0s, &w->ptab[0], &w->ptab[1], &w->gtab[0], &w->gtab[1], &w->work_p, &w->first_p, &w->tiles_p,
                    &w->work_g, &w->first_g, &w->tiles_g, &w->part_p, &w->part_g, &w->AtA_p, &w->Atb_p,
                    &w->stats_p, &w->AtA_g, &w->Atb_g, &w->stats_g, &w->adj_start, &w->adj, &w->link_edges,
                    &w->packed, &w->errbuf};
  for (DevBuf *b : bufs)
    b->release();
  std::free(w->rccl_hook); // (the communicator itself belongs to the caller)
  sage_shard_plan_destroy(w->shard);
  w->sepbuf.release();
  solver_destroy(w->solver);
  if (w->pipe_flags)
    (void)hipHostFree(w->pipe_flags);
  if (w->ev_geo)
    (void)hipEventDestroy(w->ev_geo);
  if (w->pstream)
    (void)hipStreamDestroy(w->pstream);
  w->pipe_group.release(); w->pipe_cnt.release(); w->pipe_total.release(); w->pipe_blk_list.release();
  if (w->h_err)
    (void)hipHostFree(w->h_err);
  if (w->ev_fork)
    (void)hipEventDestroy(w->ev_fork);
  if (w->ev_join)
    (void)hipEventDestroy(w->ev_join);
  if (w->stream2)
    (void)hipStreamDestroy(w->stream2);
  delete w;
}

extern "C" int sage_window_add_keyframe(SageWindow *w, const SageKeyframeView *v, const float *pose12,
                                        const float *code, float scale)
{
  if (!w || !v || !pose12 || !code || w->finalized)
    return SAGE_E_INVALID;
  if (!v->feat_pyr || !v->grad_pyr || !v->bias || !v->basis || !v->loc1d || !v->homo || v->N < 0)
    return SAGE_E_INVALID;
  w->views.push_back(*v);
  for (int s = 0; s < 2; ++s)
  {
    w->pose[s].insert(w->pose[s].end(), pose12, pose12 + 12);
    w->code[s].insert(w->code[s].end(), code, code + w->cfg.CS);
    w->scale[s].push_back(scale);
  }
  w->pose_init.insert(w->pose_init.end(), pose12, pose12 + 12);
  w->code_added.insert(w->code_added.end(), code, code + w->cfg.CS);
  w->scale_init.push_back(scale);
  return w->K++;
}

extern "C" int sage_window_add_link(SageWindow *w, int a, int b)
{
  if (!w || w->finalized || a == b || a < 0 || b < 0 || a >= w->K || b >= w->K)
    return SAGE_E_INVALID;
  w->links.emplace_back(std::min(a, b), std::max(a, b));
  w->link_geo_loss.push_back(0.f);
  return (int)w->links.size() - 1;
}

extern "C" int sage_window_set_link_geo_loss(SageWindow *w, int link, float loss_param)
{
  if (!w || w->finalized || link < 0 || link >= (int)w->links.size() || !(loss_param >= 0.f))
    return w && w->finalized ? SAGE_E_STATE : SAGE_E_INVALID;
  w->link_geo_loss[link] = loss_param;
  return SAGE_OK;
}

static bool device_local_cpulist(int device, char *buf, size_t n)
{
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess)
    return false;
  for (char *p = bdf; *p; ++p)
    *p = (char)tolower((unsigned char)*p);
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  FILE *f = fopen(path, "r");
  if (!f)
    return false;
  buf[0] = 0;
  const bool got = fgets(buf, (int)n, f) != nullptr;
  fclose(f);
  return got;
}

static std::vector<int> parse_cpulist(const char *buf)
{
  std::vector<int> out;
  for (const char *p = buf; *p;)
  {
    char *end;
    const long a = strtol(p, &end, 10);
    if (end == p)
      break;
    long b = a;
    p = end;
    if (*p == '-')
    {
      b = strtol(p + 1, &end, 10);
      p = end;
    }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
      out.push_back((int)c);
    if (*p == ',')
      ++p;
  }
  return out;
}

// One process per GPU: keep the driving thread on the CPUs the GPU hangs off (its NUMA node: the window solve reads
// freshly DMA'd pinned memory), and -- when several GPUs share that node -- on its own L3 domain (CCX) of the node: the
// solve pins its helper / worker threads to the other cores of the caller's CCX (host_math.cpp), so two ranks whose
// driving threads shared a CCX would share those cores.  Returns the number of CPUs the thread is bound to (0: unchanged).
extern "C" int sage_bind_thread_to_device(int device)
{
  char buf[4096] = {0};
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return SAGE_E_INVALID;
  if (!device_local_cpulist(device, buf, sizeof(buf)))
    return 0;
  cpu_set_t allowed, want;
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0)
    return 0;
  std::vector<int> cpus;
  for (int c : parse_cpulist(buf))
    if (CPU_ISSET(c, &allowed))
      cpus.push_back(c);
  if (cpus.empty())
    return 0;
  // devices on the same node, this one's position among them
  int on_node = 0, my_pos = 0;
  for (int d = 0; d < ndev; ++d)
  {
    char other[4096] = {0};
    if (d == device || (device_local_cpulist(d, other, sizeof(other)) && strcmp(other, buf) == 0))
    {
      if (d < device)
        ++my_pos;
      ++on_node;
    }
  }
  if (on_node > 1)
  {
    // L3 domains of the node, in the order of their first CPU
    std::vector<std::vector<int>> groups;
    std::vector<char> seen(CPU_SETSIZE, 0);
    for (int c : cpus)
    {
      if (seen[c])
        continue;
      char path[160], lb[4096] = {0};
      snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", c);
      FILE *f = fopen(path, "r");
      std::vector<int> g;
      if (f)
      {
        if (fgets(lb, sizeof(lb), f))
          for (int x : parse_cpulist(lb))
            if (x < CPU_SETSIZE && CPU_ISSET(x, &allowed) && std::find(cpus.begin(), cpus.end(), x) != cpus.end())
              g.push_back(x);
        fclose(f);
      }
      if (g.empty())
        g.push_back(c);
      for (int x : g)
        seen[x] = 1;
      groups.push_back(g);
    }
    if (groups.size() > 1)
    {
      const size_t stride = std::max<size_t>(1, groups.size() / (size_t)on_node);
      cpus = groups[((size_t)my_pos * stride) % groups.size()];
    }
  }
  CPU_ZERO(&want);
  for (int c : cpus)
    CPU_SET(c, &want);
  if (sched_setaffinity(0, sizeof(want), &want) != 0)
    return 0;
  return (int)cpus.size();
}

extern "C" int sage_window_set_allreduce(SageWindow *w, SageAllReduceFn fn, void *user)
{
  if (!w)
    return SAGE_E_INVALID;
  w->allreduce = fn;
  w->allreduce_user = user;
  return SAGE_OK;
}

// =====================================================================================================
// native RCCL: the all-reduce of a sharded window as an ncclAllReduce on the window's own stream (xGMI), no Python
// and no torch in the loop.  RCCL is bound with dlopen so that the library loads on hosts without it; when the
// process already has an RCCL mapped (PyTorch-ROCm bundles one) that instance is reused.
// =====================================================================================================
namespace
{
struct RcclApi
{
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

RcclApi &rccl()
{
  static RcclApi api = [] {
    RcclApi a;
    const char *already[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : already)
      if (!a.handle)
        a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char *fresh[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *n : fresh)
      if (!a.handle)
        a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!a.handle)
      return a;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.handle, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.handle, "ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce;
    return a;
  }();
  return api;
}

struct RcclHook
{
  ncclComm_t comm;
  hipStream_t stream;
};

int rccl_allreduce_cb(double *buf, size_t n, void *user)
{
  RcclHook *h = static_cast<RcclHook *>(user);
  const ncclResult_t r = rccl().AllReduce(buf, buf, n, ncclDouble, ncclSum, h->comm, h->stream);
  if (r != ncclSuccess)
  {
    fprintf(stderr, "[sage] ncclAllReduce: %s\n", rccl().GetErrorString ? rccl().GetErrorString(r) : "error");
    return 1;
  }
  return 0;
}
} // namespace

extern "C" int sage_rccl_unique_id(unsigned char *id128)
{
  if (!id128)
    return SAGE_E_INVALID;
  if (!rccl().ok)
    return SAGE_E_UNSUPPORTED;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess)
    return SAGE_E_STATE;
  static_assert(sizeof(id) == SAGE_RCCL_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id128, &id, sizeof(id));
  return SAGE_OK;
}

extern "C" int sage_rccl_comm_create(const unsigned char *id128, int rank, int world, void **comm_out)
{
  if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world)
    return SAGE_E_INVALID;
  if (!rccl().ok)
    return SAGE_E_UNSUPPORTED;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t r = rccl().CommInitRank(&c, world, id, rank);
  if (r != ncclSuccess)
  {
    fprintf(stderr, "[sage] ncclCommInitRank: %s\n", rccl().GetErrorString ? rccl().GetErrorString(r) : "error");
    return SAGE_E_STATE;
  }
  *comm_out = c;
  return SAGE_OK;
}

extern "C" void sage_rccl_comm_destroy(void *comm)
{
  if (comm && rccl().ok)
    (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm));
}

extern "C" int sage_window_use_rccl(SageWindow *w, void *nccl_comm)
{
  if (!w || !nccl_comm)
    return SAGE_E_INVALID;
  if (!rccl().ok)
    return SAGE_E_UNSUPPORTED;
  std::free(w->rccl_hook);
  RcclHook *h = static_cast<RcclHook *>(std::malloc(sizeof(RcclHook)));
  if (!h)
    return SAGE_E_STATE;
  h->comm = static_cast<ncclComm_t>(nccl_comm);
  h->stream = w->stream;
  w->rccl_hook = h;
  w->allreduce = rccl_allreduce_cb;
  w->allreduce_user = h;
  return SAGE_OK;
}

extern "C" int sage_window_set_shard(SageWindow *w, int rank, int world)
{
  if (!w || w->finalized || world < 1 || rank < 0 || rank >= world)
    return SAGE_E_INVALID;
  w->rank = rank;
  w->world = world;
  return SAGE_OK;
}

extern "C" int sage_window_num_keyframes(const SageWindow *w) { return w ? w->K : 0; }
extern "C" int sage_window_num_links(const SageWindow *w) { return w ? (int)w->links.size() : 0; }
extern "C" int sage_window_block_size(const SageWindow *w) { return w ? w->B : 0; }
extern "C" size_t sage_window_packed_count(const SageWindow *w)
{
  if (!w)
    return 0;
  const size_t BB = (size_t)w->B * w->B;
  return (size_t)w->K * BB + w->links.size() * BB + (size_t)w->K * w->B + 4;
}
extern "C" double *sage_window_packed_dev(SageWindow *w) { return w ? w->packed.as<double>() : nullptr; }
extern "C" double *sage_window_error_dev(SageWindow *w) { return w ? w->errbuf.as<double>() : nullptr; }
extern "C" double sage_window_residuals_per_linearize(const SageWindow *w) { return w ? w->residuals_per_lin : 0; }
extern "C" double sage_window_bytes_per_linearize(const SageWindow *w) { return w ? w->bytes_per_lin : 0; }

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

static int pipe_setup(SageWindow *w, const std::vector<int32_t> &group_of_work);
static bool pipe_wanted(const SageWindow *w);
static std::vector<int> pipe_link_sequence(int nl, int *mid = nullptr);

// Launch order of a window work list.  The list itself is edge-major (an edge's sub-tile runs are contiguous: its
// partial records must be).  mode 1: the edges of the links that share their newer keyframe (3 links = 6 directed edges
// in a temporal window: 3 INTO that keyframe, sharing their destination pyramids, 3 OUT of it, sharing source samples,
// basis rows and pre-sampled features) are walked BAND BY BAND -- run t of every edge of the group before run t + 1 --
// so the ~770 workgroups in flight work on one or two keyframe groups (tens of MB: L2/MALL resident) instead of on 24
// edges' worth of keyframes.  mode 2: additionally transposes blocks of 8 bands x (edges of the group) so that the
// workgroups of one band go to the same XCD (ids congruent mod 8).
static std::vector<int32_t> build_launch_order(const SageWindow *w, const WorkList &wl, int mode)
{
  std::vector<int32_t> order;
  const int n_edges = (int)wl.edge_first.size();
  if (mode <= 0 || n_edges == 0)
    return order;
  order.reserve(wl.work.size());
  int e0 = 0;
  while (e0 < n_edges)
  {
    // group: consecutive local edges whose links have the same newer keyframe
    const int kb = w->links[w->local_links[e0 / 2]].second;
    int e1 = e0;
    while (e1 < n_edges && w->links[w->local_links[e1 / 2]].second == kb)
      ++e1;
    int max_runs = 0;
    for (int e = e0; e < e1; ++e)
      max_runs = std::max(max_runs, (int)wl.edge_tiles[e]);
    const int ne = e1 - e0;
    if (mode == 1)
    {
      for (int t = 0; t < max_runs; ++t)
        for (int e = e0; e < e1; ++e)
          if (t < wl.edge_tiles[e])
            order.push_back(wl.edge_first[e] + t);
    }
    else
    {
      for (int t0 = 0; t0 < max_runs; t0 += 8)
        for (int j = 0; j < ne; ++j)       // position 8 * j + i  <-  (band t0 + i, edge j): XCD i gets band t0 + i
          for (int i = 0; i < 8; ++i)
            if (t0 + i < wl.edge_tiles[e0 + j])
              order.push_back(wl.edge_first[e0 + j] + t0 + i);
    }
    e0 = e1;
  }
  return order;
}

extern "C" int sage_window_finalize(SageWindow *w)
{
  if (!w || w->finalized || w->K < 1)
    return SAGE_E_INVALID;
  const SageWindowConfig &c = w->cfg;
  const int CS = c.CS, FS = c.FS, K = w->K;
  const int H = (int)c.pyr.cam[0].h, W = (int)c.pyr.cam[0].w, HW = H * W;
  int rc;
  // ---- variables & depth buffers
  for (int s = 0; s < 2; ++s)
  {
    if ((rc = w->vars[s].reserve((size_t)K * w->VS * sizeof(float))))
      return rc;
    if ((rc = upload_vars(w, s)))
      return rc;
  }
  if ((rc = w->dpt.reserve((size_t)K * HW * sizeof(float))) || (rc = w->dgrad.reserve((size_t)K * 2 * HW * sizeof(float))))
    return rc;
  // ---- local links: rank r owns the contiguous range [r*n/world, (r+1)*n/world) of the link list.  Links are added
  //      keyframe by keyframe, so a contiguous range touches ~K/world + (back links) keyframes: only those need depth
  //      maps on this rank
  w->local_links.clear();
  {
    const long long nl = (long long)w->links.size();
    const int lo = (int)(nl * w->rank / w->world), hi = (int)(nl * (w->rank + 1) / w->world);
    for (int l = lo; l < hi; ++l)
      w->local_links.push_back(l);
  }
  std::vector<char> needed(K, 0);
  for (int l : w->local_links)
    needed[w->links[l].first] = needed[w->links[l].second] = 1;
  w->n_depth = 0;
  for (int k = 0; k < K; ++k)
    w->n_depth += needed[k];
  for (int s = 0; s < 2; ++s)
  {
    std::vector<DepthItem> items;
    for (int k = 0; k < K; ++k)
    {
      if (!needed[k])
        continue;
      const float *vp = w->vars[s].as<float>() + (size_t)k * w->VS;
      items.push_back(DepthItem{w->views[k].bias, w->views[k].basis, vp + 13, vp + 12,
                                w->dpt.as<float>() + (size_t)k * HW, w->dgrad.as<float>() + (size_t)k * 2 * HW});
    }
    if ((rc = upload(w->depth_items[s], items, w->stream)))
      return rc;
    SAGE_HIP(hipStreamSynchronize(w->stream));
  }
  // ---- engine-internal relayout, once per keyframe: [FS][P] -> [FS/4][P][4] for feat, grad-x, grad-y
  const size_t plane_f = (size_t)FS * c.pyr.P;
  if ((rc = w->pk.reserve((size_t)K * 3 * plane_f * sizeof(float))))
    return rc;
  for (int k = 0; k < K; ++k)
  {
    float *base = w->pk.as<float>() + (size_t)k * 3 * plane_f;
    SAGE_HIP(launch_repack_groups(w->stream, base, w->views[k].feat_pyr, FS, c.pyr.P));
    SAGE_HIP(launch_repack_groups(w->stream, base + plane_f, w->views[k].grad_pyr, FS, c.pyr.P, 1, &c.pyr));
    SAGE_HIP(launch_repack_groups(w->stream, base + 2 * plane_f, w->views[k].grad_pyr + plane_f, FS, c.pyr.P, 2, &c.pyr));
  }
  // ---- sampled locations: validated (the kernels index depth maps / basis rows with them unchecked) and relaid in
  //      raster order (engine-owned copies; see producers.hip: the sums are order independent, the L1 is not)
  {
    std::vector<size_t> soff(K + 1, 0);
    int max_n = 0;
    for (int k = 0; k < K; ++k)
    {
      soff[k + 1] = soff[k] + (size_t)std::max(1, w->views[k].N);
      max_n = std::max(max_n, w->views[k].N);
    }
    if ((rc = w->sorted_loc.reserve(soff[K] * sizeof(int64_t))) || (rc = w->sorted_homo.reserve(soff[K] * 3 * sizeof(float))))
      return rc;
    if ((int)w->user_samples.size() != K) // (a retried finalize must not sort the sorted copies onto themselves)
    {
      w->user_samples.resize(K);
      for (int k = 0; k < K; ++k)
        w->user_samples[k] = {w->views[k].loc1d, w->views[k].homo};
    }
    std::vector<SortItem> items(K);
    for (int k = 0; k < K; ++k)
      items[k] = SortItem{reinterpret_cast<const long long *>(w->user_samples[k].first), w->user_samples[k].second,
                          w->sorted_loc.as<long long>() + soff[k], w->sorted_homo.as<float>() + 3 * soff[k],
                          w->views[k].N};
    DevBuf d_items, d_mark, d_status;
    std::vector<int> status((size_t)2 * K, 0);
    rc = upload(d_items, items, w->stream);
    if (!rc)
      rc = d_mark.reserve((size_t)K * HW * sizeof(int));
    if (!rc)
      rc = d_status.reserve((size_t)2 * K * sizeof(int));
    hipError_t he = hipSuccess;
    {
      // walk order of the samples: image tiles of 8 x 8 pixels -- a wave's 64 consecutive samples then warp to a compact
      // footprint in every destination keyframe, which is what the LDS-staged sampler of the photometric linearize
      // needs (photo_kernels.hip).  SAGE_SAMPLE_TILE=WxH picks another tile, 0x0 the raster walk.
      static const std::pair<int, int> tile = [] {
        int tw = 8, th = 8;
        if (const char *e = getenv("SAGE_SAMPLE_TILE"))
          if (sscanf(e, "%dx%d", &tw, &th) != 2 || tw < 1 || th < 1)
            tw = th = 0;
        return std::make_pair(tw, th);
      }();
      if (!rc)
        he = launch_sort_locations(w->stream, d_items.as<SortItem>(), K, max_n, HW, d_mark.as<int>(), d_status.as<int>(),
                                   (int)c.pyr.cam[0].w, tile.first, tile.second);
    }
    if (!rc && he == hipSuccess)
      he = hipMemcpyAsync(status.data(), d_status.p, status.size() * sizeof(int), hipMemcpyDeviceToHost, w->stream);
    if (!rc && he == hipSuccess)
      he = hipStreamSynchronize(w->stream);
    d_items.release();
    d_mark.release();
    d_status.release();
    if (rc)
      return rc;
    if (he != hipSuccess)
      return (int)he;
    static const bool no_sort = sage::env_flag("SAGE_NO_SAMPLE_SORT");
    for (int k = 0; k < K; ++k)
    {
      if (status[2 * k] > 0)
        return SAGE_E_INVALID; // a location outside the image
      if (no_sort || status[2 * k + 1] != w->views[k].N)
        continue; // (a pixel sampled twice: the compaction dropped a sample -> keep the caller's order)
      w->views[k].loc1d = reinterpret_cast<const int64_t *>(items[k].loc_out);
      w->views[k].homo = items[k].homo_out;
    }
  }
  // ---- pose-independent pre-sampled source features, once per keyframe
  std::vector<size_t> f0s_off(K + 1, 0);
  for (int k = 0; k < K; ++k)
    f0s_off[k + 1] = f0s_off[k] + (size_t)c.pyr.levels * FS * std::max(1, w->views[k].N);
  if ((rc = w->f0s.reserve(f0s_off[K] * sizeof(float))))
    return rc;
  for (int k = 0; k < K; ++k)
    SAGE_HIP(launch_presample_source(w->stream, w->f0s.as<float>() + f0s_off[k],
                                     w->pk.as<float>() + (size_t)k * 3 * plane_f, w->views[k].homo, w->views[k].N, FS,
                                     c.pyr));
  // ---- local edges
  w->n_edges = 2 * (int)w->local_links.size();
  std::vector<LinkEdges> le(w->links.size(), LinkEdges{-1, -1});
  std::vector<int> Nedge(w->n_edges);
  std::vector<std::vector<AdjEntry>> adjv(K);
  double residuals = 0, bytes = 0;
  const double rho = (double)c.pyr.P / (double)HW;
  for (int s = 0; s < 2; ++s)
  {
    std::vector<PhotoEdge> pt(w->n_edges);
    std::vector<GeoEdge> gt(w->n_edges);
    for (size_t li = 0; li < w->local_links.size(); ++li)
    {
      const int l = w->local_links[li];
      const int ab[2] = {w->links[l].first, w->links[l].second};
      for (int dir = 0; dir < 2; ++dir)
      {
        const int e = 2 * (int)li + dir;
        const int k0 = ab[dir], k1 = ab[1 - dir];
        const SageKeyframeView &v0 = w->views[k0], &v1 = w->views[k1];
        const float *x0 = w->vars[s].as<float>() + (size_t)k0 * w->VS;
        const float *x1 = w->vars[s].as<float>() + (size_t)k1 * w->VS;
        PhotoEdge pe{};
        pe.feat0 = v0.feat_pyr; pe.feat1 = v1.feat_pyr; pe.grad1 = v1.grad_pyr; pe.bias0 = v0.bias;
        pe.feat0_pk = w->pk.as<float>() + (size_t)k0 * 3 * plane_f;
        pe.feat1_pk = w->pk.as<float>() + (size_t)k1 * 3 * plane_f;
        pe.f0s = w->f0s.as<float>() + f0s_off[k0];
        pe.dpt0 = w->dpt.as<float>() + (size_t)k0 * HW;
        pe.dpt1_geo = (c.use_photo && c.use_geo) ? w->dpt.as<float>() + (size_t)k1 * HW : nullptr;
        pe.geo_loss = w->link_geo_loss[l];
        pe.basis0 = v0.basis; pe.mask1 = c.mask_dev; pe.homo = v0.homo; pe.loc = v0.loc1d; pe.loc_is_i64 = 1;
        pe.R0 = x0; pe.t0 = x0 + 9; pe.R1 = x1; pe.t1 = x1 + 9; pe.R10 = nullptr; pe.t10 = nullptr;
        pe.code0 = x0 + 13; pe.scale0 = x0 + 12; pe.N = v0.N;
        pt[e] = pe;
        GeoEdge ge{};
        ge.dpt0 = w->dpt.as<float>() + (size_t)k0 * HW;
        ge.bias0 = v0.bias; ge.basis0 = v0.basis; ge.dpt1 = w->dpt.as<float>() + (size_t)k1 * HW;
        ge.dgrad1 = w->dgrad.as<float>() + (size_t)k1 * 2 * HW; ge.basis1 = v1.basis; ge.mask1 = c.mask_dev;
        ge.homo = v0.homo; ge.loc = v0.loc1d; ge.loc_is_i64 = 1;
        ge.R0 = x0; ge.t0 = x0 + 9; ge.R1 = x1; ge.t1 = x1 + 9; ge.R10 = nullptr; ge.t10 = nullptr;
        ge.code0 = x0 + 13; ge.scale0 = x0 + 12; ge.scale1 = x1 + 12; ge.N = v0.N;
        ge.loss_param = w->link_geo_loss[l];
        gt[e] = ge;
        if (s == 0)
        {
          Nedge[e] = v0.N;
          if (c.use_photo)
          {
            adjv[k0].push_back(AdjEntry{0, e, 0});
            adjv[k1].push_back(AdjEntry{0, e, 1});
            residuals += (double)c.pyr.levels * v0.N * FS;
            bytes += (double)v0.N * 4.0 * (4.0 * FS * rho + CS + 6.0);
          }
          if (c.use_geo)
          {
            adjv[k0].push_back(AdjEntry{1, e, 0});
            adjv[k1].push_back(AdjEntry{1, e, 1});
            residuals += (double)v0.N;
            bytes += (double)v0.N * 4.0 * (2.0 * CS + 9.0);
          }
        }
      }
      if (s == 0)
        le[l] = LinkEdges{2 * (int)li, 2 * (int)li + 1};
    }
    if ((rc = upload(w->ptab[s], pt, w->stream)) || (rc = upload(w->gtab[s], gt, w->stream)))
      return rc;
    SAGE_HIP(hipStreamSynchronize(w->stream));
  }
  w->residuals_per_lin = residuals;
  w->bytes_per_lin = bytes;
  // ---- work lists
  std::vector<int32_t> wp_group_of_work; // photometric work item -> local link (progress signalling, pipe_setup)
  WorkList wl;
  {
    // geometric linearize: the two wave groups of a workgroup alternate over its sub-tiles (geo_kernels.hip), so a
    // workgroup wants an even, longish run of them: the pipeline fill/drain costs one half-step per workgroup
    long long total = 0;
    for (int n : Nedge)
      total += (n + kTile - 1) / kTile;
    int tpb = total >= 8192 ? 16 : (total >= 2048 ? 8 : (total >= 512 ? 4 : 2));
    if (const char *e = getenv("SAGE_GEO_TPB"))
      tpb = std::max(1, atoi(e));
    wl.build(Nedge, tpb);
  }
  w->n_work_g = (int)wl.work.size();
  w->tpb_g = wl.tiles_per_block;
  static const int order_mode = [] { const char *e = getenv("SAGE_WORK_ORDER"); return e ? atoi(e) : 0; }();
  {
    const std::vector<int32_t> og = build_launch_order(w, wl, order_mode);
    if (!og.empty() && og.size() == wl.work.size() && (rc = upload(w->order_g, og, w->stream)))
      return rc;
    if (og.empty())
      w->order_g.release();
  }
  if ((rc = upload(w->work_g, wl.work, w->stream)) || (rc = upload(w->first_g, wl.edge_first, w->stream)) ||
      (rc = upload(w->tiles_g, wl.edge_tiles, w->stream)))
    return rc;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  {
    // photometric work list: its own sub-tile run length
    WorkList wp;
    long long total = 0;
    for (int n : Nedge)
      total += (n + kTile - 1) / kTile;
    // run length of a workgroup (sub-tiles it walks: prologue amortisation, vertical L1/L2 reuse between its bands) and,
    // separately, the number of sub-tiles it accumulates in fp32 before a partial record goes out to the double sums
    // (MFMA chains of 64 fmaf per sub-tile and accumulator): the LM step's distance from the exact step grows with the
    // chain length (K = 64 window, tests/tools/tpb_noise_probe.py: 8 -> 2.1e-4, 4 -> 1.2e-4, 2 -> 7.6e-5, 1 -> 4.9e-5 rel-L2;
    // the fp32 oracle itself sits at 5.5e-5).  Records every 2 sub-tiles keep the step inside the 1e-4 parity bar.
    // (r03, one rank's shard of the K = 64 window at world 8 / 4 = 2.9 k / 5.8 k sub-tiles: runs of 4 / 8 are 19 % / 8 % faster
    //  than the 1 / 2 the first heuristic picked; >= ~3 workgroups per CU stay in flight)
    int tpb = total >= 4096 ? 8 : (total >= 1536 ? 4 : (total >= 768 ? 2 : 1));
    {
      // even runs: an edge of T sub-tiles is cut into ceil(T / tpb) workgroups of ceil(T / that) sub-tiles each -- with
      // T = 12 (3072 samples: the reference's default) runs of 8 leave a half-length second workgroup per edge and the
      // linearize 25 % slower than runs of 6 (BASELINE config 5: 1.85 -> 1.39 ms, error pass 0.49 -> 0.39 ms)
      std::vector<int> tiles;
      for (int n : Nedge)
        tiles.push_back((n + kTile - 1) / kTile);
      if (!tiles.empty())
      {
        std::nth_element(tiles.begin(), tiles.begin() + tiles.size() / 2, tiles.end());
        const int T = std::max(1, tiles[tiles.size() / 2]); // the typical edge
        const int nwg = (T + tpb - 1) / tpb, rem = T % tpb;
        if (rem != 0 && 4 * rem < 3 * tpb) // (a nearly full last run is left alone: T = 63 stays at runs of 8 -- 7 x 9 and
          tpb = (T + nwg - 1) / nwg;       //  9 x 7 measured 5-7 % slower on the headline window)
      }
    }
    if (const char *e = getenv("SAGE_PHOTO_TPB"))
      tpb = std::max(1, atoi(e));
    // a partial record every 4 sub-tiles of a run of 8 (0 = one per workgroup): with the second level of the
    // noise-critical tiles and their split accumulators in the kernel this puts the K = 64 LM step 7.0-8.2e-5 from the fp32
    // oracle's on four windows (r03: tests/tools/delta_probe.py; one record per workgroup: 8.8-9.8e-5) for +2 % of the kernel
    int flush = tpb >= 8 ? 4 : 0;
    if (const char *e = getenv("SAGE_PHOTO_FLUSH"))
      flush = std::max(0, atoi(e));
    std::vector<int> edge_order;
    if (pipe_wanted(w))
      for (int li : pipe_link_sequence((int)w->local_links.size()))
      {
        edge_order.push_back(2 * li); // both directed edges of a link stay together
        edge_order.push_back(2 * li + 1);
      }
    wp.build(Nedge, tpb, edge_order.empty() ? nullptr : &edge_order, flush);
    wp_group_of_work.resize(wp.work.size());
    for (size_t i = 0; i < wp.work.size(); ++i)
      wp_group_of_work[i] = wp.work[i].edge / 2; // group = local link of the edge
    w->n_work_p = (int)wp.work.size();
    w->tpb_p = wp.tiles_per_block;
    {
      const std::vector<int32_t> op = build_launch_order(w, wp, edge_order.empty() ? order_mode : 0);
      if (!op.empty() && op.size() == wp.work.size() && (rc = upload(w->order_p, op, w->stream)))
        return rc;
      if (op.empty())
        w->order_p.release();
    }
    w->flush_p = wp.flush;
    w->n_rec_p = wp.n_records;
    if ((rc = upload(w->work_p, wp.work, w->stream)) || (rc = upload(w->first_p, wp.edge_first, w->stream)) ||
        (rc = upload(w->tiles_p, wp.edge_tiles, w->stream)) || (rc = upload(w->rec_first_p, wp.rec_first, w->stream)) ||
        (rc = upload(w->rec_count_p, wp.rec_count, w->stream)))
      return rc;
  }
  SAGE_HIP(hipStreamSynchronize(w->stream));
  const size_t Dp = 13 + CS, Dg = 14 + 2 * CS;
  const size_t ne = std::max(1, w->n_edges);
  if ((rc = w->part_p.reserve(std::max<size_t>(1, std::max(w->n_work_p, w->n_rec_p)) * photo_partial_floats(CS) * sizeof(float))) ||
      (rc = w->part_g.reserve(std::max<size_t>(1, w->n_work_g) * geo_partial_floats(CS) * sizeof(float))) ||
      (rc = w->AtA_p.reserve(ne * Dp * Dp * sizeof(float))) || (rc = w->Atb_p.reserve(ne * Dp * sizeof(float))) ||
      (rc = w->stats_p.reserve(ne * 2 * sizeof(float))) || (rc = w->AtA_g.reserve(ne * Dg * Dg * sizeof(float))) ||
      (rc = w->Atb_g.reserve(ne * Dg * sizeof(float))) || (rc = w->stats_g.reserve(ne * 2 * sizeof(float))))
    return rc;
  if (!sage::env_flag("SAGE_NO_WIDE_EDGES") &&
      ((rc = w->wide_p.reserve(ne * (Dp * Dp + Dp) * sizeof(double))) ||
       (rc = w->wide_g.reserve(ne * (Dg * Dg + Dg) * sizeof(double)))))
    return rc;
  // ---- adjacency for the assembly
  std::vector<int32_t> adj_start(K + 1, 0);
  std::vector<AdjEntry> adj;
  for (int k = 0; k < K; ++k)
  {
    adj_start[k] = (int32_t)adj.size();
    adj.insert(adj.end(), adjv[k].begin(), adjv[k].end());
  }
  adj_start[K] = (int32_t)adj.size();
  if ((rc = upload(w->adj_start, adj_start, w->stream)) || (rc = upload(w->adj, adj, w->stream)) ||
      (rc = upload(w->link_edges, le, w->stream)))
    return rc;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  if ((rc = w->packed.reserve(sage_window_packed_count(w) * sizeof(double))) || (rc = w->errbuf.reserve(4 * sizeof(double))))
    return rc;
  SAGE_HIP(hipMemsetAsync(w->packed.p, 0, sage_window_packed_count(w) * sizeof(double), w->stream));
  SAGE_HIP(hipMemsetAsync(w->errbuf.p, 0, 4 * sizeof(double), w->stream));
  if (!w->h_err)
  {
    SAGE_HIP(hipHostMalloc(reinterpret_cast<void **>(&w->h_err), 8 * sizeof(double), hipHostMallocDefault));
    std::memset(w->h_err, 0, 8 * sizeof(double));
  }
  w->host_packed.assign(sage_window_packed_count(w), 0.0);
  w->delta.assign((size_t)K * w->B, 0.0);
  if (!sage::env_flag("SAGE_HOST_SOLVE"))
  {
    rc = solver_create(&w->solver, K, w->B, w->VS, w->links, w->stream);
    if (rc != SAGE_OK && rc != SAGE_E_UNSUPPORTED)
      return rc;
  }
  if ((rc = pipe_setup(w, wp_group_of_work)))
    return rc;
  if (w->world > 1)
  {
    // domain-decomposed solve for sharded windows: on by request (SAGE_SHARD_SCHUR=1) or for long windows, where the
    // replicated factorisation of all K keyframes dominates the iteration (DESIGN s7: K = 512 on 8 ranks: 5x less solve)
    const char *e = getenv("SAGE_SHARD_SCHUR");
    const bool want = e ? atoi(e) != 0 : w->K >= 256;
    if (want)
    {
      std::vector<int32_t> lk(2 * w->links.size());
      for (size_t l = 0; l < w->links.size(); ++l)
      {
        lk[2 * l] = w->links[l].first;
        lk[2 * l + 1] = w->links[l].second;
      }
      if ((rc = sage_shard_plan_create(w->K, (int)w->links.size(), lk.data(), w->B, w->rank, w->world, &w->shard)))
        return rc;
      const size_t ns = sage_shard_sep_count(w->shard);
      w->h_sep.assign(ns, 0.0);
      if ((rc = w->sepbuf.reserve(ns * sizeof(double))))
        return rc;
      w->host_packed.resize(sage_window_packed_count(w));
    }
  }
  w->finalized = true;
  return SAGE_OK;
}

static LaunchCommon window_lc(SageWindow *w, bool photo, bool photo_linearize = false)
{
  LaunchCommon lc{};
  lc.work = (photo ? w->work_p : w->work_g).as<WorkItem>();
  lc.edge_first = (photo ? w->first_p : w->first_g).as<int32_t>();
  lc.edge_tiles = (photo ? w->tiles_p : w->tiles_g).as<int32_t>();
  lc.n_work = photo ? w->n_work_p : w->n_work_g;
  lc.n_edges = w->n_edges;
  lc.partials = photo ? w->part_p.as<float>() : w->part_g.as<float>();
  lc.tiles_per_block = photo ? w->tpb_p : w->tpb_g;
  lc.packed = photo;
  if (photo_linearize && w->flush_p > 0)
  {
    // the linearize (and its per-edge finalize) count partial RECORDS, the error pass work items
    lc.edge_first = w->rec_first_p.as<int32_t>();
    lc.edge_tiles = w->rec_count_p.as<int32_t>();
    lc.flush = w->flush_p;
  }
  // opt-in, measured NEGATIVE on the K = 64 headline window (r02): giving every XCD a contiguous eighth of the work list
  // makes the eight L2s work on eight different edge sets at once -- L2 hit rate 59 % -> 25 %, HBM fetch 2.3 -> 5.5 GB per
  // launch, photometric linearize 0.88 -> 1.03 ms.  With the dispatcher's round-robin all XCDs walk the same edges
  // together and the MALL serves the duplicates.
  static const int xcd = [] { const char *e = getenv("SAGE_XCD_ORDER"); return e ? atoi(e) : 0; }();
  lc.xcd_order = xcd != 0 && lc.n_work >= 64 && !w->pipe_enabled; // (the pipelined launch orders links itself)
  if (!w->pipe_enabled)
    lc.order = photo ? (w->order_p.p ? w->order_p.as<int32_t>() : nullptr) : (w->order_g.p ? w->order_g.as<int32_t>() : nullptr);
  return lc;
}

static AssembleParams window_assemble_params(SageWindow *w)
{
  const SageWindowConfig &c = w->cfg;
  AssembleParams ap{};
  const bool has = w->n_edges > 0;
  ap.AtA_p = (has && c.use_photo) ? w->AtA_p.as<float>() : nullptr;
  ap.Atb_p = w->Atb_p.as<float>();
  ap.stats_p = (has && c.use_photo) ? w->stats_p.as<float>() : nullptr;
  ap.AtA_g = (has && c.use_geo) ? w->AtA_g.as<float>() : nullptr;
  ap.wide_p = (has && c.use_photo) ? w->wide_p.as<double>() : nullptr;
  ap.wide_g = (has && c.use_geo) ? w->wide_g.as<double>() : nullptr;
  ap.Atb_g = w->Atb_g.as<float>();
  ap.stats_g = (has && c.use_geo) ? w->stats_g.as<float>() : nullptr;
  ap.adj_start = w->adj_start.as<int32_t>();
  ap.adj = w->adj.as<AdjEntry>();
  ap.links = w->link_edges.as<LinkEdges>();
  ap.packed = w->packed.as<double>();
  ap.tail_mirror = w->world == 1 ? w->h_err : nullptr;
  ap.K = w->K;
  ap.nlinks = (int)w->links.size();
  ap.CS = c.CS;
  ap.n_edges_p = w->n_edges;
  ap.n_edges_g = w->n_edges;
  ap.blk_list = nullptr;
  ap.split = 1;
  return ap;
}

// linearize every local edge at variable set `set` (0 = current estimate, 1 = candidate) and assemble the packed system
static int window_linearize_set(SageWindow *w, int set)
{
  if (!w || !w->finalized)
    return SAGE_E_STATE;
  const SageWindowConfig &c = w->cfg;
  const int H = (int)c.pyr.cam[0].h, W = (int)c.pyr.cam[0].w;
  if (w->n_edges > 0)
  {
    // depth maps of every keyframe at the current variables: both factor types read their sample depths from them
    // (an accepted candidate's maps from the error pass are still valid: only the gradients are missing then)
    static const bool no_reuse = sage::env_flag("SAGE_NO_DEPTH_REUSE");
    const bool have_depth = w->dpt_set == set && !no_reuse;
    SAGE_HIP(launch_depth_batch(w->stream, c.CS, w->depth_items[set].as<DepthItem>(), w->n_depth, H, W, !have_depth,
                                !(have_depth && w->dgrad_valid)));
    w->dpt_set = set;
    w->dgrad_valid = true;
    const bool fork = w->two_streams && c.use_photo && c.use_geo;
    hipStream_t gs = fork ? w->stream2 : w->stream;
    if (fork)
    {
      SAGE_HIP(hipEventRecord(w->ev_fork, w->stream));
      SAGE_HIP(hipStreamWaitEvent(gs, w->ev_fork, 0));
    }
    // geometric first: its per-edge finalize (17 us) then hides between the two big kernels and only the shorter
    // photometric finalize (9 us) sits between the last kernel and the assembly
    if (c.use_geo)
    {
      EdgeOut out{w->AtA_g.as<float>(), w->Atb_g.as<float>(), w->stats_g.as<float>(), w->wide_g.as<double>()};
      LaunchCommon lc = window_lc(w, false);
      prof_attach(w, 1, lc);
      SAGE_HIP(launch_geo_linearize(gs, c.CS, nullptr, w->gtab[set].as<GeoEdge>(), lc, c.pyr.cam[0], c.eps,
                                    c.geo_loss_param, c.geo_weight, out));
      if (fork)
        SAGE_HIP(hipEventRecord(w->ev_join, gs));
    }
    if (c.use_photo)
    {
      EdgeOut out{w->AtA_p.as<float>(), w->Atb_p.as<float>(), w->stats_p.as<float>(), w->wide_p.as<double>()};
      LaunchCommon lc = window_lc(w, true, true);
      prof_attach(w, 0, lc);
      SAGE_HIP(launch_photo_linearize(w->stream, c.CS, c.FS, nullptr, w->ptab[set].as<PhotoEdge>(), lc, c.pyr,
                                      c.photo_weights, c.eps, out));
    }
    if (fork)
      SAGE_HIP(hipStreamWaitEvent(w->stream, w->ev_join, 0));
  }
  AssembleParams ap = window_assemble_params(w);
  // four workgroups of 512 threads per output block: one element per thread (the kernel is a chain of dependent
  // gathers per element -- 17 us; one 1024-thread workgroup per block with two elements per thread took 27 us)
  ap.split = 4;
  hipLaunchKernelGGL(assemble_kernel, dim3((w->K + ap.nlinks + 1) * ap.split), dim3(512), 0, w->stream, ap);
  SAGE_HIP(hipGetLastError());
  w->have_lin = true;
  w->lin_epoch = set == 0 ? w->vars_epoch : 0; // (a candidate's system becomes current only through lm_step's accept)
  w->spec_err_valid = false;
  w->packed_reduced = false;
  return SAGE_OK;
}

extern "C" int sage_window_linearize(SageWindow *w) { return window_linearize_set(w, 0); }

extern "C" int sage_window_error(SageWindow *w, int which)
{
  if (!w || !w->finalized || which < 0 || which > 1)
    return SAGE_E_STATE;
  const SageWindowConfig &c = w->cfg;
  const int H = (int)c.pyr.cam[0].h, W = (int)c.pyr.cam[0].w;
  const bool has = w->n_edges > 0;
  if (has && w->dpt_set != which)
  {
    SAGE_HIP(launch_depth_batch(w->stream, c.CS, w->depth_items[which].as<DepthItem>(), w->n_depth, H, W, true, false));
    w->dpt_set = which;
    w->dgrad_valid = false;
  }
  ErrorTotalsSide ph{}, ge{};
  // both factor types: ONE kernel -- the photometric error kernel also evaluates the geometric edge at the same warp
  // (PhotoEdge::dpt1_geo), which saves the geometric launch (39 us + a gap) of the error pass
  static const bool no_fusion = sage::env_flag("SAGE_NO_ERROR_FUSION");
  const bool fused = has && c.use_photo && c.use_geo && !no_fusion;
  if (has && c.use_photo)
  {
    LaunchCommon lc = window_lc(w, true);
    prof_attach(w, 2, lc);
    lc.stage = 1; // main kernel only: the per-edge statistics are formed by error_totals_kernel below
    lc.fused_geo_loss_param = fused ? c.geo_loss_param : 0.f;
    SAGE_HIP(launch_photo_error(w->stream, c.CS, c.FS, nullptr, w->ptab[which].as<PhotoEdge>(), lc, c.pyr,
                                c.photo_weights, c.eps, w->stats_p.as<float>()));
    float wsum = 0.f;
    for (int l = 0; l < c.pyr.levels; ++l)
      wsum += c.photo_weights[l];
    ph = ErrorTotalsSide{lc.edge_first, lc.edge_tiles, lc.partials, w->stats_p.as<float>(), 10.0f * wsum, 1.0f, w->n_edges,
                         fused ? 4 : 2, 0, 1};
    if (fused)
      ge = ErrorTotalsSide{lc.edge_first, lc.edge_tiles, lc.partials, w->stats_g.as<float>(), 10.0f * c.geo_weight,
                           c.geo_weight, w->n_edges, 4, 2, 3};
  }
  if (has && c.use_geo && !fused)
  {
    LaunchCommon lc = window_lc(w, false);
    prof_attach(w, 3, lc);
    lc.stage = 1;
    SAGE_HIP(launch_geo_error(w->stream, c.CS, nullptr, w->gtab[which].as<GeoEdge>(), lc, c.pyr.cam[0], c.eps,
                              c.geo_loss_param, c.geo_weight, w->stats_g.as<float>()));
    ge = ErrorTotalsSide{lc.edge_first, lc.edge_tiles, lc.partials, w->stats_g.as<float>(), 10.0f * c.geo_weight,
                         c.geo_weight, w->n_edges, 2, 0, 1};
  }
  hipLaunchKernelGGL(error_totals_kernel, dim3(1), dim3(1024), 0, w->stream, ph, ge, w->errbuf.as<double>(),
                     w->world == 1 && w->h_err ? w->h_err + 4 : nullptr);
  SAGE_HIP(hipGetLastError());
  return SAGE_OK;
}

// After a device solve the candidate variables / delta live in the solver's pinned buffers until the stream has
// drained: refresh the host mirrors (set 1) here.  Returns SAGE_E_NOT_PSD when the factorisation hit a non-positive
// pivot (the candidate is then meaningless).
static int sync_candidate(SageWindow *w)
{
  if (!w->cand_pending)
    return SAGE_OK;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  w->cand_pending = false;
  const DeviceSolver *S = w->last_solver ? w->last_solver : w->solver;
  if (solver_host_status(S) != 0)
    return SAGE_E_NOT_PSD;
  const int K = w->K, CS = w->cfg.CS, VS = w->VS;
  const float *v = solver_host_vars(S);
  for (int k = 0; k < K; ++k)
  {
    std::memcpy(&w->pose[1][(size_t)k * 12], v + (size_t)k * VS, 12 * sizeof(float));
    w->scale[1][k] = v[(size_t)k * VS + 12];
    std::memcpy(&w->code[1][(size_t)k * CS], v + (size_t)k * VS + 13, CS * sizeof(float));
  }
  std::memcpy(w->delta.data(), solver_host_delta(S), w->delta.size() * sizeof(double));
  return SAGE_OK;
}

// prior error terms at a variable set (a9): code prior w*||c||^2/CS per keyframe (code_factor.cpp:99-104, zero
// prior code), scale prior on keyframe 0 w*(ln s0 - ln s)^2 (scale_factor.cpp:102-129), pose prior on kf 0.
static void pose_local(const float *origin, const float *other, double out[6])
{
  // gtsam_traits.h:78-89 : [t1 - R1 R0^T t0, log(R1 R0^T)]
  double Rr[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      Rr[i * 3 + j] = (double)other[i * 3 + 0] * origin[j * 3 + 0] + (double)other[i * 3 + 1] * origin[j * 3 + 1] +
                      (double)other[i * 3 + 2] * origin[j * 3 + 2];
  for (int i = 0; i < 3; ++i)
    out[i] = other[9 + i] - (Rr[i * 3 + 0] * origin[9] + Rr[i * 3 + 1] * origin[10] + Rr[i * 3 + 2] * origin[11]);
  const double tr = Rr[0] + Rr[4] + Rr[8];
  const double cs = std::min(1.0, std::max(-1.0, 0.5 * (tr - 1.0)));
  const double th = std::acos(cs);
  const double k = th < 1e-8 ? 0.5 : th / (2.0 * std::sin(th));
  out[3] = k * (Rr[7] - Rr[5]);
  out[4] = k * (Rr[2] - Rr[6]);
  out[5] = k * (Rr[3] - Rr[1]);
}

static double prior_error(const SageWindow *w, int set)
{
  const SageWindowConfig &c = w->cfg;
  double e = 0;
  for (int k = 0; k < w->K; ++k)
  {
    double s = 0;
    for (int i = 0; i < c.CS; ++i)
      s += (double)w->code[set][(size_t)k * c.CS + i] * w->code[set][(size_t)k * c.CS + i];
    e += c.code_prior_weight * s / c.CS;
  }
  if (c.scale_prior_weight > 0)
  {
    const double d = std::log((double)w->scale_init[0]) - std::log((double)w->scale[set][0]);
    e += c.scale_prior_weight * d * d;
  }
  if (c.pose_prior_weight > 0)
  {
    double loc[6];
    pose_local(&w->pose[set][0], &w->pose_init[0], loc);
    for (int i = 0; i < 6; ++i)
      e += c.pose_prior_weight * loc[i] * loc[i];
  }
  return e;
}

extern "C" int sage_window_total_error(SageWindow *w, int from_linearize, double *err)
{
  if (!w || !w->finalized || !err)
    return SAGE_E_STATE;
  double t[4];
  int rcs = sync_candidate(w);
  if (rcs && rcs != SAGE_E_NOT_PSD)
    return rcs;
  // a failed factorisation only invalidates the CANDIDATE: the error at the linearisation point is still served
  const int rc_out = from_linearize ? SAGE_OK : rcs;
  if (w->world == 1 && w->h_err)
  {
    // single-rank window: the kernels mirrored the totals into pinned host memory
    SAGE_HIP(hipStreamSynchronize(w->stream));
    const double *m = w->h_err + (from_linearize ? 0 : 4);
    *err = m[0] + m[1] + prior_error(w, from_linearize ? 0 : 1);
    return rc_out;
  }
  if (from_linearize)
  {
    const size_t off = sage_window_packed_count(w) - 4;
    SAGE_HIP(hipMemcpyAsync(t, w->packed.as<double>() + off, 4 * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  }
  else
    SAGE_HIP(hipMemcpyAsync(t, w->errbuf.p, 4 * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  SAGE_HIP(hipStreamSynchronize(w->stream));
  *err = t[0] + t[1] + prior_error(w, from_linearize ? 0 : 1);
  return rc_out;
}

extern "C" int sage_window_solve(SageWindow *w, double damp, double *step_norm)
{
  if (!w || !w->finalized || !w->have_lin)
    return SAGE_E_STATE;
  const SageWindowConfig &c = w->cfg;
  const int K = w->K, B = w->B, CS = c.CS, BB = B * B, n = K * B;
  if (w->solver)
  {
    // device path: nothing leaves HBM but the candidate's host mirror (pinned, async); no synchronisation here
    // unless the caller asks for the step norm
    int rc = sync_candidate(w); // an unconsumed earlier candidate (a re-solve with another damping)
    if (rc && rc != SAGE_E_NOT_PSD)
      return rc;
    if (w->dpt_set == 1)
      w->dpt_set = -1; // the solve rewrites the candidate set
    rc = solver_run(w->solver, w->stream, w->packed.as<double>(), w->vars[0].as<float>(), w->vars[1].as<float>(), CS,
                    damp, c.code_prior_weight, c.scale_prior_weight, c.pose_prior_weight, w->scale_init[0],
                    &w->pose_init[0]);
    if (rc)
      return rc;
    w->cand_pending = true;
    w->last_solver = w->solver;
    if (step_norm)
    {
      if ((rc = sync_candidate(w)))
        return rc;
      *step_norm = std::sqrt(solver_host_step_norm2(w->last_solver ? w->last_solver : w->solver));
    }
    return SAGE_OK;
  }
  const size_t np = sage_window_packed_count(w);
  static const bool dbg = sage::env_flag("SAGE_DEBUG_TIMING");
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_a = tnow();
  SAGE_HIP(hipStreamSynchronize(w->stream));
  auto t_b = tnow();
  SAGE_HIP(hipMemcpyAsync(w->host_packed.data(), w->packed.p, np * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  SAGE_HIP(hipStreamSynchronize(w->stream));
  auto t_c = tnow();
  // diagonal priors (a9): code prior on every keyframe, scale / pose priors on keyframe 0
  std::vector<double> dadd((size_t)n, 0.0), gadd((size_t)n, 0.0);
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < CS; ++i)
    {
      dadd[k * B + 6 + i] += c.code_prior_weight;
      gadd[k * B + 6 + i] += c.code_prior_weight * (0.0 - (double)w->code[0][(size_t)k * CS + i]);
    }
  if (c.scale_prior_weight > 0)
  {
    const double s = w->scale[0][0];
    dadd[6 + CS] += c.scale_prior_weight / (s * s);
    gadd[6 + CS] += c.scale_prior_weight / s * (std::log((double)w->scale_init[0]) - std::log(s));
  }
  if (c.pose_prior_weight > 0)
  {
    double loc[6];
    pose_local(&w->pose[0][0], &w->pose_init[0], loc);
    for (int i = 0; i < 6; ++i)
    {
      dadd[i] += c.pose_prior_weight;
      gadd[i] += c.pose_prior_weight * loc[i];
    }
  }
  std::vector<int32_t> lk(2 * w->links.size());
  for (size_t l = 0; l < w->links.size(); ++l)
  {
    lk[2 * l] = w->links[l].first;
    lk[2 * l + 1] = w->links[l].second;
  }
  std::vector<double> rhs((size_t)n);
  (void)BB;
  int rcs = sage_block_solve(w->host_packed.data(), K, (int)w->links.size(), lk.data(), B, damp, dadd.data(),
                             gadd.data(), rhs.data());
  if (rcs)
    return rcs;
  auto t_d = tnow();
  w->delta = rhs;
  double nrm = 0;
  for (double v : rhs)
    nrm += v * v;
  if (step_norm)
    *step_norm = std::sqrt(nrm);
  // candidate = retract(current, delta)
  for (int k = 0; k < K; ++k)
  {
    float d6[6];
    for (int i = 0; i < 6; ++i)
      d6[i] = (float)rhs[k * B + i];
    sage_pose_retract(&w->pose[0][(size_t)k * 12], d6, &w->pose[1][(size_t)k * 12]);
    for (int i = 0; i < CS; ++i)
      w->code[1][(size_t)k * CS + i] = w->code[0][(size_t)k * CS + i] + (float)rhs[k * B + 6 + i];
    w->scale[1][k] = w->scale[0][k] + (float)rhs[k * B + 6 + CS];
  }
  const int rcu = upload_vars(w, 1);
  if (dbg)
  {
    auto t_e = tnow();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[sage solve] wait-kernels %.3f d2h %.3f block_solve %.3f retract+h2d %.3f ms\n", ms(t_a, t_b),
            ms(t_b, t_c), ms(t_c, t_d), ms(t_d, t_e));
  }
  return rcu;
}

extern "C" int sage_window_accept(SageWindow *w)
{
  if (!w || !w->finalized)
    return SAGE_E_STATE;
  int rcs = sync_candidate(w);
  if (rcs)
    return rcs;
  w->pose[0] = w->pose[1];
  w->code[0] = w->code[1];
  w->scale[0] = w->scale[1];
  ++w->vars_epoch;
  w->dpt_set = w->dpt_set == 1 ? 0 : -1; // depth maps evaluated at the candidate now belong to the current set
  // (a kernel, not hipMemcpyAsync: a device-to-device copy of 11 KB costs ~10 us of API time on the step's critical path)
  const int nv = w->K * w->VS;
  hipLaunchKernelGGL(copy_floats_kernel, dim3((nv + 255) / 256), dim3(256), 0, w->stream, w->vars[1].as<float>(),
                     w->vars[0].as<float>(), nv);
  SAGE_HIP(hipGetLastError());
  return SAGE_OK;
}

extern "C" int sage_window_reset(SageWindow *w)
{
  if (!w || !w->finalized)
    return SAGE_E_STATE;
  (void)sync_candidate(w);
  for (int s = 0; s < 2; ++s)
  {
    w->pose[s] = w->pose_init;
    w->code[s] = w->code_added;
    w->scale[s] = w->scale_init;
  }
  int rc;
  if ((rc = upload_vars(w, 0)) || (rc = upload_vars(w, 1)))
    return rc;
  w->have_lin = false;
  return SAGE_OK;
}

extern "C" int sage_window_get_keyframe(const SageWindow *w, int kf, float *pose12, float *code, float *scale)
{
  if (!w || kf < 0 || kf >= w->K)
    return SAGE_E_INVALID;
  if (pose12)
    std::memcpy(pose12, &w->pose[0][(size_t)kf * 12], 12 * sizeof(float));
  if (code)
    std::memcpy(code, &w->code[0][(size_t)kf * w->cfg.CS], w->cfg.CS * sizeof(float));
  if (scale)
    *scale = w->scale[0][kf];
  return SAGE_OK;
}

extern "C" int sage_window_set_keyframe(SageWindow *w, int kf, const float *pose12, const float *code, float scale)
{
  if (!w || kf < 0 || kf >= w->K || !pose12 || !code)
    return SAGE_E_INVALID;
  (void)sync_candidate(w);
  for (int s = 0; s < 2; ++s)
  {
    std::memcpy(&w->pose[s][(size_t)kf * 12], pose12, 12 * sizeof(float));
    std::memcpy(&w->code[s][(size_t)kf * w->cfg.CS], code, w->cfg.CS * sizeof(float));
    w->scale[s][kf] = scale;
  }
  if (w->finalized)
  {
    int rc;
    if ((rc = upload_vars(w, 0)) || (rc = upload_vars(w, 1)))
      return rc;
  }
  return SAGE_OK;
}

extern "C" int sage_window_get_delta(const SageWindow *w, double *delta)
{
  if (!w || !delta)
    return SAGE_E_INVALID;
  int rcs = sync_candidate(const_cast<SageWindow *>(w));
  if (rcs)
    return rcs;
  std::memcpy(delta, w->delta.data(), w->delta.size() * sizeof(double));
  return SAGE_OK;
}

extern "C" int sage_window_get_edge(const SageWindow *w, int type, int e, float *AtA, float *Atb, float *err,
                                    float *n_in)
{
  if (!w || !w->finalized || (type != 0 && type != 1))
    return SAGE_E_INVALID;
  // e is the global directed-edge index: link e/2, direction e%2 ; map to the local index
  const int l = e / 2, dir = e % 2;
  int li = -1;
  for (size_t i = 0; i < w->local_links.size(); ++i)
    if (w->local_links[i] == l)
      li = (int)i;
  if (li < 0)
    return SAGE_E_INVALID;
  const int le = 2 * li + dir;
  const size_t D = type == 0 ? 13 + w->cfg.CS : 14 + 2 * w->cfg.CS;
  const DevBuf &A = type == 0 ? w->AtA_p : w->AtA_g, &b = type == 0 ? w->Atb_p : w->Atb_g,
               &st = type == 0 ? w->stats_p : w->stats_g;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  if (AtA)
    SAGE_HIP(hipMemcpy(AtA, A.as<float>() + (size_t)le * D * D, D * D * sizeof(float), hipMemcpyDeviceToHost));
  if (Atb)
    SAGE_HIP(hipMemcpy(Atb, b.as<float>() + (size_t)le * D, D * sizeof(float), hipMemcpyDeviceToHost));
  float s2[2];
  SAGE_HIP(hipMemcpy(s2, st.as<float>() + (size_t)le * 2, 2 * sizeof(float), hipMemcpyDeviceToHost));
  if (err)
    *err = s2[0];
  if (n_in)
    *n_in = s2[1];
  return SAGE_OK;
}

// ------------------------------------------------------------------------------------------------
// f2: the batched per-Values prepass behind the gtsam factors.  ISAM2 asks every factor of the window for
// linearize(values) / error(values) one at a time with the SAME Values; the reference answers each with its own kernel
// launches, .item() syncs and a NearestPsd (photometric_factor.cpp:72-219, geometric_factor.cpp:41-233).  Here the first
// factor that sees new values triggers ONE sage_window_linearize (or sage_window_error) for the whole window and one
// device-to-host copy of the per-edge results; every other factor is served from the host cache.
static int local_edge_index(const SageWindow *w, int e)
{
  const int l = e / 2, dir = e % 2;
  for (size_t i = 0; i < w->local_links.size(); ++i)
    if (w->local_links[i] == l)
      return 2 * (int)i + dir;
  return -1;
}

extern "C" int sage_window_prepass(SageWindow *w, const float *pose12, const float *codes, const float *scales,
                                   int jacobians, int *recomputed)
{
  if (!w || !w->finalized || !pose12 || !codes || !scales)
    return SAGE_E_INVALID;
  const int K = w->K, CS = w->cfg.CS;
  SageWindow::FactorCache &fc = w->fc;
  const size_t np = (size_t)K * 12, nc = (size_t)K * CS;
  const bool same = fc.pose.size() == np && std::memcmp(fc.pose.data(), pose12, np * sizeof(float)) == 0 &&
                    std::memcmp(fc.code.data(), codes, nc * sizeof(float)) == 0 &&
                    std::memcmp(fc.scale.data(), scales, (size_t)K * sizeof(float)) == 0;
  if (recomputed)
    *recomputed = 0;
  if (same && (fc.lin || (!jacobians && fc.err)))
    return SAGE_OK; // a cached linearisation also carries the errors (a1 returns the same error as a2)
  int rc;
  if (!same)
  {
    (void)sync_candidate(w);
    fc.lin = fc.err = false;
    fc.psd_mode = -1;
    fc.pose.assign(pose12, pose12 + np);
    fc.code.assign(codes, codes + nc);
    fc.scale.assign(scales, scales + K);
  }
  // the window's CURRENT variables become the requested values (both sets: a later solve starts from them)
  if (std::memcmp(w->pose[0].data(), pose12, np * sizeof(float)) != 0 ||
      std::memcmp(w->code[0].data(), codes, nc * sizeof(float)) != 0 ||
      std::memcmp(w->scale[0].data(), scales, (size_t)K * sizeof(float)) != 0)
  {
    (void)sync_candidate(w);
    for (int s = 0; s < 2; ++s)
    {
      w->pose[s].assign(pose12, pose12 + np);
      w->code[s].assign(codes, codes + nc);
      w->scale[s].assign(scales, scales + K);
    }
    if ((rc = upload_vars(w, 0)) || (rc = upload_vars(w, 1)))
      return rc;
    w->have_lin = false;
  }
  const size_t ne = (size_t)w->n_edges, Dp = 13 + CS, Dg = 14 + 2 * CS;
  if (jacobians)
  {
    if ((rc = sage_window_linearize(w)))
      return rc;
  }
  else if ((rc = sage_window_error(w, 0)))
    return rc;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  auto pull = [&](std::vector<float> &dst, const DevBuf &src, size_t n) -> hipError_t {
    dst.resize(n);
    return n ? hipMemcpy(dst.data(), src.p, n * sizeof(float), hipMemcpyDeviceToHost) : hipSuccess;
  };
  if (w->cfg.use_photo)
  {
    if (jacobians)
    {
      SAGE_HIP(pull(fc.Ap, w->AtA_p, ne * Dp * Dp));
      SAGE_HIP(pull(fc.bp, w->Atb_p, ne * Dp));
    }
    SAGE_HIP(pull(fc.sp, w->stats_p, ne * 2));
  }
  if (w->cfg.use_geo)
  {
    if (jacobians)
    {
      SAGE_HIP(pull(fc.Ag, w->AtA_g, ne * Dg * Dg));
      SAGE_HIP(pull(fc.bg, w->Atb_g, ne * Dg));
    }
    SAGE_HIP(pull(fc.sg, w->stats_g, ne * 2));
  }
  fc.lin = jacobians != 0;
  fc.err = true;
  if (recomputed)
    *recomputed = 1;
  return SAGE_OK;
}

extern "C" int sage_window_factor_error(const SageWindow *w, int type, int e, double *err_out)
{
  if (!w || !w->finalized || (type != 0 && type != 1) || !err_out)
    return SAGE_E_INVALID;
  const SageWindow::FactorCache &fc = w->fc;
  if (!fc.err)
    return SAGE_E_STATE;
  const int le = local_edge_index(w, e);
  const std::vector<float> &st = type == 0 ? fc.sp : fc.sg;
  if (le < 0 || (size_t)le * 2 + 1 >= st.size())
    return SAGE_E_INVALID;
  *err_out = (double)st[(size_t)le * 2];
  return SAGE_OK;
}

extern "C" int sage_window_factor(const SageWindow *w, int type, int e, int psd_mode, double *G_out, double *g_out,
                                  double *f_out, int *dims_out, int *nkeys_out)
{
  if (!w || !w->finalized || (type != 0 && type != 1))
    return SAGE_E_INVALID;
  const SageWindow::FactorCache &fc = w->fc;
  if (!fc.lin)
    return SAGE_E_STATE;
  const int le = local_edge_index(w, e);
  const int CS = w->cfg.CS;
  const size_t D = type == 0 ? 13 + CS : 14 + 2 * CS;
  const std::vector<float> &A = type == 0 ? fc.Ap : fc.Ag, &b = type == 0 ? fc.bp : fc.bg, &st = type == 0 ? fc.sp : fc.sg;
  if (le < 0 || ((size_t)le + 1) * D * D > A.size())
    return SAGE_E_INVALID;
  if (f_out)
    *f_out = (double)st[(size_t)le * 2];
  const std::vector<double> &Cc = type == 0 ? fc.Cp : fc.Cg;
  if (fc.psd_mode == psd_mode && ((size_t)le + 1) * D * D <= Cc.size()) // prepared on the host threads already
    return sage_factor_cut_blocks(type, CS, Cc.data() + (size_t)le * D * D, b.data() + (size_t)le * D, G_out, g_out,
                                  dims_out, nkeys_out);
  return sage_factor_hessian_blocks(type, CS, A.data() + (size_t)le * D * D, b.data() + (size_t)le * D, psd_mode, G_out,
                                    g_out, dims_out, nkeys_out);
}

// NearestPsd of EVERY cached factor on `n_threads` host threads (0 = a quarter of the host's hardware threads, at most 64): the per-factor
// projection is the host cost of the gtsam path (an SVD / eigen-decomposition of a 45 x 45 and a 78 x 78 matrix per link
// direction, photometric_factor.cpp:142-149) -- ISAM2 pays it factor by factor, here it is paid once per Values in
// parallel and sage_window_factor only cuts blocks afterwards.
extern "C" int sage_window_prepare_factors(SageWindow *w, int psd_mode, int n_threads)
{
  if (!w || !w->finalized || psd_mode < 0 || psd_mode > 2)
    return SAGE_E_INVALID;
  SageWindow::FactorCache &fc = w->fc;
  if (!fc.lin)
    return SAGE_E_STATE;
  if (fc.psd_mode == psd_mode)
    return SAGE_OK;
  const int CS = w->cfg.CS;
  const size_t Dp = 13 + CS, Dg = 14 + 2 * CS;
  const size_t nep = fc.Ap.size() / (Dp * Dp), neg = fc.Ag.size() / (Dg * Dg);
  fc.Cp.assign(nep * Dp * Dp, 0.0);
  fc.Cg.assign(neg * Dg * Dg, 0.0);
  const size_t total = nep + neg;
  if (n_threads <= 0)
    n_threads = (int)std::min<unsigned>(64u, std::max(1u, std::thread::hardware_concurrency() / 4)); // a quarter of the host, <= 64
  n_threads = (int)std::min<size_t>((size_t)n_threads, std::max<size_t>(1, total));
  std::atomic<size_t> next{0};
  std::atomic<int> bad{0};
  auto work = [&]() {
    for (;;)
    {
      // geometric factors first: they are the long jobs (78 x 78)
      const size_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= total)
        return;
      int rc;
      if (i < neg)
        rc = sage_factor_psd(1, CS, fc.Ag.data() + i * Dg * Dg, psd_mode, fc.Cg.data() + i * Dg * Dg);
      else
        rc = sage_factor_psd(0, CS, fc.Ap.data() + (i - neg) * Dp * Dp, psd_mode, fc.Cp.data() + (i - neg) * Dp * Dp);
      if (rc)
        bad.store(rc, std::memory_order_relaxed);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < n_threads; ++t)
    th.emplace_back(work);
  work();
  for (auto &t : th)
    t.join();
  if (bad.load())
    return bad.load();
  fc.psd_mode = psd_mode;
  return SAGE_OK;
}

// ------------------------------------------------------------------------------------------------
// Pipelined LM iteration (single-rank windows, opt-in: SAGE_PIPELINE=1).  In the classic sequence the host
// factorisation starts when the whole window has been linearised, finalised, assembled and scattered.  Here
//   * the geometric linearize runs first; the photometric one second, over the links from BOTH ENDS of the window
//     inwards (link 0, link n-1, link 1, ...), and it signals every finished link through a pinned flag;
//   * the two-halves factorisation eliminates keyframes 0,1,2,.. on one core and K-1,K-2,.. on the other, so both
//     cores' next rows become final at the same pace.  Rows are grouped in chunks of "pair rows"; when a thread is
//     about to need a chunk it waits for the links touching its keyframes and launches, on a second stream, the
//     per-edge finalize of the newly finished links, the assembly of the chunk's blocks and their scatter to pinned
//     host memory (tickets) -- while the photometric kernel keeps linearising the links further inside;
//   * after the last link only the innermost rows, the separator and the back substitution remain.
// Links added in keyframe order (the mapper's temporal links) pipeline well; a loop closure prevents the two-halves
// split in the first place (plan_blocks) and with it this path.
// Measured (MI355X + EPYC 9575F, K = 64 headline window, same box, alternating runs): 2.20 / 2.23 ms per step against
// 2.30 / 2.24 classic -- the host is done 0.28 ms after the last link instead of 0.44, but the photometric kernel
// pays 0.09 ms for it (0.03 the write-through records and counters, 0.06 the small kernels squeezing in next to its
// workgroups: 0.92 instead of 0.84 ms per launch).  Opt-in: 2-3 % of the step for 9 % of the dominant kernel's
// roofline fraction is not a trade the default should make -- and since the host factorisation got faster (0.29 ms)
// the two paths are within 1 % of each other (2.07 vs 2.09 ms per step with two chunks of pair rows).
// ------------------------------------------------------------------------------------------------
static bool pipe_wanted(const SageWindow *w)
{
  return w->world == 1 && w->cfg.use_photo && sage::env_flag("SAGE_PIPELINE") && !sage::env_flag("SAGE_DEVICE_SOLVE") &&
         !sage::env_flag("SAGE_HOST_SOLVE");
}

// the order the photometric work list walks the local links in: both ends inwards
static std::vector<int> pipe_link_sequence(int nl, int *mid)
{
  // runs of `block` links alternately from the low and the high end (not link by link: consecutive links share
  // keyframes, and the workgroups in flight should keep sharing them in the caches)
  int block = 3;
  if (const char *e = getenv("SAGE_PIPE_BLOCK"))
    block = std::max(1, atoi(e));
  std::vector<int> seq;
  int a = 0, b = nl - 1;
  while (a <= b)
  {
    for (int i = 0; i < block && a <= b; ++i)
      seq.push_back(a++);
    for (int i = 0; i < block && a <= b; ++i)
      seq.push_back(b--);
  }
  if (mid)
    *mid = a; // links [0, a) were walked upwards, [a, nl) downwards
  return seq;
}

static int pipe_setup(SageWindow *w, const std::vector<int32_t> &group_of_work)
{
  w->pipe_enabled = false;
  const int K = w->K, nl = (int)w->local_links.size();
  int n1 = 0, n2 = 0, npo = 0;
  const int32_t *pos = nullptr, *perm = nullptr, *pair_off = nullptr;
  if (!pipe_wanted(w) || !w->solver || nl < 2 || !solver_split_info(w->solver, &n1, &n2, &pos, &perm, &pair_off, &npo))
    return SAGE_OK;
  int rc;
  SAGE_HIP(hipGetDevice(&w->device));
  if (!w->pstream)
  {
    // highest priority: the post-processing kernels are tiny and on the critical path of the host factorisation
    int lo = 0, hi = 0;
    SAGE_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    SAGE_HIP(hipStreamCreateWithPriority(&w->pstream, hipStreamNonBlocking, hi));
  }
  if (!w->ev_geo)
    SAGE_HIP(hipEventCreateWithFlags(&w->ev_geo, hipEventDisableTiming));
  // group (= local link) sizes in photometric work items
  std::vector<int32_t> total(nl, 0);
  for (int32_t g : group_of_work)
    total[g] += 1;
  if ((rc = upload(w->pipe_group, group_of_work, w->stream)) || (rc = upload(w->pipe_total, total, w->stream)) ||
      (rc = w->pipe_cnt.reserve((size_t)nl * sizeof(int32_t))))
    return rc;
  if (w->pipe_flags)
    (void)hipHostFree(w->pipe_flags);
  SAGE_HIP(hipHostMalloc(reinterpret_cast<void **>(&w->pipe_flags), (size_t)nl * sizeof(unsigned), hipHostMallocDefault));
  std::memset(w->pipe_flags, 0, (size_t)nl * sizeof(unsigned));
  // chunks of pair rows.  Every chunk costs one round of small launches (~50 us from "links done" to "rows on the
  // host") and disturbs the photometric kernel a little, so the early ones are long (both cores have slack under the
  // device's shadow) and the last ones short (what is left after the last link = latency + the innermost rows + the
  // separator + the back substitution).  SAGE_PIPE_ROWS=n: uniform chunks of n pair rows.
  const int T = std::max(n1, n2); // pair rows; npo == T + 2
  if (npo != T + 2)
    return SAGE_OK;
  std::vector<int> sizes;
  if (const char *e = getenv("SAGE_PIPE_ROWS"))
  {
    const int rows = std::max(1, atoi(e));
    for (int r = 0; r < T; r += rows)
      sizes.push_back(std::min(rows, T - r));
  }
  else
  {
    const double frac[] = {0.70}; // (then the rest: with the 0.29 ms factorisation two chunks of pair rows measured best)
    int left = T;
    for (double f : frac)
    {
      const int n = std::min(left, std::max(1, (int)(f * T + 0.5)));
      if (n > 0)
        sizes.push_back(n);
      left -= n;
    }
    if (left > 0)
      sizes.push_back(left);
  }
  // links touching a keyframe, as the extreme indices on the low / high side of the both-ends sequence
  int mid = 0; // links [0, mid) are walked upwards, [mid, nl) downwards
  (void)pipe_link_sequence(nl, &mid);
  std::vector<int> lo_of_kf(K, -1), hi_of_kf(K, nl);
  for (int li = 0; li < nl; ++li)
  {
    const auto &l = w->links[w->local_links[li]];
    for (int k : {l.first, l.second})
    {
      if (li < mid)
        lo_of_kf[k] = std::max(lo_of_kf[k], li);
      else
        hi_of_kf[k] = std::min(hi_of_kf[k], li);
    }
  }
  w->pipe_chunks.clear();
  w->pipe_chunk_of_pos.assign(K, 0);
  std::vector<int32_t> blk_list;
  const int nlinks = (int)w->links.size();
  int need_lo = -1, need_hi = nl, t0 = 0;
  auto add_chunk = [&](const std::vector<int> &positions, int ord_first, int ord_count, bool last) {
    SageWindow::PipeChunk ch{};
    ch.ord_first = ord_first;
    ch.ord_count = ord_count;
    std::vector<char> in_chunk(K, 0);
    for (int q : positions)
    {
      const int k = perm[q];
      need_lo = std::max(need_lo, lo_of_kf[k]);
      need_hi = std::min(need_hi, hi_of_kf[k]);
      w->pipe_chunk_of_pos[q] = (int)w->pipe_chunks.size();
      in_chunk[q] = 1;
    }
    if (last)
    {
      need_lo = mid - 1;
      need_hi = mid;
    }
    ch.need_lo = need_lo;
    ch.need_hi = need_hi;
    ch.blk_first = (int)blk_list.size();
    for (int q : positions)
      blk_list.push_back(perm[q]); // diagonal block + gradient of the keyframe
    for (int l = 0; l < nlinks; ++l)
    {
      // a link block sits in the row of the endpoint that is eliminated later
      const int row = std::max(pos[w->links[l].first], pos[w->links[l].second]);
      if (in_chunk[row])
        blk_list.push_back(K + l);
    }
    if (last)
      blk_list.push_back(K + nlinks); // error / inlier totals
    ch.blk_count = (int)blk_list.size() - ch.blk_first;
    w->pipe_chunks.push_back(ch);
  };
  for (int rows : sizes)
  {
    const int t1 = std::min(T, t0 + rows);
    std::vector<int> positions;
    for (int t = t0; t < t1; ++t)
    {
      if (t < n1)
        positions.push_back(t);
      if (t < n2)
        positions.push_back(n1 + t);
    }
    add_chunk(positions, pair_off[t0], pair_off[t1] - pair_off[t0], false);
    t0 = t1;
  }
  {
    std::vector<int> positions;
    for (int q = n1 + n2; q < K; ++q)
      positions.push_back(q);
    add_chunk(positions, pair_off[T], pair_off[T + 1] - pair_off[T], true);
  }
  if ((rc = upload(w->pipe_blk_list, blk_list, w->stream)))
    return rc;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  w->pipe_enabled = true;
  return SAGE_OK;
}

// enqueue the linearisation of a pipelined iteration: depth maps, geometric linearize (+ its finalize on the second
// stream), photometric linearize with progress signalling.  Finalize / assembly / scatter follow chunk by chunk.
static int pipe_linearize(SageWindow *w)
{
  const SageWindowConfig &c = w->cfg;
  const int H = (int)c.pyr.cam[0].h, W = (int)c.pyr.cam[0].w;
  const int nl = (int)w->local_links.size();
  SAGE_HIP(hipStreamSynchronize(w->pstream)); // (idle unless an earlier iteration failed half way)
  w->pipe_t0 = std::chrono::steady_clock::now();
  SAGE_HIP(hipMemsetAsync(w->pipe_cnt.p, 0, (size_t)nl * sizeof(int32_t), w->stream));
  w->pipe_epoch += 1;
  if (w->pipe_epoch == 0)
    w->pipe_epoch = 1;
  static const bool no_reuse = sage::env_flag("SAGE_NO_DEPTH_REUSE");
  const bool have_depth = w->dpt_set == 0 && !no_reuse;
  SAGE_HIP(launch_depth_batch(w->stream, c.CS, w->depth_items[0].as<DepthItem>(), w->n_depth, H, W, !have_depth,
                              !(have_depth && w->dgrad_valid)));
  w->dpt_set = 0;
  w->dgrad_valid = true;
  if (c.use_geo)
  {
    EdgeOut out{w->AtA_g.as<float>(), w->Atb_g.as<float>(), w->stats_g.as<float>(), w->wide_g.as<double>()};
    LaunchCommon lc = window_lc(w, false);
    prof_attach(w, 1, lc);
    lc.stage = 1;
    SAGE_HIP(launch_geo_linearize(w->stream, c.CS, nullptr, w->gtab[0].as<GeoEdge>(), lc, c.pyr.cam[0], c.eps,
                                  c.geo_loss_param, c.geo_weight, out));
    SAGE_HIP(hipEventRecord(w->ev_geo, w->stream));
    SAGE_HIP(hipStreamWaitEvent(w->pstream, w->ev_geo, 0));
    LaunchCommon lf = window_lc(w, false);
    lf.stage = 2;
    lf.fin_block = 256;
    lf.edge_base = 0;
    lf.edge_count = w->n_edges;
    SAGE_HIP(launch_geo_linearize(w->pstream, c.CS, nullptr, w->gtab[0].as<GeoEdge>(), lf, c.pyr.cam[0], c.eps,
                                  c.geo_loss_param, c.geo_weight, out));
  }
  {
    EdgeOut out{w->AtA_p.as<float>(), w->Atb_p.as<float>(), w->stats_p.as<float>(), w->wide_p.as<double>()};
    LaunchCommon lc = window_lc(w, true, true);
    prof_attach(w, 0, lc);
    lc.stage = 1;
    lc.sig_group = w->pipe_group.as<int32_t>();
    lc.sig_cnt = w->pipe_cnt.as<int32_t>();
    lc.sig_total = w->pipe_total.as<int32_t>();
    lc.sig_flag_host = w->pipe_flags;
    lc.sig_epoch = w->pipe_epoch;
    SAGE_HIP(launch_photo_linearize(w->stream, c.CS, c.FS, nullptr, w->ptab[0].as<PhotoEdge>(), lc, c.pyr,
                                    c.photo_weights, c.eps, out));
  }
  w->pipe_next.store(0, std::memory_order_release);
  w->pipe_fin_lo = 0;
  w->pipe_fin_hi = nl;
  w->have_lin = true;
  w->lin_epoch = w->vars_epoch;
  w->spec_err_valid = false;
  return SAGE_OK;
}

static bool pipe_chunk_ready(const SageWindow *w, int chunk)
{
  const volatile unsigned *f = w->pipe_flags;
  const SageWindow::PipeChunk &ch = w->pipe_chunks[chunk];
  for (int li = w->pipe_fin_lo; li <= ch.need_lo; ++li)
    if (f[li] != w->pipe_epoch)
      return false;
  for (int li = ch.need_hi; li < w->pipe_fin_hi; ++li)
    if (f[li] != w->pipe_epoch)
      return false;
  return true;
}

// post-processing of one chunk on the second stream: finalize the newly finished links' photometric edges, assemble
// the chunk's blocks, scatter its rows to the host.  (256-thread workgroups here and in the finalize kernels: they have
// to find room next to the photometric kernel's resident workgroups -- a 1024-thread workgroup only fits a CU that has
// drained completely.)
static int pipe_launch_chunk(SageWindow *w, int chunk)
{
  const SageWindowConfig &c = w->cfg;
  const SageWindow::PipeChunk &ch = w->pipe_chunks[chunk];
  std::atomic_thread_fence(std::memory_order_acquire);
  EdgeOut out{w->AtA_p.as<float>(), w->Atb_p.as<float>(), w->stats_p.as<float>(), w->wide_p.as<double>()};
  const int ranges[2][2] = {{w->pipe_fin_lo, ch.need_lo + 1}, {ch.need_hi, w->pipe_fin_hi}};
  for (const auto &r : ranges)
    if (r[1] > r[0])
    {
      LaunchCommon lf = window_lc(w, true, true);
      lf.stage = 2;
      lf.fin_block = 256;
      lf.edge_base = 2 * r[0];
      lf.edge_count = 2 * (r[1] - r[0]);
      SAGE_HIP(launch_photo_linearize(w->pstream, c.CS, c.FS, nullptr, w->ptab[0].as<PhotoEdge>(), lf, c.pyr,
                                      c.photo_weights, c.eps, out));
    }
  w->pipe_fin_lo = std::max(w->pipe_fin_lo, ch.need_lo + 1);
  w->pipe_fin_hi = std::min(w->pipe_fin_hi, ch.need_hi);
  AssembleParams ap = window_assemble_params(w);
  ap.blk_list = w->pipe_blk_list.as<int32_t>() + ch.blk_first;
  ap.split = 4;
  hipLaunchKernelGGL(assemble_kernel, dim3(ch.blk_count * ap.split), dim3(256), 0, w->pstream, ap);
  SAGE_HIP(hipGetLastError());
  return solver_pipe_scatter(w->solver, w->pstream, w->packed.as<double>(), w->vars[0].as<float>(), c.CS, ch.ord_first,
                             ch.ord_count);
}

// make sure the chunks up to `chunk` have been launched (and look a little ahead); called by both factorisation threads
static int pipe_ensure(SageWindow *w, int chunk)
{
  if (w->pipe_next.load(std::memory_order_acquire) > chunk)
    return 0;
  static thread_local int device_set = -1;
  if (device_set != w->device)
  {
    if (hipSetDevice(w->device) != hipSuccess) // (the helper thread starts on the default device)
      return 1;
    device_set = w->device;
  }
  while (w->pipe_lock.test_and_set(std::memory_order_acquire))
    __builtin_ia32_pause();
  int rc = 0;
  const int nch = (int)w->pipe_chunks.size();
  int next = w->pipe_next.load(std::memory_order_relaxed);
  while (!rc && next <= chunk)
  {
    // wait for the chunk's links (bounded: the photometric kernel is running)
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (!pipe_chunk_ready(w, next))
    {
      __builtin_ia32_pause();
      if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2000))
      {
        rc = 1;
        break;
      }
    }
    if (rc)
      break;
    const auto t1 = std::chrono::steady_clock::now();
    rc = pipe_launch_chunk(w, next) ? 1 : 0;
    if (sage::env_flag("SAGE_DEBUG_TIMING"))
      fprintf(stderr, "[sage pipeline] chunk %d launched at %.3f ms (waited %.3f for its links)\n", next,
              1e3 * std::chrono::duration<double>(t1 - w->pipe_t0).count(),
              1e3 * std::chrono::duration<double>(t1 - t0).count());
    w->pipe_t_wait += std::chrono::duration<double>(t1 - t0).count();
    w->pipe_t_launch += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    next += 1;
    w->pipe_next.store(next, std::memory_order_release);
  }
  // look ahead: a later chunk whose links are already done is launched now, so that its rows are on the host when the
  // factorisation gets there
  while (!rc && next < nch && next <= chunk + 1 && pipe_chunk_ready(w, next))
  {
    if (sage::env_flag("SAGE_DEBUG_TIMING"))
      fprintf(stderr, "[sage pipeline] chunk %d launched ahead at %.3f ms\n", next,
              1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - w->pipe_t0).count());
    rc = pipe_launch_chunk(w, next) ? 1 : 0;
    next += 1;
    w->pipe_next.store(next, std::memory_order_release);
  }
  w->pipe_lock.clear(std::memory_order_release);
  return rc;
}

// BlockEnvelope::idle: a thread that waits for tickets launches the next chunk as soon as its links are done (eager
// launches keep the ~0.1 ms a chunk's small kernels need next to the photometric kernel off the critical path)
static void pipe_idle(void *user)
{
  SageWindow *w = static_cast<SageWindow *>(user);
  const int nch = (int)w->pipe_chunks.size();
  if (w->pipe_next.load(std::memory_order_acquire) >= nch || w->pipe_lock.test_and_set(std::memory_order_acquire))
    return;
  static thread_local int device_set = -1;
  bool ok = true;
  if (device_set != w->device)
  {
    ok = hipSetDevice(w->device) == hipSuccess;
    if (ok)
      device_set = w->device;
  }
  int next = w->pipe_next.load(std::memory_order_relaxed);
  if (ok && next < nch && pipe_chunk_ready(w, next))
  {
    if (sage::env_flag("SAGE_DEBUG_TIMING"))
      fprintf(stderr, "[sage pipeline] chunk %d launched eagerly at %.3f ms\n", next,
              1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - w->pipe_t0).count());
    if (pipe_launch_chunk(w, next) == 0)
      w->pipe_next.store(next + 1, std::memory_order_release);
  }
  w->pipe_lock.clear(std::memory_order_release);
}

// BlockEnvelope::before_row of the pipelined factorisation (row = elimination position)
static int pipe_before_row(void *user, int row)
{
  SageWindow *w = static_cast<SageWindow *>(user);
  pipe_idle(user); // between two rows: a chunk whose links have finished in the meantime goes out now
  return pipe_ensure(w, w->pipe_chunk_of_pos[row]);
}

// solve of a pipelined iteration (after pipe_linearize)
static int pipe_solve(SageWindow *w, double damp)
{
  const SageWindowConfig &c = w->cfg;
  int rc = sync_candidate(w);
  if (rc && rc != SAGE_E_NOT_PSD)
    return rc;
  if (w->dpt_set == 1)
    w->dpt_set = -1; // the solve rewrites the candidate set
  if ((rc = solver_pipe_begin(w->solver, damp, c.code_prior_weight, c.scale_prior_weight, c.pose_prior_weight,
                              w->scale_init[0], &w->pose_init[0])))
    return rc;
  static const bool dbgt = sage::env_flag("SAGE_DEBUG_TIMING");
  w->pipe_t_wait = w->pipe_t_launch = 0;
  const auto tp0 = std::chrono::steady_clock::now();
  rc = solver_pipe_factor(w->solver, w->stream, pipe_before_row, pipe_idle, w, w->vars[0].as<float>(),
                          w->vars[1].as<float>(), c.CS);
  if (dbgt)
    fprintf(stderr, "[sage pipeline] factorisation done at %.3f ms\n",
            1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - w->pipe_t0).count());
  if (dbgt)
    fprintf(stderr, "[sage pipelined solve] %.3f ms in the factorisation call: %.3f waiting for links, %.3f launching chunks\n",
            1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count(), 1e3 * w->pipe_t_wait,
            1e3 * w->pipe_t_launch);
  // whatever happened, leave no chunk behind: the packed system / per-edge results must be complete afterwards
  const int nch = (int)w->pipe_chunks.size();
  if (w->pipe_next.load(std::memory_order_acquire) < nch)
  {
    SAGE_HIP(hipStreamSynchronize(w->stream)); // every link is done once the photometric kernel has finished
    const int rl = pipe_ensure(w, nch - 1);
    if (rl)
      return SAGE_E_STATE;
  }
  SAGE_HIP(hipStreamSynchronize(w->pstream));
  if (rc)
    return rc;
  w->cand_pending = true;
  w->last_solver = w->solver;
  return SAGE_OK;
}

// ---- sharded windows, domain-decomposed solve -------------------------------------------------------------------
// priors (a9) as diagonal / gradient additions over all K keyframes; the shard plan applies them on the owner rank only
static void window_priors(const SageWindow *w, std::vector<double> &dadd, std::vector<double> &gadd)
{
  const SageWindowConfig &c = w->cfg;
  const int K = w->K, B = w->B, CS = c.CS;
  dadd.assign((size_t)K * B, 0.0);
  gadd.assign((size_t)K * B, 0.0);
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < CS; ++i)
    {
      dadd[k * B + 6 + i] += c.code_prior_weight;
      gadd[k * B + 6 + i] += c.code_prior_weight * (0.0 - (double)w->code[0][(size_t)k * CS + i]);
    }
  if (c.scale_prior_weight > 0)
  {
    const double s = w->scale[0][0];
    dadd[6 + CS] += c.scale_prior_weight / (s * s);
    gadd[6 + CS] += c.scale_prior_weight / s * (std::log((double)w->scale_init[0]) - std::log(s));
  }
  if (c.pose_prior_weight > 0)
  {
    double loc[6];
    pose_local(&w->pose[0][0], &w->pose_init[0], loc);
    for (int i = 0; i < 6; ++i)
    {
      dadd[i] += c.pose_prior_weight;
      gadd[i] += c.pose_prior_weight * loc[i];
    }
  }
}

// prior error terms of the keyframes THIS rank owns (the other ranks' copies of their variables are stale here)
static double prior_error_owned(const SageWindow *w, int set)
{
  const SageWindowConfig &c = w->cfg;
  double e = 0;
  for (int k = 0; k < w->K; ++k)
  {
    if (sage_shard_keyframe_owner(w->shard, k) != w->rank)
      continue;
    double s2 = 0;
    for (int i = 0; i < c.CS; ++i)
      s2 += (double)w->code[set][(size_t)k * c.CS + i] * w->code[set][(size_t)k * c.CS + i];
    e += c.code_prior_weight * s2 / c.CS;
    if (k == 0 && c.scale_prior_weight > 0)
    {
      const double d = std::log((double)w->scale_init[0]) - std::log((double)w->scale[set][0]);
      e += c.scale_prior_weight * d * d;
    }
    if (k == 0 && c.pose_prior_weight > 0)
    {
      double loc[6];
      pose_local(&w->pose[set][0], &w->pose_init[0], loc);
      for (int i = 0; i < 6; ++i)
        e += c.pose_prior_weight * loc[i] * loc[i];
    }
  }
  return e;
}

__global__ void add_to_double_kernel(double *p, double v) { p[0] += v; }

// local elimination -> all-reduce of the separator system -> separator solve + back substitution of this rank's
// keyframes -> candidate variables of those keyframes.  *lin_error (optional) receives the total error at the
// linearisation point (edge totals ride in the payload tail, prior terms are contributed by their owners).
// Returns SAGE_E_NOT_PSD consistently on every rank (a rank whose local elimination fails poisons the payload).
static int schur_solve(SageWindow *w, double damp, double *lin_error)
{
  const SageWindowConfig &c = w->cfg;
  const int K = w->K, B = w->B, CS = c.CS;
  const size_t np = sage_window_packed_count(w), ns = w->h_sep.size();
  SAGE_HIP(hipMemcpyAsync(w->host_packed.data(), w->packed.p, np * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  SAGE_HIP(hipStreamSynchronize(w->stream));
  std::vector<double> dadd, gadd;
  window_priors(w, dadd, gadd);
  int rc = sage_shard_eliminate(w->shard, w->host_packed.data(), damp, dadd.data(), gadd.data(), w->h_sep.data());
  if (rc && rc != SAGE_E_NOT_PSD)
    return rc;
  if (rc == SAGE_E_NOT_PSD)
  {
    // a failed local elimination is flagged in the spare tail slot [ns-3] (a positive count after the sum: every rank
    // sees it); the separator blocks of this rank are void, the error totals at the linearisation point (tail[0..4],
    // written by sage_shard_eliminate before it factorises) stay finite so that st->error is valid on every rank
    std::fill(w->h_sep.begin(), w->h_sep.end() - 8, 0.0);
    w->h_sep[ns - 3] = 1.0;
  }
  w->h_sep[ns - 4] = prior_error_owned(w, 0);
  SAGE_HIP(hipMemcpyAsync(w->sepbuf.p, w->h_sep.data(), ns * sizeof(double), hipMemcpyHostToDevice, w->stream));
  if (w->allreduce(w->sepbuf.as<double>(), ns, w->allreduce_user))
    return SAGE_E_STATE;
  SAGE_HIP(hipMemcpyAsync(w->h_sep.data(), w->sepbuf.p, ns * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  SAGE_HIP(hipStreamSynchronize(w->stream));
  if (lin_error)
    *lin_error = w->h_sep[ns - 8] + w->h_sep[ns - 7] + w->h_sep[ns - 4];
  if (w->h_sep[ns - 3] > 0.0 || std::isnan(w->h_sep[0]))
    return SAGE_E_NOT_PSD;
  w->delta.assign((size_t)K * B, 0.0);
  rc = sage_shard_solve(w->shard, w->h_sep.data(), w->delta.data());
  if (rc)
    return rc; // SAGE_E_NOT_PSD of the separator system: identical on every rank
  // candidate = retract(current, delta) for the keyframes this rank touches; the others keep their (stale) values
  w->pose[1] = w->pose[0];
  w->code[1] = w->code[0];
  w->scale[1] = w->scale[0];
  for (int k = 0; k < K; ++k)
  {
    if (!sage_shard_keyframe_is_local(w->shard, k))
      continue;
    float d6[6];
    for (int i = 0; i < 6; ++i)
      d6[i] = (float)w->delta[(size_t)k * B + i];
    sage_pose_retract(&w->pose[0][(size_t)k * 12], d6, &w->pose[1][(size_t)k * 12]);
    for (int i = 0; i < CS; ++i)
      w->code[1][(size_t)k * CS + i] = w->code[0][(size_t)k * CS + i] + (float)w->delta[(size_t)k * B + 6 + i];
    w->scale[1][k] = w->scale[0][k] + (float)w->delta[(size_t)k * B + 6 + CS];
  }
  w->cand_pending = false;
  return upload_vars(w, 1);
}

// after a Schur-mode run every rank holds current variables only for the keyframes it touches: sum the owners' copies
extern "C" int sage_window_sync_variables(SageWindow *w)
{
  if (!w || !w->finalized)
    return SAGE_E_STATE;
  if (!w->shard)
    return SAGE_OK; // every rank solves the whole system: nothing to exchange
  if (!w->allreduce)
    return SAGE_E_STATE;
  const int K = w->K, CS = w->cfg.CS, VS = 13 + CS;
  std::vector<double> buf((size_t)K * VS, 0.0);
  for (int k = 0; k < K; ++k)
    if (sage_shard_keyframe_owner(w->shard, k) == w->rank)
    {
      double *b = &buf[(size_t)k * VS];
      for (int i = 0; i < 12; ++i)
        b[i] = w->pose[0][(size_t)k * 12 + i];
      b[12] = w->scale[0][k];
      for (int i = 0; i < CS; ++i)
        b[13 + i] = w->code[0][(size_t)k * CS + i];
    }
  DevBuf d;
  int rc = d.reserve(buf.size() * sizeof(double));
  if (rc)
    return rc;
  SAGE_HIP(hipMemcpyAsync(d.p, buf.data(), buf.size() * sizeof(double), hipMemcpyHostToDevice, w->stream));
  if (w->allreduce(d.as<double>(), buf.size(), w->allreduce_user))
  {
    d.release();
    return SAGE_E_STATE;
  }
  SAGE_HIP(hipMemcpyAsync(buf.data(), d.p, buf.size() * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  SAGE_HIP(hipStreamSynchronize(w->stream));
  d.release();
  for (int s = 0; s < 2; ++s)
    for (int k = 0; k < K; ++k)
    {
      const double *b = &buf[(size_t)k * VS];
      for (int i = 0; i < 12; ++i)
        w->pose[s][(size_t)k * 12 + i] = (float)b[i];
      w->scale[s][k] = (float)b[12];
      for (int i = 0; i < CS; ++i)
        w->code[s][(size_t)k * CS + i] = (float)b[13 + i];
    }
  if ((rc = upload_vars(w, 0)) || (rc = upload_vars(w, 1)))
    return rc;
  return SAGE_OK;
}

__global__ void copy_doubles_kernel(const double *__restrict__ src, double *__restrict__ dst, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    dst[i] = src[i];
}

// SageLmConfig::linearize_at_candidate: one LM iteration in which the candidate is evaluated by the linearize kernels.
// `packed` holds the system at the current estimate (kept from the previous accepted iteration); per evaluation: damped
// solve -> candidate; the current system is set aside (device copy, 3 MB); linearize at the candidate (its finalize
// kernels deliver the error); accepted: the candidate's system IS the next iteration's, nothing is re-evaluated;
// rejected: the saved system comes back and the damping goes up.  Decisions, damping schedule and iterates are those of
// the default sequence; sage_window_get_edge afterwards returns the per-edge results of the LAST evaluation.
static int lm_step_at_candidate(SageWindow *w, SageLmState *st, const SageLmConfig *cfg, bool sharded)
{
  int rc;
  auto clampd = [&](double d) { return std::min(std::max((double)cfg->min_damp, d), (double)cfg->max_damp); };
  const size_t np = sage_window_packed_count(w);
  auto reduce_packed = [&]() -> int {
    if (sharded && w->allreduce(w->packed.as<double>(), np, w->allreduce_user))
      return SAGE_E_STATE;
    w->packed_reduced = true;
    return SAGE_OK;
  };
  // the system at the current estimate is reused only if it is the GLOBAL one: sage_window_linearize / _prepass leave a
  // rank-local `packed` behind on a sharded window (every rank sees the same flags: same call sequence on all ranks)
  if (!(w->have_lin && w->lin_epoch == w->vars_epoch && (!sharded || w->packed_reduced)))
  {
    if ((rc = window_linearize_set(w, 0)) || (rc = reduce_packed()))
      return rc;
  }
  if (!w->spec_err_valid)
  {
    if (sharded)
    { // (the single-rank mirror h_err is not maintained for reduced totals: read the tail of the reduced buffer)
      double t[4];
      SAGE_HIP(hipMemcpyAsync(t, w->packed.as<double>() + np - 4, sizeof(t), hipMemcpyDeviceToHost, w->stream));
      SAGE_HIP(hipStreamSynchronize(w->stream));
      w->spec_error = t[0] + t[1] + prior_error(w, 0);
    }
    else if ((rc = sage_window_total_error(w, 1, &w->spec_error)))
      return rc;
    w->spec_err_valid = true;
  }
  st->error = w->spec_error;
  if ((rc = w->packed_save.reserve(np * sizeof(double))))
    return rc;
  int evals = 0;
  st->accepted = 0;
  const bool mirror = !sharded && w->h_err != nullptr; // single-rank windows: totals mirrored into pinned host memory
  for (;;)
  {
    rc = sage_window_solve(w, st->damp, nullptr);
    bool not_psd = rc == SAGE_E_NOT_PSD;
    if (rc && !not_psd)
      return rc;
    bool replaced = false;
    double cur_tot[4] = {0, 0, 0, 0};
    st->candidate_error = INFINITY;
    if (!not_psd)
    {
      const uint64_t lin_epoch = w->lin_epoch;
      if (mirror)
        std::memcpy(cur_tot, w->h_err, sizeof(cur_tot)); // (the stream drained at the end of the previous evaluation)
      hipLaunchKernelGGL(copy_doubles_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, w->stream,
                         w->packed.as<double>(), w->packed_save.as<double>(), np);
      SAGE_HIP(hipGetLastError());
      if ((rc = window_linearize_set(w, 1)) || (rc = reduce_packed()))
        return rc;
      replaced = true;
      w->lin_epoch = lin_epoch; // (not the current variables' system unless accepted below)
      w->spec_err_valid = true; // spec_error still is the error at the current estimate
      double t[4];
      if (!mirror)
        SAGE_HIP(hipMemcpyAsync(t, w->packed.as<double>() + np - 4, sizeof(t), hipMemcpyDeviceToHost, w->stream));
      rc = sync_candidate(w); // a non-positive pivot of the factorisation shows up here
      if (rc && rc != SAGE_E_NOT_PSD)
        return rc;
      not_psd = rc == SAGE_E_NOT_PSD;
      SAGE_HIP(hipStreamSynchronize(w->stream));
      if (mirror)
        std::memcpy(t, w->h_err, sizeof(t));
      if (!not_psd)
        st->candidate_error = t[0] + t[1] + prior_error(w, 1);
    }
    ++evals;
    if (st->candidate_error < st->error)
    {
      st->accepted = 1;
      if ((rc = sage_window_accept(w)))
        return rc;
      w->lin_epoch = w->vars_epoch; // `packed` is the system at the (new) current estimate
      w->spec_error = st->candidate_error;
      st->damp = clampd(st->damp / cfg->damp_dec_factor);
      break;
    }
    if (replaced)
    {
      // rejected: the system at the current estimate comes back
      hipLaunchKernelGGL(copy_doubles_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, w->stream,
                         w->packed_save.as<double>(), w->packed.as<double>(), np);
      SAGE_HIP(hipGetLastError());
      if (mirror)
        std::memcpy(w->h_err, cur_tot, sizeof(cur_tot));
    }
    const bool give_up = st->damp >= cfg->max_damp || (cfg->max_inner_evals > 0 && evals >= cfg->max_inner_evals);
    st->damp = clampd(st->damp * cfg->damp_inc_factor);
    if (give_up)
      break;
  }
  st->iters += 1;
  return SAGE_OK;
}

extern "C" int sage_window_lm_step(SageWindow *w, SageLmState *st, const SageLmConfig *cfg)
{
  if (!w || !st || !cfg)
    return SAGE_E_INVALID;
  int rc;
  if (w->world > 1 && !w->allreduce)
    return SAGE_E_STATE;
  const bool sharded = w->allreduce != nullptr; // (a hook on a single-rank window is honoured too)
  if (st->iters == 0 && st->damp <= 0)
    st->damp = cfg->init_damp;
  auto clampd = [&](double d) { return std::min(std::max((double)cfg->min_damp, d), (double)cfg->max_damp); };
  const bool pipelined = w->pipe_enabled && !sharded && w->n_edges > 0;
  const bool schur = sharded && w->shard != nullptr;
  // (rank-independent decision: the window's link count, not this rank's share of it -- a rank without links must
  //  issue the same collectives as the others)
  if (cfg->linearize_at_candidate && !pipelined && !schur && !w->links.empty())
    return lm_step_at_candidate(w, st, cfg, sharded);
  if ((rc = pipelined ? pipe_linearize(w) : sage_window_linearize(w)))
    return rc;
  if (sharded && !schur && w->allreduce(w->packed.as<double>(), sage_window_packed_count(w), w->allreduce_user))
    return SAGE_E_STATE;
  int evals = 0;
  st->accepted = 0;
  while (schur)
  {
    // domain-decomposed iteration: the collectives are the separator system and the 4-double error totals
    double lin_error = 0;
    rc = schur_solve(w, st->damp, &lin_error);
    if (rc && rc != SAGE_E_NOT_PSD)
      return rc;
    if (evals == 0)
      st->error = lin_error;
    if (rc == SAGE_E_NOT_PSD)
      st->candidate_error = INFINITY;
    else
    {
      if ((rc = sage_window_error(w, 1)))
        return rc;
      hipLaunchKernelGGL(add_to_double_kernel, dim3(1), dim3(1), 0, w->stream, w->errbuf.as<double>(),
                         prior_error_owned(w, 1));
      if (w->allreduce(w->errbuf.as<double>(), 4, w->allreduce_user))
        return SAGE_E_STATE;
      double t4[4];
      SAGE_HIP(hipMemcpyAsync(t4, w->errbuf.p, sizeof(t4), hipMemcpyDeviceToHost, w->stream));
      SAGE_HIP(hipStreamSynchronize(w->stream));
      st->candidate_error = t4[0] + t4[1];
    }
    ++evals;
    if (st->candidate_error < st->error)
    {
      st->accepted = 1;
      break;
    }
    const bool give_up = st->damp >= cfg->max_damp || (cfg->max_inner_evals > 0 && evals >= cfg->max_inner_evals);
    st->damp = clampd(st->damp * cfg->damp_inc_factor);
    if (give_up)
      break;
  }
  while (!schur)
  {
    // everything of one evaluation is enqueued before the host looks at a number: the error at the linearisation
    // point (tail of the packed buffer) is read together with the candidate's
    rc = (pipelined && evals == 0) ? pipe_solve(w, st->damp) : sage_window_solve(w, st->damp, nullptr);
    if (rc == SAGE_E_NOT_PSD)
    {
      // the damped system has a non-positive pivot: a rejected evaluation (every rank factors the same system and
      // takes this branch together; no error pass, no collective)
      if (evals == 0 && (rc = sage_window_total_error(w, 1, &st->error)))
        return rc;
      st->candidate_error = INFINITY;
      ++evals;
      if (st->damp >= cfg->max_damp || (cfg->max_inner_evals > 0 && evals >= cfg->max_inner_evals))
      {
        st->damp = clampd(st->damp * cfg->damp_inc_factor);
        break;
      }
      st->damp = clampd(st->damp * cfg->damp_inc_factor);
      continue;
    }
    if (rc)
      return rc;
    if ((rc = sage_window_error(w, 1)))
      return rc;
    if (sharded)
    {
      if (w->allreduce(w->errbuf.as<double>(), 4, w->allreduce_user))
        return SAGE_E_STATE;
      hipLaunchKernelGGL(mirror_totals_kernel, dim3(1), dim3(64), 0, w->stream,
                         w->packed.as<double>() + sage_window_packed_count(w) - 4, w->errbuf.as<double>(), w->h_err);
      // a non-positive pivot of the damped system is a REJECTED evaluation (raise the damping), not a hard error; every
      // rank factors the same reduced system, so all of them take this branch together and the number of collectives
      // per iteration stays the same on every rank
      rc = sync_candidate(w);
      if (rc && rc != SAGE_E_NOT_PSD)
        return rc;
      const bool not_psd = rc == SAGE_E_NOT_PSD;
      SAGE_HIP(hipStreamSynchronize(w->stream));
      if (evals == 0)
        st->error = w->h_err[0] + w->h_err[1] + prior_error(w, 0);
      st->candidate_error = not_psd ? INFINITY : w->h_err[4] + w->h_err[5] + prior_error(w, 1);
    }
    else
    {
      if (evals == 0 && (rc = sage_window_total_error(w, 1, &st->error)))
        return rc;
      rc = sage_window_total_error(w, 0, &st->candidate_error);
      if (rc == SAGE_E_NOT_PSD)
        st->candidate_error = INFINITY;
      else if (rc)
        return rc;
    }
    ++evals;
    if (st->candidate_error < st->error)
    {
      st->accepted = 1;
      break;
    }
    if (st->damp >= cfg->max_damp || (cfg->max_inner_evals > 0 && evals >= cfg->max_inner_evals))
    {
      st->damp = clampd(st->damp * cfg->damp_inc_factor);
      break;
    }
    st->damp = clampd(st->damp * cfg->damp_inc_factor);
  }
  if (st->accepted)
  {
    if ((rc = sage_window_accept(w)))
      return rc;
    st->damp = clampd(st->damp / cfg->damp_dec_factor);
  }
  st->iters += 1;
  return SAGE_OK;
}

// n LM iterations in one call (the loop a C++ caller writes around sage_window_lm_step; bench.py uses it so that no Python
// runs between the iterations it times).  trace (optional): n x {error, candidate_error, accepted, damp after the step}.
// Stops early on an error code; *done (optional) = iterations completed.
extern "C" int sage_window_lm_run(SageWindow *w, SageLmState *st, const SageLmConfig *cfg, int n, double *trace, int *done)
{
  if (!w || !st || !cfg || n < 0)
    return SAGE_E_INVALID;
  int i = 0, rc = SAGE_OK;
  for (; i < n; ++i)
  {
    if ((rc = sage_window_lm_step(w, st, cfg)))
      break;
    if (trace)
    {
      trace[4 * i + 0] = st->error;
      trace[4 * i + 1] = st->candidate_error;
      trace[4 * i + 2] = (double)st->accepted;
      trace[4 * i + 3] = st->damp;
    }
  }
  if (done)
    *done = i;
  return rc;
}
