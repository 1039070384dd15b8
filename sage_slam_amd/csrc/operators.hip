// operators.hip -- workspaces and the per-edge operator API behind the C ABI: the drop-in replacements of the reference's
// df::*_calculate free functions (photometric_factor_kernels.h:9-70, geometric_factor_kernels.h:18-48, reprojection /
// match-geometry kernels) and the keyframe input producers.
#include "runtime_internal.h"

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
  std::memset(ws->host_stats, 0, 16 * sizeof(float));
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
  ws->misc.release();
  ws->dpt0.release();
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

__global__ void ticket_kernel(volatile unsigned *ticket, unsigned epoch)
{
  __threadfence_system();
  *ticket = epoch;
}

int ws_ticket_wait(SageWorkspace *ws)
{
  volatile unsigned *ticket = reinterpret_cast<volatile unsigned *>(ws->host_stats + 8);
  if (++ws->ticket_epoch == 0)
    ws->ticket_epoch = 1;
  const unsigned epoch = ws->ticket_epoch;
  hipLaunchKernelGGL(ticket_kernel, dim3(1), dim3(1), 0, ws->stream, ticket, epoch);
  SAGE_HIP(hipGetLastError());
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (*ticket != epoch)
  {
    __builtin_ia32_pause();
    if ((++spins & 0x3ff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05)
    {
      SAGE_HIP(hipStreamSynchronize(ws->stream)); // (a long queue ahead of this operator: wait the ordinary way)
      break;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return SAGE_OK;
}

// the operator's kernels have written {error, inliers} into the pinned mirror themselves (ws_stats): wait for them
static int ws_fetch_stats(SageWorkspace *ws, float *error_host, float *num_inliers_host)
{
  const int rc = ws_ticket_wait(ws);
  if (rc)
    return rc;
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
  EdgeOut out{AtA_dev, Atb_dev, ws_stats(ws)};
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
  SAGE_HIP(launch_photo_error(ws->stream, CS, FS, &e, nullptr, lc, *pyr, weights_host, eps, ws_stats(ws)));
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
  EdgeOut out{AtA_dev, Atb_dev, ws_stats(ws)};
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
  SAGE_HIP(launch_geo_error(ws->stream, CS, &e, nullptr, lc, *cam, eps, loss_param, weight, ws_stats(ws)));
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
  if ((rc = ws->misc.reserve(reproj_scratch_floats(N, D) * sizeof(float))))
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
  if ((rc = ws->misc.reserve(mg_scratch_floats(N, D) * sizeof(float))))
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

