// tracker.hip -- sage_track_frame: the tracker's LM callbacks wired to the HIP operators (camera_tracker.cpp:1034-1672).
#include "runtime_internal.h"

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

// layout of the pinned evaluation buffer trk_host (floats)
constexpr int kTrkPose = 0;
constexpr int kTrkHostOut = 16; // results start here in trk_host ([0, 12) is the pose the kernels read)

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

// Zero-copy evaluation (r04): the kernels read the pose straight from the pinned mirror and write their 116 result
// floats straight into it; a one-lane kernel posts a ticket behind them and the host spins on it (ws_ticket_wait).  Per evaluation this
// replaces a host-to-device copy, a device-to-host copy and a blocking hipStreamSynchronize (each a 5-10 us round trip
// through the runtime) by two PCIe accesses of the kernels themselves.
static int track_upload_pose(SageWorkspace *ws, const float *pose12)
{
  std::memcpy(ws->trk_host, pose12, 12 * sizeof(float)); // (the previous evaluation has been waited for: nobody reads it)
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
  const int dof = c->dof;
  int rc = track_upload_pose(ws, pose12);
  if (rc)
    return rc;
  const float *R = ws->trk_host + kTrkPose, *t = R + 9; // (pinned, device-visible)
  const float *dp, *kdp;
  if ((rc = track_depths(c, scale, &dp, &kdp)))
    return rc;
  float *host = ws->trk_host + kTrkHostOut;
  float *dA = host, *db = dA + 49, *dA2 = dA + 56, *db2 = dA2 + 49;
  float *st = host + 112;
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
  if ((rc = ws_ticket_wait(ws)))
    return rc;
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
  const float *R = ws->trk_host + kTrkPose, *t = R + 9;
  const float *dp, *kdp;
  if ((rc = track_depths(c, scale, &dp, &kdp)))
    return rc;
  float *host = ws->trk_host + kTrkHostOut;
  float *st = host + 112;
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
  if ((rc = ws_ticket_wait(ws)))
    return rc;
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
  if (!ws->trk_host)
  {
    SAGE_HIP(hipHostMalloc((void **)&ws->trk_host, (kTrkHostOut + 112 + 4 + 12) * sizeof(float), hipHostMallocDefault));
    std::memset(ws->trk_host, 0, (kTrkHostOut + 112 + 4 + 12) * sizeof(float));
  }
  if (dof == 7 && ((prob->use_photo && (rc = ws->trk_dpts.reserve((size_t)prob->N * sizeof(float)))) ||
                   (prob->use_keypoints && (rc = ws->trk_kp_dpts.reserve((size_t)prob->NK * sizeof(float))))))
    return rc;
  rc = sage_track_lm(cfg, dof, track_lin_cb, track_err_cb, &ctx, pose12, scale, final_error, iters, trace, trace_cap,
                     trace_len);
  ws->defer_fetch = false;
  ws->stats_ptr = nullptr;
  return rc;
}

