/*
 * sage_ba.h -- C ABI of the MI355X-native dense bundle-adjustment engine.
 *
 * This is the drop-in boundary for the hot path of lppllppl920/SAGE-SLAM: the
 * per-pixel feature-metric (photometric) and geometric residual/Jacobian
 * evaluation and the Gauss-Newton/LM reduction into the pose x depth-code x
 * scale normal equations.  It replaces the free functions declared in
 *     system/sources/cuda/photometric_factor_kernels.h
 *     system/sources/cuda/geometric_factor_kernels.h
 * (namespace df, templated on CS/FS) and adds the batched window engine the
 * reference does not have (one launch over all factor-graph edges, block-sparse
 * normal equations, edge sharding across GPUs).
 *
 * Conventions
 *   - every `dev` pointer is a plain HIP device pointer (fp32 unless noted);
 *     layouts are the reference's own: feature pyramids
 *     [FS,P] channel-major with levels concatenated fine->coarse, gradient
 *     pyramids [2,FS,P] (x then y), bias [H*W], basis [H*W,CS] row-major,
 *     mask [H,W] float 0/1, homo [N,3], loc1d [N].
 *   - poses are world-from-keyframe; a pose buffer is 12 floats: R row-major
 *     (9) then t (3).  Tangent order [translation(3), rotation(3)], update is a
 *     LEFT multiplication T <- exp(delta)*T  (core/gtsam/gtsam_traits.h:45-70).
 *   - no torch types, no exceptions; every function returns 0 on success or a
 *     negative SAGE_E_* / positive hipError_t code (the reference calls exit()
 *     on a launch error: photometric_factor_kernels.cpp:18-31).
 *   - "no inliers" is NOT an error: error = 10*sum(w) (photometric) or
 *     10*weight (geometric), AtA = Atb = 0 (photometric...cpp:1156-1161,
 *     geometric...cpp:942-947).
 *   - all entry points are re-entrant; a SageWorkspace (stream + scratch) must
 *     not be used from two host threads at once (the reference is called from
 *     up to 4 host threads: deepfactors.cpp:1497-1505 -> one workspace each).
 *
 * The product path has no CPU fallback: without a HIP device every compute
 * entry point fails with the hipError from the runtime.
 */
#ifndef SAGE_BA_H_
#define SAGE_BA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAGE_MAX_LEVELS 8
#define SAGE_POSE_FLOATS 12 /* R (9, row-major) then t (3) */

#define SAGE_OK 0
#define SAGE_E_INVALID (-1)     /* bad argument (null pointer, unsupported CS/FS/L ...) */
#define SAGE_E_UNSUPPORTED (-2) /* combination not instantiated */
#define SAGE_E_NOT_PSD (-3)     /* host solve: matrix not positive definite even after damping */
#define SAGE_E_STATE (-4)       /* call order violated (e.g. solve before linearize) */
#define SAGE_E_NO_OVERLAP (-5)  /* tracker LM: "no overlap between frame to track and keyframe" (camera_tracker.cpp:1515-1519; the reference returns false) */

/* ---- cameras: replaces df::PinholeCamera<float> / df::CameraPyramid<float>
 *      (common/pinhole_camera.h:44-131, common/camera_pyramid.h:18-32) ---- */
typedef struct SageCamera
{
  float fx, fy, cx, cy, w, h;
} SageCamera;

typedef struct SagePyramid
{
  int32_t levels;
  int32_t P; /* total texels over all levels */
  int32_t level_offsets[SAGE_MAX_LEVELS];
  SageCamera cam[SAGE_MAX_LEVELS];
} SagePyramid;

/* CameraPyramid ctor: level i = level i-1 resized to (size_t)(w/2),(size_t)(h/2). Host only. */
/* (Level coordinates: a pyramid whose per-level focal ratios fx_l/fx_0, fy_l/fy_0 are exact powers of two -- every pyramid
 * this function / the reference's CameraPyramid builds while the level sizes stay even, i.e. the only sizes for which the
 * reference's own conv / camera / mask pyramids agree -- is sampled with the host-side quotient (bit-identical to the
 * reference's ((p + 0.5) * fx_l) / fx_0 - 0.5, photometric_factor_kernels.cpp:101-103, there).  r06: any other pyramid (an odd
 * level size, w = 250 -> 125 -> 62, or hand-made focal lengths) is accepted too: the kernels then evaluate the reference's
 * expression per pixel -- true multiplication, true division -- on the texture-path sampler; slower, same results.) */
int sage_camera_pyramid(const SageCamera *base, int levels, SagePyramid *out);

const char *sage_version(void);
const char *sage_error_string(int code);

/* ---- workspace: one per host thread / stream ---- */
typedef struct SageWorkspace SageWorkspace;
/* hip_stream: a hipStream_t (NULL = default stream). */
int sage_workspace_create(void *hip_stream, SageWorkspace **out);
void sage_workspace_destroy(SageWorkspace *ws);

/* =====================================================================
 * Per-edge operator API == the reference's free functions, raw pointers.
 * AtA/Atb are DEVICE outputs (row-major D x D and D), `error` is a HOST
 * output (the call synchronises the stream, like the reference's .item()).
 * ===================================================================== */

/* df::photometric_jac_error_calculate<CS,FS>  (photometric_factor_kernels.h:20-33,
 * .cpp:1061-1164).  D = 13+CS, column order [pose0(6) pose1(6) code0(CS) scale0].
 * weights: HOST float[L] (the reference passes a CPU tensor: photometric_factor.cpp:31-32). */
int sage_photometric_jac_error_calculate(
    SageWorkspace *ws, float *AtA_dev, float *Atb_dev, float *error_host, float *num_inliers_host,
    const float *R10_dev, const float *t10_dev, const float *R0_dev, const float *t0_dev,
    const float *R1_dev, const float *t1_dev,
    const float *bias0_dev, const float *basis0_dev, const float *code0_dev, const float *mask1_dev,
    const int64_t *loc1d_dev, const float *homo_dev,
    const float *feat0_dev, const float *feat1_dev, const float *grad1_dev,
    float scale0, const SagePyramid *pyr, float eps, const float *weights_host,
    int N, int FS, int CS);

/* df::photometric_error_calculate<FS>  (photometric_factor_kernels.h:9-18, .cpp:990-1059) */
int sage_photometric_error_calculate(
    SageWorkspace *ws, float *error_host, float *num_inliers_host,
    const float *R10_dev, const float *t10_dev,
    const float *bias0_dev, const float *basis0_dev, const float *code0_dev, const float *mask1_dev,
    const int64_t *loc1d_dev, const float *homo_dev,
    const float *feat0_dev, const float *feat1_dev,
    float scale0, const SagePyramid *pyr, float eps, const float *weights_host,
    int N, int FS, int CS);

/* df::tracker_photo_jac_error_calculate<FS> (dof = 6, photometric_factor_kernels.h:35-46, .cpp:1166-1245)
 * df::tracker_photo_jac_error_calculate_with_scale<FS> (dof = 7, .h:48-59, .cpp:1247-1325).
 * feat0s = pre-sampled source features [L,N,FS]; weights: DEVICE float[L] (tracker passes a device tensor). */
int sage_tracker_photo_jac_error_calculate(
    SageWorkspace *ws, int dof, float *AtA_dev, float *Atb_dev, float *error_host, float *num_inliers_host,
    const float *R_dev, const float *t_dev, const float *mask1_dev,
    const float *dpts0_dev, const float *homo_dev, const float *feat0s_dev,
    const float *feat1_dev, const float *grad1_dev,
    const SagePyramid *pyr, float scale0, float eps, const float *weights_dev, int N, int FS);

/* df::tracker_photo_error_calculate<FS> (photometric_factor_kernels.h:61-70, .cpp:1327-1384) */
int sage_tracker_photo_error_calculate(
    SageWorkspace *ws, float *error_host, float *num_inliers_host,
    const float *R_dev, const float *t_dev, const float *mask1_dev,
    const float *dpts0_dev, const float *homo_dev, const float *feat0s_dev, const float *feat1_dev,
    const SagePyramid *pyr, float eps, const float *weights_dev, int N, int FS);

/* df::geometric_jac_error_calculate<CS> (geometric_factor_kernels.h:38-48, .cpp:882-950).
 * D = 14+2CS, column order [pose0 pose1 code0 code1 scale0 scale1].  dpt1 = s1*(bias1+basis1*code1) [H,W],
 * dpt_grad1 [2,H,W], basis1 [H,W,CS]; loc1d is int32 here (geometric_factor.cpp:344). */
int sage_geometric_jac_error_calculate(
    SageWorkspace *ws, float *AtA_dev, float *Atb_dev, float *error_host, float *num_inliers_host,
    const float *R10_dev, const float *t10_dev, const float *R0_dev, const float *t0_dev,
    const float *R1_dev, const float *t1_dev,
    const float *bias0_dev, const float *basis0_dev, const float *code0_dev,
    const float *dpt1_dev, const float *dpt_grad1_dev, const float *basis1_dev, const float *mask1_dev,
    const int32_t *loc1d_dev, const float *homo_dev,
    float scale0, float scale1, const SageCamera *cam, float eps, float loss_param, float weight,
    int N, int CS);

/* df::geometric_error_calculate<CS> (geometric_factor_kernels.h:18-24, .cpp:837-880) */
int sage_geometric_error_calculate(
    SageWorkspace *ws, float *error_host, float *num_inliers_host,
    const float *R10_dev, const float *t10_dev,
    const float *bias0_dev, const float *basis0_dev, const float *code0_dev,
    const float *dpt1_dev, const float *mask1_dev, const int32_t *loc1d_dev, const float *homo_dev,
    float scale0, const SageCamera *cam, float eps, float loss_param, float weight, int N, int CS);

/* ---- input producers ---- */
/* UpdateDepth + ComputeSpatialGrad (core/mapping/mapping_utils.h:215-252; caller side of the geometric
 * factor, geometric_factor.cpp:317-347): dpt[H,W] = scale*(bias+basis*code), grad[2,H,W] = scale*centraldiff.
 * The producers are ASYNCHRONOUS on the workspace's stream (unlike the operator calls, which end in a stream
 * synchronise): every device input -- code_dev included -- must stay valid until the stream has passed the call. */
int sage_depth_and_grad(SageWorkspace *ws, float *dpt_dev, float *dpt_grad_dev,
                        const float *bias_dev, const float *basis_dev, const float *code_dev,
                        float scale, int H, int W, int CS);
/* GenerateGaussianPyramidWithGrad (core/mapping/mapper.cpp:1384-1426): feat [FS,H,W], mask [H,W]
 * -> pyr [FS,P], grad [2,FS,P]. */
int sage_gaussian_pyramid_with_grad(SageWorkspace *ws, float *pyr_dev, float *grad_dev,
                                    const float *feat_dev, const float *mask_dev,
                                    const SagePyramid *pyr, int FS);

/* =====================================================================
 * Host-side helpers (pure CPU, no device needed)
 * ===================================================================== */
/* Bind the CALLING thread to the CPUs of the NUMA node the HIP device hangs off (sysfs local_cpulist of its PCI
 * function), intersected with the thread's current affinity mask.  The hybrid window solve reads 3 MB of freshly DMA'd
 * normal equations per LM iteration and the pinned buffers live on the device's node: from the far socket of a
 * two-socket host the iteration is ~10 % slower.  Opt-in (one process per GPU: call it once from the thread that
 * drives the window); returns the number of CPUs bound to, 0 if the topology is not exposed, < 0 on error.
 * r05: the call also watches the host's load for 250 ms (/proc/stat) and keeps to physical cores that are quiet on all their
 * hardware threads -- alone on the NUMA node it narrows the thread's mask to the L3 domain with the most of them -- and the
 * solve's helper threads are placed on quiet cores only (SAGE_BIND_NO_PROBE=1: the plain NUMA-node binding). */
int sage_bind_thread_to_device(int device);
/* Diagnostics of the hybrid solve's thread placement: the CPUs its (up to three) helper threads are pinned to (-1: not
 * placed yet; returns how many entries were written), and how often the placement monitor has moved a helper off a core
 * that another process crowded.
 * r06: the placement monitor -- a background thread that looks at the helpers' run-queue delay and at the load on their
 * cores' other hardware threads every 250 ms and re-pins a crowded helper -- is OPT-IN: SAGE_PLACEMENT_MONITOR=1 in the
 * environment or sage_placement_monitor(1) before the first solve (sage_placement_monitor(0) stops and joins it).  A
 * drop-in library does not edit thread affinities from the background unless asked to. */
int sage_solver_helper_cpus(int *cpus, int n);
int sage_solver_placement_moves(void);
int sage_placement_monitor(int enable);
/* Host threads of the library (r06).  The window solve runs on up to 3 helper threads (+ up to 10 pool workers for
 * loop-closure plans, + the opt-in monitor).  They are started by the first solve that wants them and are JOINABLE:
 * the last sage_window_destroy of the process, sage_shutdown() and process exit (atexit) stop and join them -- nothing
 * is detached, no thread of this library outlives its last window.  sage_shutdown() may be called at any time no
 * solve is in flight (e.g. before dlclose); a later solve starts the threads again.
 * sage_host_threads_running() = how many of them are alive (diagnostic). */
void sage_shutdown(void);
int sage_host_threads_running(void);

/* se3_exp (core/mapping/mapping_utils.h:316-346): R[9], t[3] from omega[3], v[3].  (The void helpers -- this one,
 * sage_pose_retract, sage_lm_config_default -- treat a null argument as a no-op; every int entry point answers a null /
 * out-of-range argument with a status code: tests/test_host_logic.py calls all of them with NULL / 0.) */
void sage_se3_exp(const float *omega, const float *v, float *R, float *t);
/* left retraction T <- exp([v,w]) T (core/gtsam/gtsam_traits.h:45-70; camera_tracker.cpp:491-512);
 * pose/out are 12 floats (R then t), delta = [v(3), w(3)]. */
void sage_pose_retract(const float *pose, const float *delta6, float *out);
/* Higham nearest-PSD of a symmetric-ish n x n matrix in double (the algorithm
 * core/mapping/mapping_utils.h:104-128 intends; see DESIGN.md for the reference's V^T S V slip). */
int sage_nearest_psd(const double *M, int n, double *out);
/* NearestPsd AS THE REFERENCE WROTE IT (mapping_utils.h:104-128): H = V^T diag(sigma) V with the V of Eigen 3.3.9's
 * two-sided Jacobi SVD (its rotation sequence is restated, because the expression depends on V's sign and order
 * conventions), then the LDLT / minimum-eigenvalue bump loop.  Reproduces the reference (fixtures generated with the
 * vendored Eigen) wherever the reference itself is reproducible: see DESIGN.md s6 -- on the gauge-deficient systems
 * the dense factors actually produce, a 1e-15 relative change of the input moves the reference's result by 20 %. */
int sage_nearest_psd_reference(const double *M, int n, double *out);
/* a6 / a7: the HessianFactor blocks of one per-edge system, as PhotometricFactor::linearize
 * (core/gtsam/photometric_factor.cpp:142-218: keys {p0, p1, c0, s0}, block sizes {6, 6, CS, 1}) and
 * GeometricFactor::linearize (core/gtsam/geometric_factor.cpp:120-218: keys {p0, p1, c0, c1, s0, s1}, sizes
 * {6, 6, CS, CS, 1, 1}) cut them: AtA (fp32, row-major D x D, HOST) is widened to double, passed through NearestPsd
 * (psd_mode 0: none, 1: sage_nearest_psd, 2: sage_nearest_psd_reference = as the reference wrote it) and split into
 * the upper-triangular blocks G_ij, i <= j, in the order the reference pushes them (G11 G12 .. G1n G22 ..), each
 * block row-major, concatenated in G_out (sage_factor_block_count doubles); g_out receives Atb widened (D doubles: the
 * g_i are its consecutive segments).  type 0 = photometric (D = 13 + CS), 1 = geometric (D = 14 + 2 CS).
 * dims_out (optional, 6 ints) receives the key dimensions, *nkeys_out their number. */
int sage_factor_block_count(int type, int CS);
int sage_factor_hessian_blocks(int type, int CS, const float *AtA_host, const float *Atb_host, int psd_mode,
                               double *G_out, double *g_out, int32_t *dims_out, int32_t *nkeys_out);
/* solve (A + damp*diag(A)) x = b, column-pivoted Householder QR in fp32 (camera_tracker.cpp:1182-1183). */
int sage_damped_solve_qr_f32(const float *A, const float *b, int n, float damp, float *x);

/* damped solve of the block-sparse normal equations in double (envelope Cholesky):
 *   (H + diag_add + damp*diag(H + diag_add)) delta = g + g_add
 * packed = [diag K*B*B | link nlinks*B*B (rows = links[2l], cols = links[2l+1], links[2l] < links[2l+1]) | g K*B | 4]
 * (host memory, the layout of sage_window_packed_dev); diag_add / g_add (K*B doubles, may be NULL) carry the
 * diagonal priors.  Returns SAGE_E_NOT_PSD if the damped matrix is not positive definite. */
int sage_block_solve(const double *packed_host, int K, int nlinks, const int32_t *links, int B, double damp,
                     const double *diag_add, const double *g_add, double *delta);
/* diagnostics: how many half-factorisations of split windows ran as two stages (a look-ahead thread + the chain
 * through the previous row) in this process so far; the factor is the same bit for bit either way. */
long long sage_solve_lookahead_count(void);

/* ---- tracker LM (core/system/camera_tracker.cpp:1034-1310 / 1312-1672) ---- */
typedef struct SageLmConfig
{
  int max_num_iters;         /* tracking_max_num_iters       = 40   */
  float min_grad_thresh;     /* tracking_min_grad_thresh     = 1e-4 */
  float min_param_inc_thresh;/* tracking_min_param_inc_thresh= 1e-2 */
  float init_damp;           /* tracking_init_damp           = 1e-4 */
  float min_damp, max_damp;  /* tracking_min_max_damp        = 1e-6, 1e-2 */
  float damp_dec_factor;     /* tracking_damp_dec_inc_factor = 10, 100 */
  float damp_inc_factor;
  float jac_update_err_inc_threshold; /* 1e-2 */
  int max_inner_evals;       /* window LM only: cap on candidate evaluations per iteration (0 = reference policy: retry until accepted or max_damp) */
  float no_overlap_error;    /* tracker LM: > 0 -> stop with SAGE_E_NO_OVERLAP once the error at the current estimate is >= this
                              * (TrackFrame without the match-geometry term: 9.9 * sum(photo weights), camera_tracker.cpp:1515); 0 = off */
  int linearize_at_candidate;/* window LM only (sage_window_lm_step).  1: the candidate is evaluated by the LINEARIZE
                              * kernels (error and normal equations from one pass, the system at the current estimate kept
                              * aside): an accepted iteration costs one linearize + one solve and no separate error pass, a
                              * rejected one costs a linearize instead of an error pass.  Same accept / reject rule and the
                              * same iterates as the classic sequence (the two kernels' errors agree to fp32 rounding).
                              * sage_window_get_edge then returns the per-edge results of the LAST evaluation (the
                              * candidate's after a rejected one); the packed system is always the current estimate's.
                              * 0 (default) = automatic: this sequence for windows that are reduced over ranks (one
                              * collective per iteration instead of two and no error pass on the shard's critical path),
                              * the classic sequence on a single rank.  -1: the classic sequence always. */
} SageLmConfig;
void sage_lm_config_default(SageLmConfig *cfg);

/* evaluation back-end of the tracker LM: the product wires these to the HIP kernels above; tests may
 * wire them to anything.  Return 0 on success. */
typedef int (*SageTrackLinearizeFn)(void *ctx, const float *pose12, float scale, float *AtA, float *Atb, float *error);
typedef int (*SageTrackErrorFn)(void *ctx, const float *pose12, float scale, float *error);

typedef struct SageLmTraceEntry
{
  float damp, error, candidate_error;
  int accepted, relinearized;
} SageLmTraceEntry;

/* dof = 6 (TrackNewFrame) or 7 (TrackFrame, + scale).  pose12/scale are in/out.  trace (optional) receives
 * up to trace_cap entries, *trace_len the number written. */
int sage_track_lm(const SageLmConfig *cfg, int dof, SageTrackLinearizeFn lin, SageTrackErrorFn err, void *ctx,
                  float *pose12, float *scale, float *final_error, int *iters,
                  SageLmTraceEntry *trace, int trace_cap, int *trace_len);

/* product wiring of the two callbacks to the HIP kernels: CameraTracker::TrackNewFrame (dof 6) / TrackFrame (dof 7)
 * with the reference's term composition (camera_tracker.cpp:220-374):
 *   dof 6: error / AtA / Atb = photometric (tracker_photo_*)            [use_photo]
 *                            + reprojection (tracker_reproj_*)           [use_keypoints]      (:282-328, :220-248)
 *   dof 7: photometric with scale (tracker_photo_*_with_scale)          [use_photo]
 *                            + match geometry with scale                 [use_keypoints]      (:330-374, :250-280)
 * dof 7 hands over UNSCALED depths (dpt_map_0 / dpt_scale_0, camera_tracker.cpp:1397,1418): every Jacobian and every
 * candidate evaluation multiplies them by the scale being evaluated (guess_scale_0 * unscaled_*_dpts_0, :264,:273,:431,
 * :453) -- the 7th variable moves the depths, not only the Jacobian column.  dof 6 hands over metric depths.
 * The sums are fp32, term by term, like the reference's `AtA += photo_AtA` on fp32 tensors. */
typedef struct SageTrackProblem
{
  SageWorkspace *ws;
  /* photometric term */
  int32_t use_photo;
  const float *mask1_dev, *dpts0_dev, *homo_dev, *feat0s_dev, *feat1_dev, *grad1_dev;
  const float *weights_dev;
  SagePyramid pyr;
  float eps;
  int32_t N, FS;
  /* keypoint term: NK matched keypoints of frame 0 (depths: metric for dof 6, unscaled for dof 7) */
  int32_t use_keypoints, NK;
  const float *kp_dpts0_dev;          /* [NK]   */
  const float *kp_homo0_dev;          /* [NK,3] */
  const float *kp_matched_2d_dev;     /* dof 6: [NK,2] matched pixel locations in frame 1 (reprojection)       */
  const float *kp_matched_dpts1_dev;  /* dof 7: [NK]   depths of the matched points in frame 1 (match geometry) */
  const float *kp_matched_homo1_dev;  /* dof 7: [NK,3] */
  float kp_loss_param;                /* reproj_loss_param_ / match_geom_loss_param                              */
  float kp_weight;                    /* inlier_multiplier_ * {reproj,match_geom}_factor_weight                  */
} SageTrackProblem;
/* Returns SAGE_OK, or SAGE_E_NO_OVERLAP when TrackFrame's zero-overlap exit fires (cfg->no_overlap_error).  trace
 * (optional) as in sage_track_lm. */
int sage_track_frame(const SageLmConfig *cfg, int dof, const SageTrackProblem *prob,
                     float *pose12, float *scale, float *final_error, int *iters,
                     SageLmTraceEntry *trace, int trace_cap, int *trace_len);

/* =====================================================================
 * Batched window engine (no reference counterpart; parity = sum of per-edge results)
 * ===================================================================== */
typedef struct SageWindow SageWindow;

typedef struct SageKeyframeView /* device pointers, reference layouts (core/mapping/frame.h:17-125) */
{
  const float *feat_pyr;  /* [FS,P]   */
  const float *grad_pyr;  /* [2,FS,P] */
  const float *bias;      /* [H*W]    */
  const float *basis;     /* [H*W,CS] */
  const int64_t *loc1d;   /* [N]      */
  const float *homo;      /* [N,3]    */
  int32_t N;
} SageKeyframeView;

typedef struct SageWindowConfig
{
  SagePyramid pyr;
  int32_t FS, CS;
  const float *mask_dev;        /* shared video mask [H,W] */
  float photo_weights[SAGE_MAX_LEVELS];
  float geo_weight, geo_loss_param, eps;
  float code_prior_weight;      /* code_factor_weight (slam_run.flags:104) */
  float scale_prior_weight;     /* init_scale_prior_weight on keyframe 0 */
  float pose_prior_weight;      /* init_pose_prior_weight on keyframe 0 */
  int32_t use_photo, use_geo;
} SageWindowConfig;

int sage_window_create(const SageWindowConfig *cfg, void *hip_stream, SageWindow **out);
void sage_window_destroy(SageWindow *w);
/* variables: pose12 (R,t), code[CS], scale -- HOST pointers, copied. Returns keyframe id >= 0. */
int sage_window_add_keyframe(SageWindow *w, const SageKeyframeView *view, const float *pose12,
                             const float *code, float scale);
/* a link contributes both directed edges of every enabled factor type (mapper.cpp:346-374). */
int sage_window_add_link(SageWindow *w, int kf_a, int kf_b);
/* the Cauchy parameter of ONE link's two geometric edges (before finalize; 0 = the window's geo_loss_param): the mapper
 * derives it per link from the newer keyframe, geo_loss_param_factor * kf->avg_squared_dpt_bias (mapper.cpp:367-373) */
int sage_window_set_link_geo_loss(SageWindow *w, int link, float loss_param);
/* edge sharding for multi-GPU: this process evaluates the contiguous range [rank*2n/world, (rank+1)*2n/world) of the 2n
 * DIRECTED edges (edge 2l = link l a -> b, 2l + 1 = b -> a, links in the order they were added; both factor types of a
 * direction together) -- r05: the two directions of a link may sit on two ranks (42 links on 8 ranks are 5 or 6 each, 84
 * directed edges 10 or 11).  Windows that use the domain-decomposed solve (SAGE_SHARD_SCHUR, default from 256 keyframes)
 * shard by whole links, [rank*n/world, (rank+1)*n/world), as sage_shard_plan_create assumes.  Default (0,1). */
int sage_window_set_shard(SageWindow *w, int rank, int world);
/* must be called once after the last add_keyframe/add_link and before linearize/error. */
int sage_window_finalize(SageWindow *w);

int sage_window_num_keyframes(const SageWindow *w);
int sage_window_num_links(const SageWindow *w);
int sage_window_block_size(const SageWindow *w);       /* B = 7 + CS: [pose6, code CS, scale] */
/* packed normal-equation buffer (device, DOUBLE), the all-reduce payload:
 *   [ diag blocks K*B*B | link blocks nlinks*B*B (row = older kf, col = newer kf) | g K*B | err_photo err_geo n_photo n_geo ]
 * fp32 per-edge results are summed in double, like the reference widens AtA/Atb to double before gtsam adds
 * the factors (core/gtsam/photometric_factor.cpp:305-306); keeping the payload in double keeps the sum exact
 * across ranks too. */
size_t sage_window_packed_count(const SageWindow *w);
double *sage_window_packed_dev(SageWindow *w);
/* number of residuals one linearize evaluates on this shard (E_photo*L*N*FS + E_geo*N) and its algorithmic bytes */
double sage_window_residuals_per_linearize(const SageWindow *w);
double sage_window_bytes_per_linearize(const SageWindow *w);

/* linearize every local edge at the current estimate and assemble the packed buffer (async on the stream). */
int sage_window_linearize(SageWindow *w);
/* total error of every local edge at the CANDIDATE (or current, which = 0/1) variables -> 4-double device
 * buffer [err_photo err_geo n_photo n_geo]; async. */
int sage_window_error(SageWindow *w, int which);
/* (no reference counterpart) run-length tuning of the photometric kernels on the window's own data: times the photometric
 * linearize + error pass with workgroup runs of 4 ... 16 sub-tiles at the current estimate (three timed evaluations per candidate) and keeps
 * the fastest when it beats the static rule's choice by >= 4 %.  Opt-in -- call it once after sage_window_finalize (or set
 * SAGE_AUTOTUNE=1, then finalize calls it); SAGE_PHOTO_TPB pins the run length and turns this into a no-op, as does a sharded
 * window.  Results are bit-reproducible for a given run length, not across run lengths (different fp32 summation order).
 * Side effects: the window is left linearized and error-evaluated at its current variables, and the kernel-time records of
 * sage_window_set_profiling are consumed (the profiling switch itself is restored).
 * Outputs (any may be NULL): the run length now in use, the rule's, and the two timings in ms (0 when nothing was measured). */
int sage_window_tune_runs(SageWindow *w, int *tpb_out, int *tpb_rule_out, float *ms_rule_out, float *ms_best_out);
/* re-apply a run length found by sage_window_tune_runs on an earlier window of the same geometry (1 <= tpb <= 64). */
int sage_window_set_runs(SageWindow *w, int tpb);
double *sage_window_error_dev(SageWindow *w);
/* after (optional) all-reduce of the packed buffer: add priors, D2H, damped solve in double on the host,
 * write the candidate variables (retracted) and upload them.  Returns the predicted step norm. */
int sage_window_solve(SageWindow *w, double damp, double *step_norm);
/* total error (photo + geo + priors) from the (all-reduced) buffers; synchronises. */
int sage_window_total_error(SageWindow *w, int from_linearize, double *err);
int sage_window_accept(SageWindow *w);   /* candidate -> current */
int sage_window_reset(SageWindow *w);    /* restore the variables every keyframe was added with */
/* read back current variables (HOST outputs; any may be NULL) */
int sage_window_get_keyframe(const SageWindow *w, int kf, float *pose12, float *code, float *scale);
int sage_window_set_keyframe(SageWindow *w, int kf, const float *pose12, const float *code, float scale);
/* host copy of the last solve's delta (K*B doubles), for parity tests */
int sage_window_get_delta(const SageWindow *w, double *delta);
/* host copy of per-edge results of the last linearize, reference layouts (for parity tests):
 * type 0 = photometric (D=13+CS), 1 = geometric (D=14+2CS); edge index e in [0, 2*nlinks): link e/2,
 * direction e%2 (0: a->b, 1: b->a).  After sage_window_linearize / sage_window_prepass: one factor type per result,
 * exactly the reference's per-edge AtA / Atb.  After sage_window_lm_step (windows with both factor types): the iteration's
 * MERGED linearize -- the geometric edge's blocks that involve code0 through kappa*b0 (code0-code0, pose-code0, scale0-code0,
 * code0 gradient) ride in the PHOTOMETRIC edge's result of the same pair and read zero in the geometric one; the sum of
 * the two -- what the assembly forms -- is the same normal equations. */
int sage_window_get_edge(const SageWindow *w, int type, int e, float *AtA, float *Atb, float *err, float *n_in);

/* f2 (core/gtsam/photometric_factor.cpp:72-219, geometric_factor.cpp:41-233, mapper.cpp:544-551): the
 * batched per-Values prepass behind the gtsam factors.  ISAM2 calls linearize(values) / error(values) factor by factor
 * with the same Values; the adapter's factor hands the window's values over and the engine evaluates the WHOLE window
 * once per distinct Values:
 *   sage_window_prepass(win, pose12[K][12], codes[K][CS], scales[K], jacobians, &recomputed)
 *       values bit-identical to the cached ones and the cache holds what is asked for -> nothing is launched
 *       (*recomputed = 0); otherwise they become the window's current variables, ONE sage_window_linearize
 *       (jacobians != 0) or ONE sage_window_error (jacobians == 0) runs, and the per-edge results are copied to the host
 *       cache (*recomputed = 1).  A cached linearisation also answers error() at the same values.
 *   sage_window_factor(win, type, e, psd_mode, G, g, f, dims, nkeys)
 *       the HessianFactor of directed edge e (as sage_window_get_edge numbers them) from the cache: blocks and g as
 *       sage_factor_hessian_blocks cuts them (NearestPsd per psd_mode), *f = the factor's error_ (the constant term the
 *       reference passes to gtsam::HessianFactor).  SAGE_E_STATE when the cache holds no linearisation.
 *   sage_window_factor_error(win, type, e, &err)     PhotometricFactor::error / GeometricFactor::error from the cache.
 * Host pointers throughout; type 0 = photometric, 1 = geometric.  One window is not re-entrant: callers on several
 * host threads serialise prepass + factor reads per window (integration/sage_gtsam_prepass.h holds a mutex). */
int sage_window_prepass(SageWindow *w, const float *pose12, const float *codes, const float *scales, int jacobians,
                        int *recomputed);
int sage_window_factor(const SageWindow *w, int type, int e, int psd_mode, double *G_out, double *g_out, double *f_out,
                       int *dims_out, int *nkeys_out);
int sage_window_factor_error(const SageWindow *w, int type, int e, double *err_out);
/* Optional, after a prepass with jacobians: NearestPsd (psd_mode as above) of EVERY cached factor on n_threads host threads
 * (0 = a quarter of the host's hardware threads, at most 64).  The projection -- an SVD / eigen-decomposition of a (13+CS)^2 and a (14+2CS)^2 matrix per link
 * direction -- is the host cost of the gtsam path (photometric_factor.cpp:142-149); ISAM2 pays it factor by factor, this
 * pays it once per Values in parallel, and sage_window_factor with the same psd_mode then only cuts blocks.  The next
 * prepass that recomputes invalidates it. */
int sage_window_prepare_factors(SageWindow *w, int psd_mode, int n_threads);
/* the two halves of sage_factor_hessian_blocks: projection of one factor's AtA (double D x D out), block cutting */
int sage_factor_psd(int type, int CS, const float *AtA, int psd_mode, double *C_out);
int sage_factor_cut_blocks(int type, int CS, const double *C, const float *Atb, double *G_out, double *g_out,
                           int32_t *dims_out, int32_t *nkeys_out);

/* kernel timing with HIP events on the engine's own stream (bench.py's roofline): when enabled every launch of
 * the four hot kernels is bracketed by an event pair.  which: 0 photometric linearize, 1 geometric linearize,
 * 2 photometric error, 3 geometric error.  get_kernel_time synchronises, returns the accumulated milliseconds and
 * launch count since the last reset, and resets them. */
/* on: 0 off, 1 every hot kernel + the phase marks, 2 the photometric linearize only (two event records per iteration
 * instead of eleven: each record is a few microseconds of the stream's time). */
int sage_window_set_profiling(SageWindow *w, int on);
int sage_window_get_kernel_time(SageWindow *w, int which, double *total_ms, int *launches);
/* phases of the LM iterations run through sage_window_lm_step / _lm_run since the last call (profiling on), on the
 * stream's own timeline (HIP events): ms4 = {linearize (depth maps .. assembled system), all-reduce of the system,
 * solve (scatter + host factorisation + retract), error pass}, summed over `iterations`; whatever a step takes beyond
 * their sum is host time with the device idle (accept / reject decision, second all-reduce, launch gaps). */
int sage_window_get_phase_time(SageWindow *w, double *ms4, int *iterations);

/* ---- f3 (first part): sparse reprojection factor with the fair loss ----------------------------------------------
 * Replaces cuda/reprojection_factor_kernels.h:8-46 (reference: reprojection_factor_kernels.cpp):
 *   sage_reprojection_jac_error_calculate   <- reprojection_jac_error_calculate<CS>   (:468-531; kernel :27-213)
 *   sage_reprojection_error_calculate       <- reprojection_error_calculate<CS>       (:417-466; kernel :215-286)
 *   sage_tracker_reproj_jac_error_calculate <- tracker_reproj_jac_error_calculate     (:533-593; kernel :288-366)
 *   sage_tracker_reproj_error_calculate     <- tracker_reproj_error_calculate         (:595-628; kernel :367-415)
 * matched_2d [N,2] are the matched keypoint locations in keyframe 1 (pixels); loc1d is int32 as in the reference;
 * AtA [D,D] / Atb [D] are device arrays, D = 13+CS (mapper: [pose0 pose1 code0 scale0]) or 6 (tracker);
 * error / num_inliers are host scalars (the call synchronises the stream, like the reference's .item<float>()).
 * Without a keypoint in front of the camera: error = 10*weight, AtA = Atb = 0. */
int sage_reprojection_jac_error_calculate(SageWorkspace *ws, float *AtA_dev, float *Atb_dev, float *error_host,
                                          float *num_inliers_host, const float *R10, const float *t10, const float *R0,
                                          const float *t0, const float *R1, const float *t1, const float *bias0,
                                          const float *basis0, const float *code0, const int32_t *loc1d,
                                          const float *homo, const float *matched_2d, float scale0,
                                          const SageCamera *cam, float eps, float loss_param, float weight, int N, int CS);
int sage_reprojection_error_calculate(SageWorkspace *ws, float *error_host, float *num_inliers_host, const float *R10,
                                      const float *t10, const float *bias0, const float *basis0, const float *code0,
                                      const int32_t *loc1d, const float *homo, const float *matched_2d, float scale0,
                                      const SageCamera *cam, float eps, float loss_param, float weight, int N, int CS);
int sage_tracker_reproj_jac_error_calculate(SageWorkspace *ws, float *AtA_dev, float *Atb_dev, float *error_host,
                                            float *num_inliers_host, const float *R, const float *t,
                                            const float *sampled_dpts0, const float *homo, const float *matched_2d,
                                            const SageCamera *cam, float eps, float loss_param, float weight, int N);
int sage_tracker_reproj_error_calculate(SageWorkspace *ws, float *error_host, float *num_inliers_host, const float *R,
                                        const float *t, const float *sampled_dpts0, const float *homo,
                                        const float *matched_2d, const SageCamera *cam, float eps, float loss_param,
                                        float weight, int N);

/* ---- f3 (second part): 3-D match-geometry factors ----------------------------------------------------------------
 * Replaces cuda/match_geometry_factor_kernels.h:9-78 (reference: match_geometry_factor_kernels.cpp).  One entry point
 * pair per reference function family; `loss` replaces the reference's robust_loss_type string:
 *   sage_match_geometry_{jac_,}error_calculate  <- match_geometry_{jac_,}error_calculate<CS>   (:1674-1858, :1565-1672)
 *        D = 14+2CS [pose0 pose1 code0 code1 scale0 scale1]; loss SAGE_LOSS_FAIR | _L2 | _HUBER | _UNBIASED
 *   sage_loop_mg_{jac_,}error_calculate          <- loop_mg_{jac_,}error_calculate             (:1510-1563, :1475-1508)
 *        D = 14 [pose0 pose1 scale0 scale1], unscaled depths of the matched points handed over, fair loss
 *   sage_tracker_match_geom_jac_error_calculate  <- tracker_match_geom_jac_error_calculate     (:1388-1431)   D = 6
 *        with_scale != 0: ..._with_scale (:1433-1473), D = 7 (relative pose, scale0)
 *   sage_tracker_match_geom_error_calculate      <- tracker_match_geom_error_calculate         (:1352-1386)
 * All point arrays have N rows (N >= 1: the reference's mean over zero keypoints is NaN; N < 1 is SAGE_E_INVALID);
 * loc1d arrays are int32; error = weight * mean(per-keypoint loss); AtA/Atb device, error host (synchronises). */
enum { SAGE_LOSS_FAIR = 0, SAGE_LOSS_L2 = 1, SAGE_LOSS_HUBER = 2, SAGE_LOSS_UNBIASED = 3 };
int sage_match_geometry_jac_error_calculate(SageWorkspace *ws, float *AtA_dev, float *Atb_dev, float *error_host,
                                            const float *R10, const float *t10, const float *R0, const float *t0,
                                            const float *R1, const float *t1, const float *bias0, const float *bias1,
                                            const float *basis0, const float *basis1, const float *code0,
                                            const float *code1, const float *homo0, const float *matched_homo1,
                                            const int32_t *loc1d_0, const int32_t *matched_loc1d_1, float scale0,
                                            float scale1, float loss_param, float weight, int loss, int N, int CS);
int sage_match_geometry_error_calculate(SageWorkspace *ws, float *error_host, const float *R10, const float *t10,
                                        const float *bias0, const float *bias1, const float *basis0, const float *basis1,
                                        const float *code0, const float *code1, const float *homo0,
                                        const float *matched_homo1, const int32_t *loc1d_0,
                                        const int32_t *matched_loc1d_1, float scale0, float scale1, float loss_param,
                                        float weight, int loss, int N, int CS);
int sage_loop_mg_jac_error_calculate(SageWorkspace *ws, float *AtA_dev, float *Atb_dev, float *error_host,
                                     const float *R10, const float *t10, const float *R0, const float *t0,
                                     const float *R1, const float *t1, const float *unscaled_dpts0,
                                     const float *matched_unscaled_dpts1, const float *homo0, const float *matched_homo1,
                                     float scale0, float scale1, float loss_param, float weight, int N);
int sage_loop_mg_error_calculate(SageWorkspace *ws, float *error_host, const float *R10, const float *t10,
                                 const float *unscaled_dpts0, const float *matched_unscaled_dpts1, const float *homo0,
                                 const float *matched_homo1, float scale0, float scale1, float loss_param, float weight,
                                 int N);
int sage_tracker_match_geom_jac_error_calculate(SageWorkspace *ws, float *AtA_dev, float *Atb_dev, float *error_host,
                                                const float *R, const float *t, const float *sampled_dpts0,
                                                const float *matched_dpts1, const float *homo0,
                                                const float *matched_homo1, float scale0, float loss_param,
                                                float weight, int with_scale, int N);
int sage_tracker_match_geom_error_calculate(SageWorkspace *ws, float *error_host, const float *R, const float *t,
                                            const float *sampled_dpts0, const float *matched_dpts1, const float *homo0,
                                            const float *matched_homo1, float loss_param, float weight, int N);

/* ---- f4 (matching core): descriptor response argmax with cycle consistency ------------------------------------------
 * Replaces the tensor expression of core/gtsam/match_geometry_factor.cpp:62-97 and
 * core/system/camera_tracker.cpp:608-633: for K keypoints of frame 0 (flat indices kp_loc1d_0, int64) the best match in
 * desc1 [C,H,W] under -sum_c (d0 - d1)^2, the best match of THAT descriptor back in desc0, and the inlier flag
 * |kp - cyc|_2 <= cyc_thresh (pixels).  The K x H*W response maps are never materialised.  Outputs are device arrays
 * of K entries; *n_inliers_host receives the number of flags set (synchronises).  TEASER++ filtering stays on the host. */
int sage_cycle_match(SageWorkspace *ws, const float *desc0_dev, const float *desc1_dev, const int64_t *kp_loc1d_0,
                     int K, int C, int H, int W, float cyc_thresh, int64_t *raw_matched_loc1d_1,
                     int64_t *cyc_matched_loc1d_0, int32_t *inlier_flags, int *n_inliers_host);

/* ---- f1 producers: valid-pixel enumeration and seeded keyframe sampling ------------------------------------
 * sage_valid_locations: core/mapping/mapping_utils.h:254-287 (GenerateValidLocations): flat indices of mask > 0.5 in
 *   ascending order and their normalised homogeneous coordinates ((x-u0)/fx, (y-v0)/fy, 1).  Outputs are device
 *   arrays with room for H*W entries; *n_valid_host receives the count (synchronises the stream).
 * sage_shuffle_indices (host): the permutation mapper.cpp:1326-1333 draws -- std::iota, std::mt19937 seeded with
 *   (long)timestamp, std::shuffle -- through the same standard-library calls.
 * sage_sample_locations: mapper.cpp:1334-1340: the first min(num_samples, n_valid) shuffled indices gathered from the
 *   valid arrays into the keyframe's sampled_locations_1d / sampled_locations_homo.
 * sage_sort_locations (no reference counterpart): raster-orders a keyframe's sampled locations (and their homogeneous
 *   coordinates) out of place.  The factor sums do not depend on the order of the samples, the GPU's vector L1 does: on
 *   the shuffled list the reference keeps (mapper.cpp:1326-1340) the photometric kernels are 5-7x slower than on the
 *   sorted one.  The window engine (sage_window_finalize) does this internally; callers of the per-edge operators
 *   should sort once per keyframe.  *sorted_host = 0 (outputs = copy of the inputs) when a pixel is listed twice;
 *   SAGE_E_INVALID for a location outside H*W.  Synchronises the stream. */
int sage_sort_locations(SageWorkspace *ws, const int64_t *loc1d_dev, const float *homo_dev, int n, int H, int W,
                        int64_t *loc1d_out_dev, float *homo_out_dev, int *sorted_host);
int sage_valid_locations(SageWorkspace *ws, const float *mask_dev, const SageCamera *cam, int64_t *loc1d_dev,
                         float *homo_dev, int *n_valid_host);
int sage_shuffle_indices(int64_t seed, int64_t n, int64_t *idx_host);
int sage_sample_locations(SageWorkspace *ws, const int64_t *valid_loc1d_dev, const float *valid_homo_dev, int n_valid,
                          int64_t seed, int num_samples, int64_t *loc1d_dev, float *homo_dev, int *n_out_host);

/* sharded windows (sage_window_set_shard, world > 1): the in-place SUM all-reduce of a device buffer of n doubles
 * over all ranks, enqueued so that it is ordered with the window's stream (an RCCL ncclAllReduce on that stream, or
 * torch.distributed.all_reduce when the window runs on torch's current stream).  Returns 0 on success.  With the
 * hook installed sage_window_lm_step drives a sharded window as well: the host code between the launches stays
 * native, the hook is entered twice per iteration (packed buffer, 4-double error buffer). */
typedef int (*SageAllReduceFn)(double *dev_buf, size_t n, void *user);
int sage_window_set_allreduce(SageWindow *w, SageAllReduceFn fn, void *user);

/* ---- domain-decomposed solve of a link-sharded window (host, double; csrc/shard_solve.cpp) ---------------------------
 * Link ranges are contiguous (sage_window_set_shard), so a rank's slice of the normal equations is complete for the
 * keyframes only its own links touch: it eliminates those locally (sage_shard_eliminate) and contributes the Schur
 * complement on the keyframes it shares with other ranks to the separator buffer -- the ONE all-reduced payload of an
 * LM iteration (sage_shard_sep_count doubles; 1.2 MB instead of the 3.0 MB packed system at K = 64 on 8 ranks).  After
 * the sum every rank factors the (identical) separator system and back-substitutes its own keyframes
 * (sage_shard_solve).  packed_local_host: this rank's UN-reduced packed normal equations in the layout of
 * sage_window_packed_dev; diag_add / g_add: the K*B diagonal priors of sage_block_solve, applied by the keyframe's
 * designated owner (the lowest rank touching it).  The separator buffer ends with 8 doubles: [0..3] the rank's error
 * / inlier totals at the linearisation point (tail of the packed buffer), [4..7] free for the caller (prior errors).
 * delta (K*B doubles): entries of the keyframes this rank touches are written, the others left alone. */
typedef struct SageShardPlan SageShardPlan;
int sage_shard_plan_create(int K, int nlinks, const int32_t *links, int B, int rank, int world, SageShardPlan **out);
void sage_shard_plan_destroy(SageShardPlan *p);
size_t sage_shard_sep_count(const SageShardPlan *p);
int sage_shard_num_separators(const SageShardPlan *p);
int sage_shard_num_interior(const SageShardPlan *p);
int sage_shard_keyframe_owner(const SageShardPlan *p, int kf);
int sage_shard_keyframe_is_local(const SageShardPlan *p, int kf);
int sage_shard_eliminate(SageShardPlan *p, const double *packed_local_host, double damp, const double *diag_add,
                         const double *g_add, double *sep_out);
int sage_shard_solve(SageShardPlan *p, const double *sep_reduced, double *delta);
/* The same decomposition inside ONE process: keyframe k belongs to domain k*ndomains/K, the newer endpoint of every
 * link that crosses a domain boundary is a separator (the first 3 keyframes of a domain in a temporal window, the far
 * end of a loop closure).  packed_host is then the ASSEMBLED system (domain 0 contributes the separator blocks).
 * sage_block_solve_domains = sage_block_solve computed that way on `ndomains` host threads: for long windows and for
 * loop closures, whose envelope rows would otherwise span the whole window. */
int sage_shard_plan_create_domains(int K, int nlinks, const int32_t *links, int B, int domain, int ndomains,
                                   SageShardPlan **out);
int sage_block_solve_domains(const double *packed_host, int K, int nlinks, const int32_t *links, int B, double damp,
                             const double *diag_add, const double *g_add, int ndomains, double *delta);

/* Native collective: RCCL (backend of record on MI355X: ring / tree over xGMI) bound at run time.  One process per GPU:
 *   rank 0:  sage_rccl_unique_id(id);   broadcast the 128 bytes by any means (MPI, a socket, torch.distributed);
 *   every rank, with its GPU current:   sage_rccl_comm_create(id, rank, world, &comm);
 *   sage_window_use_rccl(win, comm)     -> the window's two all-reduces per LM iteration become
 *        ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, comm, <the window's stream>)   -- no Python, no torch in the loop.
 * `comm` may equally be an ncclComm_t the host application already owns.  SAGE_E_UNSUPPORTED when no librccl is found. */
/* Development aid (bench.py --emulate-shard; csrc/window_dist.hip): adds the share of ranks that are not there after every
 * all-reduce of the window -- rest_dev = n_iterates packed systems (sage_window_packed_count doubles each, device memory,
 * kept alive by the caller), entry i evaluated at the i-th LM iterate since sage_window_reset.  One rank of an N-rank
 * job then walks the job's real trajectory on a one-GPU box with a one-rank communicator. */
int sage_window_emulate_peers(SageWindow *w, const double *rest_dev, int n_iterates);
#define SAGE_RCCL_ID_BYTES 128
int sage_rccl_unique_id(unsigned char *id128);
int sage_rccl_comm_create(const unsigned char *id128, int rank, int world, void **comm_out);
void sage_rccl_comm_destroy(void *comm);
/* what the communicator itself reports (ncclCommCount / ncclCommUserRank): bench.py prints it so that the first multi-GPU
 * line is self-verifying (r06) */
int sage_rccl_comm_info(void *comm, int *ranks_out, int *rank_out);
int sage_window_use_rccl(SageWindow *w, void *nccl_comm);

/* one full LM iteration: linearize -> (all-reduce) -> solve -> error at candidate -> (all-reduce) -> accept/reject
 * (policy of camera_tracker.cpp:1156-1279); windows that are reduced over ranks take the one-collective sequence
 * solve -> linearize at the candidate -> all-reduce -> accept/reject by themselves (SageLmConfig::linearize_at_candidate).
 * Sharded windows need sage_window_set_allreduce / sage_window_use_rccl first. */
typedef struct SageLmState
{
  double damp, error, candidate_error;
  int accepted, iters;
} SageLmState;
int sage_window_lm_step(SageWindow *w, SageLmState *st, const SageLmConfig *cfg);
/* n iterations of sage_window_lm_step in one call; trace (optional): n x {error, candidate_error, accepted, damp after the
 * step}; *done (optional) = iterations completed (an error code ends the run early). */
int sage_window_lm_run(SageWindow *w, SageLmState *st, const SageLmConfig *cfg, int n, double *trace, int *done);
/* the same + step_seconds[n]: host wall time of every iteration (entry of the call / return of the previous iteration ->
 * this iteration's accept / reject decision) */
int sage_window_lm_run_timed(SageWindow *w, SageLmState *st, const SageLmConfig *cfg, int n, double *trace, int *done,
                             double *step_seconds);
/* Sharded windows with the domain-decomposed solve (sage_shard_*: chosen at sage_window_finalize for world > 1 when
 * SAGE_SHARD_SCHUR=1, or by default from K >= 256 keyframes): sage_window_lm_step all-reduces the separator system
 * instead of the packed normal equations, and a rank only updates the keyframes its own links touch.  Call this (a
 * collective: every rank, through the window's all-reduce) before reading variables of other keyframes with
 * sage_window_get_keyframe; a no-op for windows that solve the whole system on every rank. */
int sage_window_sync_variables(SageWindow *w);

#ifdef __cplusplus
}
#endif
#endif /* SAGE_BA_H_ */
