// sage_internal.h -- structs shared by the host runtime and the kernels (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <utility>
#include <vector>

#include "sage_ba.h"

// The photometric kernels form a level's pixel coordinate as fma(p + 0.5, fx_l / fx_0, -0.5) with the quotient precomputed on
// the host; the reference evaluates ((p + 0.5) * fx_l) / fx_0 - 0.5 per pixel (photometric_factor_kernels.cpp:101-103,
// :142-144).  The two agree bit for bit when the quotient is a power of two -- every CameraPyramid whose level sizes stay
// even (common/camera_pyramid.h:18-32; the only sizes for which the reference's conv / camera / mask pyramids agree) -- and
// can differ in floor() at texel boundaries otherwise.  r06: such pyramids are no longer refused: the kernels then form the
// coordinate with the reference's own expression (PhotoParams::exact_coord, TrackParams::exact_coord, the source pre-sampler)
// on the texture-path sampler.
static inline bool pyramid_is_dyadic(const SagePyramid &pyr)
{
  for (int l = 0; l < pyr.levels && l < SAGE_MAX_LEVELS; ++l)
  {
    int e;
    const float rx = pyr.cam[l].fx / pyr.cam[0].fx, ry = pyr.cam[l].fy / pyr.cam[0].fy;
    if (!(rx > 0.f) || !(ry > 0.f) || std::frexp(rx, &e) != 0.5f || std::frexp(ry, &e) != 0.5f)
      return false;
  }
  return true;
}


namespace sage
{

constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kFinalizeBlock = 1024; // per-edge finalize kernels: one workgroup per edge, latency bound -> wide workgroups
constexpr int kWaves = kBlock / 64;
constexpr int kTile = 256;           // source pixels per sub-tile (one per lane)

// One directed photometric edge kf0 -> kf1 (a1/a2).  All pointers are device pointers.
struct PhotoEdge
{
  const float *feat0;   // [FS,P]  source pyramid
  const float *feat1;   // [FS,P]  destination pyramid
  const float *grad1;   // [2,FS,P]
  // engine-internal channel-group layout [FS/4][P][4] (float4 per texel per group of 4 channels): one
  // buffer_load_dwordx4 per tap and group, fully coalesced across the wave.  nullptr -> use the reference layout.
  // feat1_pk heads three consecutive arrays of the destination keyframe: feat | fx_l d/dx | fy_l d/dy (the gradient
  // pyramids pre-multiplied by their level's focal lengths)
  const float *feat0_pk, *feat1_pk;
  // pose-independent pre-sampled source features of the source keyframe [L][FS/4][N][4] (what the reference's
  // tracker calls cat_sampled_features_0, camera_tracker.cpp:1104-1123), built once per keyframe -- stored NEGATED (the
  // samplers start their interpolation chain sum_k w_k t_k - f0 from it)
  const float *f0s;
  const float *dpt0;    // [H,W]   s0*(bias0+basis0*code0): depth map of the SOURCE keyframe at the evaluated variables
  // window error pass only: depth map of the DESTINATION keyframe -> the error kernel also forms the geometric edge's
  // error (geometric_factor_kernels.cpp:127-218: same warp, same mask lookup) and no separate geometric launch runs
  const float *dpt1_geo;
  float geo_loss; // > 0: the geometric edge's own Cauchy parameter (GeoEdge::loss_param), else the launch's
  // merged linearize (LaunchCommon::merge_geo_weight): per-pixel hand-over from the geometric kernel of the same
  // (kf0, kf1) pair, [N][4] = {omega, D, dD/dx, dD/dy} -- the photometric kernel folds the geometric edge's code0 blocks
  // into its own contractions (GeoEdge::px_out is the same buffer)
  const float *geo_px;
  const float *bias0;   // [H*W]
  const float *basis0;  // [H*W,CS]
  const float *mask1;   // [H,W]
  const float *homo;    // [N,3]
  const void *loc;      // [N] int64 or int32
  const float *R0, *t0; // world-from-kf0 (9 row-major, 3)
  const float *R1, *t1; // world-from-kf1
  const float *R10, *t10; // optional relative pose; nullptr -> computed in-kernel from (R0,t0),(R1,t1)
  const float *code0;   // [CS]
  const float *scale0;  // 1 float (device) or nullptr -> scale0_val
  float scale0_val;
  int32_t N;
  int32_t loc_is_i64;
};

// One directed geometric edge kf0 -> kf1 (a4/a5).
struct GeoEdge
{
  const float *dpt0;    // [H,W]   s0*(bias0+basis0*code0): depth map of the SOURCE keyframe at the evaluated variables
  const float *bias0;   // [H*W]
  const float *basis0;  // [H*W,CS]
  const float *dpt1;    // [H,W]   s1*(bias1+basis1*code1)
  const float *dgrad1;  // [2,H,W]
  const float *basis1;  // [H,W,CS]
  const float *mask1;   // [H,W]
  const float *homo;    // [N,3]
  const void *loc;      // [N]
  const float *R0, *t0, *R1, *t1, *R10, *t10;
  const float *code0;
  const float *scale0, *scale1; // device scalars or nullptr -> *_val
  float scale0_val, scale1_val;
  int32_t N;
  int32_t loc_is_i64;
  float loss_param; // > 0: this edge's Cauchy parameter (the mapper's geo_loss_param_factor * avg_squared_dpt_bias of the
                    // link's newer keyframe, mapper.cpp:369), else the launch's
  float *px_out;    // merged linearize: [N][4] = {omega, D, dD/dx, dD/dy} per source pixel (PhotoEdge::geo_px)
};

// Tracker edge (a3): relative pose only, pre-sampled source features.
struct TrackEdge
{
  const float *feat0s;  // [L,N,FS]
  const float *feat1;   // [FS,P]
  const float *grad1;   // [2,FS,P]
  const float *mask1;
  const float *homo;    // [N,3]
  const float *dpts0;   // [N]
  const float *R, *t;   // relative pose T10 (device)
  const float *weights; // [L] device
  float scale0;
  int32_t N;
};

struct WorkItem
{
  int32_t edge;
  int32_t tile;
};

// partial-sum layouts (floats per workgroup)
//   photometric: 40 scalars (37 used) + NT tiles of 16x16 MFMA accumulators (raw [tile][reg][lane])
//     CS=32: tiles {cc00, cc01, cc11, X_lo, X_hi};  CS=16: {cc00, X_lo}
//   geometric:   48 scalars (48 used) + NT tiles
//     t = [kappa*b0 (CS) ; beta (CS)] -> n16 = 2CS/16 column blocks; tiles: upper triangle of T (n16*(n16+1)/2)
//     then n16 cross tiles (rows = y, 16 padded)
constexpr int kPhotoScalars = 40;
constexpr int kGeoScalars = 48;
__host__ __device__ constexpr int photo_tiles(int CS) { return CS == 32 ? 5 : 2; }
// the first photo_cc_tiles(CS) tiles are the code-code blocks (fp32 in the record); the cross tiles X and the 40 scalar
// slots (pose tile, error, inliers) follow the fp32 part as DOUBLES: summed over the workgroup's waves in double and never
// rounded to fp32 again (r04: with the samples in 8 x 8 tile order a wave's / a sub-tile's sums are spatially coherent and
// large against the edge total they cancel into -- their fp32 roundings were the step noise of the small windows)
__host__ __device__ constexpr int photo_cc_tiles(int CS) { return CS == 32 ? 3 : 1; }
__host__ __device__ constexpr int photo_partial_double_offset(int CS) { return kPhotoScalars + photo_tiles(CS) * 256; } // in floats, even
__host__ __device__ constexpr int photo_partial_doubles(int CS) { return kPhotoScalars + (photo_tiles(CS) - photo_cc_tiles(CS)) * 256; }
__host__ __device__ constexpr int photo_partial_floats(int CS) { return photo_partial_double_offset(CS) + 2 * photo_partial_doubles(CS); }
__host__ __device__ constexpr int geo_n16(int CS) { return 2 * CS / 16; }
__host__ __device__ constexpr int geo_tiles(int CS) { return geo_n16(CS) * (geo_n16(CS) + 1) / 2 + geo_n16(CS); }
__host__ __device__ constexpr int geo_partial_floats(int CS) { return kGeoScalars + geo_tiles(CS) * 256; }
constexpr int kTrackScalars = 40; // 28 (7x7 upper) + 7 + err + valid

struct LaunchCommon
{
  const WorkItem *work;      // [n_work]
  const int32_t *edge_first; // [n_edges] first work index of each edge
  const int32_t *edge_tiles; // [n_edges]
  int32_t n_work;
  int32_t n_edges;
  float *partials;           // [n_work][partial_floats]
  int32_t tiles_per_block;   // consecutive kTile sub-tiles per work item (work[i].tile = first sub-tile)
  hipEvent_t ev_start = nullptr, ev_stop = nullptr; // optional: recorded around the main kernel only
  bool packed = false;       // edges carry the channel-group (float4) pyramids
  // linearize launchers: stage 0 = main kernel + per-edge finalize (default), 1 = main kernel only,
  // 2 = finalize only, for the edges [edge_base, edge_base + edge_count)
  int32_t stage = 0, edge_base = 0, edge_count = 0;
  int32_t fin_block = kFinalizeBlock; // threads per workgroup of the finalize kernel
  // photometric error launches: > 0 -> the kernel also evaluates the geometric error of every edge (PhotoEdge::dpt1_geo)
  // with this Cauchy parameter; its partial record is then 4 floats {err_photo, n, err_geo, n} instead of 2
  float fused_geo_loss_param = 0.f;
  // photometric linearize: > 0 -> one partial record per `flush` sub-tiles (edge_first / edge_tiles then count RECORDS:
  // record = edge_first[edge] + tile / flush); 0 -> one record per work item
  int32_t flush = 0;
  // merged linearize of a window's two factor types (r05): > 0 = the geometric factor weight.  Both factor types of a link
  // see the same source pixels, the same warp and -- pixel by pixel -- the same inliers, and every block of the geometric
  // edge that involves code0 only through t0 = kappa*b0 has the form  sum_n (weight_n) x b_n^T  the photometric kernel
  // contracts anyway: the geometric kernel hands {omega, D, grad D} of every pixel to the photometric kernel
  // (GeoEdge::px_out = PhotoEdge::geo_px), drops its 5 code0 tiles (of 15 at CS = 32), and the photometric kernel adds
  // w_g omega kappa^2 to its code-code weight, w_g omega kappa [a; rho] to its cross rows and carries one more cross row
  // (w_g omega kappa D: the scale1-code0 block, read back by the geometric finalize).  The per-edge results are then
  // MIXED (the photometric edge holds the pair's code0 blocks, the geometric edge zeros there); their sum -- what the
  // assembly forms -- is the same normal equations.
  float merge_geo_weight = 0.f;
  // geometric finalize of a merged launch: the photometric launch's partial records (row 8 of the cross tiles)
  const float *merge_photo_partials = nullptr;
  const int32_t *merge_photo_rec_first = nullptr, *merge_photo_rec_count = nullptr;
};

// per-edge results, reference layouts
struct EdgeOut
{
  float *AtA;   // [n_edges][D*D]
  float *Atb;   // [n_edges][D]
  float *stats; // [n_edges][2] = {error, num_inliers}
  // optional (window engine): the same results before their rounding to fp32, [n_edges][D*D + D] doubles per edge
  // (AtA then Atb).  The assembly sums these, so the reference-layout fp32 outputs stay what the operator API returns
  // while the window's normal equations skip one fp32 rounding per entry.
  double *wide = nullptr;
};

// ---- launchers (implemented in the .hip files) ----
hipError_t launch_photo_linearize(hipStream_t s, int CS, int FS, const PhotoEdge *single, const PhotoEdge *table,
                                  const LaunchCommon &lc, const SagePyramid &pyr, const float *weights_host,
                                  float eps, const EdgeOut &out);
hipError_t launch_photo_error(hipStream_t s, int CS, int FS, const PhotoEdge *single, const PhotoEdge *table,
                              const LaunchCommon &lc, const SagePyramid &pyr, const float *weights_host,
                              float eps, float *stats /*[n_edges][2]*/);
hipError_t launch_geo_linearize(hipStream_t s, int CS, const GeoEdge *single, const GeoEdge *table,
                                const LaunchCommon &lc, const SageCamera &cam, float eps, float loss_param,
                                float weight, const EdgeOut &out);
hipError_t launch_geo_error(hipStream_t s, int CS, const GeoEdge *single, const GeoEdge *table,
                            const LaunchCommon &lc, const SageCamera &cam, float eps, float loss_param,
                            float weight, float *stats);
hipError_t launch_track_linearize(hipStream_t s, int dof, int FS, const TrackEdge &edge, const LaunchCommon &lc,
                                  const SagePyramid &pyr, float eps, const EdgeOut &out);
hipError_t launch_track_error(hipStream_t s, int FS, const TrackEdge &edge, const LaunchCommon &lc,
                              const SagePyramid &pyr, float eps, float *stats);
hipError_t launch_depth_and_grad(hipStream_t s, int CS, float *dpt, float *grad, const float *bias,
                                 const float *basis, const float *code, const float *scale_dev, float scale,
                                 int H, int W);
struct DepthItem
{
  const float *bias, *basis, *code, *scale;
  float *dpt, *grad;
};
hipError_t launch_presample_source(hipStream_t s, float *f0s, const float *feat_pk, const float *homo, int N, int FS,
                                   const SagePyramid &pyr);
struct RepackScale // optional per-level factor of launch_repack_groups
{
  int on;
  float s[SAGE_MAX_LEVELS];
};
hipError_t launch_repack_groups(hipStream_t s, float *dst, const float *src, int C, int P, int axis = 0,
                                const SagePyramid *pyr = nullptr, const float *level_scale = nullptr);
// raster-order relayout of sampled locations (producers.hip)
struct SortItem
{
  const long long *loc;
  const float *homo;
  long long *loc_out;
  float *homo_out;
  int n;
};
hipError_t launch_sort_locations(hipStream_t s, const SortItem *items_dev, int K, int max_n, int HW, int *mark_dev,
                                 int *status_dev, int W = 0, int tile_w = 0, int tile_h = 0, const int *pad_flags_dev = nullptr,
                                 int *tiles_out_dev = nullptr, bool keep_marks = false);
// depth maps (and, for the Jacobian pass, their central-difference gradients) of all keyframes of a window
hipError_t launch_depth_batch(hipStream_t s, int CS, const DepthItem *items_dev, int K, int H, int W, bool with_depth,
                              bool with_grad);
hipError_t launch_stats_finalize(hipStream_t s, const LaunchCommon &lc, float *stats, float fallback, float scale);
size_t reproj_scratch_floats(int N, int D);
hipError_t launch_reproj(hipStream_t s, int CS, bool tracker, bool jac, const float *R10, const float *t10, const float *R0,
                         const float *t0, const float *R1, const float *t1, const float *bias0, const float *basis0,
                         const float *code0, const int32_t *loc, const float *dpts0, const float *homo,
                         const float *matched, float scale0, const SageCamera &cam, float eps, float loss_param,
                         float weight, int N, float *scratch, float *AtA, float *Atb, float *stats);
hipError_t launch_cycle_match(hipStream_t s, const float *desc0, const float *desc1, const long long *kp_loc0, int K,
                              int C, int H, int W, float cyc_thresh, long long *raw_matched1, long long *cyc_matched0,
                              int32_t *inlier, int *n_inliers_dev);
size_t mg_scratch_floats(int N, int D);
hipError_t launch_match_geom(hipStream_t s, int mode, int loss, int CS, bool jac, const float *R10, const float *t10,
                             const float *R0, const float *t0, const float *R1, const float *t1, const float *bias0,
                             const float *bias1, const float *basis0, const float *basis1, const float *code0,
                             const float *code1, const float *dpts0, const float *dpts1, const float *homo0,
                             const float *homo1, const int32_t *loc0, const int32_t *loc1, float scale0, float scale1,
                             float loss_param, float weight, int N, float *scratch, float *AtA, float *Atb, float *stats);
hipError_t launch_valid_locations(hipStream_t s, const float *mask, const SageCamera &cam, long long *loc1d, float *homo,
                                  int *n_out_dev);
hipError_t launch_gather_locations(hipStream_t s, const long long *vloc, const float *vhomo, const long long *index_dev,
                                   int n, long long *loc1d, float *homo);
hipError_t launch_depth_samples(hipStream_t s, int CS, float *dpt, const float *bias, const float *basis,
                                const float *code, float scale, const void *loc, int loc_is_i64, int N, int HW);
hipError_t launch_scale_array(hipStream_t s, float *out, const float *in, float mult, int n);
hipError_t launch_gaussian_pyramid_with_grad(hipStream_t s, float *pyr, float *grad, const float *feat,
                                             const float *mask, const SagePyramid &p, int FS, float *scratch_mask);


// ---- device solver of the window's damped normal equations (solve_kernels.hip) ----
struct DeviceSolver;
// SAGE_E_UNSUPPORTED when the block envelope is wider than the LDS panel (the caller keeps the host solver)
int solver_create(DeviceSolver **out, int K, int B, int VS, const std::vector<std::pair<int, int>> &links,
                  hipStream_t stream, bool allow_split = true);
void solver_destroy(DeviceSolver *S);
int solver_run(DeviceSolver *S, hipStream_t stream, const double *packed_dev, const float *vars0, float *vars1, int CS,
               double damp, double code_w, double scale_w, double pose_w, float scale_init0, const float *pose_init0);
// valid after the stream has been synchronised
const float *solver_host_vars(const DeviceSolver *S);
const double *solver_host_delta(const DeviceSolver *S);
double solver_host_step_norm2(const DeviceSolver *S);
int solver_host_status(const DeviceSolver *S);

} // namespace sage
