// sage_adapter_keypoints.cpp -- replacement translation unit for
//     system/sources/cuda/reprojection_factor_kernels.cpp     (the 4 functions of reprojection_factor_kernels.h:10-38)
//     system/sources/cuda/match_geometry_factor_kernels.cpp   (the 7 functions of match_geometry_factor_kernels.h:9-66)
// of lppllppl920/SAGE-SLAM -- the sparse keypoint factors.  Like sage_adapter.cpp it lives in the REFERENCE
// tree next to the two headers it includes and defines the same `namespace df` symbols, so
// core/gtsam/{reprojection,match_geometry,loop_mg}_factor.cpp and core/system/camera_tracker.cpp compile unchanged
// against PyTorch-ROCm's libtorch; every call forwards raw device pointers to the C ABI of sage_ba.h (libsage_ba.so).
//
// Semantics kept: fresh device AtA / Atb tensors assigned into the references ([D, D] and [D, 1], like the reference's
// matmul results), `error` on the host (the call synchronises, like .item<float>()), exit(code) on a runtime error,
// int32 location tensors as the reference's packed_accessor32<int> demands (match_geometry_factor.cpp:107-109 converts to
// kInt32 before the call; any other integer type is converted here), and the robust_loss_type STRING dispatch of
// match_geometry_factor_kernels.cpp:1589-1649 / :1704-1782 -- "fair", "L2", "huber", "unbiased"; any other string launches
// no kernel in the reference: the error is 0 and AtA / Atb are zero matrices of the right shape.
#include <cstdio>
#include <cstdlib>
#include <string>

#include <torch/torch.h>
#include <c10/hip/HIPStream.h>

#include "reprojection_factor_kernels.h" // the reference's own declarations, unchanged
#include "match_geometry_factor_kernels.h"
#include "sage_ba.h"

#ifndef DF_CODE_SIZE
#error "DF_CODE_SIZE comes from the reference's build (cuda/CMakeLists.txt)"
#endif

namespace df
{
namespace
{
void chk(int rc, const char *where)
{
  if (rc)
  {
    std::fprintf(stderr, "[sage_adapter] %s: %s (%d)\n", where, sage_error_string(rc), rc);
    std::exit(rc); // the reference: gpuErrchk -> exit(code)
  }
}

SageWorkspace *ws()
{
  thread_local SageWorkspace *w = [] {
    SageWorkspace *p = nullptr;
    chk(sage_workspace_create(c10::hip::getCurrentHIPStream().stream(), &p), "sage_workspace_create");
    return p;
  }();
  return w;
}

SageCamera to_cam(const PinholeCamera<float> &c) // common/pinhole_camera.h:44-131
{
  return SageCamera{c.fx(), c.fy(), c.u0(), c.v0(), (float)c.width(), (float)c.height()};
}

const float *f32(const at::Tensor &t, const char *what)
{
  TORCH_CHECK(t.is_cuda() && t.is_contiguous() && t.scalar_type() == at::kFloat, "sage_adapter: ", what,
              " must be a contiguous fp32 device tensor");
  return t.data_ptr<float>();
}

at::Tensor i32(const at::Tensor &t) { return t.to(at::kInt).contiguous(); }

// robust_loss_type -> SAGE_LOSS_*; -1: a string the reference has no branch for
int loss_of(const std::string &s)
{
  if (s == "fair")
    return SAGE_LOSS_FAIR;
  if (s == "L2")
    return SAGE_LOSS_L2;
  if (s == "huber")
    return SAGE_LOSS_HUBER;
  if (s == "unbiased")
    return SAGE_LOSS_UNBIASED;
  return -1;
}

void fresh(at::Tensor &AtA, at::Tensor &Atb, int D, const at::Tensor &like, bool zero = false)
{
  const auto opts = like.options().dtype(at::kFloat);
  AtA = zero ? torch::zeros({D, D}, opts) : torch::empty({D, D}, opts);
  Atb = zero ? torch::zeros({D, 1}, opts) : torch::empty({D, 1}, opts);
}
} // namespace

// ---------------------------------------------------------------------------------------------- reprojection (f3)
void tracker_reproj_jac_error_calculate(at::Tensor &AtA, at::Tensor &Atb, float &error, const at::Tensor rotation,
                                        const at::Tensor translation, const at::Tensor sampled_dpts_0,
                                        const at::Tensor sampled_locations_homo_0,
                                        const at::Tensor matched_locations_2d_1, const PinholeCamera<float> &camera,
                                        const float eps, const float loss_param, const float weight)
{
  const SageCamera cam = to_cam(camera);
  fresh(AtA, Atb, 6, sampled_locations_homo_0);
  chk(sage_tracker_reproj_jac_error_calculate(ws(), AtA.data_ptr<float>(), Atb.data_ptr<float>(), &error, nullptr,
                                              f32(rotation, "rotation"), f32(translation, "translation"),
                                              f32(sampled_dpts_0, "sampled_dpts"),
                                              f32(sampled_locations_homo_0, "locations_homo"),
                                              f32(matched_locations_2d_1, "matched_locations_2d"), &cam, eps, loss_param,
                                              weight, (int)sampled_locations_homo_0.size(0)),
      "tracker_reproj_jac_error_calculate");
}

float tracker_reproj_error_calculate(const at::Tensor rotation, const at::Tensor translation,
                                     const at::Tensor sampled_dpts_0, const at::Tensor sampled_locations_homo_0,
                                     const at::Tensor matched_locations_2d_1, const PinholeCamera<float> &camera,
                                     const float eps, const float loss_param, const float weight)
{
  const SageCamera cam = to_cam(camera);
  float err = 0.f;
  chk(sage_tracker_reproj_error_calculate(ws(), &err, nullptr, f32(rotation, "rotation"), f32(translation, "translation"),
                                          f32(sampled_dpts_0, "sampled_dpts"),
                                          f32(sampled_locations_homo_0, "locations_homo"),
                                          f32(matched_locations_2d_1, "matched_locations_2d"), &cam, eps, loss_param,
                                          weight, (int)sampled_locations_homo_0.size(0)),
      "tracker_reproj_error_calculate");
  return err;
}

template <int CS>
void reprojection_jac_error_calculate(at::Tensor &AtA, at::Tensor &Atb, float &error, const at::Tensor rotation10,
                                      const at::Tensor translation10, const at::Tensor rotation0,
                                      const at::Tensor translation0, const at::Tensor rotation1,
                                      const at::Tensor translation1, const at::Tensor flatten_dpt_map_bias_0,
                                      const at::Tensor flatten_dpt_jac_code_0, const at::Tensor code_0,
                                      const at::Tensor sampled_locations_1d_0, const at::Tensor sampled_locations_homo_0,
                                      const at::Tensor matched_locations_2d_1, const float scale_0,
                                      const PinholeCamera<float> &camera, const float eps, const float loss_param,
                                      const float weight)
{
  const SageCamera cam = to_cam(camera);
  const at::Tensor loc = i32(sampled_locations_1d_0);
  fresh(AtA, Atb, 13 + CS, sampled_locations_homo_0);
  chk(sage_reprojection_jac_error_calculate(
          ws(), AtA.data_ptr<float>(), Atb.data_ptr<float>(), &error, nullptr, f32(rotation10, "rotation10"),
          f32(translation10, "translation10"), f32(rotation0, "rotation0"), f32(translation0, "translation0"),
          f32(rotation1, "rotation1"), f32(translation1, "translation1"), f32(flatten_dpt_map_bias_0, "dpt_map_bias"),
          f32(flatten_dpt_jac_code_0, "dpt_jac_code"), f32(code_0, "code"), loc.data_ptr<int32_t>(),
          f32(sampled_locations_homo_0, "locations_homo"), f32(matched_locations_2d_1, "matched_locations_2d"), scale_0,
          &cam, eps, loss_param, weight, (int)sampled_locations_homo_0.size(0), CS),
      "reprojection_jac_error_calculate");
}

template <int CS>
float reprojection_error_calculate(const at::Tensor rotation10, const at::Tensor translation10,
                                   const at::Tensor flatten_dpt_map_bias_0, const at::Tensor flatten_dpt_jac_code_0,
                                   const at::Tensor code_0, const at::Tensor sampled_locations_1d_0,
                                   const at::Tensor sampled_locations_homo_0, const at::Tensor matched_locations_2d_1,
                                   const float scale_0, const PinholeCamera<float> &camera, const float eps,
                                   const float loss_param, const float weight)
{
  const SageCamera cam = to_cam(camera);
  const at::Tensor loc = i32(sampled_locations_1d_0);
  float err = 0.f;
  chk(sage_reprojection_error_calculate(ws(), &err, nullptr, f32(rotation10, "rotation10"),
                                        f32(translation10, "translation10"), f32(flatten_dpt_map_bias_0, "dpt_map_bias"),
                                        f32(flatten_dpt_jac_code_0, "dpt_jac_code"), f32(code_0, "code"),
                                        loc.data_ptr<int32_t>(), f32(sampled_locations_homo_0, "locations_homo"),
                                        f32(matched_locations_2d_1, "matched_locations_2d"), scale_0, &cam, eps,
                                        loss_param, weight, (int)sampled_locations_homo_0.size(0), CS),
      "reprojection_error_calculate");
  return err;
}

// ---------------------------------------------------------------------------------------------- match geometry (f3)
float tracker_match_geom_error_calculate(const at::Tensor rotation, const at::Tensor translation,
                                         const at::Tensor sampled_dpts_0, const at::Tensor matched_dpts_1,
                                         const at::Tensor sampled_locations_homo_0,
                                         const at::Tensor matched_locations_homo_1, const float loss_param,
                                         const float weight)
{
  float err = 0.f;
  chk(sage_tracker_match_geom_error_calculate(ws(), &err, f32(rotation, "rotation"), f32(translation, "translation"),
                                              f32(sampled_dpts_0, "sampled_dpts_0"), f32(matched_dpts_1, "matched_dpts_1"),
                                              f32(sampled_locations_homo_0, "locations_homo_0"),
                                              f32(matched_locations_homo_1, "locations_homo_1"), loss_param, weight,
                                              (int)sampled_locations_homo_0.size(0)),
      "tracker_match_geom_error_calculate");
  return err;
}

void tracker_match_geom_jac_error_calculate(at::Tensor &AtA, at::Tensor &Atb, float &error, const at::Tensor rotation,
                                            const at::Tensor translation, const at::Tensor sampled_dpts_0,
                                            const at::Tensor matched_dpts_1, const at::Tensor sampled_locations_homo_0,
                                            const at::Tensor matched_locations_homo_1, const float loss_param,
                                            const float weight)
{
  fresh(AtA, Atb, 6, sampled_locations_homo_0);
  chk(sage_tracker_match_geom_jac_error_calculate(
          ws(), AtA.data_ptr<float>(), Atb.data_ptr<float>(), &error, f32(rotation, "rotation"),
          f32(translation, "translation"), f32(sampled_dpts_0, "sampled_dpts_0"), f32(matched_dpts_1, "matched_dpts_1"),
          f32(sampled_locations_homo_0, "locations_homo_0"), f32(matched_locations_homo_1, "locations_homo_1"), 1.0f,
          loss_param, weight, 0, (int)sampled_locations_homo_0.size(0)),
      "tracker_match_geom_jac_error_calculate");
}

void tracker_match_geom_jac_error_calculate_with_scale(at::Tensor &AtA, at::Tensor &Atb, float &error,
                                                       const at::Tensor rotation, const at::Tensor translation,
                                                       const at::Tensor sampled_dpts_0, const at::Tensor matched_dpts_1,
                                                       const at::Tensor sampled_locations_homo_0,
                                                       const at::Tensor matched_locations_homo_1, const float scale_0,
                                                       const float loss_param, const float weight)
{
  fresh(AtA, Atb, 7, sampled_locations_homo_0);
  chk(sage_tracker_match_geom_jac_error_calculate(
          ws(), AtA.data_ptr<float>(), Atb.data_ptr<float>(), &error, f32(rotation, "rotation"),
          f32(translation, "translation"), f32(sampled_dpts_0, "sampled_dpts_0"), f32(matched_dpts_1, "matched_dpts_1"),
          f32(sampled_locations_homo_0, "locations_homo_0"), f32(matched_locations_homo_1, "locations_homo_1"), scale_0,
          loss_param, weight, 1, (int)sampled_locations_homo_0.size(0)),
      "tracker_match_geom_jac_error_calculate_with_scale");
}

template <int CS>
float match_geometry_error_calculate(const at::Tensor rotation, const at::Tensor translation,
                                     const at::Tensor flatten_dpt_map_bias_0, const at::Tensor flatten_dpt_map_bias_1,
                                     const at::Tensor flatten_dpt_jac_code_0, const at::Tensor flatten_dpt_jac_code_1,
                                     const at::Tensor code_0, const at::Tensor code_1,
                                     const at::Tensor sampled_locations_homo_0, const at::Tensor matched_locations_homo_1,
                                     const at::Tensor sampled_locations_1d_0, const at::Tensor matched_locations_1d_1,
                                     const float scale_0, const float scale_1, const float loss_param, const float weight,
                                     const std::string robust_loss_type)
{
  const int loss = loss_of(robust_loss_type);
  if (loss < 0)
    return 0.f; // (no branch of :1589-1649 taken: weight * mean(zeros))
  const at::Tensor l0 = i32(sampled_locations_1d_0), l1 = i32(matched_locations_1d_1);
  float err = 0.f;
  chk(sage_match_geometry_error_calculate(
          ws(), &err, f32(rotation, "rotation"), f32(translation, "translation"), f32(flatten_dpt_map_bias_0, "bias_0"),
          f32(flatten_dpt_map_bias_1, "bias_1"), f32(flatten_dpt_jac_code_0, "jac_code_0"),
          f32(flatten_dpt_jac_code_1, "jac_code_1"), f32(code_0, "code_0"), f32(code_1, "code_1"),
          f32(sampled_locations_homo_0, "locations_homo_0"), f32(matched_locations_homo_1, "locations_homo_1"),
          l0.data_ptr<int32_t>(), l1.data_ptr<int32_t>(), scale_0, scale_1, loss_param, weight, loss,
          (int)sampled_locations_homo_0.size(0), CS),
      "match_geometry_error_calculate");
  return err;
}

template <int CS>
void match_geometry_jac_error_calculate(at::Tensor &AtA, at::Tensor &Atb, float &error, const at::Tensor rotation10,
                                        const at::Tensor translation10, const at::Tensor rotation0,
                                        const at::Tensor translation0, const at::Tensor rotation1,
                                        const at::Tensor translation1, const at::Tensor flatten_dpt_map_bias_0,
                                        const at::Tensor flatten_dpt_map_bias_1, const at::Tensor flatten_dpt_jac_code_0,
                                        const at::Tensor flatten_dpt_jac_code_1, const at::Tensor code_0,
                                        const at::Tensor code_1, const at::Tensor sampled_locations_homo_0,
                                        const at::Tensor matched_locations_homo_1,
                                        const at::Tensor sampled_locations_1d_0, const at::Tensor matched_locations_1d_1,
                                        const float scale_0, const float scale_1, const float loss_param,
                                        const float weight, const std::string robust_loss_type)
{
  const int loss = loss_of(robust_loss_type);
  if (loss < 0)
  {
    // no branch of :1704-1782 taken: the Jacobian and difference buffers stay zero
    fresh(AtA, Atb, 14 + 2 * CS, sampled_locations_homo_0, true);
    error = 0.f;
    return;
  }
  const at::Tensor l0 = i32(sampled_locations_1d_0), l1 = i32(matched_locations_1d_1);
  fresh(AtA, Atb, 14 + 2 * CS, sampled_locations_homo_0);
  chk(sage_match_geometry_jac_error_calculate(
          ws(), AtA.data_ptr<float>(), Atb.data_ptr<float>(), &error, f32(rotation10, "rotation10"),
          f32(translation10, "translation10"), f32(rotation0, "rotation0"), f32(translation0, "translation0"),
          f32(rotation1, "rotation1"), f32(translation1, "translation1"), f32(flatten_dpt_map_bias_0, "bias_0"),
          f32(flatten_dpt_map_bias_1, "bias_1"), f32(flatten_dpt_jac_code_0, "jac_code_0"),
          f32(flatten_dpt_jac_code_1, "jac_code_1"), f32(code_0, "code_0"), f32(code_1, "code_1"),
          f32(sampled_locations_homo_0, "locations_homo_0"), f32(matched_locations_homo_1, "locations_homo_1"),
          l0.data_ptr<int32_t>(), l1.data_ptr<int32_t>(), scale_0, scale_1, loss_param, weight, loss,
          (int)sampled_locations_homo_0.size(0), CS),
      "match_geometry_jac_error_calculate");
}

float loop_mg_error_calculate(const at::Tensor rotation, const at::Tensor translation,
                              const at::Tensor sampled_unscaled_dpts_0, const at::Tensor matched_unscaled_dpts_1,
                              const at::Tensor sampled_locations_homo_0, const at::Tensor matched_locations_homo_1,
                              const float scale_0, const float scale_1, const float loss_param, const float weight)
{
  float err = 0.f;
  chk(sage_loop_mg_error_calculate(ws(), &err, f32(rotation, "rotation"), f32(translation, "translation"),
                                   f32(sampled_unscaled_dpts_0, "unscaled_dpts_0"),
                                   f32(matched_unscaled_dpts_1, "unscaled_dpts_1"),
                                   f32(sampled_locations_homo_0, "locations_homo_0"),
                                   f32(matched_locations_homo_1, "locations_homo_1"), scale_0, scale_1, loss_param, weight,
                                   (int)sampled_locations_homo_0.size(0)),
      "loop_mg_error_calculate");
  return err;
}

void loop_mg_jac_error_calculate(at::Tensor &AtA, at::Tensor &Atb, float &error, const at::Tensor rotation10,
                                 const at::Tensor translation10, const at::Tensor rotation0,
                                 const at::Tensor translation0, const at::Tensor rotation1,
                                 const at::Tensor translation1, const at::Tensor sampled_unscaled_dpts_0,
                                 const at::Tensor matched_unscaled_dpts_1, const at::Tensor sampled_locations_homo_0,
                                 const at::Tensor matched_locations_homo_1, const float scale_0, const float scale_1,
                                 const float loss_param, const float weight)
{
  fresh(AtA, Atb, 14, sampled_locations_homo_0);
  chk(sage_loop_mg_jac_error_calculate(
          ws(), AtA.data_ptr<float>(), Atb.data_ptr<float>(), &error, f32(rotation10, "rotation10"),
          f32(translation10, "translation10"), f32(rotation0, "rotation0"), f32(translation0, "translation0"),
          f32(rotation1, "rotation1"), f32(translation1, "translation1"), f32(sampled_unscaled_dpts_0, "unscaled_dpts_0"),
          f32(matched_unscaled_dpts_1, "unscaled_dpts_1"), f32(sampled_locations_homo_0, "locations_homo_0"),
          f32(matched_locations_homo_1, "locations_homo_1"), scale_0, scale_1, loss_param, weight,
          (int)sampled_locations_homo_0.size(0)),
      "loop_mg_jac_error_calculate");
}

// ---------------------------------------------------------------------------------------------- explicit instantiations
// (reprojection_factor_kernels.cpp:630-660, match_geometry_factor_kernels.cpp:1828-1858: DF_CODE_SIZE only)
#define SAGE_T const at::Tensor
template void reprojection_jac_error_calculate<DF_CODE_SIZE>(at::Tensor &, at::Tensor &, float &, SAGE_T, SAGE_T, SAGE_T,
                                                             SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T,
                                                             SAGE_T, SAGE_T, const float, const PinholeCamera<float> &,
                                                             const float, const float, const float);
template float reprojection_error_calculate<DF_CODE_SIZE>(SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T,
                                                          const float, const PinholeCamera<float> &, const float,
                                                          const float, const float);
template float match_geometry_error_calculate<DF_CODE_SIZE>(SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T,
                                                            SAGE_T, SAGE_T, SAGE_T, SAGE_T, const float, const float,
                                                            const float, const float, const std::string);
template void match_geometry_jac_error_calculate<DF_CODE_SIZE>(at::Tensor &, at::Tensor &, float &, SAGE_T, SAGE_T, SAGE_T,
                                                               SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T,
                                                               SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T,
                                                               const float, const float, const float, const float,
                                                               const std::string);
#undef SAGE_T

} // namespace df
