// sage_adapter.cpp -- replacement translation unit for
//     system/sources/cuda/photometric_factor_kernels.cpp   (7 - 2 = 5 function templates of photometric_factor_kernels.h)
//     system/sources/cuda/geometric_factor_kernels.cpp     (the 2 non-"unbiased" templates of geometric_factor_kernels.h)
// of lppllppl920/SAGE-SLAM.  It lives in the REFERENCE tree (next to the two headers it includes) and defines the same
// `namespace df` symbols, so core/gtsam/{photometric,geometric}_factor.cpp and core/system/camera_tracker.cpp compile
// unchanged against PyTorch-ROCm's libtorch; every call forwards raw device pointers to the C ABI of sage_ba.h
// (libsage_ba.so).  INTEGRATION.md has the CMake change.
//
// Semantics kept: fresh device AtA/Atb tensors assigned into the references, `error` on the host (the
// call synchronises the stream, like the reference's .item<float>()), the zero-overlap fallback values, exit(code) on
// a runtime error (photometric_factor_kernels.cpp:18-31), re-entrancy per host thread (one workspace per thread; the
// reference calls these functions from up to four threads, deepfactors.cpp:1497-1505).
// Supported instantiations: CS in {16, 32}, FS in {16, 32}, <= 8 pyramid levels.
#include <cstdio>
#include <cstdlib>

#include <torch/torch.h>
#include <c10/hip/HIPStream.h>

#include "photometric_factor_kernels.h" // the reference's own declarations, unchanged
#include "geometric_factor_kernels.h"
#include "sage_ba.h"

#ifndef DF_CODE_SIZE
#error "DF_CODE_SIZE / DF_FEAT_SIZE come from the reference's build (cuda/CMakeLists.txt)"
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

SagePyramid to_pyr(const CameraPyramid<float> &cp) // common/camera_pyramid.h:18-32; offsets = the reference's level_offsets
{
  SagePyramid p{};
  p.levels = (int)cp.Levels();
  TORCH_CHECK(p.levels >= 1 && p.levels <= SAGE_MAX_LEVELS, "sage_adapter: unsupported number of pyramid levels");
  int off = 0;
  for (int l = 0; l < p.levels; ++l)
  {
    p.cam[l] = to_cam(cp[l]);
    p.level_offsets[l] = off;
    off += (int)cp[l].width() * (int)cp[l].height();
  }
  p.P = off;
  return p;
}

const float *f32(const at::Tensor &t, const char *what)
{
  TORCH_CHECK(t.is_cuda() && t.is_contiguous() && t.scalar_type() == at::kFloat, "sage_adapter: ", what,
              " must be a contiguous fp32 device tensor");
  return t.data_ptr<float>();
}

at::Tensor host_weights(const at::Tensor &w) // PhotometricFactor hands a CPU tensor (photometric_factor.cpp:31-32)
{
  return w.to(at::kCPU).to(at::kFloat).contiguous();
}
} // namespace

// ---------------------------------------------------------------------------------------------- photometric (a1, a2)
template <int FS>
float photometric_error_calculate(const at::Tensor rotation, const at::Tensor translation,
                                  const at::Tensor flatten_dpt_map_bias_0, const at::Tensor flatten_dpt_jac_code_0,
                                  const at::Tensor code_0, const at::Tensor valid_mask_1,
                                  const at::Tensor sampled_locations_1d_0, const at::Tensor sampled_locations_homo_0,
                                  const at::Tensor feat_map_pyramid_0, const at::Tensor feat_map_pyramid_1,
                                  const at::Tensor /*level_offsets: implied by the camera pyramid*/, const float scale_0,
                                  const CameraPyramid<float> &camera_pyramid, const float eps,
                                  const at::Tensor weights_tensor)
{
  const SagePyramid pyr = to_pyr(camera_pyramid);
  const at::Tensor w = host_weights(weights_tensor);
  TORCH_CHECK(sampled_locations_1d_0.scalar_type() == at::kLong && sampled_locations_1d_0.is_contiguous());
  float err = 0.f;
  chk(sage_photometric_error_calculate(
          ws(), &err, nullptr, f32(rotation, "rotation"), f32(translation, "translation"),
          f32(flatten_dpt_map_bias_0, "dpt_map_bias"), f32(flatten_dpt_jac_code_0, "dpt_jac_code"),
          f32(code_0, "code"), f32(valid_mask_1, "valid_mask"), sampled_locations_1d_0.data_ptr<int64_t>(),
          f32(sampled_locations_homo_0, "locations_homo"), f32(feat_map_pyramid_0, "feat_map_pyramid_0"),
          f32(feat_map_pyramid_1, "feat_map_pyramid_1"), scale_0, &pyr, eps, w.data_ptr<float>(),
          (int)sampled_locations_homo_0.size(0), FS, (int)flatten_dpt_jac_code_0.size(1)),
      "photometric_error_calculate");
  return err;
}

template <int CS, int FS>
void photometric_jac_error_calculate(at::Tensor &AtA, at::Tensor &Atb, float &error, const at::Tensor rotation10,
                                     const at::Tensor translation10, const at::Tensor rotation0,
                                     const at::Tensor translation0, const at::Tensor rotation1,
                                     const at::Tensor translation1, const at::Tensor flatten_dpt_map_bias_0,
                                     const at::Tensor flatten_dpt_jac_code_0, const at::Tensor code_0,
                                     const at::Tensor valid_mask_1, const at::Tensor sampled_locations_1d_0,
                                     const at::Tensor sampled_locations_homo_0, const at::Tensor feat_map_pyramid_0,
                                     const at::Tensor feat_map_pyramid_1, const at::Tensor feat_map_grad_pyramid_1,
                                     const at::Tensor /*level_offsets*/, const float scale_0,
                                     const CameraPyramid<float> &camera_pyramid, const float eps,
                                     const at::Tensor weights_tensor)
{
  const SagePyramid pyr = to_pyr(camera_pyramid);
  const at::Tensor w = host_weights(weights_tensor);
  TORCH_CHECK(sampled_locations_1d_0.scalar_type() == at::kLong && sampled_locations_1d_0.is_contiguous());
  const auto opts = sampled_locations_homo_0.options();
  AtA = torch::empty({13 + CS, 13 + CS}, opts); // fresh outputs, like the reference (:1147-1152)
  Atb = torch::empty({13 + CS, 1}, opts);
  chk(sage_photometric_jac_error_calculate(
          ws(), AtA.data_ptr<float>(), Atb.data_ptr<float>(), &error, nullptr, f32(rotation10, "rotation10"),
          f32(translation10, "translation10"), f32(rotation0, "rotation0"), f32(translation0, "translation0"),
          f32(rotation1, "rotation1"), f32(translation1, "translation1"), f32(flatten_dpt_map_bias_0, "dpt_map_bias"),
          f32(flatten_dpt_jac_code_0, "dpt_jac_code"), f32(code_0, "code"), f32(valid_mask_1, "valid_mask"),
          sampled_locations_1d_0.data_ptr<int64_t>(), f32(sampled_locations_homo_0, "locations_homo"),
          f32(feat_map_pyramid_0, "feat_map_pyramid_0"), f32(feat_map_pyramid_1, "feat_map_pyramid_1"),
          f32(feat_map_grad_pyramid_1, "feat_map_grad_pyramid_1"), scale_0, &pyr, eps, w.data_ptr<float>(),
          (int)sampled_locations_homo_0.size(0), FS, CS),
      "photometric_jac_error_calculate");
}

// ---------------------------------------------------------------------------------------------- tracker trio (a3)
// the tracker passes its level weights as a DEVICE tensor (camera_tracker.cpp:177-188): handed through unchanged
template <int FS>
void tracker_photo_jac_error_calculate(at::Tensor &AtA, at::Tensor &Atb, float &error, const at::Tensor rotation,
                                       const at::Tensor translation, const at::Tensor valid_mask_1,
                                       const at::Tensor sampled_dpts_0, const at::Tensor sampled_locations_homo_0,
                                       const at::Tensor sampled_features_0, const at::Tensor feat_map_pyramid_1,
                                       const at::Tensor feat_map_grad_pyramid_1, const at::Tensor /*level_offsets*/,
                                       const CameraPyramid<float> &camera_pyramid, const float eps,
                                       const at::Tensor weights_tensor)
{
  const SagePyramid pyr = to_pyr(camera_pyramid);
  const auto opts = sampled_locations_homo_0.options();
  AtA = torch::empty({6, 6}, opts);
  Atb = torch::empty({6, 1}, opts);
  chk(sage_tracker_photo_jac_error_calculate(
          ws(), 6, AtA.data_ptr<float>(), Atb.data_ptr<float>(), &error, nullptr, f32(rotation, "rotation"),
          f32(translation, "translation"), f32(valid_mask_1, "valid_mask"), f32(sampled_dpts_0, "sampled_dpts"),
          f32(sampled_locations_homo_0, "locations_homo"), f32(sampled_features_0, "sampled_features"),
          f32(feat_map_pyramid_1, "feat_map_pyramid_1"), f32(feat_map_grad_pyramid_1, "feat_map_grad_pyramid_1"), &pyr,
          1.0f, eps, f32(weights_tensor, "weights"), (int)sampled_locations_homo_0.size(0), FS),
      "tracker_photo_jac_error_calculate");
}

template <int FS>
void tracker_photo_jac_error_calculate_with_scale(at::Tensor &AtA, at::Tensor &Atb, float &error,
                                                  const at::Tensor rotation, const at::Tensor translation,
                                                  const at::Tensor valid_mask_1, const at::Tensor sampled_dpts_0,
                                                  const at::Tensor sampled_locations_homo_0,
                                                  const at::Tensor sampled_features_0,
                                                  const at::Tensor feat_map_pyramid_1,
                                                  const at::Tensor feat_map_grad_pyramid_1,
                                                  const at::Tensor /*level_offsets*/,
                                                  const CameraPyramid<float> &camera_pyramid, const float scale_0,
                                                  const float eps, const at::Tensor weights_tensor)
{
  const SagePyramid pyr = to_pyr(camera_pyramid);
  const auto opts = sampled_locations_homo_0.options();
  AtA = torch::empty({7, 7}, opts);
  Atb = torch::empty({7, 1}, opts);
  chk(sage_tracker_photo_jac_error_calculate(
          ws(), 7, AtA.data_ptr<float>(), Atb.data_ptr<float>(), &error, nullptr, f32(rotation, "rotation"),
          f32(translation, "translation"), f32(valid_mask_1, "valid_mask"), f32(sampled_dpts_0, "sampled_dpts"),
          f32(sampled_locations_homo_0, "locations_homo"), f32(sampled_features_0, "sampled_features"),
          f32(feat_map_pyramid_1, "feat_map_pyramid_1"), f32(feat_map_grad_pyramid_1, "feat_map_grad_pyramid_1"), &pyr,
          scale_0, eps, f32(weights_tensor, "weights"), (int)sampled_locations_homo_0.size(0), FS),
      "tracker_photo_jac_error_calculate_with_scale");
}

template <int FS>
float tracker_photo_error_calculate(const at::Tensor rotation, const at::Tensor translation,
                                    const at::Tensor valid_mask_1, const at::Tensor sampled_dpts_0,
                                    const at::Tensor sampled_locations_homo_0, const at::Tensor sampled_features_0,
                                    const at::Tensor feat_map_pyramid_1, const at::Tensor /*level_offsets*/,
                                    const CameraPyramid<float> &camera_pyramid, const float eps,
                                    const at::Tensor weights_tensor)
{
  const SagePyramid pyr = to_pyr(camera_pyramid);
  float err = 0.f;
  chk(sage_tracker_photo_error_calculate(
          ws(), &err, nullptr, f32(rotation, "rotation"), f32(translation, "translation"),
          f32(valid_mask_1, "valid_mask"), f32(sampled_dpts_0, "sampled_dpts"),
          f32(sampled_locations_homo_0, "locations_homo"), f32(sampled_features_0, "sampled_features"),
          f32(feat_map_pyramid_1, "feat_map_pyramid_1"), &pyr, eps, f32(weights_tensor, "weights"),
          (int)sampled_locations_homo_0.size(0), FS),
      "tracker_photo_error_calculate");
  return err;
}

// ---------------------------------------------------------------------------------------------- geometric (a4, a5)
// sampled_locations_1d_0 arrives as int32 here (geometric_factor.cpp:344: .to(torch::kInt32))
template <int CS>
float geometric_error_calculate(const at::Tensor rotation, const at::Tensor translation,
                                const at::Tensor flatten_dpt_map_bias_0, const at::Tensor flatten_dpt_jac_code_0,
                                const at::Tensor code_0, const at::Tensor dpt_map_1, const at::Tensor valid_mask_1,
                                const at::Tensor sampled_locations_1d_0, const at::Tensor sampled_locations_homo_0,
                                const float scale_0, const PinholeCamera<float> &camera, const float eps,
                                const float loss_param, const float weight)
{
  const SageCamera cam = to_cam(camera);
  const at::Tensor loc = sampled_locations_1d_0.to(at::kInt).contiguous();
  const at::Tensor d1 = dpt_map_1.contiguous();
  float err = 0.f;
  chk(sage_geometric_error_calculate(ws(), &err, nullptr, f32(rotation, "rotation"), f32(translation, "translation"),
                                     f32(flatten_dpt_map_bias_0, "dpt_map_bias"),
                                     f32(flatten_dpt_jac_code_0, "dpt_jac_code"), f32(code_0, "code"),
                                     f32(d1, "dpt_map_1"), f32(valid_mask_1, "valid_mask"), loc.data_ptr<int32_t>(),
                                     f32(sampled_locations_homo_0, "locations_homo"), scale_0, &cam, eps, loss_param,
                                     weight, (int)sampled_locations_homo_0.size(0), CS),
      "geometric_error_calculate");
  return err;
}

template <int CS>
void geometric_jac_error_calculate(at::Tensor &AtA, at::Tensor &Atb, float &error, const at::Tensor rotation10,
                                   const at::Tensor translation10, const at::Tensor rotation0,
                                   const at::Tensor translation0, const at::Tensor rotation1,
                                   const at::Tensor translation1, const at::Tensor flatten_dpt_map_bias_0,
                                   const at::Tensor flatten_dpt_jac_code_0, const at::Tensor code_0,
                                   const at::Tensor dpt_map_1, const at::Tensor dpt_map_grad_1,
                                   const at::Tensor dpt_jac_code_1, const at::Tensor valid_mask_1,
                                   const at::Tensor sampled_locations_1d_0, const at::Tensor sampled_locations_homo_0,
                                   const float scale_0, const float scale_1, const PinholeCamera<float> &camera,
                                   const float eps, const float loss_param, const float weight)
{
  const SageCamera cam = to_cam(camera);
  const at::Tensor loc = sampled_locations_1d_0.to(at::kInt).contiguous();
  const at::Tensor d1 = dpt_map_1.contiguous(), g1 = dpt_map_grad_1.contiguous();
  const auto opts = sampled_locations_homo_0.options();
  AtA = torch::empty({14 + 2 * CS, 14 + 2 * CS}, opts);
  Atb = torch::empty({14 + 2 * CS, 1}, opts);
  chk(sage_geometric_jac_error_calculate(
          ws(), AtA.data_ptr<float>(), Atb.data_ptr<float>(), &error, nullptr, f32(rotation10, "rotation10"),
          f32(translation10, "translation10"), f32(rotation0, "rotation0"), f32(translation0, "translation0"),
          f32(rotation1, "rotation1"), f32(translation1, "translation1"), f32(flatten_dpt_map_bias_0, "dpt_map_bias"),
          f32(flatten_dpt_jac_code_0, "dpt_jac_code"), f32(code_0, "code"), f32(d1, "dpt_map_1"),
          f32(g1, "dpt_map_grad_1"), f32(dpt_jac_code_1, "dpt_jac_code_1"), f32(valid_mask_1, "valid_mask"),
          loc.data_ptr<int32_t>(), f32(sampled_locations_homo_0, "locations_homo"), scale_0, scale_1, &cam, eps,
          loss_param, weight, (int)sampled_locations_homo_0.size(0), CS),
      "geometric_jac_error_calculate");
}

// ---------------------------------------------------------------------------------------------- explicit instantiations
// the same set the reference emits (photometric_factor_kernels.cpp:1388-1449, geometric_factor_kernels.cpp:954-990)
#define SAGE_T const at::Tensor
template float photometric_error_calculate<DF_FEAT_SIZE>(SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T,
                                                         SAGE_T, SAGE_T, SAGE_T, const float,
                                                         const CameraPyramid<float> &, const float, SAGE_T);
template void photometric_jac_error_calculate<DF_CODE_SIZE, DF_FEAT_SIZE>(
    at::Tensor &, at::Tensor &, float &, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T,
    SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, const float, const CameraPyramid<float> &, const float, SAGE_T);
template void tracker_photo_jac_error_calculate<DF_FEAT_SIZE>(at::Tensor &, at::Tensor &, float &, SAGE_T, SAGE_T,
                                                              SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T,
                                                              const CameraPyramid<float> &, const float, SAGE_T);
template void tracker_photo_jac_error_calculate_with_scale<DF_FEAT_SIZE>(
    at::Tensor &, at::Tensor &, float &, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T,
    const CameraPyramid<float> &, const float, const float, SAGE_T);
template float tracker_photo_error_calculate<DF_FEAT_SIZE>(SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T,
                                                           const CameraPyramid<float> &, const float, SAGE_T);
template float geometric_error_calculate<DF_CODE_SIZE>(SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T,
                                                       SAGE_T, const float, const PinholeCamera<float> &, const float,
                                                       const float, const float);
template void geometric_jac_error_calculate<DF_CODE_SIZE>(at::Tensor &, at::Tensor &, float &, SAGE_T, SAGE_T, SAGE_T,
                                                          SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T,
                                                          SAGE_T, SAGE_T, SAGE_T, SAGE_T, SAGE_T, const float,
                                                          const float, const PinholeCamera<float> &, const float,
                                                          const float, const float);
#undef SAGE_T

} // namespace df
