// adapter_run -- EXECUTES the two replacement translation units integration/sage_adapter.cpp and
// integration/sage_adapter_keypoints.cpp (the boundary had only ever been compiled).  This driver is what
// the reference's callers are to those TUs: it includes the reference's OWN headers (cuda/photometric_factor_kernels.h,
// geometric_factor_kernels.h, reprojection_factor_kernels.h, match_geometry_factor_kernels.h, common/camera_pyramid.h,
// pinhole_camera.h), builds at::Tensor arguments with PyTorch-ROCm's libtorch exactly as core/gtsam/*_factor.cpp and
// core/system/camera_tracker.cpp hand them over (device fp32 tensors, int64 / int32 locations, a CPU weights tensor for the
// mapper's photometric factor, a device one for the tracker's) and calls all 7 + 11 `df::` entry points.  Built in the build
// container by sage_slam_amd/build.py:build_adapter_run() (needs the reference's headers), run on the GPU by
// tests/test_gpu_adapter_run.py, which compares what comes back with the CPU oracle.
//
//   adapter_run <in.bin> <out.bin>
//
// in.bin / out.bin: a sequence of named arrays -- uint32 name length, name bytes, uint8 dtype (0 f32, 1 i64, 2 i32),
// uint32 ndim, int64 dims[ndim], raw data.  Scalars are 1-element f32 arrays.
#include <torch/torch.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "photometric_factor_kernels.h"
#include "geometric_factor_kernels.h"
#include "reprojection_factor_kernels.h"
#include "match_geometry_factor_kernels.h"

#ifndef DF_CODE_SIZE
#error "compile with -DDF_CODE_SIZE=32 -DDF_FEAT_SIZE=16 like the reference's build"
#endif

using Map = std::map<std::string, at::Tensor>;

static Map load(const char *path)
{
  Map m;
  FILE *f = fopen(path, "rb");
  if (!f)
  {
    fprintf(stderr, "adapter_run: cannot open %s\n", path);
    exit(2);
  }
  for (;;)
  {
    uint32_t nl = 0;
    if (fread(&nl, 4, 1, f) != 1)
      break;
    std::string name(nl, '\0');
    uint8_t dt = 0;
    uint32_t nd = 0;
    if (fread(&name[0], 1, nl, f) != nl || fread(&dt, 1, 1, f) != 1 || fread(&nd, 4, 1, f) != 1)
      exit(3);
    std::vector<int64_t> dims(nd);
    if (nd && fread(dims.data(), 8, nd, f) != nd)
      exit(3);
    const auto st = dt == 0 ? at::kFloat : (dt == 1 ? at::kLong : at::kInt);
    at::Tensor t = torch::empty(dims, torch::TensorOptions().dtype(st));
    const size_t bytes = (size_t)t.numel() * t.element_size();
    if (bytes && fread(t.data_ptr(), 1, bytes, f) != bytes)
      exit(3);
    m[name] = t;
  }
  fclose(f);
  return m;
}

static void put(FILE *f, const std::string &name, const at::Tensor &t_in)
{
  const at::Tensor t = t_in.detach().to(at::kCPU).contiguous();
  const uint32_t nl = (uint32_t)name.size();
  const uint8_t dt = t.scalar_type() == at::kFloat ? 0 : (t.scalar_type() == at::kLong ? 1 : 2);
  const uint32_t nd = (uint32_t)t.dim();
  fwrite(&nl, 4, 1, f);
  fwrite(name.data(), 1, nl, f);
  fwrite(&dt, 1, 1, f);
  fwrite(&nd, 4, 1, f);
  for (uint32_t i = 0; i < nd; ++i)
  {
    const int64_t d = t.size(i);
    fwrite(&d, 8, 1, f);
  }
  fwrite(t.data_ptr(), 1, (size_t)t.numel() * t.element_size(), f);
}
static void put_scalar(FILE *f, const std::string &name, float v) { put(f, name, torch::full({1}, v)); }

// what "fresh outputs" means at the boundary (photometric_factor_kernels.cpp:1147-1152: AtA / Atb are ASSIGNED matmul
// results): whatever the caller's tensors held before, they come back as new device tensors of the factor's shape
static float fresh_ok(const at::Tensor &AtA, const at::Tensor &Atb, const void *old_a, const void *old_b, int64_t D)
{
  const bool ok = AtA.is_cuda() && Atb.is_cuda() && AtA.dim() == 2 && AtA.size(0) == D && AtA.size(1) == D && Atb.dim() == 2 &&
                  Atb.size(0) == D && Atb.size(1) == 1 && AtA.data_ptr() != old_a && Atb.data_ptr() != old_b &&
                  AtA.scalar_type() == at::kFloat;
  return ok ? 1.f : 0.f;
}

int main(int argc, char **argv)
{
  if (argc < 3)
  {
    fprintf(stderr, "usage: adapter_run <in.bin> <out.bin>\n");
    return 1;
  }
  if (!torch::cuda::is_available())
  {
    fprintf(stderr, "adapter_run: no HIP device\n");
    return 5;
  }
  torch::NoGradGuard ng;
  Map in = load(argv[1]);
  auto D = [&](const char *n) -> at::Tensor {
    auto it = in.find(n);
    if (it == in.end())
    {
      fprintf(stderr, "adapter_run: input %s missing\n", n);
      exit(4);
    }
    return it->second.to(at::kCUDA).contiguous();
  };
  auto S = [&](const char *n) -> float { return in.at(n).item<float>(); };
  FILE *out = fopen(argv[2], "wb");
  constexpr int CS = DF_CODE_SIZE, FS = DF_FEAT_SIZE;

  // ---- cameras, as the reference builds them (common/camera_pyramid.h:18-32)
  const at::Tensor camv = in.at("cam");
  const float *cv_ = camv.data_ptr<float>();
  const df::PinholeCamera<float> cam(cv_[0], cv_[1], cv_[2], cv_[3], cv_[4], cv_[5]);
  const int levels = (int)S("levels");
  const df::CameraPyramid<float> pyr(cam, levels);

  // =========================================================== dense factors (sage_adapter.cpp)
  if (in.count("feat_pyr_0"))
  {
    const at::Tensor R10 = D("R10"), t10 = D("t10"), R0 = D("R0"), t0 = D("t0"), R1 = D("R1"), t1 = D("t1");
    const at::Tensor bias0 = D("bias_0"), basis0 = D("basis_0"), code0 = D("code_0"), mask = D("valid_mask_1");
    const at::Tensor loc64 = D("loc1d_0"), homo = D("homo_0"), f0 = D("feat_pyr_0"), f1 = D("feat_pyr_1"), g1 = D("grad_pyr_1");
    const at::Tensor offs = D("level_offsets");
    const at::Tensor w_cpu = in.at("photo_weights");           // PhotometricFactor hands a CPU tensor (photometric_factor.cpp:31-32)
    const at::Tensor w_dev = D("photo_weights");               // the tracker a device tensor (camera_tracker.cpp:177-188)
    const float s0 = S("scale_0"), s1 = S("scale_1"), eps = S("eps");
    for (int far = 0; far < 2; ++far)
    {
      // far = 1: a relative translation that puts every point behind keyframe 1 -- the zero-overlap fallback values
      const at::Tensor tt = far ? torch::tensor({0.f, 0.f, -100.f}).to(at::kCUDA) : t10;
      const std::string sfx = far ? "_far" : "";
      at::Tensor AtA = torch::ones({3, 3}), Atb = torch::ones({5});      // stale CPU tensors of the wrong shape
      const void *oa = AtA.data_ptr(), *ob = Atb.data_ptr();
      float err = -1.f;
      df::photometric_jac_error_calculate<CS, FS>(AtA, Atb, err, R10, tt, R0, t0, R1, t1, bias0, basis0, code0, mask, loc64,
                                                  homo, f0, f1, g1, offs, s0, pyr, eps, w_cpu);
      put(out, "photo_AtA" + sfx, AtA); put(out, "photo_Atb" + sfx, Atb); put_scalar(out, "photo_err" + sfx, err);
      put_scalar(out, "photo_fresh" + sfx, fresh_ok(AtA, Atb, oa, ob, 13 + CS));
      put_scalar(out, "photo_err_only" + sfx,
                 df::photometric_error_calculate<FS>(R10, tt, bias0, basis0, code0, mask, loc64, homo, f0, f1, offs, s0, pyr,
                                                     eps, w_cpu));
      // geometric pair: int32 locations (geometric_factor.cpp:344)
      const at::Tensor loc32 = loc64.to(at::kInt);
      at::Tensor GA = torch::ones({2}), Gb = torch::ones({2});
      oa = GA.data_ptr(); ob = Gb.data_ptr();
      float gerr = -1.f;
      df::geometric_jac_error_calculate<CS>(GA, Gb, gerr, R10, tt, R0, t0, R1, t1, bias0, basis0, code0, D("dpt_map_1"),
                                            D("dpt_map_grad_1"), D("basis_1"), mask, loc32, homo, s0, s1, cam, eps,
                                            S("geo_loss_param"), S("geo_weight"));
      put(out, "geo_AtA" + sfx, GA); put(out, "geo_Atb" + sfx, Gb); put_scalar(out, "geo_err" + sfx, gerr);
      put_scalar(out, "geo_fresh" + sfx, fresh_ok(GA, Gb, oa, ob, 14 + 2 * CS));
      put_scalar(out, "geo_err_only" + sfx,
                 df::geometric_error_calculate<CS>(R10, tt, bias0, basis0, code0, D("dpt_map_1"), mask, loc32, homo, s0, cam,
                                                   eps, S("geo_loss_param"), S("geo_weight")));
      // tracker trio (pre-sampled source features, depths handed over)
      const at::Tensor dp = D("sampled_dpts_0"), sf = D("sampled_features_0");
      at::Tensor TA = torch::ones({1}), Tb = torch::ones({1});
      oa = TA.data_ptr(); ob = Tb.data_ptr();
      float terr = -1.f;
      df::tracker_photo_jac_error_calculate<FS>(TA, Tb, terr, R10, tt, mask, dp, homo, sf, f1, g1, offs, pyr, eps, w_dev);
      put(out, "trk6_AtA" + sfx, TA); put(out, "trk6_Atb" + sfx, Tb); put_scalar(out, "trk6_err" + sfx, terr);
      put_scalar(out, "trk6_fresh" + sfx, fresh_ok(TA, Tb, oa, ob, 6));
      at::Tensor SA, Sb;
      float serr = -1.f;
      df::tracker_photo_jac_error_calculate_with_scale<FS>(SA, Sb, serr, R10, tt, mask, dp, homo, sf, f1, g1, offs, pyr, s0, eps,
                                                           w_dev);
      put(out, "trk7_AtA" + sfx, SA); put(out, "trk7_Atb" + sfx, Sb); put_scalar(out, "trk7_err" + sfx, serr);
      put_scalar(out, "trk7_fresh" + sfx, fresh_ok(SA, Sb, nullptr, nullptr, 7));
      put_scalar(out, "trk_err_only" + sfx,
                 df::tracker_photo_error_calculate<FS>(R10, tt, mask, dp, homo, sf, f1, offs, pyr, eps, w_dev));
    }
  }

  // =========================================================== keypoint factors (sage_adapter_keypoints.cpp)
  if (in.count("kp_homo_0"))
  {
    const at::Tensor R10 = D("kp_R10"), t10 = D("kp_t10"), R0 = D("kp_R0"), t0 = D("kp_t0"), R1 = D("kp_R1"), t1 = D("kp_t1");
    const at::Tensor bias0 = D("kp_bias_0"), bias1 = D("kp_bias_1"), basis0 = D("kp_basis_0"), basis1 = D("kp_basis_1");
    const at::Tensor code0 = D("kp_code_0"), code1 = D("kp_code_1"), homo0 = D("kp_homo_0"), homo1 = D("kp_homo_1");
    const at::Tensor loc0 = D("kp_loc_0"), loc1 = D("kp_loc_1");          // int32, as match_geometry_factor.cpp:107-109 makes them
    const at::Tensor matched2d = D("kp_matched_2d"), dpts0 = D("kp_dpts_0"), dpts1 = D("kp_dpts_1");
    const at::Tensor u0 = D("kp_unscaled_0"), u1 = D("kp_unscaled_1");
    const float s0 = S("kp_scale_0"), s1 = S("kp_scale_1"), eps = S("eps"), c = S("kp_loss_param"), wgt = S("kp_weight");
    // reprojection (mapper + tracker)
    {
      at::Tensor A = torch::ones({2}), b = torch::ones({2});
      const void *oa = A.data_ptr(), *ob = b.data_ptr();
      float err = -1.f;
      df::reprojection_jac_error_calculate<CS>(A, b, err, R10, t10, R0, t0, R1, t1, bias0, basis0, code0, loc0, homo0, matched2d,
                                               s0, cam, eps, c, wgt);
      put(out, "reproj_AtA", A); put(out, "reproj_Atb", b); put_scalar(out, "reproj_err", err);
      put_scalar(out, "reproj_fresh", fresh_ok(A, b, oa, ob, 13 + CS));
      put_scalar(out, "reproj_err_only",
                 df::reprojection_error_calculate<CS>(R10, t10, bias0, basis0, code0, loc0, homo0, matched2d, s0, cam, eps, c, wgt));
      // an int64 location tensor is converted, not refused
      put_scalar(out, "reproj_err_only_i64",
                 df::reprojection_error_calculate<CS>(R10, t10, bias0, basis0, code0, loc0.to(at::kLong), homo0, matched2d, s0, cam,
                                                      eps, c, wgt));
      at::Tensor TA, Tb;
      float terr = -1.f;
      df::tracker_reproj_jac_error_calculate(TA, Tb, terr, R10, t10, dpts0, homo0, matched2d, cam, eps, c, wgt);
      put(out, "trk_reproj_AtA", TA); put(out, "trk_reproj_Atb", Tb); put_scalar(out, "trk_reproj_err", terr);
      put_scalar(out, "trk_reproj_fresh", fresh_ok(TA, Tb, nullptr, nullptr, 6));
      put_scalar(out, "trk_reproj_err_only", df::tracker_reproj_error_calculate(R10, t10, dpts0, homo0, matched2d, cam, eps, c, wgt));
    }
    // match geometry: the robust_loss_type string dispatch (match_geometry_factor_kernels.cpp:1704-1807)
    for (const char *loss : {"fair", "L2", "huber", "unbiased", "no_such_loss"})
    {
      at::Tensor A = torch::ones({2}), b = torch::ones({2});
      const void *oa = A.data_ptr(), *ob = b.data_ptr();
      float err = -1.f;
      df::match_geometry_jac_error_calculate<CS>(A, b, err, R10, t10, R0, t0, R1, t1, bias0, bias1, basis0, basis1, code0, code1,
                                                 homo0, homo1, loc0, loc1, s0, s1, c, wgt, loss);
      const std::string k = std::string("mg_") + loss;
      put(out, k + "_AtA", A); put(out, k + "_Atb", b); put_scalar(out, k + "_err", err);
      put_scalar(out, k + "_fresh", fresh_ok(A, b, oa, ob, 14 + 2 * CS));
      put_scalar(out, k + "_err_only",
                 df::match_geometry_error_calculate<CS>(R10, t10, bias0, bias1, basis0, basis1, code0, code1, homo0, homo1, loc0,
                                                        loc1, s0, s1, c, wgt, loss));
    }
    {
      at::Tensor A, b;
      float err = -1.f;
      df::loop_mg_jac_error_calculate(A, b, err, R10, t10, R0, t0, R1, t1, u0, u1, homo0, homo1, s0, s1, c, wgt);
      put(out, "loop_AtA", A); put(out, "loop_Atb", b); put_scalar(out, "loop_err", err);
      put_scalar(out, "loop_fresh", fresh_ok(A, b, nullptr, nullptr, 14));
      put_scalar(out, "loop_err_only", df::loop_mg_error_calculate(R10, t10, u0, u1, homo0, homo1, s0, s1, c, wgt));
      at::Tensor TA, Tb;
      float terr = -1.f;
      df::tracker_match_geom_jac_error_calculate(TA, Tb, terr, R10, t10, dpts0, dpts1, homo0, homo1, c, wgt);
      put(out, "trk_mg6_AtA", TA); put(out, "trk_mg6_Atb", Tb); put_scalar(out, "trk_mg6_err", terr);
      put_scalar(out, "trk_mg6_fresh", fresh_ok(TA, Tb, nullptr, nullptr, 6));
      at::Tensor SA, Sb;
      float serr = -1.f;
      df::tracker_match_geom_jac_error_calculate_with_scale(SA, Sb, serr, R10, t10, dpts0, dpts1, homo0, homo1, s0, c, wgt);
      put(out, "trk_mg7_AtA", SA); put(out, "trk_mg7_Atb", Sb); put_scalar(out, "trk_mg7_err", serr);
      put_scalar(out, "trk_mg7_fresh", fresh_ok(SA, Sb, nullptr, nullptr, 7));
      put_scalar(out, "trk_mg_err_only", df::tracker_match_geom_error_calculate(R10, t10, dpts0, dpts1, homo0, homo1, c, wgt));
    }
  }
  fclose(out);
  return 0;
}
