// sage_gtsam_prepass.h -- gtsam side of SURVEY s8 f2: serve PhotometricFactor / GeometricFactor ::linearize / ::error
// (core/gtsam/photometric_factor.cpp:72-219, geometric_factor.cpp:41-233) from ONE batched window evaluation per
// gtsam::Values instead of one kernel sequence + NearestPsd per factor.
//
// NOT LINKED AGAINST gtsam IN THIS IMAGE: gtsam needs Boost, which the build container lacks; the header is
// put through a compiler with the real Eigen / Sophus and syntax-check stand-ins for gtsam / boost
// (integration/compile_check, tests/test_adapter_compiles.py).  Everything below the
// gtsam types -- value comparison, batched linearize / error pass, device-to-host copy of the per-edge results,
// NearestPsd (as written or Higham), the cut into the reference's G11..Gnn / g1..gn order -- is engine code behind the
// C ABI (sage_window_prepass / sage_window_factor / sage_window_factor_error, include/sage_ba.h) and is tested on the GPU
// against the oracle (tests/test_gpu_factor_cache.py); this header is the ~80 lines of type conversion that remain.
//
// Use: Mapper owns one SageWindowCache per active window (keyframes + links registered as in INTEGRATION.md s3) and
// hands it to the factors it creates; the two factor classes replace the bodies of linearize() / error() as shown.
#pragma once
#include <gtsam/linear/HessianFactor.h>
#include <gtsam/nonlinear/Values.h>
#include <sophus/se3.hpp>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "sage_ba.h"

namespace df
{

class SageWindowCache
{
public:
  struct Keys
  {
    gtsam::Key pose, code, scale;
  };

  // keys[k]: the gtsam keys of window keyframe k (the order of sage_window_add_keyframe)
  SageWindowCache(SageWindow *win, std::vector<Keys> keys, int code_size, int psd_mode /* 2 = NearestPsd as written */)
      : win_(win), keys_(std::move(keys)), CS_(code_size), psd_mode_(psd_mode),
        pose_(keys_.size() * 12), code_(keys_.size() * code_size), scale_(keys_.size())
  {
    for (size_t k = 0; k < keys_.size(); ++k) // start from the variables the keyframes were added with
      sage_window_get_keyframe(win_, (int)k, &pose_[k * 12], &code_[k * CS_], &scale_[k]);
  }

  // Gather the window's variables out of `c` (pose_wk as [R row-major | t], gtsam_traits.h:45-70) and make sure the
  // cache holds their linearisation (jacobians) or errors.  The first factor that sees new Values pays for the whole
  // window; every other factor of the same ISAM2 relinearisation / Dogleg trial is a cache hit.
  // (called with mutex_ held: the prepass and the read of its result must not interleave with another thread's Values)
  bool Prepare(const gtsam::Values &c, bool jacobians)
  {
    for (size_t k = 0; k < keys_.size(); ++k)
    {
      if (!c.exists(keys_[k].pose)) // keyframe marginalised out of this Values: keep the engine's current value
        continue;
      const Sophus::SE3f T = c.at<Sophus::SE3f>(keys_[k].pose);
      const Eigen::Matrix3f R = T.rotationMatrix();
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          pose_[k * 12 + i * 3 + j] = R(i, j);
      for (int i = 0; i < 3; ++i)
        pose_[k * 12 + 9 + i] = T.translation()(i);
      const gtsam::Vector code = c.at<gtsam::Vector>(keys_[k].code);
      for (int i = 0; i < CS_; ++i)
        code_[k * CS_ + i] = (float)code(i);
      scale_[k] = c.at<float>(keys_[k].scale);
    }
    int recomputed = 0;
    const int rc = sage_window_prepass(win_, pose_.data(), code_.data(), scale_.data(), jacobians ? 1 : 0, &recomputed);
    if (rc != SAGE_OK)
    {
      fprintf(stderr, "sage_window_prepass: %s\n", sage_error_string(rc)); // the reference's gpuErrchk convention
      exit(rc);
    }
    // the per-factor NearestPsd (the host cost of linearize(): an SVD of a 45 x 45 / 78 x 78 matrix each) for the whole
    // window at once on the host's cores; Linearize() below then only cuts blocks
    if (recomputed && jacobians && eager_psd_threads_ >= 0)
    {
      // a failed projection is not fatal here: the cache is marked unprepared by the engine and Linearize() projects the
      // factors it is asked for one by one (and reports their errors)
      const int rcp = sage_window_prepare_factors(win_, psd_mode_, eager_psd_threads_);
      if (rcp != SAGE_OK)
        fprintf(stderr, "sage_window_prepare_factors: %s (falling back to per-factor projection)\n", sage_error_string(rcp));
    }
    return recomputed != 0;
  }

  // PhotometricFactor::linearize (type 0, keys {p0,p1,c0,s0}) / GeometricFactor::linearize (type 1, keys
  // {p0,p1,c0,c1,s0,s1}); edge = 2 * link + direction as sage_window_get_edge numbers them
  boost::shared_ptr<gtsam::HessianFactor> Linearize(const gtsam::Values &c, int type, int edge,
                                                    const gtsam::FastVector<gtsam::Key> &factor_keys)
  {
    std::lock_guard<std::mutex> lock(mutex_); // factors are called from up to 4 host threads (deepfactors.cpp:1497-1505)
    Prepare(c, true);
    std::vector<double> G(sage_factor_block_count(type, CS_)), g(type == 0 ? 13 + CS_ : 14 + 2 * CS_);
    int dims[6], nkeys = 0;
    double f = 0.0;
    const int rc = sage_window_factor(win_, type, edge, psd_mode_, G.data(), g.data(), &f, dims, &nkeys);
    if (rc != SAGE_OK)
    {
      fprintf(stderr, "sage_window_factor: %s\n", sage_error_string(rc));
      exit(rc);
    }
    using RowMajor = Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
    std::vector<gtsam::Matrix> Gs;
    std::vector<gtsam::Vector> gs;
    size_t o = 0, go = 0;
    for (int i = 0; i < nkeys; ++i) // G11 G12 .. G1n G22 .. (photometric_factor.cpp:165-181)
      for (int j = i; j < nkeys; ++j)
      {
        Gs.emplace_back(Eigen::Map<const RowMajor>(G.data() + o, dims[i], dims[j]));
        o += (size_t)dims[i] * dims[j];
      }
    for (int i = 0; i < nkeys; ++i)
    {
      gs.emplace_back(Eigen::Map<const Eigen::VectorXd>(g.data() + go, dims[i]));
      go += dims[i];
    }
    return boost::make_shared<gtsam::HessianFactor>(factor_keys, Gs, gs, f); // f = error_ (photometric_factor.cpp:218)
  }

  // PhotometricFactor::error / GeometricFactor::error (photometric_factor.cpp:72-101, geometric_factor.cpp:41-64)
  double Error(const gtsam::Values &c, int type, int edge)
  {
    std::lock_guard<std::mutex> lock(mutex_);
    Prepare(c, false);
    double e = 0.0;
    const int rc = sage_window_factor_error(win_, type, edge, &e);
    if (rc != SAGE_OK)
    {
      fprintf(stderr, "sage_window_factor_error: %s\n", sage_error_string(rc));
      exit(rc);
    }
    return e;
  }

private:
  SageWindow *win_;
  std::vector<Keys> keys_;
  int CS_, psd_mode_;
  std::vector<float> pose_, code_, scale_;
  std::mutex mutex_;

public:
  // >= 0: project every factor right after a batched linearisation on this many host threads (0 = the engine's default: a
  // quarter of the hardware threads, at most 64 -- include/sage_ba.h); -1: lazily, factor
  // by factor inside Linearize() (what ISAM2's partial relinearisation wants when it touches a few factors only)
  int eager_psd_threads_ = 0;
};

// In core/gtsam/photometric_factor.cpp the two bodies become (geometric_factor.cpp alike with type 1):
//
//   double PhotometricFactor<Scalar, CS>::error(const gtsam::Values &c) const
//   { return this->active(c) ? cache_->Error(c, 0, edge_) : 0.0; }
//
//   boost::shared_ptr<gtsam::GaussianFactor> PhotometricFactor<Scalar, CS>::linearize(const gtsam::Values &c) const
//   {
//     if (!this->active(c)) return boost::shared_ptr<gtsam::HessianFactor>();
//     return cache_->Linearize(c, 0, edge_, {pose0_key_, pose1_key_, code0_key_, scale0_key_});
//   }
//
// with two new members (std::shared_ptr<SageWindowCache> cache_; int edge_) set by Mapper when it adds the link.

} // namespace df
