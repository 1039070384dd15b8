// compile-only driver of integration/sage_gtsam_prepass.h (tests/test_adapter_compiles.py): real Eigen + Sophus from
// the reference's thirdparty tree, stand-ins for gtsam / boost (this directory)
#include "../sage_gtsam_prepass.h"

double prepass_check(df::SageWindowCache &cache, const gtsam::Values &values)
{
  boost::shared_ptr<gtsam::HessianFactor> f = cache.Linearize(values, 0, 3, {1, 2, 3, 4});
  return cache.Error(values, 1, 3) + (f ? 1.0 : 0.0);
}
