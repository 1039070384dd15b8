// SYNTAX-CHECK STAND-IN, not gtsam: gtsam needs Boost, which the build image lacks.  It declares only the names
// integration/sage_gtsam_prepass.h uses, with the signatures of gtsam 4.0 (gtsam/linear/HessianFactor.h: the
// constructor taking keys, the upper-triangular blocks G11 G12 .. Gnn, the g_i and the constant term f), so that
// tests/test_adapter_compiles.py can put the header through a compiler.  Never shipped, never linked.
#pragma once
#include <Eigen/Dense>
#include <cstdint>
#include <memory>
#include <vector>

namespace boost
{
template <class T> using shared_ptr = std::shared_ptr<T>;
template <class T, class... A> shared_ptr<T> make_shared(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }
} // namespace boost

namespace gtsam
{
typedef std::uint64_t Key;
typedef Eigen::MatrixXd Matrix;
typedef Eigen::VectorXd Vector;
template <class T> using FastVector = std::vector<T>;
class HessianFactor
{
public:
  HessianFactor(const FastVector<Key> &, const std::vector<Matrix> &, const std::vector<Vector> &, double) {}
};
} // namespace gtsam
