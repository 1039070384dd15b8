// STAND-IN, not gtsam: gtsam needs Boost, which the build image lacks.  It declares only the names
// integration/sage_gtsam_prepass.h uses, with the signatures of gtsam 4.0 (gtsam/linear/HessianFactor.h: the
// constructor taking keys, the upper-triangular blocks G11 G12 .. Gnn, the g_i and the constant term f).  Unlike a
// syntax-only stub it RECORDS what it is constructed with and assembles the augmented information matrix
//   [ G  g ; g^T  f ]   (block (i,j), i <= j, from Gs[i*n - i(i-1)/2 + j - i]; the lower triangle by symmetry)
// exactly as gtsam's constructor does (HessianFactor.cpp: "HessianFactor(keys, Gs, gs, f)" fills info_ block by block and
// throws on a block whose shape does not match its keys' dimensions), so that integration/compile_check/prepass_run.cpp
// can execute the glue header and a test can compare what arrives on the gtsam side (tests/test_gpu_gtsam_glue.py).
// Never shipped, never linked into the engine.
#pragma once
#include <Eigen/Dense>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <vector>

namespace boost
{
template <class T> using shared_ptr = std::shared_ptr<T>;
template <class T, class... A> shared_ptr<T> make_shared(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }
} // namespace boost

namespace gtsam
{
typedef std::uint64_t Key;
typedef Eigen::MatrixXd Matrix; // column-major, like gtsam::Matrix
typedef Eigen::VectorXd Vector;
template <class T> using FastVector = std::vector<T>;
class HessianFactor
{
public:
  HessianFactor(const FastVector<Key> &keys, const std::vector<Matrix> &Gs, const std::vector<Vector> &gs, double f)
      : keys_(keys), f_(f)
  {
    const size_t n = keys.size();
    if (gs.size() != n || Gs.size() != n * (n + 1) / 2)
      throw std::invalid_argument("HessianFactor: wrong number of blocks");
    std::vector<Eigen::Index> off(n + 1, 0);
    for (size_t i = 0; i < n; ++i)
    {
      dims_.push_back((int)gs[i].size());
      off[i + 1] = off[i] + gs[i].size();
    }
    const Eigen::Index D = off[n];
    info_ = Matrix::Zero(D + 1, D + 1);
    size_t idx = 0;
    for (size_t i = 0; i < n; ++i)
      for (size_t j = i; j < n; ++j, ++idx)
      {
        if (Gs[idx].rows() != gs[i].size() || Gs[idx].cols() != gs[j].size())
          throw std::invalid_argument("HessianFactor: block shape does not match the key dimensions");
        info_.block(off[i], off[j], gs[i].size(), gs[j].size()) = Gs[idx];
        if (i != j)
          info_.block(off[j], off[i], gs[j].size(), gs[i].size()) = Gs[idx].transpose();
      }
    for (size_t i = 0; i < n; ++i)
    {
      info_.block(off[i], D, gs[i].size(), 1) = gs[i];
      info_.block(D, off[i], 1, gs[i].size()) = gs[i].transpose();
    }
    info_(D, D) = f;
  }
  const FastVector<Key> &keys() const { return keys_; }
  const std::vector<int> &dims() const { return dims_; }
  const Matrix &augmentedInformation() const { return info_; }
  double constantTerm() const { return f_; }

private:
  FastVector<Key> keys_;
  std::vector<int> dims_;
  Matrix info_;
  double f_;
};
} // namespace gtsam
