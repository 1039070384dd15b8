// Golden vectors of Eigen 3.3.9's  (AtA + damp*diag(AtA)).colPivHouseholderQr().solve(Atb)  in fp32 -- the reference's
// tracker solve (core/system/camera_tracker.cpp:1182-1183) -- produced with the Eigen vendored under
// /root/reference/system/thirdparty/eigen.  Build container only; writes JSON to stdout:
//   g++ -O2 -std=c++14 -I/root/reference/system/thirdparty/eigen tests/golden/make_colpiv_qr_golden.cpp -o /tmp/qrgen
//   /tmp/qrgen > tests/golden/colpiv_qr_eigen339.json
// Cases: well-conditioned, exactly dependent columns, a zero column, near-dependent columns over a range of gaps that
// straddles Eigen's nonzeroPivots() threshold (where a rank() style rule would cut and Eigen does not).
#include <Eigen/Dense>
#include <cstdio>
#include <random>
#include <vector>

typedef Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic> Mat;
typedef Eigen::Matrix<float, Eigen::Dynamic, 1> Vec;

static void emit(const char *name, const Mat &A, const Vec &b, float damp, bool last)
{
  const int n = (int)A.rows();
  Mat M = A;
  M.diagonal() += damp * A.diagonal();
  Eigen::ColPivHouseholderQR<Mat> qr(M);
  Vec x = qr.solve(b);
  std::printf("{\"name\":\"%s\",\"n\":%d,\"damp\":%.9g,\"nonzero_pivots\":%d,\"A\":[", name, n, damp, (int)qr.nonzeroPivots());
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      std::printf("%s%.9g", (i || j) ? "," : "", A(i, j));
  std::printf("],\"b\":[");
  for (int i = 0; i < n; ++i)
    std::printf("%s%.9g", i ? "," : "", b(i));
  std::printf("],\"x\":[");
  for (int i = 0; i < n; ++i)
    std::printf("%s%.9g", i ? "," : "", x(i));
  std::printf("]}%s\n", last ? "" : ",");
}

int main()
{
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::printf("[\n");
  char name[64];
  for (int n : {6, 7})
  {
    Mat J(40, n);
    for (int i = 0; i < 40; ++i)
      for (int j = 0; j < n; ++j)
        J(i, j) = nd(rng);
    Vec r(40);
    for (int i = 0; i < 40; ++i)
      r(i) = nd(rng);
    std::snprintf(name, sizeof name, "full_%d", n);
    emit(name, J.transpose() * J, J.transpose() * r, 1e-4f, false);
    Mat Jd = J;
    Jd.col(n - 1) = Jd.col(0) + Jd.col(1);
    std::snprintf(name, sizeof name, "dependent_%d", n);
    emit(name, Jd.transpose() * Jd, Jd.transpose() * r, 0.f, false);
    Mat Jz = J;
    Jz.col(2).setZero();
    std::snprintf(name, sizeof name, "zerocol_%d", n);
    emit(name, Jz.transpose() * Jz, Jz.transpose() * r, 1e-6f, false);
    for (int e = 1; e <= 7; ++e)
    {
      Mat Jn = J;
      Vec p(40);
      for (int i = 0; i < 40; ++i)
        p(i) = nd(rng);
      Jn.col(n - 1) = Jn.col(0) - Jn.col(1) + std::pow(10.f, -0.5f * (float)e - 0.5f) * p; // gap 1e-1 ... 1e-4
      std::snprintf(name, sizeof name, "near_%d_e%d", n, e);
      emit(name, Jn.transpose() * Jn, Jn.transpose() * r, 1e-6f, false);
    }
  }
  Mat D = Mat::Identity(6, 6);
  Vec db = Vec::Ones(6);
  emit("identity_6", D, db, 1e-4f, true);
  std::printf("]\n");
  return 0;
}
