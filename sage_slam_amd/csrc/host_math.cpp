// host_math.cpp -- pure-CPU host helpers of the C ABI (no device needed).
//
//   sage_camera_pyramid      common/camera_pyramid.h:18-32 + pinhole_camera_impl.h:120-132
//   sage_se3_exp             core/mapping/mapping_utils.h:316-346
//   sage_pose_retract        core/gtsam/gtsam_traits.h:45-70, core/system/camera_tracker.cpp:491-512
//   sage_nearest_psd         the Higham algorithm core/mapping/mapping_utils.h:104-128 intends
//   sage_damped_solve_qr_f32 core/system/camera_tracker.cpp:1182-1183 (colPivHouseholderQr in fp32)
//   sage_track_lm            core/system/camera_tracker.cpp:1156-1279 (+ LMConvergence :527-573)
#include <algorithm>
#include <map>
#include <memory>
#include <atomic>
#include <chrono>
#include <cmath>
#include <limits>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <condition_variable>
#include <cstring>
#include <ctime>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <mutex>
#include <numeric>
#include <random>
#include <vector>

#include "host_math.h"
#include "sage_ba.h"

extern "C" const char *sage_version(void) { return "sage-ba-mi355x 0.1 (gfx950)"; }

extern "C" const char *sage_error_string(int code)
{
  switch (code)
  {
  case SAGE_OK:
    return "ok";
  case SAGE_E_INVALID:
    return "invalid argument";
  case SAGE_E_UNSUPPORTED:
    return "unsupported CS/FS/levels combination";
  case SAGE_E_NOT_PSD:
    return "normal equations not positive definite";
  case SAGE_E_STATE:
    return "call order violated";
  case SAGE_E_NO_OVERLAP:
    return "tracker: no overlap between the frame to track and the keyframe";
  default:
    return code > 0 ? "HIP runtime error (hipError_t)" : "unknown error";
  }
}

extern "C" int sage_camera_pyramid(const SageCamera *base, int levels, SagePyramid *out)
{
  if (!base || !out || levels < 1 || levels > SAGE_MAX_LEVELS)
    return SAGE_E_INVALID;
  std::memset(out, 0, sizeof(*out));
  out->levels = levels;
  int off = 0;
  for (int i = 0; i < levels; ++i)
  {
    SageCamera c = (i == 0) ? *base : out->cam[i - 1];
    if (i != 0)
    {
      // new size = (size_t)(w/2), (size_t)(h/2); fx,u0 *= new_w/w ; fy,v0 *= new_h/h   (all fp32)
      const size_t nw = (size_t)(c.w / 2), nh = (size_t)(c.h / 2);
      const float xr = (float)nw / c.w, yr = (float)nh / c.h;
      c.fx *= xr;
      c.fy *= yr;
      c.cx *= xr;
      c.cy *= yr;
      c.w = (float)nw;
      c.h = (float)nh;
    }
    out->cam[i] = c;
    out->level_offsets[i] = off;
    off += (int)c.w * (int)c.h;
  }
  out->P = off;
  return SAGE_OK;
}

// mapper.cpp:1326-1333: the keyframe's sample permutation (the reference's own standard-library calls)
extern "C" int sage_shuffle_indices(int64_t seed, int64_t n, int64_t *idx)
{
  if (n < 0 || (n > 0 && !idx))
    return SAGE_E_INVALID;
  std::vector<long> indices((size_t)n);
  std::iota(indices.begin(), indices.end(), 0);
  std::mt19937 g;
  g.seed((long)seed);
  std::shuffle(indices.begin(), indices.end(), g);
  for (int64_t i = 0; i < n; ++i)
    idx[i] = indices[(size_t)i];
  return SAGE_OK;
}

extern "C" void sage_se3_exp(const float *omega, const float *v, float *R, float *t)
{
  if (!omega || !v || !R || !t) // (void helpers: a null argument is a no-op, never a fault)
    return;
  float theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  float n[3] = {1.f, 0.f, 0.f}; // "a casual rotation direction vector" when theta == 0
  if (theta > 0)
  {
    n[0] = omega[0] / theta;
    n[1] = omega[1] / theta;
    n[2] = omega[2] / theta;
  }
  theta = std::max(theta, 1.0e-14f);
  const float s = std::sin(theta), c = std::cos(theta);
  const float K[3][3] = {{0, -n[2], n[1]}, {n[2], 0, -n[0]}, {-n[1], n[0], 0}};
  float K2[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      K2[i][j] = K[i][0] * K[0][j] + K[i][1] * K[1][j] + K[i][2] * K[2][j];
  const float a = (1.0f - c) / theta, b = (theta - s) / theta;
  for (int i = 0; i < 3; ++i)
  {
    float acc = 0.f;
    for (int j = 0; j < 3; ++j)
    {
      const float id = (i == j) ? 1.f : 0.f;
      R[i * 3 + j] = id + s * K[i][j] + (1.0f - c) * K2[i][j];
      acc += (id + a * K[i][j] + b * K2[i][j]) * v[j];
    }
    t[i] = acc;
  }
}

extern "C" void sage_pose_retract(const float *pose, const float *d, float *out)
{
  if (!pose || !d || !out)
    return;
  float dR[9], dt[3];
  sage_se3_exp(d + 3, d, dR, dt); // delta = [v, omega]
  float R[9], t[3];
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j)
      R[i * 3 + j] = dR[i * 3 + 0] * pose[0 * 3 + j] + dR[i * 3 + 1] * pose[1 * 3 + j] + dR[i * 3 + 2] * pose[2 * 3 + j];
    t[i] = dR[i * 3 + 0] * pose[9] + dR[i * 3 + 1] * pose[10] + dR[i * 3 + 2] * pose[11] + dt[i];
  }
  std::memcpy(out, R, sizeof(R));
  std::memcpy(out + 9, t, sizeof(t));
}

// ---------------------------------------------------------------- dense helpers (double)
namespace sage
{

// cyclic Jacobi eigen-decomposition of a symmetric matrix: A = V diag(w) V^T (columns of V)
void sym_eig(std::vector<double> &A, int n, std::vector<double> &w, std::vector<double> &V)
{
  V.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i)
    V[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep)
  {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        (i == j ? diag : off) += A[(size_t)i * n + j] * A[(size_t)i * n + j];
    if (off <= 1e-30 * (diag + 1e-300))
      break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q)
      {
        const double apq = A[(size_t)p * n + q];
        if (std::fabs(apq) < 1e-300)
          continue;
        const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < n; ++k)
        {
          const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
          A[(size_t)k * n + p] = c * akp - s * akq;
          A[(size_t)k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k)
        {
          const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
          A[(size_t)p * n + k] = c * apk - s * aqk;
          A[(size_t)q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k)
        {
          const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = c * vkp - s * vkq;
          V[(size_t)k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  w.resize(n);
  for (int i = 0; i < n; ++i)
    w[i] = A[(size_t)i * n + i];
}

// Symmetric eigen-decomposition by Householder tridiagonalisation + implicit QL (the EISPACK tred2 / tql2 pair): the same
// A = V diag(w) V^T as sym_eig at ~1/15 of the time for the 45 x 45 / 78 x 78 factor matrices (Higham projection of
// sage_nearest_psd: the per-factor host cost of the gtsam path).  Returns false if QL does not converge (the caller falls
// back to the Jacobi sweeps).  A is destroyed.
static bool sym_eig_ql(std::vector<double> &A, int n, std::vector<double> &w, std::vector<double> &V)
{
  std::vector<double> e(n, 0.0);
  w.assign(n, 0.0);
  auto a = [&](int i, int j) -> double & { return A[(size_t)i * n + j]; };
  for (int i = n - 1; i >= 1; --i)
  {
    const int l = i - 1;
    double h = 0.0, scale = 0.0;
    if (l > 0)
    {
      for (int k = 0; k <= l; ++k)
        scale += std::fabs(a(i, k));
      if (scale == 0.0)
        e[i] = a(i, l);
      else
      {
        for (int k = 0; k <= l; ++k)
        {
          a(i, k) /= scale;
          h += a(i, k) * a(i, k);
        }
        double f = a(i, l);
        double g = f >= 0.0 ? -std::sqrt(h) : std::sqrt(h);
        e[i] = scale * g;
        h -= f * g;
        a(i, l) = f - g;
        f = 0.0;
        for (int j = 0; j <= l; ++j)
        {
          a(j, i) = a(i, j) / h;
          g = 0.0;
          for (int k = 0; k <= j; ++k)
            g += a(j, k) * a(i, k);
          for (int k = j + 1; k <= l; ++k)
            g += a(k, j) * a(i, k);
          e[j] = g / h;
          f += e[j] * a(i, j);
        }
        const double hh = f / (h + h);
        for (int j = 0; j <= l; ++j)
        {
          f = a(i, j);
          e[j] = g = e[j] - hh * f;
          for (int k = 0; k <= j; ++k)
            a(j, k) -= f * e[k] + g * a(i, k);
        }
      }
    }
    else
      e[i] = a(i, l);
    w[i] = h;
  }
  w[0] = 0.0;
  e[0] = 0.0;
  for (int i = 0; i < n; ++i)
  {
    const int l = i - 1;
    if (w[i] != 0.0)
      for (int j = 0; j <= l; ++j)
      {
        double g = 0.0;
        for (int k = 0; k <= l; ++k)
          g += a(i, k) * a(k, j);
        for (int k = 0; k <= l; ++k)
          a(k, j) -= g * a(k, i);
      }
    w[i] = a(i, i);
    a(i, i) = 1.0;
    for (int j = 0; j <= l; ++j)
      a(j, i) = a(i, j) = 0.0;
  }
  // rows of Z = the accumulated transformation transposed: the QL rotations then touch two contiguous rows
  std::vector<double> Z((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k)
      Z[(size_t)i * n + k] = a(k, i);
  for (int i = 1; i < n; ++i)
    e[i - 1] = e[i];
  e[n - 1] = 0.0;
  const double eps = std::numeric_limits<double>::epsilon();
  for (int l = 0; l < n; ++l)
  {
    int iter = 0, m;
    do
    {
      for (m = l; m < n - 1; ++m)
      {
        const double dd = std::fabs(w[m]) + std::fabs(w[m + 1]);
        if (std::fabs(e[m]) <= eps * dd)
          break;
      }
      if (m != l)
      {
        if (iter++ == 80)
          return false;
        double g = (w[l + 1] - w[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = w[m] - w[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
        double sn = 1.0, cs = 1.0, pp = 0.0;
        int i;
        for (i = m - 1; i >= l; --i)
        {
          double f = sn * e[i];
          const double b = cs * e[i];
          e[i + 1] = (r = std::hypot(f, g));
          if (r == 0.0)
          {
            w[i + 1] -= pp;
            e[m] = 0.0;
            break;
          }
          sn = f / r;
          cs = g / r;
          g = w[i + 1] - pp;
          r = (w[i] - g) * sn + 2.0 * cs * b;
          w[i + 1] = g + (pp = sn * r);
          g = cs * r - b;
          double *zi = &Z[(size_t)i * n], *zj = &Z[(size_t)(i + 1) * n];
          for (int k = 0; k < n; ++k)
          {
            f = zj[k];
            zj[k] = sn * zi[k] + cs * f;
            zi[k] = cs * zi[k] - sn * f;
          }
        }
        if (r == 0.0 && i >= l)
          continue;
        w[l] -= pp;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  V.resize((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k)
      V[(size_t)k * n + i] = Z[(size_t)i * n + k];
  return true;
}

// sum_k a[k] * b[k] with four independent partial sums (strict fp semantics keep the compiler from splitting one chain)
static inline double dot4(const double *a, const double *b, int n)
{
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int k = 0;
  for (; k + 4 <= n; k += 4)
  {
    s0 += a[k] * b[k];
    s1 += a[k + 1] * b[k + 1];
    s2 += a[k + 2] * b[k + 2];
    s3 += a[k + 3] * b[k + 3];
  }
  for (; k < n; ++k)
    s0 += a[k] * b[k];
  return (s0 + s1) + (s2 + s3);
}

// LDLT-style positive (semi-)definiteness test (Eigen::LDLT::isPositive): all pivots >= 0.
static bool is_psd(const std::vector<double> &M, int n)
{
  std::vector<double> L(M), Ld((size_t)n * n, 0.0); // Ld[j][k] = L[j][k] * d_k: every inner sum is a contiguous dot product
  for (int j = 0; j < n; ++j)
  {
    const double d = L[(size_t)j * n + j] - dot4(&L[(size_t)j * n], &Ld[(size_t)j * n], j);
    if (d < 0.0)
      return false;
    L[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i)
    {
      const double sv = L[(size_t)i * n + j] - dot4(&L[(size_t)i * n], &Ld[(size_t)j * n], j);
      const double lij = (d != 0.0) ? sv / d : 0.0;
      L[(size_t)i * n + j] = lij;
      Ld[(size_t)i * n + j] = lij * d;
    }
  }
  return true;
}

} // namespace sage

extern "C" int sage_nearest_psd(const double *M, int n, double *out)
{
  if (!M || !out || n < 1)
    return SAGE_E_INVALID;
  // B = (M + M^T)/2 ; H = polar factor of B = V |Lambda| V^T ; A2 = (B+H)/2 ; A3 = sym(A2)
  std::vector<double> B((size_t)n * n), w, V;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      B[(size_t)i * n + j] = 0.5 * (M[(size_t)i * n + j] + M[(size_t)j * n + i]);
  std::vector<double> tmp(B);
  if (!sage::sym_eig_ql(tmp, n, w, V))
  {
    tmp = B;
    sage::sym_eig(tmp, n, w, V);
  }
  std::vector<double> A3((size_t)n * n, 0.0), VW((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k)
      VW[(size_t)i * n + k] = V[(size_t)i * n + k] * std::fabs(w[k]);
  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) // H = V |Lambda| V^T is symmetric: one triangle, mirrored (= the symmetrisation of A2)
    {
      const double h = sage::dot4(&VW[(size_t)i * n], &V[(size_t)j * n], n);
      A3[(size_t)i * n + j] = A3[(size_t)j * n + i] = 0.5 * (B[(size_t)i * n + j] + h);
    }
  // bump by (-min_eig*k + spacing) until LDLT-positive (mapping_utils.h:119-126).  The eigenvalues of A3 + c I are those
  // of A3 plus c: ONE decomposition serves every round of the loop.
  double k = 1; // (a double: 60 doublings of an int would overflow)
  const double spacing = 1e-15;
  bool have_min = false;
  double mn = 0.0;
  for (int it = 0; it < 60 && !sage::is_psd(A3, n); ++it)
  {
    if (!have_min)
    {
      std::vector<double> t2(A3), w2, V2;
      if (!sage::sym_eig_ql(t2, n, w2, V2))
      {
        t2 = A3;
        sage::sym_eig(t2, n, w2, V2);
      }
      mn = *std::min_element(w2.begin(), w2.end());
      have_min = true;
    }
    // a bump below the rounding granularity of the diagonal would move the tracked minimum but not the matrix: never
    // less than one ulp of the largest diagonal entry, and never negative (the tracked minimum turns positive after the
    // first round; the matrix is re-measured then instead of being walked back)
    double dmax = 0.0;
    for (int i = 0; i < n; ++i)
      dmax = std::max(dmax, std::fabs(A3[(size_t)i * n + i]));
    const double ulp = dmax * 2.220446049250313e-16;
    double bump = -mn * k + spacing;
    if (bump < ulp)
    {
      if (mn > 0.0) // the estimate says PSD but the LDLT test disagrees: measure again on the matrix as it is now
        have_min = false;
      bump = ulp;
    }
    for (int i = 0; i < n; ++i)
      A3[(size_t)i * n + i] += bump;
    mn += bump;
    k *= 2;
  }
  std::memcpy(out, A3.data(), sizeof(double) * n * n);
  return SAGE_OK;
}

// ---------------------------------------------------------------- NearestPsd exactly as the reference wrote it
namespace sage
{
// Two-sided Jacobi SVD of a square real matrix the way Eigen 3.3.9's JacobiSVD runs it (the reference's
// `Eigen::JacobiSVD<T> svd(B, ComputeThinV)`, mapping_utils.h:111): work = B / max|B|; sweeps over (p, q < p) while any
// off-diagonal pair exceeds max(DBL_MIN, 2 eps * max diagonal); each pair: a left rotation that symmetrises the 2x2
// block, then a symmetric Jacobi rotation on both sides; |diagonal| = singular values sorted descending with the columns
// of V swapped alongside.  Only V (what `matrixV()` returns) and sigma are produced.  The point of restating the
// procedure instead of calling any SVD: the reference's  H = V^T diag(sigma) V  is NOT invariant under the sign / order
// conventions of V, so it can only be reproduced by walking the same rotations.
static void eigen_jacobi_svd_v(const std::vector<double> &B, int n, std::vector<double> &V, std::vector<double> &sv)
{
  double scale = 0.0;
  for (double v : B)
    scale = std::max(scale, std::fabs(v));
  if (scale == 0.0)
    scale = 1.0;
  std::vector<double> Wk(B);
  for (double &v : Wk)
    v /= scale;
  V.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i)
    V[(size_t)i * n + i] = 1.0;
  auto W = [&](int i, int j) -> double & { return Wk[(size_t)i * n + j]; };
  const double tiny = std::numeric_limits<double>::min(), prec = 2.0 * std::numeric_limits<double>::epsilon();
  double max_diag = 0.0;
  for (int i = 0; i < n; ++i)
    max_diag = std::max(max_diag, std::fabs(W(i, i)));
  bool finished = false;
  while (!finished)
  {
    finished = true;
    for (int p = 1; p < n; ++p)
      for (int q = 0; q < p; ++q)
      {
        const double thr = std::max(tiny, prec * max_diag);
        if (!(std::fabs(W(p, q)) > thr || std::fabs(W(q, p)) > thr))
          continue;
        finished = false;
        // 2x2 block [[W(p,p), W(p,q)], [W(q,p), W(q,q)]]: rot1 makes it symmetric ...
        double m00 = W(p, p), m01 = W(p, q), m10 = W(q, p), m11 = W(q, q);
        double c1, s1;
        {
          const double t = m00 + m11, d = m10 - m01;
          if (std::fabs(d) < tiny)
          {
            s1 = 0.0;
            c1 = 1.0;
          }
          else
          {
            const double u = t / d, tmp = std::sqrt(1.0 + u * u);
            s1 = 1.0 / tmp;
            c1 = u / tmp;
          }
        }
        {
          const double a0 = c1 * m00 + s1 * m10, a1 = c1 * m01 + s1 * m11;
          const double b0 = -s1 * m00 + c1 * m10, b1 = -s1 * m01 + c1 * m11;
          m00 = a0; m01 = a1; m10 = b0; m11 = b1;
        }
        // ... j_right diagonalises the symmetric block (makeJacobi(x = m00, y = m01, z = m11))
        double cr, sr;
        {
          const double deno = 2.0 * std::fabs(m01);
          if (deno < tiny)
          {
            cr = 1.0;
            sr = 0.0;
          }
          else
          {
            const double tau = (m00 - m11) / deno, w = std::sqrt(tau * tau + 1.0);
            const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
            const double sign_t = t > 0.0 ? 1.0 : -1.0, nn = 1.0 / std::sqrt(t * t + 1.0);
            sr = -sign_t * (m01 / std::fabs(m01)) * std::fabs(t) * nn;
            cr = nn;
          }
        }
        // j_left = rot1 * j_right^T
        const double cl = c1 * cr + s1 * sr, sl = -c1 * sr + s1 * cr;
        for (int k = 0; k < n; ++k) // rows p, q from the left
        {
          const double x = W(p, k), y = W(q, k);
          W(p, k) = cl * x + sl * y;
          W(q, k) = -sl * x + cl * y;
        }
        for (int k = 0; k < n; ++k) // columns p, q from the right (work matrix and V)
        {
          const double x = W(k, p), y = W(k, q);
          W(k, p) = cr * x - sr * y;
          W(k, q) = sr * x + cr * y;
          const double vx = V[(size_t)k * n + p], vy = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = cr * vx - sr * vy;
          V[(size_t)k * n + q] = sr * vx + cr * vy;
        }
        max_diag = std::max(max_diag, std::max(std::fabs(W(p, p)), std::fabs(W(q, q))));
      }
  }
  sv.resize(n);
  for (int i = 0; i < n; ++i)
    sv[i] = std::fabs(W(i, i)) * scale; // (negative diagonals flip columns of U, which the reference never asks for)
  for (int i = 0; i < n; ++i)
  {
    int pos = i;
    for (int j = i + 1; j < n; ++j)
      if (sv[j] > sv[pos])
        pos = j;
    if (sv[pos] == 0.0)
      break;
    if (pos != i)
    {
      std::swap(sv[i], sv[pos]);
      for (int k = 0; k < n; ++k)
        std::swap(V[(size_t)k * n + i], V[(size_t)k * n + pos]);
    }
  }
}

// Eigen::LDLT::isPositive(): the pivoted LDL^T (largest remaining |diagonal| first) meets no negative pivot
static bool eigen_ldlt_is_positive(const std::vector<double> &M, int n)
{
  std::vector<double> A(M);
  auto a = [&](int i, int j) -> double & { return A[(size_t)i * n + j]; };
  for (int k = 0; k < n; ++k)
  {
    int piv = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(a(i, i)) > std::fabs(a(piv, piv)))
        piv = i;
    if (piv != k)
    {
      for (int j = 0; j < n; ++j)
        std::swap(a(k, j), a(piv, j));
      for (int i = 0; i < n; ++i)
        std::swap(a(i, k), a(i, piv));
    }
    const double d = a(k, k);
    if (d < 0.0)
      return false;
    if (d == 0.0)
      continue;
    for (int i = k + 1; i < n; ++i)
    {
      const double l = a(i, k) / d;
      for (int j = k + 1; j < n; ++j)
        a(i, j) -= l * a(k, j);
    }
  }
  return true;
}
} // namespace sage

extern "C" int sage_nearest_psd_reference(const double *M, int n, double *out)
{
  if (!M || !out || n < 1)
    return SAGE_E_INVALID;
  std::vector<double> B((size_t)n * n), V, sv;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      B[(size_t)i * n + j] = (M[(size_t)i * n + j] + M[(size_t)j * n + i]) / 2;
  sage::eigen_jacobi_svd_v(B, n, V, sv);
  // H = V^T diag(sigma) V  (mapping_utils.h:112, as written): H_ij = sum_k sigma_k V_ki V_kj
  std::vector<double> A3((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
    {
      double h = 0.0;
      for (int k = 0; k < n; ++k)
        h += V[(size_t)k * n + i] * sv[k] * V[(size_t)k * n + j];
      A3[(size_t)i * n + j] = (B[(size_t)i * n + j] + h) / 2;
    }
  std::vector<double> S(A3);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      A3[(size_t)i * n + j] = (S[(size_t)i * n + j] + S[(size_t)j * n + i]) / 2;
  double k = 1; // (a double: 60 doublings of an int would overflow)
  const double spacing = 1e-15;
  for (int it = 0; it < 60 && !sage::eigen_ldlt_is_positive(A3, n); ++it)
  {
    std::vector<double> t2(A3), w2, V2;
    sage::sym_eig(t2, n, w2, V2);
    const double mn = *std::min_element(w2.begin(), w2.end());
    for (int i = 0; i < n; ++i)
      A3[(size_t)i * n + i] += -mn * k + spacing;
    k *= 2;
  }
  std::memcpy(out, A3.data(), sizeof(double) * n * n);
  return SAGE_OK;
}

// ---------------------------------------------------------------- HessianFactor blocks (a6 / a7)
static int factor_dims(int type, int CS, int dims[6])
{
  if (CS < 1 || (type != 0 && type != 1))
    return 0;
  if (type == 0)
  {
    const int d[4] = {6, 6, CS, 1};
    std::copy(d, d + 4, dims);
    return 4;
  }
  const int d[6] = {6, 6, CS, CS, 1, 1};
  std::copy(d, d + 6, dims);
  return 6;
}

extern "C" int sage_factor_block_count(int type, int CS)
{
  int dims[6];
  const int nk = factor_dims(type, CS, dims);
  if (!nk)
    return SAGE_E_INVALID;
  int total = 0;
  for (int i = 0; i < nk; ++i)
    for (int j = i; j < nk; ++j)
      total += dims[i] * dims[j];
  return total;
}

// widen to double and project: psd_mode 0 none, 1 Higham, 2 NearestPsd as the reference wrote it
extern "C" int sage_factor_psd(int type, int CS, const float *AtA, int psd_mode, double *C_out)
{
  int dims[6];
  const int nk = factor_dims(type, CS, dims);
  if (!nk || !AtA || !C_out || psd_mode < 0 || psd_mode > 2)
    return SAGE_E_INVALID;
  int D = 0;
  for (int i = 0; i < nk; ++i)
    D += dims[i];
  std::vector<double> M((size_t)D * D);
  for (size_t i = 0; i < M.size(); ++i)
    M[i] = (double)AtA[i]; // AtA_.cast<double>() (photometric_factor.cpp:305, :142)
  if (psd_mode == 1)
    return sage_nearest_psd(M.data(), D, C_out);
  if (psd_mode == 2)
    return sage_nearest_psd_reference(M.data(), D, C_out);
  std::memcpy(C_out, M.data(), M.size() * sizeof(double));
  return SAGE_OK;
}

// the upper-triangular blocks G11 G12 .. Gnn of a (projected) D x D matrix and g = Atb in the reference's push order
extern "C" int sage_factor_cut_blocks(int type, int CS, const double *C, const float *Atb, double *G_out, double *g_out,
                                      int32_t *dims_out, int32_t *nkeys_out)
{
  int dims[6];
  const int nk = factor_dims(type, CS, dims);
  if (!nk || !C || !Atb || !G_out || !g_out)
    return SAGE_E_INVALID;
  int D = 0, off[6];
  for (int i = 0; i < nk; ++i)
  {
    off[i] = D;
    D += dims[i];
  }
  double *o = G_out;
  for (int i = 0; i < nk; ++i)
    for (int j = i; j < nk; ++j)
      for (int r = 0; r < dims[i]; ++r)
        for (int c = 0; c < dims[j]; ++c)
          *o++ = C[(size_t)(off[i] + r) * D + off[j] + c];
  for (int i = 0; i < D; ++i)
    g_out[i] = (double)Atb[i];
  if (dims_out)
    for (int i = 0; i < nk; ++i)
      dims_out[i] = dims[i];
  if (nkeys_out)
    *nkeys_out = nk;
  return SAGE_OK;
}

extern "C" int sage_factor_hessian_blocks(int type, int CS, const float *AtA, const float *Atb, int psd_mode,
                                          double *G_out, double *g_out, int32_t *dims_out, int32_t *nkeys_out)
{
  int dims[6];
  const int nk = factor_dims(type, CS, dims);
  if (!nk || !AtA || !Atb || !G_out || !g_out || psd_mode < 0 || psd_mode > 2)
    return SAGE_E_INVALID;
  int D = 0;
  for (int i = 0; i < nk; ++i)
    D += dims[i];
  std::vector<double> C((size_t)D * D);
  const int rc = sage_factor_psd(type, CS, AtA, psd_mode, C.data());
  if (rc)
    return rc;
  return sage_factor_cut_blocks(type, CS, C.data(), Atb, G_out, g_out, dims_out, nkeys_out);
}

extern "C" int sage_damped_solve_qr_f32(const float *A, const float *b, int n, float damp, float *x)
{
  if (!A || !b || !x || n < 1 || n > 64)
    return SAGE_E_INVALID;
  // (AtA + damp*diag(AtA)).colPivHouseholderQr().solve(Atb) in fp32 (camera_tracker.cpp:1182-1183), restated after Eigen
  // 3.3.9 step by step so that the RANK DECISION is Eigen's: ColPivHouseholderQR::computeInPlace (ColPivHouseholderQR.h:
  // 480-570) pivots on DOWNDATED column norms (LAPACK xGEQPF rule, recomputed when the downdate loses accuracy), counts
  // m_nonzero_pivots = the first k whose largest remaining squared norm is < (eps * max initial norm)^2 / rows * (rows-k),
  // makeHouseholderInPlace's (tau, beta) convention (Householder.h:65-93), and _solve_impl (:589-610): only the first
  // nonzero_pivots reflectors touch the right-hand side, the leading triangle is solved, the other components are zero.
  // Pinned by tests/golden/colpiv_qr_eigen339.json (produced with the vendored Eigen).  Column-major like Eigen.
  const float eps = std::numeric_limits<float>::epsilon();
  std::vector<float> M((size_t)n * n), c(b, b + n), hco(n, 0.f), upd(n), dir(n), tmp(n);
  auto at = [&](int i, int j) -> float & { return M[(size_t)j * n + i]; };
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      at(i, j) = A[(size_t)i * n + j] + (i == j ? damp * A[(size_t)i * n + i] : 0.f);
  auto col_norm = [&](int j, int from) {
    float sq = 0.f;
    for (int i = from; i < n; ++i)
      sq += at(i, j) * at(i, j);
    return std::sqrt(sq);
  };
  float maxcol = 0.f;
  for (int j = 0; j < n; ++j)
  {
    dir[j] = upd[j] = col_norm(j, 0);
    maxcol = std::max(maxcol, upd[j]);
  }
  const float thr_helper = (maxcol * eps) * (maxcol * eps) / (float)n;
  const float downdate_thr = std::sqrt(eps);
  std::vector<int> transp(n);
  int rank = n;
  for (int k = 0; k < n; ++k)
  {
    int best = k;
    for (int j = k + 1; j < n; ++j) // maxCoeff: the first maximal entry
      if (upd[j] > upd[best])
        best = j;
    const float biggest_sq = upd[best] * upd[best];
    if (rank == n && biggest_sq < thr_helper * (float)(n - k))
      rank = k;
    transp[k] = best;
    if (best != k)
    {
      for (int i = 0; i < n; ++i)
        std::swap(at(i, k), at(i, best));
      std::swap(upd[k], upd[best]);
      std::swap(dir[k], dir[best]);
    }
    // makeHouseholderInPlace on column k, rows k..n-1
    float tail_sq = 0.f;
    for (int i = k + 1; i < n; ++i)
      tail_sq += at(i, k) * at(i, k);
    const float c0 = at(k, k);
    float tau, beta;
    if (tail_sq <= std::numeric_limits<float>::min())
    {
      tau = 0.f;
      beta = c0;
      for (int i = k + 1; i < n; ++i)
        at(i, k) = 0.f;
    }
    else
    {
      beta = std::sqrt(c0 * c0 + tail_sq);
      if (c0 >= 0.f)
        beta = -beta;
      for (int i = k + 1; i < n; ++i)
        at(i, k) /= (c0 - beta);
      tau = (beta - c0) / beta;
    }
    hco[k] = tau;
    at(k, k) = beta;
    // applyHouseholderOnTheLeft to the trailing columns
    if (n - k == 1)
    { /* no trailing block */ }
    else if (tau != 0.f)
      for (int j = k + 1; j < n; ++j)
      {
        float t = 0.f;
        for (int i = k + 1; i < n; ++i)
          t += at(i, k) * at(i, j);
        t += at(k, j);
        at(k, j) -= tau * t;
        for (int i = k + 1; i < n; ++i)
          at(i, j) -= tau * at(i, k) * t;
      }
    for (int j = k + 1; j < n; ++j)
      if (upd[j] != 0.f)
      {
        float t = std::fabs(at(k, j)) / upd[j];
        t = (1.f + t) * (1.f - t);
        t = t < 0.f ? 0.f : t;
        const float q = upd[j] / dir[j];
        const float t2 = t * (q * q);
        if (t2 <= downdate_thr)
          dir[j] = upd[j] = col_norm(j, k + 1);
        else
          upd[j] *= std::sqrt(t);
      }
  }
  std::vector<int> perm(n);
  for (int i = 0; i < n; ++i)
    perm[i] = i;
  for (int k = 0; k < n; ++k)
    std::swap(perm[k], perm[transp[k]]);
  for (int i = 0; i < n; ++i)
    x[i] = 0.f;
  if (rank == 0)
    return SAGE_OK;
  for (int k = 0; k < rank; ++k) // c = H_{rank-1} ... H_0 c
  {
    if (hco[k] == 0.f)
      continue;
    float t = c[k];
    for (int i = k + 1; i < n; ++i)
      t += at(i, k) * c[i];
    c[k] -= hco[k] * t;
    for (int i = k + 1; i < n; ++i)
      c[i] -= hco[k] * at(i, k) * t;
  }
  for (int i = rank - 1; i >= 0; --i)
  {
    float t = c[i];
    for (int j = i + 1; j < rank; ++j)
      t -= at(i, j) * tmp[j];
    tmp[i] = t / at(i, i);
  }
  for (int i = 0; i < rank; ++i)
    x[perm[i]] = tmp[i];
  return SAGE_OK;
}

// ---------------------------------------------------------------- tracker LM policy
extern "C" void sage_lm_config_default(SageLmConfig *c)
{
  if (!c)
    return;
  // system/configs/slam_run.flags:17-23
  c->max_num_iters = 40;
  c->min_grad_thresh = 1.0e-4f;
  c->min_param_inc_thresh = 1.0e-2f;
  c->init_damp = 1.0e-4f;
  c->min_damp = 1.0e-6f;
  c->max_damp = 1.0e-2f;
  c->damp_dec_factor = 10.f;
  c->damp_inc_factor = 100.f;
  c->jac_update_err_inc_threshold = 1.0e-2f;
  c->max_inner_evals = 0;
  c->no_overlap_error = 0.f;
  c->linearize_at_candidate = 0;
}

namespace sage
{

// RotationToAngleAxis(R, 1e-6) as written in core/mapping/mapping_utils.h:143-212, including its quirks:
// the returned vector uses the HALF angle atan2(sin, cos) (the "2.0 *" of the torchgeometry original is
// missing) and case c0 divides by sqrt(0).  Only used by the convergence test.
void rotation_to_angle_axis_as_reference(const float *R, float eps, float *out)
{
  // rmat_t = R^T
  auto rt = [&](int i, int j) { return R[j * 3 + i]; };
  const bool d2 = rt(2, 2) < eps;
  const bool d0_d1 = rt(0, 0) > rt(1, 1);
  const bool d0_nd1 = rt(0, 0) < -rt(1, 1);
  const float t0 = 1.0f + rt(0, 0) - rt(1, 1) - rt(2, 2);
  const float t1 = 1.0f - rt(0, 0) + rt(1, 1) - rt(2, 2);
  const float t2 = 1.0f - rt(0, 0) - rt(1, 1) + rt(2, 2);
  const float t3 = 1.0f + rt(0, 0) + rt(1, 1) + rt(2, 2);
  float q[4], den;
  if (d2 && d0_d1)
  {
    q[0] = rt(1, 2) - rt(2, 1); q[1] = t0; q[2] = rt(0, 1) + rt(1, 0); q[3] = rt(2, 0) + rt(0, 2);
    den = 0.f; // t0*mask_c1 (sic) -> 0 in case c0
  }
  else if (d2)
  {
    q[0] = rt(2, 0) - rt(0, 2); q[1] = rt(0, 1) + rt(1, 0); q[2] = t1; q[3] = rt(1, 2) + rt(2, 1);
    den = t0 + t1; // t0*mask_c1 + t1*mask_c1 (sic)
  }
  else if (d0_nd1)
  {
    q[0] = rt(0, 1) - rt(1, 0); q[1] = rt(2, 0) + rt(0, 2); q[2] = rt(1, 2) + rt(2, 1); q[3] = t2;
    den = t2;
  }
  else
  {
    q[0] = t3; q[1] = rt(1, 2) - rt(2, 1); q[2] = rt(2, 0) - rt(0, 2); q[3] = rt(0, 1) - rt(1, 0);
    den = t3;
  }
  const float sq = std::sqrt(den);
  for (int i = 0; i < 4; ++i)
    q[i] = 0.5f * q[i] / sq;
  const float ss = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const float sn = std::sqrt(ss), cs = q[0];
  const float two_theta = cs < 0.0f ? std::atan2(-sn, -cs) : std::atan2(sn, cs);
  const float k = ss > 0.0f ? two_theta / sn : 2.0f;
  out[0] = k * q[1];
  out[1] = k * q[2];
  out[2] = k * q[3];
}

static bool lm_converged(const SageLmConfig &cfg, int dof, const float *pose, float scale, const float *Atb,
                         const float *sol)
{
  float rv[3];
  rotation_to_angle_axis_as_reference(pose, 1.0e-6f, rv);
  float max_grad = 0.f;
  for (int i = 0; i < dof; ++i)
    max_grad = std::max(max_grad, std::fabs(Atb[i]));
  // max( solution / (|[t, rotvec, scale]| + 1e-8) ) -- signed numerator, as written (:531-536)
  const float den[7] = {std::fabs(pose[9]), std::fabs(pose[10]), std::fabs(pose[11]),
                        std::fabs(rv[0]), std::fabs(rv[1]), std::fabs(rv[2]), std::fabs(scale)};
  float max_inc = -INFINITY;
  bool nan = false;
  for (int i = 0; i < dof; ++i)
  {
    const float r = sol[i] / (den[i] + 1.0e-8f);
    if (r != r)
      nan = true;
    max_inc = std::max(max_inc, r);
  }
  if (nan)
    max_inc = NAN;
  return max_grad < cfg.min_grad_thresh || max_inc < cfg.min_param_inc_thresh;
}

} // namespace sage

extern "C" int sage_track_lm(const SageLmConfig *cfgp, int dof, SageTrackLinearizeFn lin, SageTrackErrorFn errf,
                             void *ctx, float *pose12, float *scale, float *final_error, int *iters,
                             SageLmTraceEntry *trace, int trace_cap, int *trace_len)
{
  if (!cfgp || !lin || !errf || !pose12 || (dof != 6 && dof != 7) || (dof == 7 && !scale))
    return SAGE_E_INVALID;
  const SageLmConfig cfg = *cfgp;
  float AtA[49], Atb[7], sol[7] = {0, 0, 0, 0, 0, 0, 0};
  float guess[12], cand[12];
  std::memcpy(guess, pose12, sizeof(guess));
  float guess_scale = scale ? *scale : 1.0f, cand_scale = guess_scale;
  bool update_jac = true;
  float prev_error = 0.f, curr_error = 1.f, cand_error = 0.f;
  long curr_iter = 0;
  float damp = cfg.init_damp;
  int ntrace = 0;
  auto clampd = [&](float d) { return std::min(std::max(cfg.min_damp, d), cfg.max_damp); };
  int rc = 0;
  while (true)
  {
    // skip the Jacobian when the last step changed the error too little (:1159)
    if (std::fabs(curr_error - prev_error) / prev_error > cfg.jac_update_err_inc_threshold)
    {
      float lin_error = 0.f;
      if ((rc = lin(ctx, guess, guess_scale, AtA, Atb, &lin_error)) != 0)
        return rc;
      if (curr_iter == 0) // update_error only on the first pass (:1166, :1491); later the accepted candidate's error stands
        curr_error = lin_error;
      update_jac = true;
    }
    else
      update_jac = false;
    // TrackFrame without the match-geometry term: "no overlap" ends the tracking with a failure (:1515-1519)
    if (cfg.no_overlap_error > 0.f && curr_error >= cfg.no_overlap_error)
    {
      rc = SAGE_E_NO_OVERLAP;
      break;
    }
    curr_iter += 1;
    if ((rc = sage_damped_solve_qr_f32(AtA, Atb, dof, damp, sol)) != 0)
      return rc;
    if (sage::lm_converged(cfg, dof, guess, guess_scale, Atb, sol))
      break;
    bool accepted = false;
    while (true)
    {
      sage_pose_retract(guess, sol, cand); // UpdateVariables (:467-512)
      cand_scale = dof == 7 ? guess_scale + sol[6] : guess_scale;
      if ((rc = errf(ctx, cand, cand_scale, &cand_error)) != 0)
        return rc;
      if (cand_error < curr_error)
      {
        accepted = true;
        break;
      }
      else if (damp < cfg.max_damp)
      {
        damp = clampd(damp * cfg.damp_inc_factor);
        if ((rc = sage_damped_solve_qr_f32(AtA, Atb, dof, damp, sol)) != 0)
          return rc;
      }
      else
        break;
    }
    if (trace && ntrace < trace_cap)
      trace[ntrace++] = SageLmTraceEntry{damp, curr_error, cand_error, accepted ? 1 : 0, update_jac ? 1 : 0};
    if (cand_error >= curr_error && damp >= cfg.max_damp)
      break;
    std::memcpy(guess, cand, sizeof(guess));
    guess_scale = cand_scale;
    if (update_jac)
      prev_error = curr_error;
    curr_error = cand_error;
    damp = clampd(damp / cfg.damp_dec_factor);
    if (curr_iter >= cfg.max_num_iters)
      break;
  }
  std::memcpy(pose12, guess, sizeof(guess));
  if (scale)
    *scale = guess_scale;
  if (final_error)
    *final_error = curr_error;
  if (iters)
    *iters = (int)curr_iter;
  if (trace_len)
    *trace_len = ntrace;
  return rc;
}

// ---------------------------------------------------------------- envelope Cholesky (window solve)
namespace sage
{

void EnvelopeMatrix::init(int n_, const std::vector<int> &first_)
{
  n = n_;
  first = first_;
  rowptr.resize(n + 1);
  size_t off = 0;
  for (int r = 0; r < n; ++r)
  {
    rowptr[r] = off;
    off += (size_t)(r - first[r] + 1);
  }
  rowptr[n] = off;
  data.assign(off, 0.0);
}

// dot product with reassociation allowed (SIMD + several accumulators); runtime-dispatched to the widest ISA
// of the host (the LM step is host-bound on this solve once the kernels are fast).
__attribute__((target_clones("avx512f", "avx2", "default"))) static double env_dot(const double *a, const double *b,
                                                                                   int len)
{
#pragma clang fp reassociate(on)
  double acc = 0.0;
#pragma clang loop vectorize(enable) interleave_count(4)
  for (int k = 0; k < len; ++k)
    acc += a[k] * b[k];
  return acc;
}

__attribute__((target_clones("avx512f", "avx2", "default"))) static void env_axpy(double *y, const double *x, double a,
                                                                                  int len)
{
#pragma clang loop vectorize(enable) interleave_count(4)
  for (int k = 0; k < len; ++k)
    y[k] -= a * x[k];
}

namespace
{
struct SpinBarrier
{
  std::atomic<int> count{0};
  std::atomic<int> gen{0};
  int n;
  explicit SpinBarrier(int n_) : n(n_) {}
  void wait()
  {
    if (n <= 1)
      return;
    const int g = gen.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1)
    {
      count.store(0, std::memory_order_relaxed);
      gen.fetch_add(1, std::memory_order_release);
    }
    else
      while (gen.load(std::memory_order_acquire) == g)
        __builtin_ia32_pause();
  }
};

typedef double v8d __attribute__((vector_size(64), aligned(8)));

// (macros, not functions: a v8d crossing a function boundary would need the AVX-512 ABI in every clone)
#define SAGE_LOADU(dst, p) __builtin_memcpy(&(dst), (p), sizeof(v8d))
#define SAGE_HSUM(v) ((((v)[0] + (v)[4]) + ((v)[2] + (v)[6])) + (((v)[1] + (v)[5]) + ((v)[3] + (v)[7])))

// C_i[j] -= dot(A_i[0:len], B_j[0:len]) for i < ni, j < nj with a 4x4 register-blocked micro-kernel, k vectorised
// (8 doubles: one zmm, two ymm or four xmm depending on the clone the resolver picks).
__attribute__((target_clones("avx512f", "avx2", "default"))) static void gemm_nt_sub(double *const *C, int coff,
                                                                                     const double *const *A,
                                                                                     const double *const *B, int ni,
                                                                                     int nj, int len_full,
                                                                                     bool b_lower_tri, int lower_only_row0)
{
  // b_lower_tri: row j of B is zero beyond column j (inverse of a Cholesky factor) -> the k range of tile column j0
  //              stops at j0+4 (rounded up to whole vectors);
  // lower_only_row0 >= 0: only outputs with j <= lower_only_row0 + i are needed (lower triangle of a diagonal block)
  for (int i0 = 0; i0 < ni; i0 += 4)
  {
    const int mi = ni - i0 < 4 ? ni - i0 : 4;
    for (int j0 = 0; j0 < nj; j0 += 4)
    {
      if (lower_only_row0 >= 0 && j0 > lower_only_row0 + i0 + 3)
        break;
      const int mj = nj - j0 < 4 ? nj - j0 : 4;
      int len = len_full;
      if (b_lower_tri)
      {
        const int need = ((j0 + 4 + 7) / 8) * 8;
        len = need < len_full ? need : len_full;
      }
      const double *a[4], *b[4];
      for (int t = 0; t < 4; ++t)
      {
        a[t] = A[i0 + (t < mi ? t : 0)];
        b[t] = B[j0 + (t < mj ? t : 0)];
      }
      v8d acc[4][4];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
          acc[i][j] = v8d{0, 0, 0, 0, 0, 0, 0, 0};
      int k = 0;
      for (; k + 8 <= len; k += 8)
      {
        v8d a0, a1, a2, a3, b0, b1, b2, b3;
        SAGE_LOADU(a0, a[0] + k); SAGE_LOADU(a1, a[1] + k); SAGE_LOADU(a2, a[2] + k); SAGE_LOADU(a3, a[3] + k);
        SAGE_LOADU(b0, b[0] + k); SAGE_LOADU(b1, b[1] + k); SAGE_LOADU(b2, b[2] + k); SAGE_LOADU(b3, b[3] + k);
        acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[0][2] += a0 * b2; acc[0][3] += a0 * b3;
        acc[1][0] += a1 * b0; acc[1][1] += a1 * b1; acc[1][2] += a1 * b2; acc[1][3] += a1 * b3;
        acc[2][0] += a2 * b0; acc[2][1] += a2 * b1; acc[2][2] += a2 * b2; acc[2][3] += a2 * b3;
        acc[3][0] += a3 * b0; acc[3][1] += a3 * b1; acc[3][2] += a3 * b2; acc[3][3] += a3 * b3;
      }
      double tail[4][4] = {{0}};
      for (; k < len; ++k)
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j)
            tail[i][j] += a[i][k] * b[j][k];
      for (int i = 0; i < mi; ++i)
        for (int j = 0; j < mj; ++j)
          C[i0 + i][coff + j0 + j] -= SAGE_HSUM(acc[i][j]) + tail[i][j];
    }
  }
}
} // namespace

// In-place Cholesky of an nb x nb diagonal block given by row pointers (row i holds columns 0..i) and the inverse
// W of its factor (dense nb x nb row-major, lower triangular).  One multiversioned function so the short inner loops
// are compiled for the widest host ISA without a per-call dispatch.  Returns false if not positive definite.
__attribute__((target_clones("avx512f", "avx2", "default"))) static bool diag_factor(double *const *L, int nb, double *W,
                                                                                 double *col)
{
  for (int j = 0; j < nb; ++j)
  {
    const double d = L[j][j];
    if (!(d > 0.0))
      return false;
    const double sj = std::sqrt(d), inv = 1.0 / sj;
    L[j][j] = sj;
    for (int i = j + 1; i < nb; ++i)
    {
      L[i][j] *= inv;
      col[i] = L[i][j];
    }
    for (int i = j + 1; i < nb; ++i) // right-looking update of the trailing rows, contiguous in k
    {
      const double lij = col[i];
      double *Li = L[i];
#pragma clang loop vectorize(enable)
      for (int k = j + 1; k <= i; ++k)
        Li[k] -= lij * col[k];
    }
  }
  for (int i = 0; i < nb; ++i) // W = inv(L): row i = (e_i - sum_{k<i} L[i][k] W[k][:]) / L[i][i]
  {
    double *Wi = W + (size_t)i * nb;
    for (int j = 0; j < nb; ++j)
      Wi[j] = 0.0;
    Wi[i] = 1.0;
    for (int k = 0; k < i; ++k)
    {
      const double lik = L[i][k];
      const double *Wk = W + (size_t)k * nb;
#pragma clang loop vectorize(enable)
      for (int j = 0; j <= k; ++j)
        Wi[j] -= lik * Wk[j];
    }
    const double inv = 1.0 / L[i][i];
#pragma clang loop vectorize(enable)
    for (int j = 0; j <= i; ++j)
      Wi[j] *= inv;
  }
  return true;
}

// Blocked left-looking Cholesky on the row-contiguous envelope.  With block > 1 the rows come in aligned groups
// of `block` rows sharing `first` (the window's keyframe blocks); then for block row I and block column J < I
//     S    = A_IJ - L_I[:, k0:cJ] * L_J[:, k0:cJ]^T          (GEMM, k contiguous in both operands)
//     L_IJ = S * inv(L_JJ)^T                                   (GEMM against the cached inverse of the diagonal factor)
// and the diagonal block is a dense block x block Cholesky of A_II - L_I[:, f:r0] L_I[:, f:r0]^T.
// The GEMMs are split over `threads` by rows of the block row.
bool EnvelopeMatrix::cholesky_inplace(int block, int threads)
{
  bool uniform = block > 1 && n % block == 0;
  if (uniform)
    for (int r = 0; r < n && uniform; ++r)
      uniform = first[r] == first[(r / block) * block] && first[r] % block == 0;
  if (!uniform)
  {
    // generic envelope: plain row-wise left-looking factorisation
    for (int r = 0; r < n; ++r)
    {
      double *Lr = &data[rowptr[r]];
      const int fr = first[r];
      for (int c = fr; c <= r; ++c)
      {
        const double *Lc = &data[rowptr[c]];
        const int fc = first[c];
        const int k0 = fr > fc ? fr : fc;
        const double s = Lr[c - fr] - env_dot(Lr + (k0 - fr), Lc + (k0 - fc), c - k0);
        if (c < r)
          Lr[c - fr] = s / Lc[c - fc];
        else
        {
          if (!(s > 0.0))
            return false;
          Lr[c - fr] = std::sqrt(s);
        }
      }
    }
    return true;
  }
  const int nb = block, NB = n / nb;
  threads = std::max(1, std::min(threads, nb / 4));
  if (nb < 16)
    threads = 1;
  std::vector<double> winv((size_t)NB * nb * nb, 0.0); // inverse of every diagonal factor block (lower triangular)
  std::atomic<bool> ok{true};
  SpinBarrier bar(threads);
  double t_gemm1 = 0, t_gemm2 = 0, t_diag = 0, t_scalar = 0;
  auto tk = [] { return std::chrono::steady_clock::now(); };
  auto dtm = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  auto worker = [&](int tid) {
    std::vector<double> tmp((size_t)nb * nb);
    std::vector<double *> crow(nb);
    std::vector<const double *> arow(nb), brow(nb), trow(nb);
    // this thread's slice of the block row
    const int per = (nb + threads - 1) / threads;
    const int i_lo = std::min(nb, tid * per), i_hi = std::min(nb, i_lo + per), ni = i_hi - i_lo;
    for (int I = 0; I < NB; ++I)
    {
      const int r0 = I * nb, f = first[r0];
      for (int J = f / nb; J < I; ++J)
      {
        const int c0 = J * nb, fJ = first[c0], k0 = f > fJ ? f : fJ, len = c0 - k0;
        if (ni > 0)
        {
          for (int i = 0; i < ni; ++i)
          {
            double *row = &data[rowptr[r0 + i_lo + i]];
            crow[i] = row;          // column c is at row[c - f]
            arow[i] = row + (k0 - f);
          }
          for (int j = 0; j < nb; ++j)
            brow[j] = &data[rowptr[c0 + j]] + (k0 - fJ);
          auto q0 = tk();
          gemm_nt_sub(crow.data(), c0 - f, arow.data(), brow.data(), ni, nb, len, false, -1);
          if (tid == 0) t_gemm1 += dtm(q0, tk());
          auto q1 = tk();
          // L_IJ = S * Winv_J^T : copy S, clear the destination, accumulate with the (negated) GEMM
          const double *W = &winv[(size_t)J * nb * nb];
          for (int i = 0; i < ni; ++i)
          {
            double *dst = crow[i] + (c0 - f);
            for (int j = 0; j < nb; ++j)
            {
              tmp[(size_t)i * nb + j] = -dst[j];
              dst[j] = 0.0;
            }
            trow[i] = &tmp[(size_t)i * nb];
          }
          for (int j = 0; j < nb; ++j)
            brow[j] = W + (size_t)j * nb;
          gemm_nt_sub(crow.data(), c0 - f, trow.data(), brow.data(), ni, nb, nb, true, -1);
          if (tid == 0) t_gemm2 += dtm(q1, tk());
        }
        // no barrier needed between block columns: thread t only touches its own rows of block row I,
        // and block rows < I are final
      }
      // diagonal block: subtract the left part (own rows x all rows of the block -> needs everyone's left parts)
      bar.wait();
      auto q2 = tk();
      if (ni > 0)
      {
        for (int i = 0; i < ni; ++i)
        {
          double *row = &data[rowptr[r0 + i_lo + i]];
          crow[i] = row;
          arow[i] = row;
        }
        for (int j = 0; j < nb; ++j)
          brow[j] = &data[rowptr[r0 + j]];
        // only columns j <= i are stored: accumulate the full ni x nb slice into tmp, then fold the lower part back
        for (int i = 0; i < ni; ++i)
        {
          for (int j = 0; j < nb; ++j)
            tmp[(size_t)i * nb + j] = 0.0;
          crow[i] = &tmp[(size_t)i * nb];
        }
        gemm_nt_sub(crow.data(), 0, arow.data(), brow.data(), ni, std::min(nb, i_hi), r0 - f, false, i_lo);
        for (int i = 0; i < ni; ++i)
        {
          double *dst = &data[rowptr[r0 + i_lo + i]] + (r0 - f);
          for (int j = 0; j <= i_lo + i; ++j)
            dst[j] += tmp[(size_t)i * nb + j];
        }
      }
      bar.wait();
      if (tid == 0) t_diag += dtm(q2, tk());
      auto q3 = tk();
      if (tid == 0)
      {
        // dense Cholesky of the nb x nb diagonal block (lower part, in place) + the inverse of its factor
        std::vector<double *> drow(nb);
        std::vector<double> colbuf(nb);
        for (int i = 0; i < nb; ++i)
          drow[i] = &data[rowptr[r0 + i]] + (r0 - f);
        if (!diag_factor(drow.data(), nb, &winv[(size_t)I * nb * nb], colbuf.data()))
          ok.store(false);
      }
      bar.wait();
      if (tid == 0) t_scalar += dtm(q3, tk());
      if (!ok.load())
        return;
    }
  };
  if (threads == 1)
    worker(0);
  else
  {
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t)
      pool.emplace_back(worker, t);
    worker(0);
    for (auto &th : pool)
      th.join();
  }
  if (sage::env_flag("SAGE_DEBUG_TIMING"))
    fprintf(stderr, "[sage cholesky] gemm(S) %.3f gemm(trsm) %.3f diag-gemm %.3f diag-chol+inv %.3f ms\n", t_gemm1,
            t_gemm2, t_diag, t_scalar);
  return ok.load();
}

void EnvelopeMatrix::solve_inplace(std::vector<double> &b) const
{
  // L y = b
  for (int r = 0; r < n; ++r)
  {
    const double *Lr = &data[rowptr[r]];
    const int fr = first[r];
    b[r] = (b[r] - env_dot(Lr, &b[fr], r - fr)) / Lr[r - fr];
  }
  // L^T x = y
  for (int r = n - 1; r >= 0; --r)
  {
    const double *Lr = &data[rowptr[r]];
    const int fr = first[r];
    const double x = b[r] / Lr[r - fr];
    b[r] = x;
    env_axpy(&b[fr], Lr, x, r - fr);
  }
}

} // namespace sage

// ------------------------------------------------------------------------------------------------
// Fixed-block-size Cholesky on transposed block storage (the window solve's host leg).
//
// Every finished block is kept as T_ij = L_ij^T (row t of T = column t of L), the diagonal factors as U = L^T and
// X = U^-1.  With that layout every contraction is "broadcast one scalar, multiply a contiguous row":
//   C_ij^T[c][:]  -= sum_k sum_t T_jk[t][c] * T_ik[t][:]         (trailing update)
//   T_ij[c][:]     = sum_{t<=c} X_j[t][c] * C_ij^T[t][:]          (L_ij = C_ij L_jj^-T)
// so the micro-kernel is 4 output rows x NV vectors of accumulators, NV loads + 4 broadcasts + 4*NV FMAs per step,
// no horizontal sums (the envelope code above pays one per output).
// ------------------------------------------------------------------------------------------------
namespace sage
{
static void host_threads_atexit_once(); // (defined with host_threads_shutdown below)
namespace
{
#define SAGE_STOREU(p, v) __builtin_memcpy((p), &(v), sizeof(v8d))

// The blocks of a row arrive by DMA straight into DRAM (no cache allocation on this platform): a row's first touch of
// its ~4 fresh blocks (51 KB) would stall the core for ~2 us.  The contraction loops of row i therefore prefetch row
// i+1's blocks, two cache lines per inner step.
struct RowPrefetch // (plain aggregate: the multi-versioned callers must not need an out-of-line constructor)
{
  const char *p, *end;
  // two more ranges, taken up when the first is through (r05, arrow-row chains: besides the chain's own next block the
  // half's blocks and inverse of the NEXT column -- written by another core, 64 KB a chain used to wait for column by column)
  const char *p2 = nullptr, *end2 = nullptr, *p3 = nullptr, *end3 = nullptr;
  inline __attribute__((always_inline)) void step()
  {
    if (p < end)
    {
      __builtin_prefetch(p, 0, 3);
      __builtin_prefetch(p + 64, 0, 3);
      p += 128;
    }
    else if (p2 < end2)
    {
      __builtin_prefetch(p2, 0, 3);
      __builtin_prefetch(p2 + 64, 0, 3);
      p2 += 128;
    }
    else if (p3 < end3)
    {
      __builtin_prefetch(p3, 0, 3);
      __builtin_prefetch(p3 + 64, 0, 3);
      p3 += 128;
    }
  }
};

// CT[c][8*V0 ..] -= sum_t Tj[t][c] * Ti[t][8*V0 ..]  for the 4 rows c0..c0+3 ; vectors V0..NV-1 only
template <int NV, int V0>
static inline __attribute__((always_inline)) void tn_sub_rows4(double *CT, const double *Tj, const double *Ti, int c0,
                                                               RowPrefetch &pf)
{
  constexpr int BP = NV * 8;
  v8d acc[4][NV];
  for (int u = 0; u < 4; ++u)
    for (int v = V0; v < NV; ++v)
      acc[u][v] = v8d{0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = 0; t < BP; ++t)
  {
    const double *ti = Ti + t * BP, *tj = Tj + t * BP + c0;
    pf.step();
    v8d b[NV];
    for (int v = V0; v < NV; ++v)
      SAGE_LOADU(b[v], ti + 8 * v);
    for (int u = 0; u < 4; ++u)
    {
      const double sc = tj[u];
      const v8d s = {sc, sc, sc, sc, sc, sc, sc, sc};
      for (int v = V0; v < NV; ++v)
        acc[u][v] += s * b[v];
    }
  }
  for (int u = 0; u < 4; ++u)
    for (int v = V0; v < NV; ++v)
    {
      v8d c;
      SAGE_LOADU(c, CT + (c0 + u) * BP + 8 * v);
      c -= acc[u][v];
      SAGE_STOREU(CT + (c0 + u) * BP + 8 * v, c);
    }
}

// CT -= Tj^T-contraction with Ti over the whole block; upper_only: only entries [c][r >= c] are needed (diagonal
// block, symmetric) -> the vectors left of the diagonal are skipped
template <int NV>
static inline __attribute__((always_inline)) void tn_sub(double *CT, const double *Tj, const double *Ti, bool upper_only,
                                                         RowPrefetch &pf)
{
  constexpr int BP = NV * 8;
  for (int c0 = 0; c0 < BP; c0 += 4)
  {
    const int v0 = upper_only ? c0 / 8 : 0;
    switch (v0)
    {
    case 0: tn_sub_rows4<NV, 0>(CT, Tj, Ti, c0, pf); break;
    case 1: tn_sub_rows4<NV, (NV > 1 ? 1 : 0)>(CT, Tj, Ti, c0, pf); break;
    case 2: tn_sub_rows4<NV, (NV > 2 ? 2 : 0)>(CT, Tj, Ti, c0, pf); break;
    case 3: tn_sub_rows4<NV, (NV > 3 ? 3 : 0)>(CT, Tj, Ti, c0, pf); break;
    default: tn_sub_rows4<NV, (NV > 4 ? 4 : 0)>(CT, Tj, Ti, c0, pf); break;
    }
  }
}

// in place: CT[c][:] <- sum_{t<=c} X[t][c] * CT[t][:]   (rows in descending groups of 4: a group reads rows <= its own)
template <int NV>
static inline __attribute__((always_inline)) void apply_inverse(double *CT, const double *X)
{
  constexpr int BP = NV * 8;
  for (int c0 = BP - 4; c0 >= 0; c0 -= 4)
  {
    v8d acc[4][NV];
    for (int u = 0; u < 4; ++u)
      for (int v = 0; v < NV; ++v)
        acc[u][v] = v8d{0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t <= c0 + 3; ++t)
    {
      const double *ct = CT + t * BP, *xs = X + t * BP + c0; // X[t][c] = 0 for c < t
      v8d b[NV];
      for (int v = 0; v < NV; ++v)
        SAGE_LOADU(b[v], ct + 8 * v);
      for (int u = 0; u < 4; ++u)
      {
        const double sc = xs[u];
        const v8d s = {sc, sc, sc, sc, sc, sc, sc, sc};
        for (int v = 0; v < NV; ++v)
          acc[u][v] += s * b[v];
      }
    }
    for (int u = 0; u < 4; ++u)
      for (int v = 0; v < NV; ++v)
        SAGE_STOREU(CT + (c0 + u) * BP + 8 * v, acc[u][v]);
  }
}

// One 8-row panel of the Cholesky factorisation below, everything with compile-time indices so that the panel's
// diagonal block lives in eight registers: the pivot chain (sqrt, divide, seven multiplier broadcasts) never goes
// through memory -- a scalar reload of a just-stored vector element does not forward and cost more than the sqrt.
template <int NV, int P>
static inline __attribute__((always_inline)) bool factor_panel(double *S, double *rinv)
{
  constexpr int BP = NV * 8;
  v8d D[8];
  for (int cc = 0; cc < 8; ++cc)
    SAGE_LOADU(D[cc], S + (8 * P + cc) * BP + 8 * P);
  double rsv[8];
#pragma unroll
  for (int cc = 0; cc < 8; ++cc)
  {
    const double d = D[cc][cc];
    if (!(d > 0.0))
      return false;
    const double rs = 1.0 / std::sqrt(d);
    rsv[cc] = rs;
    rinv[8 * P + cc] = rs; // 1 / U[c][c]
    const v8d rv = {rs, rs, rs, rs, rs, rs, rs, rs};
    D[cc] *= rv;
#pragma unroll
    for (int c2 = cc + 1; c2 < 8; ++c2)
    {
      const double f = D[cc][c2];
      const v8d fv = {f, f, f, f, f, f, f, f};
      D[c2] -= fv * D[cc];
    }
  }
  // U is upper triangular: clear what the updates left below the diagonal of the block, store the rows
#pragma unroll
  for (int cc = 0; cc < 8; ++cc)
  {
#pragma unroll
    for (int r = 0; r < cc; ++r)
      D[cc][r] = 0.0;
    SAGE_STOREU(S + (8 * P + cc) * BP + 8 * P, D[cc]);
  }
  // the same eliminations on the panel's other columns (independent of the pivot chain)
#pragma unroll
  for (int v = P + 1; v < NV; ++v)
  {
    v8d R[8];
    for (int cc = 0; cc < 8; ++cc)
      SAGE_LOADU(R[cc], S + (8 * P + cc) * BP + 8 * v);
#pragma unroll
    for (int cc = 0; cc < 8; ++cc)
    {
      const v8d rv = {rsv[cc], rsv[cc], rsv[cc], rsv[cc], rsv[cc], rsv[cc], rsv[cc], rsv[cc]};
      R[cc] *= rv;
#pragma unroll
      for (int c2 = cc + 1; c2 < 8; ++c2)
      {
        const double f = D[cc][c2];
        const v8d fv = {f, f, f, f, f, f, f, f};
        R[c2] -= fv * R[cc];
      }
    }
    for (int cc = 0; cc < 8; ++cc)
      SAGE_STOREU(S + (8 * P + cc) * BP + 8 * v, R[cc]);
  }
  // rank-8 update of the trailing rows
  for (int c2 = 8 * P + 8; c2 < BP; ++c2)
  {
    double *row2 = S + c2 * BP;
    v8d f[8];
    for (int t = 0; t < 8; ++t)
    {
      const double ft = S[(8 * P + t) * BP + c2];
      f[t] = v8d{ft, ft, ft, ft, ft, ft, ft, ft};
    }
    for (int v = c2 / 8; v < NV; ++v)
    {
      v8d acc;
      SAGE_LOADU(acc, row2 + 8 * v);
      for (int t = 0; t < 8; ++t)
      {
        v8d x;
        SAGE_LOADU(x, S + (8 * P + t) * BP + 8 * v);
        acc -= f[t] * x;
      }
      SAGE_STOREU(row2 + 8 * v, acc);
    }
  }
  if constexpr (P + 1 < NV)
    return factor_panel<NV, P + 1>(S, rinv);
  else
    return true;
}

// X = U^-1 by back substitution on rows: X[c][:] = (e_c - sum_{t>c} U[c][t] X[t][:]) / U[c][c].  X[t][:] is zero left
// of column t, so a block of eight t only touches the vectors from its own on -- with the block index a template
// parameter every vector loop has compile-time bounds (a run-time start index would move the accumulators from
// registers to the stack).  Two accumulator sets (even / odd t) keep enough independent FMA chains in flight.
template <int NV, int TB>
static inline __attribute__((always_inline)) void inverse_accumulate(const double *u, const double *X, v8d *a0, v8d *a1)
{
  constexpr int BP = NV * 8;
#pragma unroll
  for (int tt = 0; tt < 8; tt += 2)
  {
    const int t = 8 * TB + tt;
    const double f0 = u[t], f1 = u[t + 1];
    const v8d fv0 = {f0, f0, f0, f0, f0, f0, f0, f0}, fv1 = {f1, f1, f1, f1, f1, f1, f1, f1};
#pragma unroll
    for (int v = TB; v < NV; ++v)
    {
      v8d xa, xb;
      SAGE_LOADU(xa, X + t * BP + 8 * v);
      SAGE_LOADU(xb, X + (t + 1) * BP + 8 * v);
      a0[v] -= fv0 * xa;
      a1[v] -= fv1 * xb;
    }
  }
  if constexpr (TB + 1 < NV)
    inverse_accumulate<NV, TB + 1>(u, X, a0, a1);
}

template <int NV, int CB>
static inline __attribute__((always_inline)) void inverse_rows(const double *S, double *X, const double *rinv)
{
  constexpr int BP = NV * 8;
  for (int cc = 7; cc >= 0; --cc)
  {
    const int c = 8 * CB + cc;
    double e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e[cc] = 1.0;
    v8d a0[NV], a1[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
      a0[v] = a1[v] = v8d{0, 0, 0, 0, 0, 0, 0, 0};
    SAGE_LOADU(a0[CB], e);
    const double *u = S + c * BP;
    for (int t = c + 1; t < 8 * CB + 8; ++t) // the rest of the row's own block
    {
      const double f = u[t];
      const v8d fv = {f, f, f, f, f, f, f, f};
#pragma unroll
      for (int v = CB; v < NV; ++v)
      {
        v8d x;
        SAGE_LOADU(x, X + t * BP + 8 * v);
        a0[v] -= fv * x;
      }
    }
    if constexpr (CB + 1 < NV)
      inverse_accumulate<NV, CB + 1>(u, X, a0, a1);
    const double inv = rinv[c];
    const v8d iv = {inv, inv, inv, inv, inv, inv, inv, inv};
#pragma unroll
    for (int v = 0; v < NV; ++v)
    {
      v8d r = (a0[v] + a1[v]) * iv; // (zero for v < CB)
      SAGE_STOREU(X + c * BP + 8 * v, r);
    }
  }
  if constexpr (CB > 0)
    inverse_rows<NV, CB - 1>(S, X, rinv);
}

// S (symmetric, entries [c][r >= c] valid) -> U = L^T in place (A = U^T U), X = U^-1 (upper triangular, zeros below).
// Blocked by panels of 8 rows (factor_panel), one rank-8 register-accumulated update per trailing row.
template <int NV>
static inline __attribute__((always_inline)) bool factor_diag(double *S, double *X)
{
  constexpr int BP = NV * 8;
  double rinv[BP];
  if (!factor_panel<NV, 0>(S, rinv))
    return false;
  inverse_rows<NV, NV - 1>(S, X, rinv);
  return true;
}

// The device streams the blocks in row order (solve_kernels.hip): wait until every block of row i carries this solve's
// ticket.  (Not inlined: keeps <chrono> out of the target_clones bodies.)
static double mono_seconds()
{
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
// from_col: only the blocks (i, j >= from_col) -- the others have been consumed already (the arrow-row tasks take their
// blocks one by one).  Structural fill blocks (E.fill) are zeroed here instead of waited for: this is their first touch.
static __attribute__((noinline)) bool wait_row_tickets(const BlockEnvelope &E, int i, double *T, int from_col = 0,
                                                       int skip_lo = 0, int skip_hi = 0)
{
  const int na = E.a_cnt ? E.a_cnt[i] : 0;
  const int b0[2] = {na ? E.a_off[i] : 0, E.row_off[i]}, nb[2] = {na, i - E.row_first[i] + 1};
  const int c0[2] = {na ? E.a_first[i] : 0, E.row_first[i]};
  const size_t BB = (size_t)E.Bp * E.Bp;
  for (int rg = 0; rg < 2; ++rg)
    for (int q = 0; q < nb[rg]; ++q)
    {
      if (c0[rg] + q < from_col || (c0[rg] + q >= skip_lo && c0[rg] + q < skip_hi))
        continue; // (skip range: blocks the separator pre-pass has taken care of, tickets and fill included)
      if (E.fill && E.fill[b0[rg] + q])
      {
        std::memset(T + (size_t)(b0[rg] + q) * BB, 0, BB * sizeof(double));
        continue;
      }
      const volatile unsigned *f = E.ready + b0[rg] + q;
      if (*f == E.epoch)
        continue;
      const double t0 = mono_seconds();
      unsigned spins = 0;
      while (*f != E.epoch)
      {
        __builtin_ia32_pause();
        if (E.idle && (spins & 0x3f) == 0)
          E.idle(E.user);
        if ((++spins & 0xfff) == 0 && mono_seconds() - t0 > 2.0)
          return false;
      }
      if (E.t_ticket_wait)
        *E.t_ticket_wait += mono_seconds() - t0;
    }
  std::atomic_thread_fence(std::memory_order_acquire);
  return true;
}

// One pass over a range of block rows.  phase 0: factorise rows [lo,hi) ascending and forward-substitute y;
// phase 1: back-substitute rows [lo,hi) descending.  Rows only touch the blocks of their own column ranges, so two
// row ranges that do not reference each other can run on two cores.
// role (phase 0 only): 0 one thread does everything; 1 / 2 the two stages of E.pipe (1: the chain through row i-1 and
// the diagonal, 2: the look-ahead).  Rows with an A range are not split (no such rows in the halves of a plan).
template <int NV>
static inline __attribute__((always_inline)) int block_chol_pass(const BlockEnvelope &E, double *T, double *X, double *y,
                                                                 int phase, int lo, int hi, int role = 0)
{
  constexpr int BP = NV * 8, BB = BP * BP;
  const int K = E.K;
  const int32_t *row_first = E.row_first, *row_off = E.row_off;
  auto afirst = [&](int i) { return E.a_cnt ? E.a_first[i] : 0; };
  auto acnt = [&](int i) { return E.a_cnt ? E.a_cnt[i] : 0; };
  auto has = [&](int i, int j) { return (j >= row_first[i] && j <= i) || (j >= afirst(i) && j < afirst(i) + acnt(i)); };
  auto blk = [&](int i, int j) {
    return T + (size_t)(j < row_first[i] ? E.a_off[i] + j - E.a_first[i] : row_off[i] + j - row_first[i]) * BB;
  };
  if (phase == 0)
  {
    static const bool prof = sage::env_flag("SAGE_CHOL_PROFILE");
    unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, tl = __builtin_readcyclecounter();
#define LAP(k) do { if (prof) { const unsigned long long t_ = __builtin_readcyclecounter(); tp[k] += t_ - tl; tl = t_; } } while (0)
    BlockEnvelope::RowPipe *pipe = role ? E.pipe + (lo >= E.n1 && E.n1 > 0 ? 1 : 0) : nullptr;
    auto pipe_wait = [&](std::atomic<int> &c, int need) -> bool { // false: the other stage gave up (negative count)
      int v;
      while ((v = c.load(std::memory_order_acquire)) < need)
      {
        if (v < 0)
          return false;
        __builtin_ia32_pause();
      }
      return true;
    };
    if (role == 2 && E.sep_pre)
      pipe->pre.store(1, std::memory_order_release); // (before this stage's first `early`: the chain thread cannot finish its half unseen)
    for (int i = lo; i < hi; ++i)
    {
      LAP(5);
      if (E.before_row && E.before_row(E.user, i))
        return -2;
      // separator row of a plain split window: its blocks against a half's columns may have been formed already by that
      // half's look-ahead thread (sep_pre) -- tickets, fill and arithmetic; they are skipped below
      int pre_lo[2] = {0, 0}, pre_hi[2] = {0, 0};
      if (role == 0 && E.sep_pre && E.pipe && i >= E.n1 + E.n2)
      {
        for (int hf = 0; hf < 2; ++hf)
        {
          int v;
          while ((v = E.pipe[hf].pre.load(std::memory_order_acquire)) == 1)
            __builtin_ia32_pause();
          if (v < 0)
            return -2;
          if (v == 2)
          {
            pre_lo[hf] = hf == 0 ? afirst(i) : row_first[i];
            pre_hi[hf] = hf == 0 ? afirst(i) + acnt(i) : std::min(E.n1 + E.n2, i);
          }
        }
      }
      if (role != 1 && E.ready && !wait_row_tickets(E, i, T, pre_hi[0] > pre_lo[0] ? pre_hi[0] : 0, pre_lo[1], pre_hi[1]))
      {
        if (pipe)
          pipe->early.store(-1, std::memory_order_release);
        if (role == 2 && E.sep_pre)
          pipe->pre.store(-1, std::memory_order_release);
        return -2;
      }
      LAP(0);
      // the column ranges of row i, in ascending order
      const int r0[2] = {afirst(i), row_first[i]}, r1[2] = {afirst(i) + acnt(i), i};
      RowPrefetch pf{nullptr, nullptr};
      if (i + 1 < hi && role != 1)
      {
        // row i+1's blocks are contiguous in the storage: [A range | B range]
        const size_t b0 = (size_t)(acnt(i + 1) ? E.a_off[i + 1] : row_off[i + 1]);
        const size_t nb = (size_t)acnt(i + 1) + (size_t)(i + 1 - row_first[i + 1] + 1);
        pf.p = reinterpret_cast<const char *>(T + b0 * BB);
        pf.end = pf.p + nb * BB * sizeof(double);
      }
      if (role == 2)
      {
        // look-ahead: needs the rows <= i-2 complete (and its own earlier rows)
        if (!pipe_wait(pipe->late, i - 1 - lo))
        {
          if (E.sep_pre)
            pipe->pre.store(-1, std::memory_order_release);
          return -2;
        }
        for (int j = row_first[i]; j < i; ++j)
        {
          double *CT = blk(i, j);
          const int kend = j == i - 1 ? i - 2 : j; // (i, i-1): the product with row i-1's last block is the other stage's
          for (int k = row_first[i]; k < kend; ++k)
            if (has(j, k))
              tn_sub<NV>(CT, blk(j, k), blk(i, k), false, pf);
          if (j < i - 1)
            apply_inverse<NV>(CT, X + (size_t)j * BB);
        }
        double *S = blk(i, i);
        for (int k = row_first[i]; k < i - 1; ++k)
          tn_sub<NV>(S, blk(i, k), blk(i, k), true, pf);
        while (pf.p < pf.end) // (short rows: the prefetch of the next row's blocks is part of this stage's job)
          pf.step();
        pipe->early.store(i + 1 - lo, std::memory_order_release);
        continue;
      }
      if (role == 1)
      {
        if (!pipe_wait(pipe->early, i + 1 - lo))
          return -2;
        if (i - 1 >= row_first[i])
        {
          double *CT = blk(i, i - 1);
          if (i - 2 >= row_first[i] && has(i - 1, i - 2))
            tn_sub<NV>(CT, blk(i - 1, i - 2), blk(i, i - 2), false, pf);
          apply_inverse<NV>(CT, X + (size_t)(i - 1) * BB);
          tn_sub<NV>(blk(i, i), CT, CT, true, pf);
        }
      }
      else
      {
      for (int rg = 0; rg < 2; ++rg)
        for (int j = r0[rg]; j < r1[rg]; ++j)
        {
          if ((j >= pre_lo[0] && j < pre_hi[0]) || (j >= pre_lo[1] && j < pre_hi[1]))
            continue; // (formed by the half's look-ahead thread: same operations, same order)
          double *CT = blk(i, j);
          for (int rk = 0; rk <= rg; ++rk)
            for (int k = r0[rk]; k < std::min(r1[rk], j); ++k)
              if (has(j, k))
                tn_sub<NV>(CT, blk(j, k), blk(i, k), false, pf);
          LAP(1);
          apply_inverse<NV>(CT, X + (size_t)j * BB);
          LAP(2);
        }
      double *S0 = blk(i, i);
      for (int rg = 0; rg < 2; ++rg)
        for (int k = r0[rg]; k < r1[rg]; ++k)
          tn_sub<NV>(S0, blk(i, k), blk(i, k), true, pf);
      } // role 0
      double *S = blk(i, i);
      LAP(1);
      if (!factor_diag<NV>(S, X + (size_t)i * BB))
      {
        if (pipe)
          pipe->late.store(-1, std::memory_order_release);
        return 1 + i;
      }
      LAP(3);
      // forward substitution: y_i = L_ii^-1 (g_i - sum_k L_ik y_k)
      double w[BP];
      for (int r = 0; r < BP; ++r)
        w[r] = y[(size_t)i * BP + r];
      for (int rg = 0; rg < 2; ++rg)
        for (int k = r0[rg]; k < r1[rg]; ++k)
        {
          const double *Tk = blk(i, k), *yk = y + (size_t)k * BP;
          for (int t = 0; t < BP; ++t)
          {
            const double f = yk[t];
            for (int r = 0; r < BP; ++r)
              w[r] -= f * Tk[t * BP + r];
          }
        }
      double yi[BP];
      for (int c = 0; c < BP; ++c)
        yi[c] = 0.0;
      const double *Xi = X + (size_t)i * BB;
      for (int t = 0; t < BP; ++t)
      {
        const double f = w[t];
        for (int c = 0; c < BP; ++c) // X[t][c] = 0 for c < t
          yi[c] += f * Xi[t * BP + c];
      }
      for (int c = 0; c < BP; ++c)
        y[(size_t)i * BP + c] = yi[c];
      LAP(4);
      if (pipe)
        pipe->late.store(i + 1 - lo, std::memory_order_release);
      if (E.progress && i < E.n1 + E.n2)
        E.progress[i >= E.n1 ? 1 : 0].store(i + 1, std::memory_order_release);
    }
    if (role == 2 && E.sep_pre)
    {
      // ---- separator pre-pass (r05): the blocks L_sj of the separator rows s against THIS half's columns j only need rows of
      // this half -- L_sj = (A_sj - sum_{k < j, same range} L_jk L_sk) L_jj^-T -- and this thread has nothing left to do: it
      // forms them right behind the chain thread's row j instead of the caller forming them after both halves have joined.
      // Per block the same operations in the same order as in the separator pass (bit-identical factor).
      const int sep0 = E.n1 + E.n2, half = lo >= E.n1 && E.n1 > 0 ? 1 : 0;
      auto rng = [&](int srow, int &a, int &b) {
        a = half == 0 ? afirst(srow) : (int)row_first[srow];
        b = half == 0 ? afirst(srow) + acnt(srow) : std::min(sep0, srow);
      };
      int jlo = hi;
      for (int srow = sep0; srow < K; ++srow)
      {
        int a, b;
        rng(srow, a, b);
        if (a < b)
          jlo = std::min(jlo, a);
      }
      RowPrefetch pf0{nullptr, nullptr};
      bool ok = true;
      // the separator rows' blocks were written by the device's DMA and sit in no cache: ask for them now, while the chain
      // thread still works on the rows this pass waits for
      for (int srow = sep0; srow < K; ++srow)
      {
        int a, b;
        rng(srow, a, b);
        for (int j = std::max(a, lo); j < std::min(b, hi); ++j)
        {
          const char *pb = reinterpret_cast<const char *>(
              T + (size_t)(j < row_first[srow] ? E.a_off[srow] + j - E.a_first[srow] : row_off[srow] + j - row_first[srow]) * BB);
          for (size_t o = 0; o < (size_t)BB * sizeof(double); o += 64)
            __builtin_prefetch(pb + o, 1, 3);
        }
      }
      for (int j = std::max(jlo, lo); j < hi && ok; ++j)
      {
        if (!pipe_wait(pipe->late, j + 1 - lo)) // row j complete: its blocks and the inverse of its diagonal factor
        {
          ok = false;
          break;
        }
        for (int srow = sep0; srow < K && ok; ++srow)
        {
          int a, b;
          rng(srow, a, b);
          if (j < a || j >= b)
            continue;
          const size_t idx = (size_t)(j < row_first[srow] ? E.a_off[srow] + j - E.a_first[srow] : row_off[srow] + j - row_first[srow]);
          double *CT = T + idx * BB;
          if (E.fill && E.fill[idx])
            std::memset(CT, 0, (size_t)BB * sizeof(double)); // structural fill: first touch
          else if (E.ready)
          {
            const volatile unsigned *f = E.ready + idx;
            const double t0 = mono_seconds();
            unsigned spins = 0;
            while (*f != E.epoch)
            {
              __builtin_ia32_pause();
              if ((++spins & 0xfff) == 0 && mono_seconds() - t0 > 2.0)
              {
                ok = false;
                break;
              }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            if (!ok)
              break;
          }
          for (int k = a; k < j; ++k)
            if (has(j, k))
              tn_sub<NV>(CT, blk(j, k), blk(srow, k), false, pf0);
          apply_inverse<NV>(CT, X + (size_t)j * BB);
        }
      }
      pipe->pre.store(ok ? 2 : -1, std::memory_order_release);
      if (!ok)
        return -2;
    }
    if (prof)
      fprintf(stderr, "[chol profile] rows %d..%d kcycles: wait %.0f gemm+syrk %.0f apply-inverse %.0f factor+inverse %.0f fwd-subst %.0f other %.0f\n",
              lo, hi, tp[0] * 1e-3, tp[1] * 1e-3, tp[2] * 1e-3, tp[3] * 1e-3, tp[4] * 1e-3, tp[5] * 1e-3);
    return 0;
  }
  // back substitution: x_i = L_ii^-T (y_i - sum_{m>i, (m,i) stored} L_mi^T x_m)
  for (int i = hi - 1; i >= lo; --i)
  {
    double z[BP];
    for (int t = 0; t < BP; ++t)
      z[t] = y[(size_t)i * BP + t];
    // (explicit vectors: strict fp semantics keep the compiler from vectorising a dot product on its own, and the
    // scalar loops made this sweep a tenth of the solve)
    // z[t] -= sum over the rows m below that store a block (m, i) of  sum_r Tm[t][r] x_m[r]: 8 values of t at a time keep
    // one accumulator vector each across ALL those blocks (one horizontal sum per t and row instead of one per block --
    // the arrow rows of a loop-closure plan put half a dozen blocks into every column)
    const int nm = E.col_ptr ? E.col_ptr[i + 1] - E.col_ptr[i] : K - i - 1;
    for (int t0 = 0; t0 < BP; t0 += 8)
    {
      v8d acc[8];
      for (int u = 0; u < 8; ++u)
        acc[u] = v8d{0, 0, 0, 0, 0, 0, 0, 0};
      for (int q = 0; q < nm; ++q)
      {
        const int m = E.col_ptr ? E.col_rows[E.col_ptr[i] + q] : i + 1 + q;
        if (m >= E.bs_skip_from)
          break; // (ascending lists: the separator rows come last)
        if (!E.col_ptr && !has(m, i))
          continue;
        const double *Tm = blk(m, i) + t0 * BP, *xm = y + (size_t)m * BP;
        for (int v = 0; v < NV; ++v)
        {
          v8d xv;
          SAGE_LOADU(xv, xm + 8 * v);
          for (int u = 0; u < 8; ++u)
          {
            v8d a;
            SAGE_LOADU(a, Tm + u * BP + 8 * v);
            acc[u] += a * xv;
          }
        }
      }
      for (int u = 0; u < 8; ++u)
        z[t0 + u] -= ((acc[u][0] + acc[u][4]) + (acc[u][1] + acc[u][5])) + ((acc[u][2] + acc[u][6]) + (acc[u][3] + acc[u][7]));
    }
    const double *Xi = X + (size_t)i * BB;
    v8d zv[NV];
    for (int v = 0; v < NV; ++v)
      SAGE_LOADU(zv[v], z + 8 * v);
    for (int c = 0; c < BP; ++c)
    {
      v8d acc = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int v = c / 8; v < NV; ++v) // X[c][t] = 0 for t < c (stored zeros inside the first vector)
      {
        v8d a;
        SAGE_LOADU(a, Xi + c * BP + 8 * v);
        acc += a * zv[v];
      }
      y[(size_t)i * BP + c] = ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
    }
  }
  return 0;
}

__attribute__((target_clones("avx512f", "avx2", "default"))) static int block_chol_40(const BlockEnvelope &E, double *T,
                                                                                      double *X, double *y, int phase,
                                                                                      int lo, int hi, int role = 0)
{
  return block_chol_pass<5>(E, T, X, y, phase, lo, hi, role);
}
__attribute__((target_clones("avx512f", "avx2", "default"))) static int block_chol_24(const BlockEnvelope &E, double *T,
                                                                                      double *X, double *y, int phase,
                                                                                      int lo, int hi, int role = 0)
{
  return block_chol_pass<3>(E, T, X, y, phase, lo, hi, role);
}
// Separator rows of a partial factorisation (domain decomposition, shard_solve.cpp): the rows [nI, K) get L_ij for
// their columns j < nI; their blocks (i, j), nI <= j <= i, end as the Schur complement C_ij = A_ij - sum_{k<nI} L_ik L_jk^T
// (not factorised) and y_i as c_i = b_i - sum_{k<nI} L_ik y_k.  No A ranges in this storage.
template <int NV>
static inline __attribute__((always_inline)) void block_schur_rows(const BlockEnvelope &E, double *T, double *X, double *y,
                                                                   int nI)
{
  constexpr int BP = NV * 8, BB = BP * BP;
  const int K = E.K;
  const int32_t *row_first = E.row_first, *row_off = E.row_off;
  auto has = [&](int i, int j) { return j >= row_first[i] && j <= i; };
  auto blk = [&](int i, int j) { return T + (size_t)(row_off[i] + j - row_first[i]) * BB; };
  RowPrefetch pf{nullptr, nullptr};
  for (int i = nI; i < K; ++i)
  {
    const int f = row_first[i];
    for (int j = f; j < i; ++j)
    {
      double *CT = blk(i, j);
      for (int k = f; k < std::min(j, nI); ++k)
        if (has(j, k))
          tn_sub<NV>(CT, blk(j, k), blk(i, k), false, pf);
      if (j < nI)
        apply_inverse<NV>(CT, X + (size_t)j * BB);
    }
    double *S = blk(i, i);
    for (int k = f; k < std::min(i, nI); ++k)
      tn_sub<NV>(S, blk(i, k), blk(i, k), true, pf);
    double w[BP];
    for (int r = 0; r < BP; ++r)
      w[r] = y[(size_t)i * BP + r];
    for (int k = f; k < std::min(i, nI); ++k)
    {
      const double *Tk = blk(i, k), *yk = y + (size_t)k * BP;
      for (int t = 0; t < BP; ++t)
      {
        const double fk = yk[t];
        for (int r = 0; r < BP; ++r)
          w[r] -= fk * Tk[t * BP + r];
      }
    }
    for (int r = 0; r < BP; ++r)
      y[(size_t)i * BP + r] = w[r];
  }
}
__attribute__((target_clones("avx512f", "avx2", "default"))) static void block_schur_40(const BlockEnvelope &E, double *T,
                                                                                        double *X, double *y, int nI)
{
  block_schur_rows<5>(E, T, X, y, nI);
}
__attribute__((target_clones("avx512f", "avx2", "default"))) static void block_schur_24(const BlockEnvelope &E, double *T,
                                                                                        double *X, double *y, int nI)
{
  block_schur_rows<3>(E, T, X, y, nI);
}

// ---------------------------------------------------------------------------------------------------------------
// Separator rows on several cores (loop-closure plans, plan_blocks: cover keyframes).  A separator row i has a range in
// the first half (A = [a_first, a_first + a_cnt)), a range in the second half ([row_first, sep0)) and the separator
// columns [sep0, i).  Everything a row does inside ONE half only needs that half's factor and the row's own blocks:
//   task A (row i, half h):  L_ij for the columns j of that range (ascending, following the half's progress counter),
//                            the partial sums  -sum_j L_ij L_ij^T  (diagonal block) and  -sum_j L_ij y_j  (rhs)
//   task B (rows i > i2, half h):  -sum_k L_i2,k L_ik^T over the common columns of the two rows in that half
// all into private buffers; the separator block itself (a handful of rows) is then finished by the caller, who adds the
// partial sums in a fixed order (deterministic: the result does not depend on which thread ran which task).
// ---------------------------------------------------------------------------------------------------------------
struct SepTaskA
{
  int row, half, j0, j1; // columns [j0, j1)
  // r05: a long chain may carry a PARTNER row of the same half and column range (the partner's own entry is a `slave`: its
  // slots are filled by the master's thread): the two rows share the loads of the half's blocks column by column, and the
  // number of long chains fits the cores of the halves' L3 domain (a chain on another domain runs ~30 % slower and the
  // separator then waits a millisecond for it: config 5, 3 cover rows x 2 halves on 4 free cores)
  int partner = -1;
  bool slave = false;
};
// Which long arrow-row chains a thread takes: domain 0 = the caller's L3 domain (first half), 1 = the second half's own L3
// domain of a two-domain placement, -1 = a pool worker elsewhere on the node (short tasks and pair products only)
static thread_local int tl_domain = 0;
static std::atomic<int> g_domain_threads[2] = {{-1}, {0}}; // pool workers per domain (-1: unknown / no pool)
static std::atomic<bool> g_two_domains{false};             // the second half, its look-ahead and its chains sit on domain 1
struct SepTaskB
{
  int i, i2, half, k0, k1; // common columns [k0, k1), i2 < i
};
struct SepJob
{
  const BlockEnvelope *E = nullptr;
  double *T = nullptr, *X = nullptr, *y = nullptr;
  int sep0 = 0;
  std::vector<SepTaskA> ta;
  std::vector<SepTaskB> tb;
  std::vector<double> Sd, wd, Pp; // [ta][BB], [ta][Bp], [tb][BB]
  std::atomic<int> nextA{0}, doneA{0}, nextB{0}, doneB{0};
  std::atomic<int> nextLong[2];
  std::vector<int> long_tasks[2], short_tasks; // indices into ta: chains of > 16 columns by half (masters only) / the rest
  std::atomic<int> abort{0};
  std::atomic<int> progress[2];
  // phase C (back substitution): y_i -= sum over the separator rows m of L_mi^T x_m for the rows i of the halves, in
  // chunks of rows -- after the separator rows' x are known (goC: 0 wait, 1 go, 2 skip)
  std::vector<std::pair<int, int>> tc;
  std::atomic<int> goC{0}, nextC{0}, doneC{0};
  double t_start = 0.0; // (SAGE_DEBUG_TIMING)
  int dbg_waits[64] = {};
};

static void sep_job_build(SepJob &J)
{
  const BlockEnvelope &E = *J.E;
  const int sep0 = E.n1 + E.n2, K = E.K, BB = E.Bp * E.Bp;
  J.sep0 = sep0;
  auto rng = [&](int i, int h, int &a, int &b) { // separator row i, half h -> [a, b)
    if (h == 0)
    {
      a = E.a_cnt ? E.a_first[i] : 0;
      b = a + (E.a_cnt ? E.a_cnt[i] : 0);
    }
    else
    {
      a = std::min((int)E.row_first[i], sep0);
      b = sep0;
    }
  };
  for (int i = sep0; i < K; ++i)
    for (int h = 0; h < 2; ++h)
    {
      int a, b;
      rng(i, h, a, b);
      if (b > a)
        J.ta.push_back({i, h, a, b});
    }
  // longest first: the long arrow rows start early, the short middle-separator rows fill the gaps
  std::stable_sort(J.ta.begin(), J.ta.end(), [](const SepTaskA &x, const SepTaskA &y) { return x.j1 - x.j0 > y.j1 - y.j0; });
  {
    // pair long chains of the same half and range until they fit the fast threads (the pool workers on the caller's L3
    // domain + the caller and the second half's helper join only after their halves: not counted)
    const bool two = g_two_domains.load(std::memory_order_acquire);
    const int fast0 = g_domain_threads[0].load(std::memory_order_acquire), fast1 = g_domain_threads[1].load(std::memory_order_acquire);
    auto is_long = [&](const SepTaskA &a) { return a.j1 - a.j0 > 16; };
    int n_long[2] = {0, 0};
    for (auto &a : J.ta)
      if (is_long(a))
        ++n_long[a.half];
    // capacity per half: its own domain's workers (two domains), or the one domain's workers shared by both halves
    auto over = [&](int h) {
      if (fast0 <= 0)
        return false;
      return two ? n_long[h] > std::max(1, h == 0 ? fast0 : fast1) : n_long[0] + n_long[1] > fast0;
    };
    if (!sage::env_flag("SAGE_SOLVE_NO_PAIRING"))
      for (size_t t = 0; t < J.ta.size(); ++t)
      {
        SepTaskA &a = J.ta[t];
        if (!is_long(a) || a.slave || a.partner >= 0 || !over(a.half))
          continue;
        for (size_t u = J.ta.size(); u-- > t + 1;) // (the last rows first: they are the ones that used to end up off-domain)
        {
          SepTaskA &b = J.ta[u];
          if (is_long(b) && !b.slave && b.partner < 0 && b.half == a.half && b.j0 >= a.j0 && b.j1 == a.j1)
          {
            a.partner = (int)u;
            b.slave = true;
            --n_long[a.half];
            break;
          }
        }
      }
    J.nextLong[0].store(0, std::memory_order_relaxed);
    J.nextLong[1].store(0, std::memory_order_relaxed);
    for (size_t t = 0; t < J.ta.size(); ++t)
      if (!J.ta[t].slave)
      {
        if (is_long(J.ta[t]))
          J.long_tasks[J.ta[t].half].push_back((int)t);
        else
          J.short_tasks.push_back((int)t);
      }

  }
  // (i2 == i, r05: the row's own  sum_k L_ik L_ik^T  -- its share of the diagonal block -- as pair products too: it is not on
  //  the chain's recurrence, and one product less per column lets the arrow-row chains keep closer to the halves)
  for (int i = sep0; i < K; ++i)
    for (int i2 = sep0; i2 <= i; ++i2)
      for (int h = 0; h < 2; ++h)
      {
        int a, b, a2, b2;
        rng(i, h, a, b);
        rng(i2, h, a2, b2);
        const int k0 = std::max(a, a2), k1 = std::min(b, b2);
        // (pieces of <= 64 columns: the pair products of two long arrow rows spread over the whole pool)
        for (int c0 = k0; c0 < k1; c0 += 64)
          J.tb.push_back({i, i2, h, c0, std::min(k1, c0 + 64)});
      }
  std::stable_sort(J.tb.begin(), J.tb.end(), [](const SepTaskB &x, const SepTaskB &y) { return x.k1 - x.k0 > y.k1 - y.k0; });
  J.Sd.assign(J.ta.size() * (size_t)BB, 0.0);
  J.wd.assign(J.ta.size() * (size_t)E.Bp, 0.0);
  J.Pp.assign(J.tb.size() * (size_t)BB, 0.0);
  J.progress[0].store(0, std::memory_order_relaxed);
  J.progress[1].store(E.n1, std::memory_order_relaxed);
  if (E.col_ptr)
    for (int r0 = 0; r0 < sep0; r0 += 32)
      J.tc.push_back({r0, std::min(sep0, r0 + 32)});
}

template <int NV>
static inline __attribute__((always_inline)) void sep_run_c(SepJob &J, int t)
{
  constexpr int BP = NV * 8, BB = BP * BP;
  const BlockEnvelope &E = *J.E;
  double *T = J.T, *y = J.y;
  auto blk = [&](int r, int c) {
    return T + (size_t)(c < E.row_first[r] ? E.a_off[r] + c - E.a_first[r] : E.row_off[r] + c - E.row_first[r]) * BB;
  };
  for (int i = J.tc[t].first; i < J.tc[t].second; ++i)
  {
    const int n0 = E.col_ptr[i], n1 = E.col_ptr[i + 1];
    int q0 = n0;
    while (q0 < n1 && E.col_rows[q0] < J.sep0)
      ++q0;
    if (q0 == n1)
      continue;
    for (int t0 = 0; t0 < BP; t0 += 8)
    {
      v8d acc[8];
      for (int u = 0; u < 8; ++u)
        acc[u] = v8d{0, 0, 0, 0, 0, 0, 0, 0};
      for (int q = q0; q < n1; ++q)
      {
        const int m = E.col_rows[q];
        const double *Tm = blk(m, i) + t0 * BP, *xm = y + (size_t)m * BP;
        for (int v = 0; v < NV; ++v)
        {
          v8d xv;
          SAGE_LOADU(xv, xm + 8 * v);
          for (int u = 0; u < 8; ++u)
          {
            v8d a;
            SAGE_LOADU(a, Tm + u * BP + 8 * v);
            acc[u] += a * xv;
          }
        }
      }
      for (int u = 0; u < 8; ++u)
        y[(size_t)i * BP + t0 + u] -=
            ((acc[u][0] + acc[u][4]) + (acc[u][1] + acc[u][5])) + ((acc[u][2] + acc[u][6]) + (acc[u][3] + acc[u][7]));
    }
  }
}

template <int NV>
static inline __attribute__((always_inline)) void sep_run_a(SepJob &J, int t)
{
  constexpr int BP = NV * 8, BB = BP * BP;
  const BlockEnvelope &E = *J.E;
  double *T = J.T, *X = J.X, *y = J.y;
  const SepTaskA &a = J.ta[t];
  auto afirst = [&](int r) { return E.a_cnt ? E.a_first[r] : 0; };
  auto acnt = [&](int r) { return E.a_cnt ? E.a_cnt[r] : 0; };
  auto has = [&](int r, int c) { return (c >= E.row_first[r] && c <= r) || (c >= afirst(r) && c < afirst(r) + acnt(r)); };
  auto blk = [&](int r, int c) {
    return T + (size_t)(c < E.row_first[r] ? E.a_off[r] + c - E.a_first[r] : E.row_off[r] + c - E.row_first[r]) * BB;
  };
  // the rows this thread carries: the task's own and, for a paired chain, its partner's (same half, same columns)
  const int rows[2] = {a.row, a.partner >= 0 ? J.ta[a.partner].row : -1};
  const int row_j0[2] = {a.j0, a.partner >= 0 ? J.ta[a.partner].j0 : 0}; // (a partner may join at a later column)
  double *wds[2] = {J.wd.data() + (size_t)t * BP, a.partner >= 0 ? J.wd.data() + (size_t)a.partner * BP : nullptr};
  RowPrefetch pf{nullptr, nullptr};
  for (int j = a.j0; j < a.j1; ++j)
  {
    // the half's row j (its blocks, X_j and y_j) must be final
    unsigned spins = 0;
    while (J.progress[a.half].load(std::memory_order_acquire) <= j)
    {
      __builtin_ia32_pause();
      if ((++spins & 0xff) == 0 && J.abort.load(std::memory_order_acquire))
        return;
    }
    if (spins)
      ++J.dbg_waits[t & 63]; // (SAGE_DEBUG_TIMING: columns at which this chain had caught up with its half)
    if (j + 1 < a.j1)
    {
      // the half's row j + 1 (its blocks are contiguous in the storage) and its inverse: asked for now, used next column
      const int r1 = j + 1;
      pf.p2 = reinterpret_cast<const char *>(T + (size_t)E.row_off[r1] * BB);
      pf.end2 = pf.p2 + (size_t)(r1 - E.row_first[r1] + 1) * BB * sizeof(double);
      pf.p3 = reinterpret_cast<const char *>(X + (size_t)r1 * BB);
      pf.end3 = pf.p3 + BB * sizeof(double);
    }
    for (int q = 0; q < 2 && rows[q] >= 0; ++q)
    {
      const int i = rows[q];
      if (j < row_j0[q])
        continue;
      if (E.ready && E.fill && E.fill[(blk(i, j) - T) / BB])
        std::memset(blk(i, j), 0, BB * sizeof(double)); // structural fill: not delivered, zeroed at its first touch
      else if (E.ready) // this block of row i has arrived from the device
      {
        const volatile unsigned *f = E.ready + (blk(i, j) - T) / BB;
        const double t0 = mono_seconds();
        spins = 0;
        while (*f != E.epoch)
        {
          __builtin_ia32_pause();
          if ((++spins & 0xfff) == 0 && (J.abort.load(std::memory_order_acquire) || mono_seconds() - t0 > 2.0))
          {
            J.abort.store(2, std::memory_order_release);
            return;
          }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
      }
      double *CT = blk(i, j);
      if (j + 1 < a.j1)
      {
        pf.p = reinterpret_cast<const char *>(blk(i, j + 1));
        pf.end = pf.p + BB * sizeof(double);
      }
      for (int k = std::max(row_j0[q], (int)E.row_first[j]); k < j; ++k) // (row j is a half row: its columns are [row_first[j], j])
        if (has(j, k))
          tn_sub<NV>(CT, blk(j, k), blk(i, k), false, pf);
      apply_inverse<NV>(CT, X + (size_t)j * BB);
      // (the diagonal share  sum_j CT_j CT_j^T  is a pair product of phase B since r05: not on this chain's recurrence)
      const double *yk = y + (size_t)j * BP;
      double *wd = wds[q];
      for (int tt = 0; tt < BP; ++tt)
      {
        const double f = yk[tt];
        for (int r = 0; r < BP; ++r)
          wd[r] -= f * CT[tt * BP + r];
      }
    }
  }
}

template <int NV>
static inline __attribute__((always_inline)) void sep_run_b(SepJob &J, int t)
{
  constexpr int BP = NV * 8, BB = BP * BP;
  const BlockEnvelope &E = *J.E;
  double *T = J.T;
  const SepTaskB &b = J.tb[t];
  auto blk = [&](int r, int c) {
    return T + (size_t)(c < E.row_first[r] ? E.a_off[r] + c - E.a_first[r] : E.row_off[r] + c - E.row_first[r]) * BB;
  };
  double *P = J.Pp.data() + (size_t)t * BB;
  RowPrefetch pf{nullptr, nullptr};
  for (int k = b.k0; k < b.k1; ++k)
    tn_sub<NV>(P, blk(b.i2, k), blk(b.i, k), b.i2 == b.i, pf); // (a row with itself: the upper triangle is all the factorisation reads)
}

// the separator block: rows [sep0, K) with the partial sums of the tasks folded in; then their back substitution
template <int NV>
static inline __attribute__((always_inline)) int sep_finish(SepJob &J)
{
  constexpr int BP = NV * 8, BB = BP * BP;
  const BlockEnvelope &E = *J.E;
  double *T = J.T, *X = J.X, *y = J.y;
  const int sep0 = J.sep0, K = E.K;
  auto blk = [&](int r, int c) {
    return T + (size_t)(c < E.row_first[r] ? E.a_off[r] + c - E.a_first[r] : E.row_off[r] + c - E.row_first[r]) * BB;
  };
  RowPrefetch pf{nullptr, nullptr};
  for (int i = sep0; i < K; ++i)
  {
    // (the blocks of the columns < sep0 went through the arrow-row tasks one by one)
    if (E.ready && !wait_row_tickets(E, i, T, sep0))
      return -2;
    const int c0 = std::max((int)E.row_first[i], sep0);
    for (int j = c0; j < i; ++j)
    {
      double *CT = blk(i, j);
      for (size_t t = 0; t < J.tb.size(); ++t) // fixed order: deterministic
        if (J.tb[t].i == i && J.tb[t].i2 == j)
        {
          const double *P = J.Pp.data() + t * BB;
          for (int o = 0; o < BB; ++o)
            CT[o] += P[o];
        }
      for (int k = std::max(c0, std::max((int)E.row_first[j], sep0)); k < j; ++k)
        tn_sub<NV>(CT, blk(j, k), blk(i, k), false, pf);
      apply_inverse<NV>(CT, X + (size_t)j * BB);
    }
    double *S = blk(i, i);
    double w[BP];
    for (int r = 0; r < BP; ++r)
      w[r] = y[(size_t)i * BP + r];
    for (size_t t = 0; t < J.tb.size(); ++t) // the row's own pair products (fixed order: deterministic)
      if (J.tb[t].i == i && J.tb[t].i2 == i)
      {
        const double *P = J.Pp.data() + t * BB;
        for (int o = 0; o < BB; ++o)
          S[o] += P[o];
      }
    for (size_t t = 0; t < J.ta.size(); ++t)
      if (J.ta[t].row == i)
      {
        const double *wd = J.wd.data() + t * BP;
        for (int r = 0; r < BP; ++r)
          w[r] += wd[r];
      }
    for (int k = c0; k < i; ++k)
    {
      tn_sub<NV>(S, blk(i, k), blk(i, k), true, pf);
      const double *Tk = blk(i, k), *yk = y + (size_t)k * BP;
      for (int t = 0; t < BP; ++t)
      {
        const double f = yk[t];
        for (int r = 0; r < BP; ++r)
          w[r] -= f * Tk[t * BP + r];
      }
    }
    if (!factor_diag<NV>(S, X + (size_t)i * BB))
      return 1 + i;
    double yi[BP];
    for (int c = 0; c < BP; ++c)
      yi[c] = 0.0;
    const double *Xi = X + (size_t)i * BB;
    for (int t = 0; t < BP; ++t)
    {
      const double f = w[t];
      for (int c = 0; c < BP; ++c)
        yi[c] += f * Xi[t * BP + c];
    }
    for (int c = 0; c < BP; ++c)
      y[(size_t)i * BP + c] = yi[c];
  }
  return 0;
}

__attribute__((target_clones("avx512f", "avx2", "default"))) static void sep_run_a_40(SepJob &J, int t) { sep_run_a<5>(J, t); }
__attribute__((target_clones("avx512f", "avx2", "default"))) static void sep_run_a_24(SepJob &J, int t) { sep_run_a<3>(J, t); }
__attribute__((target_clones("avx512f", "avx2", "default"))) static void sep_run_b_40(SepJob &J, int t) { sep_run_b<5>(J, t); }
__attribute__((target_clones("avx512f", "avx2", "default"))) static void sep_run_b_24(SepJob &J, int t) { sep_run_b<3>(J, t); }
__attribute__((target_clones("avx512f", "avx2", "default"))) static void sep_run_c_40(SepJob &J, int t) { sep_run_c<5>(J, t); }
__attribute__((target_clones("avx512f", "avx2", "default"))) static void sep_run_c_24(SepJob &J, int t) { sep_run_c<3>(J, t); }
__attribute__((target_clones("avx512f", "avx2", "default"))) static int sep_finish_40(SepJob &J) { return sep_finish<5>(J); }
__attribute__((target_clones("avx512f", "avx2", "default"))) static int sep_finish_24(SepJob &J) { return sep_finish<3>(J); }

// take tasks until none is left (called by the pool's workers, the helper after its half, and the caller).  Long chains go
// to the threads of the halves' L3 domains only (tl_domain >= 0), longest first; everybody takes the short ones.
static void sep_work(SepJob &J)
{
  const bool b40 = J.E->Bp == 40;
  const int nA = (int)J.ta.size(), nB = (int)J.tb.size();
  static const bool dbg_a = sage::env_flag("SAGE_DEBUG_TIMING");
  auto run_a = [&](int t) {
    const double ta0 = dbg_a ? mono_seconds() : 0.0;
    if (!J.abort.load(std::memory_order_acquire))
      b40 ? sep_run_a_40(J, t) : sep_run_a_24(J, t);
    if (dbg_a && J.ta[t].j1 - J.ta[t].j0 > 16)
      fprintf(stderr, "[sage arrow task] row %d%s half %d cols %d: %.0f us on cpu %d (ends %.0f us after job start; caught up with its half at %d columns)\n",
              J.ta[t].row, J.ta[t].partner >= 0 ? " (+ a partner row)" : "", J.ta[t].half, J.ta[t].j1 - J.ta[t].j0,
              1e6 * (mono_seconds() - ta0), sched_getcpu(), 1e6 * (mono_seconds() - J.t_start), J.dbg_waits[t & 63]);
    J.doneA.fetch_add(J.ta[t].partner >= 0 ? 2 : 1, std::memory_order_acq_rel);
  };
  if (tl_domain >= 0)
  {
    // its own half's chains first; then whatever is left of the other half's (a chain nobody has started is better run
    // across domains than not at all: liveness does not depend on the placement)
    const int first = g_two_domains.load(std::memory_order_acquire) ? (tl_domain & 1) : 0;
    for (int pass = 0; pass < 2; ++pass)
    {
      const int h = pass == 0 ? first : 1 - first;
      for (;;)
      {
        const int q = J.nextLong[h].fetch_add(1, std::memory_order_acq_rel);
        if (q >= (int)J.long_tasks[h].size())
          break;
        run_a(J.long_tasks[h][q]);
      }
    }
  }
  for (;;)
  {
    const int q = J.nextA.fetch_add(1, std::memory_order_acq_rel);
    if (q >= (int)J.short_tasks.size())
      break;
    run_a(J.short_tasks[q]);
  }
  while (J.doneA.load(std::memory_order_acquire) < nA) // phase B reads the rows phase A completes
    __builtin_ia32_pause();
  for (;;)
  {
    const int t = J.nextB.fetch_add(1, std::memory_order_acq_rel);
    if (t >= nB)
      break;
    if (!J.abort.load(std::memory_order_acquire))
      b40 ? sep_run_b_40(J, t) : sep_run_b_24(J, t);
    J.doneB.fetch_add(1, std::memory_order_acq_rel);
  }
}

// phase C: entered by the pool's workers right after sep_work (they wait for the caller's go), by the caller when the
// separator rows are back-substituted
static void sep_work_c(SepJob &J, bool wait_for_go)
{
  if (wait_for_go)
  {
    int g;
    while ((g = J.goC.load(std::memory_order_acquire)) == 0)
      __builtin_ia32_pause();
    if (g != 1)
      return;
  }
  const bool b40 = J.E->Bp == 40;
  const int nC = (int)J.tc.size();
  for (;;)
  {
    const int t = J.nextC.fetch_add(1, std::memory_order_acq_rel);
    if (t >= nC)
      break;
    b40 ? sep_run_c_40(J, t) : sep_run_c_24(J, t);
    J.doneC.fetch_add(1, std::memory_order_acq_rel);
  }
}

// Worker pool for the arrow rows: persistent threads that sleep on a condition variable, are woken by
// block_chol_arm(true) and spin for a job for a few milliseconds (like the helper of the second half).
struct SepPool
{
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<bool> armed{false};
  std::atomic<unsigned> posted{0};
  std::atomic<bool> open{false};
  std::atomic<int> active{0};
  std::atomic<bool> busy{false};
  std::atomic<SepJob *> job{nullptr};
  std::vector<pthread_t> tids;
  std::vector<std::thread> ths; // joinable: host_threads_shutdown() stops and joins them, block_chol_arm() starts them again
  int n_workers = 0;
  std::atomic<bool> quit{false}, running{false};
  unsigned seen0 = 0;
  std::unique_ptr<std::atomic<int>[]> ktid, cpu; // per worker: kernel thread id, CPU it is pinned to (-1: none) -- placement monitor
  std::vector<int> dom; // per worker: 0 / 1 = pinned to a core of the first / second half's L3 domain, -1 elsewhere (place_pool)
  std::atomic<int> near_cpu{-1};
  std::atomic<int> near_mode{-1};
  void loop(int idx)
  {
    unsigned seen = seen0;
    for (;;)
    {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return armed.load(std::memory_order_acquire) || quit.load(std::memory_order_acquire); });
      }
      if (quit.load(std::memory_order_acquire))
        return;
      double t0 = mono_seconds();
      unsigned spins = 0;
      while (armed.load(std::memory_order_acquire) && !quit.load(std::memory_order_relaxed))
      {
        const unsigned p = posted.load(std::memory_order_acquire);
        if (p != seen)
        {
          seen = p;
          // hand-off: the owner does `open = false` THEN reads `active`; a worker does `active += 1` THEN reads `open`.
          // A store followed by a load of another variable needs sequential consistency on both sides (with release /
          // acquire the two may be reordered -- the owner reads active == 0 while a late worker still reads open == true
          // and runs a job that lives on the owner's stack after the owner has returned)
          active.fetch_add(1, std::memory_order_seq_cst);
          if (open.load(std::memory_order_seq_cst))
          {
            SepJob *j = job.load(std::memory_order_acquire);
            tl_domain = (size_t)idx < dom.size() ? dom[idx] : 0;
            sep_work(*j);
            sep_work_c(*j, true);
          }
          active.fetch_sub(1, std::memory_order_seq_cst);
          t0 = mono_seconds(); // the idle time-out counts from the last job, not from the wake-up
        }
        __builtin_ia32_pause();
        // nobody came for 20 ms: back to sleep.  Only while no job is open (the owner clears `armed` itself when its solve
        // is done): a worker that timed out in the middle of another caller's arm / post must not disarm the pool
        if ((++spins & 1023) == 0 && mono_seconds() - t0 > 20e-3 && !open.load(std::memory_order_acquire))
          armed.store(false, std::memory_order_release);
      }
    }
  }
};
// ---- life cycle of the solver's host threads (r06).  The OBJECTS (CholHelper x 3, SepPool) are made once and never freed
// -- a caller that lost a race with a shutdown still holds valid memory and, by the hand-over protocols' design, depends on
// no thread that has not claimed its job -- but the THREADS are joinable: block_chol_arm() starts them on demand,
// host_threads_shutdown() (sage_shutdown(), the last sage_window_destroy, atexit) stops and joins them.  Nothing is detached.
static std::mutex g_threads_mu;
static std::atomic<SepPool *> g_sep_pool_made{nullptr}; // (the placement monitor must not CREATE the pool by asking for it)
static SepPool *sep_pool()
{
  static SepPool *p = [] {
    const unsigned hw = std::thread::hardware_concurrency();
    // (r04: the halves run two threads each now, so the arrow-row tasks are what the halves wait for -- config 5 per LM
    //  step with 4 / 6 / 10 / 14 workers: 7.3 / 6.9 / 6.55 / 6.5 ms on one box)
    int n = getenv("SAGE_SOLVE_POOL") ? atoi(getenv("SAGE_SOLVE_POOL")) : 10;
    n = std::min(n, (int)hw - 4);
    if (n < 1)
      return (SepPool *)nullptr;
    SepPool *q = new SepPool;
    q->n_workers = n;
    q->ktid.reset(new std::atomic<int>[n]);
    q->cpu.reset(new std::atomic<int>[n]);
    for (int i = 0; i < n; ++i)
    {
      q->ktid[i].store(0);
      q->cpu[i].store(-1);
    }
    g_sep_pool_made.store(q, std::memory_order_release);
    return q;
  }();
  return p;
}
// (g_threads_mu held)
static void sep_pool_start(SepPool *q)
{
  if (q->running.load(std::memory_order_acquire))
    return;
  host_threads_atexit_once();
  q->quit.store(false, std::memory_order_release);
  q->tids.clear();
  q->ths.clear();
  q->near_cpu.store(-1, std::memory_order_release); // new threads: not pinned yet, place_pool places them again
  q->near_mode.store(-1, std::memory_order_release);
  q->seen0 = q->posted.load(std::memory_order_acquire);
  for (int i = 0; i < q->n_workers; ++i)
  {
    q->ktid[i].store(0);
    q->cpu[i].store(-1);
    q->ths.emplace_back([q, i] {
      q->ktid[i].store((int)syscall(SYS_gettid), std::memory_order_release);
      q->loop(i);
    });
    q->tids.push_back(q->ths.back().native_handle());
  }
  q->running.store(true, std::memory_order_release);
}

static int block_chol_range(const BlockEnvelope &E, double *T, double *X, double *y, int phase, int lo, int hi,
                            int role = 0)
{
  return E.Bp == 40 ? block_chol_40(E, T, X, y, phase, lo, hi, role) : block_chol_24(E, T, X, y, phase, lo, hi, role);
}

// Helper threads of a split window: [0] takes the second half, [1] / [2] are the look-ahead stages of the first / second
// half (BlockEnvelope::RowPipe).  A helper sleeps on a condition variable, is woken by block_chol_arm() (called while the
// caller still waits for the device), then spins for a job so that picking one up costs no wake-up latency.  Nobody
// depends on a helper that has not claimed its job: the second half is claimed back and run by the caller, a half without
// a look-ahead stage runs as one thread.
struct CholHelper
{
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<bool> armed{false};
  std::atomic<unsigned> posted{0};
  std::atomic<int> claim{0};   // 0 free, 1 helper, 2 caller
  std::atomic<int> p1_rc{-2};  // result of the helper's factorisation pass (-2: not finished)
  std::atomic<int> go_p2{0};   // 1: run the back substitution, 2: skip it
  std::atomic<int> p2_done{0};
  std::atomic<bool> busy{false}; // one client at a time
  const BlockEnvelope *E = nullptr;
  double *T = nullptr, *X = nullptr, *y = nullptr;
  int kind = 0; // 0: second half (factorisation, later the back substitution), 1: look-ahead stage of rows [lo, hi)
  int lo = 0, hi = 0;
  std::thread th;
  pthread_t tid{};
  std::atomic<int> ktid{0};    // kernel thread id (its /proc/self/task entry: the placement monitor reads its run-queue delay)
  std::atomic<int> cpu{-1};    // the CPU it is pinned to (-1: not pinned)
  std::atomic<int> near_cpu{-1}, near_mode{-1};
  std::atomic<bool> quit{false}, running{false}; // joinable thread: started by block_chol_arm, stopped by host_threads_shutdown
  unsigned seen0 = 0;
  static void cpu_relax() { __builtin_ia32_pause(); }
  void loop();
};
static CholHelper *chol_helper(int idx = 0)
{
  // the objects live for the life of the process (see "life cycle" above); their threads come and go
  static CholHelper **hs = [] {
    CholHelper **v = new CholHelper *[3]{nullptr, nullptr, nullptr};
    const unsigned hc = std::thread::hardware_concurrency();
    for (int i = 0; i < 3; ++i)
    {
      if (hc < (i == 0 ? 2u : 4u))
        continue;
      CholHelper *p = new CholHelper;
      p->kind = i == 0 ? 0 : 1;
      v[i] = p;
    }
    return v;
  }();
  return hs[idx];
}
// (g_threads_mu held)
static void chol_helper_start(CholHelper *p)
{
  if (p->running.load(std::memory_order_acquire))
    return;
  host_threads_atexit_once();
  p->quit.store(false, std::memory_order_release);
  p->ktid.store(0, std::memory_order_release);
  p->cpu.store(-1, std::memory_order_release);
  p->near_cpu.store(-1, std::memory_order_release);
  p->near_mode.store(-1, std::memory_order_release);
  p->seen0 = p->posted.load(std::memory_order_acquire);
  p->th = std::thread([p] {
    p->ktid.store((int)syscall(SYS_gettid), std::memory_order_release);
    p->loop();
  });
  p->tid = p->th.native_handle();
  p->running.store(true, std::memory_order_release);
}

static std::atomic<long long> g_lookahead_count{0};

// rows [lo, hi) can run as two stages: plain band rows (no A range, nothing left of lo), no per-row callback
static bool rows_can_pipe(const BlockEnvelope &E, int lo, int hi)
{
  static const bool off = sage::env_flag("SAGE_SOLVE_NO_LOOKAHEAD");
  if (off || E.no_lookahead || E.before_row || !E.pipe || hi - lo < 4)
    return false;
  for (int i = lo; i < hi; ++i)
    if ((E.a_cnt && E.a_cnt[i]) || E.row_first[i] < lo)
      return false;
  return true;
}

// hand rows [lo, hi) to look-ahead helper `idx`; true once the helper has claimed the job (it then WILL publish its
// progress in E.pipe), false when there is no armed helper or it did not answer within a few microseconds
static bool lookahead_engage(int idx, const BlockEnvelope &E, double *T, double *X, double *y, int lo, int hi)
{
  CholHelper *h = chol_helper(idx);
  if (!h || !h->armed.load(std::memory_order_acquire) || !rows_can_pipe(E, lo, hi))
    return false;
  bool expect = false;
  if (!h->busy.compare_exchange_strong(expect, true, std::memory_order_acq_rel))
    return false;
  h->E = &E; h->T = T; h->X = X; h->y = y; h->lo = lo; h->hi = hi;
  h->p1_rc.store(-2, std::memory_order_relaxed);
  h->claim.store(0, std::memory_order_release); // (a helper still looking at an older post claims only after the fields are set)
  h->posted.fetch_add(1, std::memory_order_release);
  const double t0 = mono_seconds();
  unsigned spins = 0;
  while (h->claim.load(std::memory_order_acquire) == 0)
  {
    CholHelper::cpu_relax();
    if ((++spins & 63) == 0 && mono_seconds() - t0 > 20e-6)
    {
      int e0 = 0;
      if (h->claim.compare_exchange_strong(e0, 2, std::memory_order_acq_rel))
      {
        h->busy.store(false, std::memory_order_release);
        return false;
      }
    }
  }
  g_lookahead_count.fetch_add(1, std::memory_order_relaxed);
  return true;
}
static void lookahead_release(int idx)
{
  CholHelper *h = chol_helper(idx);
  while (h->p1_rc.load(std::memory_order_acquire) == -2)
    CholHelper::cpu_relax();
  h->armed.store(false, std::memory_order_release);
  h->busy.store(false, std::memory_order_release);
}

void CholHelper::loop()
{
  unsigned seen = seen0; // (sampled by the thread that started this one, before it can post: a restarted thread does not
                         //  answer posts from before its time and cannot miss the starter's first one)
  for (;;)
  {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return armed.load(std::memory_order_acquire) || quit.load(std::memory_order_acquire); });
    }
    if (quit.load(std::memory_order_acquire))
      return;
    const double t0 = mono_seconds();
    unsigned spins = 0;
    while (armed.load(std::memory_order_acquire) && !quit.load(std::memory_order_relaxed))
    {
      const unsigned p = posted.load(std::memory_order_acquire);
      if (p != seen)
      {
        seen = p;
        int expect = 0;
        if (claim.compare_exchange_strong(expect, 1, std::memory_order_acq_rel))
        {
          const BlockEnvelope &e = *E;
          if (kind == 1)
          {
            const int rc = block_chol_range(e, T, X, y, 0, lo, hi, 2);
            p1_rc.store(rc == -2 ? -3 : rc, std::memory_order_release); // (-2 is the "not finished" value)
          }
          else
          {
            const bool piped = lookahead_engage(2, e, T, X, y, e.n1, e.n1 + e.n2);
            const int rc = block_chol_range(e, T, X, y, 0, e.n1, e.n1 + e.n2, piped ? 1 : 0);
            if (piped)
              lookahead_release(2);
            p1_rc.store(rc, std::memory_order_release);
            int g;
            while ((g = go_p2.load(std::memory_order_acquire)) == 0)
              cpu_relax();
            if (g == 1)
              block_chol_range(e, T, X, y, 1, e.n1, e.n1 + e.n2);
            p2_done.store(1, std::memory_order_release);
          }
        }
      }
      cpu_relax();
      if ((++spins & 1023) == 0 && mono_seconds() - t0 > 8e-3) // nobody came: back to sleep
        armed.store(false, std::memory_order_release);
    }
  }
}
} // namespace

// CPU list file of sysfs ("0-7,128-135") -> cpu numbers
static std::vector<int> read_cpu_list(const char *path)
{
  std::vector<int> out;
  FILE *f = fopen(path, "r");
  if (!f)
    return out;
  char buf[4096];
  if (fgets(buf, sizeof(buf), f))
  {
    const char *p = buf;
    while (*p)
    {
      char *end;
      const long a = strtol(p, &end, 10);
      if (end == p)
        break;
      long b = a;
      p = end;
      if (*p == '-')
      {
        b = strtol(p + 1, &end, 10);
        p = end;
      }
      for (long c = a; c <= b && out.size() < 4096; ++c)
        out.push_back((int)c);
      if (*p == ',')
        ++p;
    }
  }
  fclose(f);
  return out;
}

// ---- which CPUs the solve's threads may be placed on.  Default: the calling thread's affinity mask.  r05: a GPU box is a
// slice of a node whose other GPUs run other jobs -- their host threads sit on CPUs of the same NUMA node, and a helper
// pinned onto a core another tenant saturates runs its half of the factorisation at half speed for the life of the
// process (~1 process in 8 measured +60..200 us per solve).  sage_bind_thread_to_device therefore samples /proc/stat and
// hands over the CPUs of QUIET physical cores (placement_set_allowed); the caller's own mask may be narrower than that
// (its L3 domain), the loop-closure plans' second domain is looked for in the handed-over set.
static std::mutex g_place_mu;
static bool g_place_override = false;
static cpu_set_t g_place_allowed;
// (heap-allocated and never destroyed: a thread of this library may still look at them while the process runs its static
//  destructors)
static std::map<std::pair<int, bool>, std::vector<int>> &g_ccx_cache = *new std::map<std::pair<int, bool>, std::vector<int>>;
static std::map<std::pair<int, size_t>, std::vector<int>> &g_dom2_cache = *new std::map<std::pair<int, size_t>, std::vector<int>>;

void placement_set_allowed(const cpu_set_t *allowed)
{
  std::lock_guard<std::mutex> lk(g_place_mu);
  g_place_override = allowed != nullptr;
  if (allowed)
    g_place_allowed = *allowed;
  g_ccx_cache.clear();
  g_dom2_cache.clear();
}

static bool placement_allowed(cpu_set_t *out) // (g_place_mu held)
{
  if (g_place_override)
  {
    *out = g_place_allowed;
    return true;
  }
  return sched_getaffinity(0, sizeof(*out), out) == 0;
}

// CPUs that were busy (> 25 % non-idle) during a window of `ms` milliseconds; empty when /proc/stat cannot be read
std::vector<int> placement_busy_cpus(int ms)
{
  auto snap = [](std::map<int, std::pair<unsigned long long, unsigned long long>> &m) {
    FILE *f = fopen("/proc/stat", "r");
    if (!f)
      return false;
    char line[512];
    while (fgets(line, sizeof(line), f))
    {
      int cpu;
      unsigned long long v[8] = {0};
      if (sscanf(line, "cpu%d %llu %llu %llu %llu %llu %llu %llu %llu", &cpu, &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6],
                 &v[7]) >= 5)
      {
        unsigned long long tot = 0;
        for (int i = 0; i < 8; ++i)
          tot += v[i];
        m[cpu] = {tot, v[3] + v[4]};
      }
    }
    fclose(f);
    return !m.empty();
  };
  std::map<int, std::pair<unsigned long long, unsigned long long>> a, b;
  std::vector<int> out;
  if (!snap(a))
    return out;
  std::this_thread::sleep_for(std::chrono::milliseconds(ms));
  if (!snap(b))
    return out;
  for (const auto &kv : a)
  {
    auto it = b.find(kv.first);
    if (it == b.end())
      continue;
    const double tot = (double)(it->second.first - kv.second.first), idle = (double)(it->second.second - kv.second.second);
    if (tot >= 4.0 && 1.0 - idle / tot > 0.25) // (USER_HZ ticks: at least 4 in the window)
      out.push_back(kv.first);
  }
  return out;
}

// hardware threads of the physical core of `cpu`
std::vector<int> placement_core_siblings(int cpu)
{
  char path[128];
  snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
  std::vector<int> sib = read_cpu_list(path);
  if (sib.empty())
    sib.push_back(cpu);
  return sib;
}

std::vector<int> placement_l3_domain(int cpu)
{
  char path[128];
  snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
  return read_cpu_list(path);
}

// Thread placement of the solve: the caller, the helper of the second half and the worker pool each get their own
// PHYSICAL core of the caller's CCX (cores that share its L3): the halves and the arrow-row tasks then work out of one
// cache and one NUMA node (a helper on the far socket takes ~35 % longer for its half), and no two of them share a core
// through SMT (r03: a pool thread on the helper's sibling made the helper's half 35-45 % slower on config 5).
// cores[0] -> helper, cores[1 + t] -> pool thread t; threads the CCX has no core left for fall back to the rest of the
// caller's NUMA node as a set.
static std::vector<int> sibling_free_cores(const std::vector<int> &cpus, int caller_cpu, const cpu_set_t &allowed)
{
  std::vector<int> cores, seen_core;
  char path[128];
  snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", caller_cpu);
  const std::vector<int> caller_sib = read_cpu_list(path);
  const int caller_core = caller_sib.empty() ? caller_cpu : *std::min_element(caller_sib.begin(), caller_sib.end());
  for (int c : cpus)
  {
    if (c >= CPU_SETSIZE || !CPU_ISSET(c, &allowed))
      continue;
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
    const std::vector<int> sib = read_cpu_list(path);
    const int core = sib.empty() ? c : *std::min_element(sib.begin(), sib.end());
    if (core == caller_core || std::find(seen_core.begin(), seen_core.end(), core) != seen_core.end())
      continue;
    seen_core.push_back(core);
    cores.push_back(c); // the first allowed hardware thread of that core
  }
  return cores;
}

static std::vector<int> node_cpus_of(int cpu)
{
  char path[128];
  for (int node = 0; node < 64; ++node)
  {
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    const std::vector<int> nl = read_cpu_list(path);
    if (std::find(nl.begin(), nl.end(), cpu) != nl.end())
      return nl;
  }
  return {};
}

// (sysfs is read once per caller CPU: the arm call sits at the start of every solve; returned by value -- the cache is
//  cleared when the allowed set changes)
static std::vector<int> ccx_cores_of(int cpu, bool with_node)
{
  std::lock_guard<std::mutex> lk(g_place_mu);
  auto it = g_ccx_cache.find({cpu, with_node});
  if (it != g_ccx_cache.end())
    return it->second;
  std::vector<int> cores;
  char path[128];
  snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
  const std::vector<int> l3 = read_cpu_list(path);
  cpu_set_t allowed;
  if (!l3.empty() && placement_allowed(&allowed))
  {
    cores = sibling_free_cores(l3, cpu, allowed);
    if (with_node)
      for (int c : sibling_free_cores(node_cpus_of(cpu), cpu, allowed))
        if (std::find(cores.begin(), cores.end(), c) == cores.end())
          cores.push_back(c);
  }
  return g_ccx_cache.emplace(std::make_pair(cpu, with_node), std::move(cores)).first->second;
}

static void pin_one(pthread_t t, int cpu)
{
  cpu_set_t want;
  CPU_ZERO(&want);
  CPU_SET(cpu, &want);
  (void)pthread_setaffinity_np(t, sizeof(want), &want);
}

// cores of ANOTHER L3 domain of the caller's NUMA node (the next one with at least `want` free physical cores), or empty
static std::vector<int> second_domain_cores_uncached(int cpu, size_t want);
static std::vector<int> second_domain_cores(int cpu, size_t want)
{
  // (sysfs is read once per caller CPU and size: the arm call sits at the start of every solve)
  std::lock_guard<std::mutex> lk(g_place_mu);
  auto it = g_dom2_cache.find({cpu, want});
  if (it != g_dom2_cache.end())
    return it->second;
  return g_dom2_cache.emplace(std::make_pair(cpu, want), second_domain_cores_uncached(cpu, want)).first->second;
}
static std::vector<int> second_domain_cores_uncached(int cpu, size_t want)
{
  char path[128];
  snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
  const std::vector<int> l3a = read_cpu_list(path);
  cpu_set_t allowed;
  if (l3a.empty() || !placement_allowed(&allowed))
    return {};
  std::vector<int> seen = l3a;
  for (int c : node_cpus_of(cpu))
  {
    if (std::find(seen.begin(), seen.end(), c) != seen.end() || c >= CPU_SETSIZE || !CPU_ISSET(c, &allowed))
      continue;
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", c);
    const std::vector<int> l3b = read_cpu_list(path);
    if (l3b.empty())
      continue;
    seen.insert(seen.end(), l3b.begin(), l3b.end());
    std::vector<int> cores = sibling_free_cores(l3b, cpu, allowed);
    if (cores.size() >= want)
      return cores;
  }
  return {};
}

// ---- placement monitor (r05).  The box's other tenants move: a core that was quiet when the helpers were placed may carry
// somebody else's thread a minute later, and a helper that shares its hardware thread (it then waits on the run queue when the
// solve wakes it) or its physical core (SMT: ~2/3 speed) slows every solve of the process from then on.  A background thread
// looks every 250 ms at (a) the run-queue delay of the three helper threads (/proc/self/task/<tid>/schedstat) and (b) the load
// on the OTHER hardware threads of their cores (/proc/stat); a helper that is crowded in two consecutive looks is moved to
// a core of its own L3 domain (else the caller's, else the NUMA node) that is idle on all its hardware threads; the workers of
// the loop-closure plans' arrow-row pool are watched the same way once the pool exists.  SAGE_PLACEMENT_MONITOR=0 turns it off; SAGE_DEBUG_TIMING prints the moves.
static std::atomic<bool> g_monitor_started{false};
static std::atomic<int> g_monitor_moves{0};
static std::mutex g_pin_mu;
static std::mutex g_monitor_mu;
static std::condition_variable g_monitor_cv;
static bool g_monitor_stop = false;           // (g_monitor_mu)
static std::thread *g_monitor_thread = nullptr; // (g_threads_mu)

static long long read_run_delay_ns(int ktid)
{
  char path[96];
  snprintf(path, sizeof(path), "/proc/self/task/%d/schedstat", ktid);
  FILE *f = fopen(path, "r");
  if (!f)
    return -1;
  unsigned long long run = 0, delay = 0;
  const int n = fscanf(f, "%llu %llu", &run, &delay);
  fclose(f);
  return n == 2 ? (long long)delay : -1;
}

static bool stat_snapshot(std::map<int, std::pair<unsigned long long, unsigned long long>> &m)
{
  FILE *f = fopen("/proc/stat", "r");
  if (!f)
    return false;
  char line[512];
  while (fgets(line, sizeof(line), f))
  {
    int cpu;
    unsigned long long v[8] = {0};
    if (sscanf(line, "cpu%d %llu %llu %llu %llu %llu %llu %llu %llu", &cpu, &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6],
               &v[7]) >= 5)
    {
      unsigned long long tot = 0;
      for (int i = 0; i < 8; ++i)
        tot += v[i];
      m[cpu] = {tot, v[3] + v[4]};
    }
  }
  fclose(f);
  return !m.empty();
}

static void placement_monitor_loop()
{
  const bool verbose = sage::env_flag("SAGE_DEBUG_TIMING");
  std::map<int, std::pair<unsigned long long, unsigned long long>> prev, cur;
  // watched threads: slots 0..2 the helpers, 3.. the arrow-row pool's workers (once a loop-closure plan has made the pool)
  std::vector<long long> prev_delay(3, -1);
  std::vector<int> strikes(3, 0);
  stat_snapshot(prev);
  {
    // its own affinity: the CPUs the placement may use (not the one-L3 mask inherited from the LM thread that started it,
    // where it would compete with the thread that spins)
    cpu_set_t allowed;
    bool ok;
    {
      std::lock_guard<std::mutex> lk(g_place_mu);
      ok = placement_allowed(&allowed);
    }
    if (ok)
      (void)pthread_setaffinity_np(pthread_self(), sizeof(allowed), &allowed);
  }
  for (;;)
  {
    {
      std::unique_lock<std::mutex> lm(g_monitor_mu);
      if (g_monitor_cv.wait_for(lm, std::chrono::milliseconds(250), [] { return g_monitor_stop; }))
        return;
    }
    cur.clear();
    if (!stat_snapshot(cur))
      continue;
    auto busy = [&](int c) {
      auto a = prev.find(c), b = cur.find(c);
      if (a == prev.end() || b == cur.end())
        return 0.0;
      const double tot = (double)(b->second.first - a->second.first), idle = (double)(b->second.second - a->second.second);
      return tot >= 4.0 ? 1.0 - idle / tot : 0.0;
    };
    CholHelper *hs[3] = {chol_helper(0), chol_helper(1), chol_helper(2)};
    SepPool *q = g_sep_pool_made.load(std::memory_order_acquire);
    const int nq = q ? q->n_workers : 0;
    prev_delay.resize(3 + nq, -1);
    strikes.resize(3 + nq, 0);
    auto slot_cpu = [&](int i) -> std::atomic<int> * {
      return i < 3 ? ((hs[i] && hs[i]->running.load(std::memory_order_acquire)) ? &hs[i]->cpu : nullptr)
                   : (q->running.load(std::memory_order_acquire) ? &q->cpu[i - 3] : nullptr);
    };
    auto slot_ktid = [&](int i) { return i < 3 ? (hs[i] ? hs[i]->ktid.load(std::memory_order_acquire) : 0) : q->ktid[i - 3].load(std::memory_order_acquire); };
    auto slot_thread = [&](int i) { return i < 3 ? hs[i]->tid : q->tids[i - 3]; };
    const int near = hs[0] ? hs[0]->near_cpu.load(std::memory_order_acquire) : (q ? q->near_cpu.load(std::memory_order_acquire) : -1);
    for (int i = 0; i < 3 + nq; ++i)
    {
      std::atomic<int> *pc = slot_cpu(i);
      if (!pc)
        continue;
      const int c = pc->load(std::memory_order_acquire), kt = slot_ktid(i);
      if (c < 0 || kt <= 0)
        continue;
      const long long d = read_run_delay_ns(kt);
      const long long dd = (d >= 0 && prev_delay[i] >= 0) ? d - prev_delay[i] : 0;
      prev_delay[i] = d;
      bool crowded = dd > 2000000; // > 2 ms on the run queue in a quarter second: somebody shares the hardware thread
      double sib_busy = 0.0;
      for (int sib : placement_core_siblings(c))
        if (sib != c)
          sib_busy = std::max(sib_busy, busy(sib));
      crowded = crowded || sib_busy > 0.3;
      strikes[i] = crowded ? strikes[i] + 1 : 0;
      if (strikes[i] < 2 || near < 0)
        continue;
      // a quiet core: the thread's own L3 domain first (a pool worker of the second half's domain stays there), then the
      // caller's domain and node; idle on all hardware threads, not used by another watched thread
      std::vector<int> cands = placement_l3_domain(c);
      for (int x : ccx_cores_of(near, true))
        cands.push_back(x);
      cpu_set_t allowed;
      {
        std::lock_guard<std::mutex> lk(g_place_mu);
        if (!placement_allowed(&allowed))
          continue;
      }
      const std::vector<int> near_sib = placement_core_siblings(near);
      int target = -1;
      for (int cand : cands)
      {
        if (cand >= CPU_SETSIZE || !CPU_ISSET(cand, &allowed) || std::find(near_sib.begin(), near_sib.end(), cand) != near_sib.end())
          continue;
        bool ok = true;
        for (int sib : placement_core_siblings(cand))
        {
          ok = ok && busy(sib) < 0.1;
          for (int j = 0; j < 3 + nq && ok; ++j)
          {
            std::atomic<int> *pj = slot_cpu(j);
            ok = !(pj && pj->load(std::memory_order_acquire) == sib);
          }
        }
        if (ok)
        {
          target = cand;
          break;
        }
      }
      if (target < 0)
        continue;
      {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        pin_one(slot_thread(i), target);
        pc->store(target, std::memory_order_release);
      }
      g_monitor_moves.fetch_add(1, std::memory_order_relaxed);
      strikes[i] = 0;
      if (verbose)
        fprintf(stderr, "[sage placement] %s %d: cpu %d crowded (run-queue delay %.1f ms, sibling load %.0f %%) -> cpu %d\n",
                i < 3 ? "helper" : "pool worker", i < 3 ? i : i - 3, c, dd * 1e-6, 100.0 * sib_busy, target);
    }
    prev.swap(cur);
  }
}


// r06: OPT-IN (SAGE_PLACEMENT_MONITOR=1 or sage_placement_monitor(1)) -- a drop-in library does not edit thread affinities
// from a background thread unless asked to.  Joinable: host_threads_shutdown() stops it.
static std::atomic<int> g_monitor_wanted{-1}; // -1: ask the environment, 0 / 1: set through the API
static void placement_monitor_start()
{
  int want = g_monitor_wanted.load(std::memory_order_acquire);
  if (want < 0)
  {
    const char *e = getenv("SAGE_PLACEMENT_MONITOR");
    want = (e && atoi(e) != 0) ? 1 : 0;
    g_monitor_wanted.store(want, std::memory_order_release);
  }
  if (!want || g_monitor_started.load(std::memory_order_acquire))
    return;
  bool expect = false;
  if (!g_monitor_started.compare_exchange_strong(expect, true))
    return;
  std::lock_guard<std::mutex> lk(g_threads_mu);
  host_threads_atexit_once();
  {
    std::lock_guard<std::mutex> lm(g_monitor_mu);
    g_monitor_stop = false;
  }
  g_monitor_thread = new std::thread(placement_monitor_loop);
}
// (g_threads_mu held)
static void placement_monitor_stop()
{
  if (!g_monitor_thread)
    return;
  {
    std::lock_guard<std::mutex> lm(g_monitor_mu);
    g_monitor_stop = true;
  }
  g_monitor_cv.notify_all();
  g_monitor_thread->join();
  delete g_monitor_thread;
  g_monitor_thread = nullptr;
  g_monitor_started.store(false, std::memory_order_release);
}
void placement_monitor_enable(int on)
{
  g_monitor_wanted.store(on ? 1 : 0, std::memory_order_release);
  if (!on)
  {
    std::lock_guard<std::mutex> lk(g_threads_mu);
    placement_monitor_stop();
  }
}
int placement_monitor_running() { return g_monitor_started.load(std::memory_order_acquire) ? 1 : 0; }

int placement_helper_cpus(int *cpus, int n)
{
  int k = 0;
  for (int idx = 0; idx < 3 && k < n; ++idx)
    if (CholHelper *h = chol_helper(idx))
      cpus[k++] = h->cpu.load(std::memory_order_acquire);
  return k;
}
int placement_monitor_moves() { return g_monitor_moves.load(std::memory_order_relaxed); }

// mode 0: everything on the caller's L3 domain A -- A[0] second half, A[1] / A[2] look-ahead stages, pool from A[3] on;
// mode 1: no look-ahead stages -- A[0] second half, pool from A[1] on;
// mode 2 (two domains): A[0] look-ahead of the first half, pool workers for the first half's chains from A[1] on;
//         B[0] second half, B[1] its look-ahead stage, pool workers for the second half's chains from B[2] on
static void place_helper(CholHelper *h, int cpu, int idx, int mode, const std::vector<int> &B)
{
  if (cpu < 0 || (cpu == h->near_cpu.load(std::memory_order_acquire) && mode == h->near_mode.load(std::memory_order_acquire)))
    return;
  h->near_cpu.store(cpu, std::memory_order_release);
  h->near_mode.store(mode, std::memory_order_release);
  const std::vector<int> &A = ccx_cores_of(cpu, false);
  int core = -1;
  if (mode == 2)
    core = idx == 0 ? (B.size() > 0 ? B[0] : -1) : idx == 1 ? (A.size() > 0 ? A[0] : -1) : (B.size() > 1 ? B[1] : -1);
  else if ((int)A.size() > idx)
    core = A[idx];
  if (core >= 0)
  {
    std::lock_guard<std::mutex> pin_lk(g_pin_mu); // (the placement monitor re-pins under the same lock: cpu and affinity stay in step)
    pin_one(h->tid, core);
    h->cpu.store(core, std::memory_order_release);
  }
  placement_monitor_start(); // (opt-in; takes g_threads_mu -- not under g_pin_mu, which the monitor itself takes)
}

static void place_pool(SepPool *q, int cpu, int mode, const std::vector<int> &B, int chains_per_half)
{
  if (cpu < 0 || (cpu == q->near_cpu.load(std::memory_order_acquire) && mode == q->near_mode.load(std::memory_order_acquire)))
    return;
  q->near_cpu.store(cpu, std::memory_order_release);
  q->near_mode.store(mode, std::memory_order_release);
  const std::vector<int> &cores = ccx_cores_of(cpu, true); // domain A first, then the rest of the NUMA node
  const size_t n_ccx = ccx_cores_of(cpu, false).size();
  std::lock_guard<std::mutex> pin_lk(g_pin_mu);
  q->dom.assign(q->tids.size(), -1);
  int n0 = 0, n1 = 0;
  if (mode == 2)
  {
    size_t t = 0;
    for (size_t c = 1; c < n_ccx && (int)c <= chains_per_half && t < q->tids.size(); ++c, ++t, ++n0)
    {
      pin_one(q->tids[t], cores[c]);
      q->cpu[t].store(cores[c], std::memory_order_release);
      q->dom[t] = 0;
    }
    for (size_t c = 2; c < B.size() && (int)c - 1 <= chains_per_half && t < q->tids.size(); ++c, ++t, ++n1)
    {
      pin_one(q->tids[t], B[c]);
      q->cpu[t].store(B[c], std::memory_order_release);
      q->dom[t] = 1;
    }
    // the rest: the remaining cores of domain A, then of the node (short tasks, pair products, back substitution)
    for (size_t c = 1 + (size_t)n0; t < q->tids.size() && c < cores.size(); ++c)
    {
      if (std::find(B.begin(), B.end(), cores[c]) != B.end())
        continue;
      q->cpu[t].store(cores[c], std::memory_order_release);
      pin_one(q->tids[t++], cores[c]);
    }
  }
  else
  {
    const size_t off = mode == 1 ? 1 : 3;
    for (size_t t = 0; t < q->tids.size() && t + off < cores.size(); ++t)
    {
      pin_one(q->tids[t], cores[t + off]);
      q->cpu[t].store(cores[t + off], std::memory_order_release);
      q->dom[t] = t + off < n_ccx ? 0 : -1;
      n0 += q->dom[t] == 0 ? 1 : 0;
    }
  }
  g_domain_threads[0].store(n0, std::memory_order_release);
  g_domain_threads[1].store(n1, std::memory_order_release);
  g_two_domains.store(mode == 2, std::memory_order_release);
}

bool block_chol_arm(bool with_pool, int long_arrow_chains)
{
  static const bool no_la_env = sage::env_flag("SAGE_SOLVE_NO_LOOKAHEAD");
  const int cpu = sched_getcpu();
  // r05: a loop-closure plan's long arrow-row chains run ~30 % slower on another L3 domain than their half and the separator
  // then waits for them (config 5: six chains, four cores left next to the two halves and their look-ahead stages: 1.1-1.4 ms).
  // The two halves do not reference each other: when the chains do not fit domain A, the SECOND half moves to a domain B of
  // its own with its look-ahead stage and its chains (mode 2) -- every chain then sits next to the half it follows.  Without
  // a second domain the look-ahead stages' cores go to the chains when that makes them fit (mode 1).
  int mode = no_la_env ? 1 : 0;
  std::vector<int> B;
  const int per_half = (long_arrow_chains + 1) / 2;
  if (mode == 0 && with_pool && long_arrow_chains > 0 && !sage::env_flag("SAGE_SOLVE_KEEP_LOOKAHEAD"))
  {
    const int n_ccx = (int)ccx_cores_of(cpu, false).size(); // cores of domain A without the caller's
    if (long_arrow_chains > n_ccx - 3)
    {
      if (!sage::env_flag("SAGE_SOLVE_ONE_DOMAIN"))
        B = second_domain_cores(cpu, (size_t)(2 + per_half));
      if (!B.empty() && n_ccx >= 1 + per_half)
        mode = 2;
      else if (long_arrow_chains <= n_ccx - 1)
        mode = 1;
    }
  }
  const bool no_la = mode == 1;
  for (int idx = 0; idx < (no_la ? 1 : 3); ++idx)
  {
    CholHelper *h = chol_helper(idx);
    if (h && !h->running.load(std::memory_order_acquire))
    {
      std::lock_guard<std::mutex> lk(g_threads_mu);
      chol_helper_start(h);
    }
    if (h && !h->armed.load(std::memory_order_acquire))
    {
      place_helper(h, cpu, idx, mode, B);
      {
        std::lock_guard<std::mutex> lk(h->mu);
        h->armed.store(true, std::memory_order_release);
      }
      h->cv.notify_one();
    }
  }
  placement_monitor_start(); // (opt-in: two relaxed loads when it is off or already running)
  if (!with_pool)
    return no_la;
  SepPool *q = sep_pool();
  if (q && !q->running.load(std::memory_order_acquire))
  {
    std::lock_guard<std::mutex> lk(g_threads_mu);
    sep_pool_start(q);
  }
  if (q && !q->armed.load(std::memory_order_acquire))
  {
    place_pool(q, cpu, mode, B, per_half);
    {
      std::lock_guard<std::mutex> lk(q->mu);
      q->armed.store(true, std::memory_order_release);
    }
    q->cv.notify_all();
  }
  return no_la;
}

// Stop and join every host thread this library started (helpers, arrow-row pool, placement monitor).  Safe to call at any
// time no solve is in flight; the next block_chol_arm() starts them again.  Called by sage_shutdown(), by the last
// sage_window_destroy and at process exit.
void host_threads_shutdown()
{
  std::lock_guard<std::mutex> lk(g_threads_mu);
  placement_monitor_stop();
  for (int idx = 0; idx < 3; ++idx)
  {
    CholHelper *h = chol_helper(idx);
    if (!h || !h->running.load(std::memory_order_acquire))
      continue;
    {
      std::lock_guard<std::mutex> lh(h->mu);
      h->quit.store(true, std::memory_order_release);
    }
    h->cv.notify_all();
    h->th.join();
    h->armed.store(false, std::memory_order_release);
    h->cpu.store(-1, std::memory_order_release);
    h->running.store(false, std::memory_order_release);
  }
  if (SepPool *q = g_sep_pool_made.load(std::memory_order_acquire))
    if (q->running.load(std::memory_order_acquire))
    {
      {
        std::lock_guard<std::mutex> lq(q->mu);
        q->quit.store(true, std::memory_order_release);
      }
      q->cv.notify_all();
      for (std::thread &t : q->ths)
        t.join();
      q->ths.clear();
      q->tids.clear();
      q->armed.store(false, std::memory_order_release);
      q->running.store(false, std::memory_order_release);
    }
}
int host_threads_running()
{
  int n = placement_monitor_running();
  for (int idx = 0; idx < 3; ++idx)
    if (CholHelper *h = chol_helper(idx))
      n += h->running.load(std::memory_order_acquire) ? 1 : 0;
  if (SepPool *q = g_sep_pool_made.load(std::memory_order_acquire))
    n += q->running.load(std::memory_order_acquire) ? q->n_workers : 0;
  return n;
}
static void host_threads_atexit_once()
{
  static std::once_flag once;
  std::call_once(once, [] { atexit(host_threads_shutdown); });
}

// rows of the separator part whose ranges run far along a half (> 16 columns): the chains the pool's fast threads carry
int block_plan_long_arrow_chains(const BlockEnvelope &E)
{
  if (E.n1 <= 0 || E.n2 <= 0 || !E.a_cnt)
    return 0;
  const int sep0 = E.n1 + E.n2;
  int n = 0;
  for (int i = sep0; i < E.K; ++i)
  {
    if (E.a_cnt[i] > 16)
      ++n;
    if (sep0 - std::min((int)E.row_first[i], sep0) > 16)
      ++n;
  }
  return n;
}

bool block_plan_has_arrow_rows(const BlockEnvelope &E)
{
  if (E.n1 <= 0 || E.n2 <= 0 || !E.a_cnt)
    return false;
  const int sep0 = E.n1 + E.n2;
  long long reach = 0;
  for (int i = sep0; i < E.K; ++i)
    reach += E.a_cnt[i] + (sep0 - std::min((int)E.row_first[i], sep0));
  return reach > 64; // (a plain split window: <= 2 x 3 blocks per separator row)
}

int plan_blocks(int K, const std::vector<std::pair<int, int>> &links, bool allow_split, BlockPlan &out)
{
  std::vector<int32_t> &perm = out.perm, &pos = out.pos, &row_first = out.row_first, &row_off = out.row_off,
                       &a_first = out.a_first, &a_cnt = out.a_cnt, &a_off = out.a_off, &blk_row = out.blk_row,
                       &blk_col = out.blk_col, &blk_src = out.blk_src;
  int &n1 = out.n1, &n2 = out.n2, &nblk = out.nblk;
  // ---- elimination order.  A chain-like window (every keyframe linked to a few predecessors) splits at a separator of
  // w consecutive keyframes into two halves without a link between them: order = [first half ascending | second half
  // DESCENDING | separator].  The two halves are then two independent banded factorisations (the host runs them on two
  // cores), only the w separator rows see both.  Windows with long-range links (loop closures) keep the identity order.
  perm.assign(K, 0);
  pos.assign(K, 0);
  for (int k = 0; k < K; ++k)
    perm[k] = k;
  n1 = n2 = 0;
  for (auto &l : links)
    if (l.first < 0 || l.second <= l.first || l.second >= K)
      return SAGE_E_INVALID;
  if (allow_split && K >= 16)
  {
    // the helper's half runs a little slower than the caller's (it wakes from sleep for every solve): give it
    // `bias` rows less
    constexpr int bias = 1;
    // Loop closures: a link that spans more than a separator's width crosses every candidate split.  Such links are
    // covered by a small set C of keyframes (greedy: the keyframe on most still-uncovered long links first) that joins
    // the separator: order = [first half | second half descending | middle separator | C].  The rows of C are "arrow"
    // rows -- their ranges run the whole length of both halves -- but the two halves stay two independent banded
    // factorisations, and the arrow rows are independent of each other until the (small) separator block: the host
    // factorisation spreads them over a few cores (block_chol_solve_tr).  At most 8 cover keyframes; otherwise, and
    // for windows without a split point, the identity order stays.
    constexpr bool no_cover = false;
    std::vector<char> inC(K, 0);
    std::vector<int> cover;
    int best_m = -1, best_w = 0;
    for (;;)
    {
      int best_cost = 2 * K;
      best_m = -1;
      for (int m = K / 4; m <= (3 * K) / 4; ++m)
      {
        int wdt = 0;
        for (auto &l : links)
          if (!inC[l.first] && !inC[l.second] && l.first < m && l.second >= m)
            wdt = std::max(wdt, l.second - m + 1);
        if (wdt < 1 || wdt > 8 || m + wdt > K - 2)
          continue;
        const int cost = std::max(m, K - m - wdt + bias) + 2 * wdt;
        if (cost < best_cost)
        {
          best_cost = cost;
          best_m = m;
          best_w = wdt;
        }
      }
      if (best_m > 0 || no_cover || cover.size() >= 8)
        break;
      // no split point: cover one more long link (span > 8: it cannot sit inside a separator)
      std::vector<int> deg(K, 0);
      int any = 0;
      for (auto &l : links)
        if (!inC[l.first] && !inC[l.second] && l.second - l.first > 8)
        {
          ++deg[l.first];
          ++deg[l.second];
          ++any;
        }
      if (!any)
        break;
      const int pick = (int)(std::max_element(deg.begin(), deg.end()) - deg.begin()); // (first maximum: lowest keyframe)
      inC[pick] = 1;
      cover.push_back(pick);
    }
    if (best_m > 0)
    {
      int q = 0;
      for (int k = 0; k < best_m; ++k)
        if (!inC[k])
          perm[q++] = k;
      n1 = q;
      for (int k = K - 1; k >= best_m + best_w; --k)
        if (!inC[k])
          perm[q++] = k;
      n2 = q - n1;
      for (int k = best_m; k < best_m + best_w; ++k)
        if (!inC[k])
          perm[q++] = k;
      std::sort(cover.begin(), cover.end());
      for (int k : cover)
        perm[q++] = k;
      if (n1 < 1 || n2 < 1) // (degenerate: everything on one side) -> identity order
      {
        for (int k = 0; k < K; ++k)
          perm[k] = k;
        n1 = n2 = 0;
      }
    }
  }
  for (int q = 0; q < K; ++q)
    pos[perm[q]] = q;
  // block storage: row i keeps the envelope range B = [row_first[i], i]; a separator row additionally keeps a range
  // A = [a_first[i], n1) over the tail of the first half (the columns in between -- the whole second half up to its
  // own tail -- are structurally zero in the factor and are neither stored nor visited).
  const bool split = n1 > 0;
  const int sep0 = n1 + n2;
  row_first.assign(K, 0);
  row_off.assign(K, 0);
  a_first.assign(K, 0);
  a_cnt.assign(K, 0);
  a_off.assign(K, 0);
  for (int k = 0; k < K; ++k)
  {
    row_first[k] = k;
    a_first[k] = n1;
  }
  for (auto &l : links)
  {
    const int i = std::max(pos[l.first], pos[l.second]), j = std::min(pos[l.first], pos[l.second]);
    if (split && i >= sep0 && j < n1)
      a_first[i] = std::min(a_first[i], j);
    else
      row_first[i] = std::min(row_first[i], j);
  }
  if (split)
    for (int i = sep0; i < K; ++i)
    {
      row_first[i] = std::min(row_first[i], (int32_t)sep0); // separator rows couple through both halves
      a_cnt[i] = n1 - a_first[i];
    }
  nblk = 0;
  for (int k = 0; k < K; ++k)
  {
    a_off[k] = nblk;
    nblk += a_cnt[k];
    row_off[k] = nblk;
    nblk += k - row_first[k] + 1;
  }
  auto bidx = [&](int i, int j) {
    return j < row_first[i] ? a_off[i] + j - a_first[i] : row_off[i] + j - row_first[i];
  };
  blk_row.assign(nblk, 0);
  blk_col.assign(nblk, 0);
  blk_src.assign(nblk, -1);
  for (int k = 0; k < K; ++k)
  {
    for (int j = a_first[k]; j < a_first[k] + a_cnt[k]; ++j)
    {
      blk_row[bidx(k, j)] = k;
      blk_col[bidx(k, j)] = j;
    }
    for (int j = row_first[k]; j <= k; ++j)
    {
      blk_row[bidx(k, j)] = k;
      blk_col[bidx(k, j)] = j;
    }
  }
  // column lists (rows below the diagonal that store a block of the column, ascending)
  out.col_ptr.assign(K + 1, 0);
  for (int b = 0; b < nblk; ++b)
    if (blk_row[b] != blk_col[b])
      ++out.col_ptr[blk_col[b] + 1];
  for (int j = 0; j < K; ++j)
    out.col_ptr[j + 1] += out.col_ptr[j];
  out.col_rows.assign(std::max(1, (int)out.col_ptr[K]), 0);
  {
    std::vector<int32_t> fill(out.col_ptr.begin(), out.col_ptr.end() - 1);
    for (int k = 0; k < K; ++k) // rows ascending -> every column's list comes out ascending
    {
      for (int j = a_first[k]; j < a_first[k] + a_cnt[k]; ++j)
        out.col_rows[fill[j]++] = k;
      for (int j = row_first[k]; j < k; ++j)
        out.col_rows[fill[j]++] = k;
    }
  }
  for (size_t l = 0; l < links.size(); ++l)
  {
    const int a = links[l].first, b = links[l].second;
    const int i = std::max(pos[a], pos[b]), j = std::min(pos[a], pos[b]);
    int &src = blk_src[bidx(i, j)];
    if (src >= 0)
      return SAGE_E_UNSUPPORTED; // duplicate link: the host path accumulates, this one does not
    // the packed link block is H[a rows][b cols]; block (i,j) is H[perm[i] rows][perm[j] cols]
    src = (int)l | (perm[i] == a ? 0x40000000 : 0);
  }
  return SAGE_OK;
}

int block_chol_partial(const BlockEnvelope &E, double *T, double *X, double *y, int nI)
{
  if ((E.Bp != 40 && E.Bp != 24) || E.a_cnt)
    return -1;
  const int rc = block_chol_range(E, T, X, y, 0, 0, nI); // interior rows: factor + forward substitution
  if (rc)
    return rc;
  if (E.Bp == 40)
    block_schur_40(E, T, X, y, nI);
  else
    block_schur_24(E, T, X, y, nI);
  return 0;
}

int block_chol_partial_back(const BlockEnvelope &E, double *T, double *X, double *y, int nI)
{
  if ((E.Bp != 40 && E.Bp != 24) || E.a_cnt)
    return -1;
  return block_chol_range(E, T, X, y, 1, 0, nI); // x_i for the interior rows, y[nI..K) holding the separators' x
}

int block_chol_solve_tr(const BlockEnvelope &E0, double *T, double *X, double *y)
{
  if (E0.Bp != 40 && E0.Bp != 24)
    return -1;
  const int K = E0.K;
  if (E0.n1 <= 0 || E0.n2 <= 0)
  {
    const int rc = block_chol_range(E0, T, X, y, 0, 0, K);
    return rc ? rc : block_chol_range(E0, T, X, y, 1, 0, K);
  }
  const int sep0 = E0.n1 + E0.n2;
  // loop-closure plans: the long separator rows are cut into tasks for the worker pool (SepJob); the halves publish
  // their progress so that the tasks run right behind them
  BlockEnvelope E = E0;
  const bool arrow = block_plan_has_arrow_rows(E0);
  SepJob job;
  SepPool *pool = nullptr;
  bool pool_mine = false;
  if (arrow)
  {
    job.E = &E; job.T = T; job.X = X; job.y = y;
    sep_job_build(job);
    job.t_start = mono_seconds();
    E.progress = job.progress;
    pool = sep_pool();
    if (pool && pool->armed.load(std::memory_order_acquire))
    {
      bool expect = false;
      if (pool->busy.compare_exchange_strong(expect, true, std::memory_order_acq_rel))
      {
        pool_mine = true;
        pool->job.store(&job, std::memory_order_release);
        pool->open.store(true, std::memory_order_seq_cst);
        pool->posted.fetch_add(1, std::memory_order_release);
      }
    }
  }
  BlockEnvelope::RowPipe pipes[2];
  E.pipe = pipes;
  static const bool no_sep_pre = sage::env_flag("SAGE_SOLVE_NO_SEP_PRE");
  E.sep_pre = !arrow && !no_sep_pre && E.n1 > 0 && E.n2 > 0 && !E.before_row;
  CholHelper *h = chol_helper();
  bool shared = false;
  if (h && h->armed.load(std::memory_order_acquire))
  {
    bool expect = false;
    if (h->busy.compare_exchange_strong(expect, true, std::memory_order_acq_rel))
    {
      shared = true;
      h->E = &E; h->T = T; h->X = X; h->y = y;
      h->p1_rc.store(-2, std::memory_order_relaxed);
      h->go_p2.store(0, std::memory_order_relaxed);
      h->p2_done.store(0, std::memory_order_relaxed);
      h->claim.store(0, std::memory_order_release);
      h->posted.fetch_add(1, std::memory_order_release);
    }
  }
  // the first half as two stages when its look-ahead helper answers (the second half's thread asks for its own)
  const bool piped = lookahead_engage(1, E, T, X, y, 0, E.n1);
  static const bool dbg = sage::env_flag("SAGE_DEBUG_TIMING");
  double tp[6] = {0, 0, 0, 0, 0, 0};
  if (dbg)
    tp[0] = mono_seconds();
  int rc = block_chol_range(E, T, X, y, 0, 0, E.n1, piped ? 1 : 0);
  if (piped)
    lookahead_release(1);
  if (dbg)
    tp[1] = mono_seconds();
  bool helper_has_it = false;
  if (shared)
  {
    int expect = 0;
    helper_has_it = !h->claim.compare_exchange_strong(expect, 2, std::memory_order_acq_rel);
  }
  int rc2;
  if (helper_has_it)
  {
    while ((rc2 = h->p1_rc.load(std::memory_order_acquire)) == -2)
      CholHelper::cpu_relax();
  }
  else
    rc2 = rc == 0 ? block_chol_range(E, T, X, y, 0, E.n1, sep0) : 0;
  if (dbg)
    tp[2] = mono_seconds();
  if (rc == 0)
    rc = rc2;
  if (arrow)
  {
    if (rc != 0)
      job.abort.store(1, std::memory_order_release);
    sep_work(job); // the caller takes tasks too (all of them when no pool thread is around)
    while (job.doneB.load(std::memory_order_acquire) < (int)job.tb.size())
      CholHelper::cpu_relax();
    if (dbg)
      fprintf(stderr, "[sage block chol] arrow tasks (%zu row chains, %zu pair products) done %.0f us after the halves\n",
              job.ta.size(), job.tb.size(), 1e6 * (mono_seconds() - tp[2]));
    if (rc == 0 && job.abort.load(std::memory_order_acquire))
      rc = -2;
    if (rc == 0)
      rc = E.Bp == 40 ? sep_finish_40(job) : sep_finish_24(job);
  }
  else if (rc == 0)
    rc = block_chol_range(E, T, X, y, 0, sep0, K);
  if (rc == 0)
    block_chol_range(E, T, X, y, 1, sep0, K);
  if (arrow)
  {
    // the separator rows' share of every half row's back substitution, in row chunks on the pool (the arrow rows put three
    // more blocks into every column: streaming them once, in parallel, instead of inside the two sequential sweeps)
    const bool par_c = rc == 0 && !job.tc.empty();
    job.goC.store(par_c ? 1 : 2, std::memory_order_release);
    if (par_c)
    {
      sep_work_c(job, false);
      while (job.doneC.load(std::memory_order_acquire) < (int)job.tc.size())
        CholHelper::cpu_relax();
      E.bs_skip_from = sep0;
    }
    if (pool_mine)
    {
      pool->open.store(false, std::memory_order_seq_cst);
      while (pool->active.load(std::memory_order_seq_cst) != 0)
        CholHelper::cpu_relax();
      pool->armed.store(false, std::memory_order_release);
      pool->busy.store(false, std::memory_order_release);
    }
  }
  if (dbg)
    tp[3] = mono_seconds();
  if (helper_has_it)
    h->go_p2.store(rc == 0 ? 1 : 2, std::memory_order_release);
  if (rc == 0)
  {
    block_chol_range(E, T, X, y, 1, 0, E.n1);
    if (!helper_has_it)
      block_chol_range(E, T, X, y, 1, E.n1, sep0);
  }
  if (dbg)
    tp[4] = mono_seconds();
  if (helper_has_it)
    while (!h->p2_done.load(std::memory_order_acquire))
      CholHelper::cpu_relax();
  if (dbg)
  {
    tp[5] = mono_seconds();
    fprintf(stderr, "[sage block chol] us: first half %.0f (+wait for the %s %.0f) separator %.0f%s back-subst %.0f (+wait %.0f)\n",
            1e6 * (tp[1] - tp[0]), helper_has_it ? "helper" : "second half, same thread", 1e6 * (tp[2] - tp[1]),
            1e6 * (tp[3] - tp[2]), arrow ? (pool_mine ? " (arrow rows, worker pool)" : " (arrow rows, no pool)") : "",
            1e6 * (tp[4] - tp[3]), 1e6 * (tp[5] - tp[4]));
  }
  if (shared)
  {
    h->armed.store(false, std::memory_order_release); // the helper goes back to sleep until the next arm
    h->busy.store(false, std::memory_order_release);
  }
  return rc;
}
} // namespace sage

extern "C" long long sage_solve_lookahead_count(void) { return sage::g_lookahead_count.load(); }

extern "C" int sage_block_solve(const double *packed, int K, int nlinks, const int32_t *links, int B, double damp,
                                const double *diag_add, const double *g_add, double *delta)
{
  if (!packed || K < 1 || B < 1 || nlinks < 0 || (nlinks > 0 && !links) || !delta)
    return SAGE_E_INVALID;
  // the factorisation works on blocks padded to a multiple of 8 rows (identity on the padding) so that every
  // GEMM inner length is a whole number of 8-double vectors
  const int Bp = (B + 7) / 8 * 8;
  const int BB = B * B, n = K * Bp, n_out = K * B;
  const double *diag = packed;
  const double *lnk = diag + (size_t)K * BB;
  const double *g = lnk + (size_t)nlinks * BB;
  // envelope: first non-zero block column of each block row
  std::vector<int> first_blk(K);
  for (int k = 0; k < K; ++k)
    first_blk[k] = k;
  for (int l = 0; l < nlinks; ++l)
  {
    const int a = links[2 * l], b = links[2 * l + 1];
    if (a < 0 || b <= a || b >= K)
      return SAGE_E_INVALID;
    first_blk[b] = std::min(first_blk[b], a);
  }
  if (Bp == 40 || Bp == 24)
  {
    // fixed-size transposed-block path (the one the window engine runs on the device-scattered storage), with the
    // same elimination order and two-core split
    const int BBp = Bp * Bp;
    std::vector<std::pair<int, int>> lk(nlinks);
    for (int l = 0; l < nlinks; ++l)
      lk[l] = {links[2 * l], links[2 * l + 1]};
    sage::BlockPlan bp;
    {
      const int rcp = sage::plan_blocks(K, lk, !sage::env_flag("SAGE_SOLVE_NO_SPLIT"), bp);
      if (rcp == SAGE_E_UNSUPPORTED) // duplicate links accumulate on this path: plan without them
      {
        std::sort(lk.begin(), lk.end());
        lk.erase(std::unique(lk.begin(), lk.end()), lk.end());
        const int rcq = sage::plan_blocks(K, lk, !sage::env_flag("SAGE_SOLVE_NO_SPLIT"), bp);
        if (rcq != SAGE_OK)
          return rcq;
      }
      else if (rcp != SAGE_OK)
        return rcp;
    }
    const int nblk = bp.nblk;
    auto bidx = [&](int i, int j) {
      return j < bp.row_first[i] ? bp.a_off[i] + j - bp.a_first[i] : bp.row_off[i] + j - bp.row_first[i];
    };
    std::vector<double> T((size_t)nblk * BBp, 0.0), X((size_t)K * BBp), y((size_t)K * Bp, 0.0);
    for (int q = 0; q < K; ++q)
    {
      const int k = bp.perm[q];
      double *D = T.data() + (size_t)bidx(q, q) * BBp;
      for (int i = 0; i < Bp; ++i)
        for (int j = 0; j < Bp; ++j)
        {
          double v = 0.0;
          if (i < B && j < B)
          {
            v = 0.5 * (diag[(size_t)k * BB + i * B + j] + diag[(size_t)k * BB + j * B + i]);
            if (i == j)
              v = (v + (diag_add ? diag_add[k * B + i] : 0.0)) * (1.0 + damp);
          }
          else if (i == j)
            v = 1.0 + damp;
          D[i * Bp + j] = v;
        }
      for (int i = 0; i < B; ++i)
        y[(size_t)q * Bp + i] = g[(size_t)k * B + i] + (g_add ? g_add[k * B + i] : 0.0);
    }
    for (int l = 0; l < nlinks; ++l)
    {
      const int a = links[2 * l], b = links[2 * l + 1];
      const int qi = std::max(bp.pos[a], bp.pos[b]), qj = std::min(bp.pos[a], bp.pos[b]);
      // stored block is [c in column keyframe][r in row keyframe]; the packed link block is [r in a][c in b]
      double *D = T.data() + (size_t)bidx(qi, qj) * BBp;
      const bool row_is_a = bp.perm[qi] == a;
      for (int i = 0; i < B; ++i)
        for (int j = 0; j < B; ++j)
          D[row_is_a ? j * Bp + i : i * Bp + j] += lnk[(size_t)l * BB + i * B + j];
    }
    static const bool dbg2 = sage::env_flag("SAGE_DEBUG_TIMING");
    const auto t0 = std::chrono::steady_clock::now();
    sage::BlockEnvelope env;
    env.K = K; env.Bp = Bp; env.row_first = bp.row_first.data(); env.row_off = bp.row_off.data();
    env.a_first = bp.a_first.data(); env.a_cnt = bp.a_cnt.data(); env.a_off = bp.a_off.data();
    env.n1 = bp.n1; env.n2 = bp.n2;
    env.col_ptr = bp.col_ptr.data(); env.col_rows = bp.col_rows.data();
    if (bp.n1 > 0)
      env.no_lookahead = sage::block_chol_arm(sage::block_plan_has_arrow_rows(env), sage::block_plan_long_arrow_chains(env));
    const int rcf = sage::block_chol_solve_tr(env, T.data(), X.data(), y.data());
    if (dbg2)
      fprintf(stderr, "[sage block_solve] fixed-block Cholesky + substitution %.3f ms\n",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (rcf != 0)
      return SAGE_E_NOT_PSD;
    for (int k = 0; k < K; ++k)
      std::memcpy(delta + (size_t)k * B, y.data() + (size_t)bp.pos[k] * Bp, sizeof(double) * B);
    return SAGE_OK;
  }
  std::vector<int> first(n);
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < Bp; ++i)
      first[k * Bp + i] = first_blk[k] * Bp;
  static const bool dbg = sage::env_flag("SAGE_DEBUG_TIMING");
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_a = tnow();
  sage::EnvelopeMatrix M;
  M.init(n, first);
  std::vector<double> rhs(n, 0.0);
  for (int k = 0; k < K; ++k)
  {
    for (int i = 0; i < B; ++i)
    {
      for (int j = 0; j <= i; ++j)
        M.at(k * Bp + i, k * Bp + j) = 0.5 * (diag[(size_t)k * BB + i * B + j] + diag[(size_t)k * BB + j * B + i]);
      rhs[k * Bp + i] = g[(size_t)k * B + i] + (g_add ? g_add[k * B + i] : 0.0);
      if (diag_add)
        M.at(k * Bp + i, k * Bp + i) += diag_add[k * B + i];
    }
    for (int i = B; i < Bp; ++i)
      M.at(k * Bp + i, k * Bp + i) = 1.0; // padding rows: identity, rhs 0 -> delta 0
  }
  for (int l = 0; l < nlinks; ++l)
  {
    const int a = links[2 * l], b = links[2 * l + 1]; // block (a,b) -> lower-triangle rows of b
    for (int i = 0; i < B; ++i)
      for (int j = 0; j < B; ++j)
        M.at(b * Bp + j, a * Bp + i) += lnk[(size_t)l * BB + i * B + j];
  }
  for (int r = 0; r < n; ++r) // LM damping H + damp*diag(H) (camera_tracker.cpp:1182)
    M.at(r, r) *= (1.0 + damp);
  static const int n_threads = getenv("SAGE_SOLVE_THREADS") ? std::max(1, atoi(getenv("SAGE_SOLVE_THREADS")))
                                                            : 1; // EPYC 9575F: the thread split loses at n ~ 2.5k
  auto t_b = tnow();
  if (!M.cholesky_inplace(Bp, n_threads))
    return SAGE_E_NOT_PSD;
  auto t_c = tnow();
  M.solve_inplace(rhs);
  if (dbg)
  {
    auto t_d = tnow();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[sage block_solve] n %d threads %d: assemble %.3f cholesky %.3f substitution %.3f ms\n", n,
            n_threads, ms(t_a, t_b), ms(t_b, t_c), ms(t_c, t_d));
  }
  for (int k = 0; k < K; ++k)
    std::memcpy(delta + (size_t)k * B, rhs.data() + (size_t)k * Bp, sizeof(double) * B);
  (void)n_out;
  return SAGE_OK;
}
