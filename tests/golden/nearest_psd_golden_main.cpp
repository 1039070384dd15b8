// Driver of tests/golden/make_nearest_psd_golden.py: calls the REFERENCE's own NearestPsd (core/mapping/mapping_utils.h:88-128,
// extracted from the reference tree into ref_nearest_psd.inc at generation time -- not stored in this repository) with
// the vendored Eigen 3.3.9 on seeded matrices and prints inputs + outputs.  Build container only.
#include <Eigen/Dense>
#include <cstdio>
#include <cstdint>
#include <vector>

namespace df
{
#include "ref_nearest_psd.inc"
}

static uint64_t lcg_state = 88172645463325252ull;
static double urand() // xorshift64*, uniform in (-1, 1)
{
  lcg_state ^= lcg_state >> 12;
  lcg_state ^= lcg_state << 25;
  lcg_state ^= lcg_state >> 27;
  const uint64_t r = lcg_state * 2685821657736338717ull;
  return ((double)(r >> 11) / 9007199254740992.0) * 2.0 - 1.0;
}

static void dump(const char *name, const Eigen::MatrixXd &M, const Eigen::MatrixXd &A)
{
  const int n = (int)M.rows();
  printf("{\"name\": \"%s\", \"n\": %d, \"M\": [", name, n);
  for (int i = 0; i < n * n; ++i)
    printf("%s%.17g", i ? "," : "", M(i / n, i % n));
  printf("], \"A\": [");
  for (int i = 0; i < n * n; ++i)
    printf("%s%.17g", i ? "," : "", A(i / n, i % n));
  printf("]}");
}

int main()
{
  printf("[");
  bool first = true;
  auto emit = [&](const char *name, const Eigen::MatrixXd &M) {
    if (!first)
      printf(",\n");
    first = false;
    dump(name, M, df::NearestPsd(M));
  };
  const int sizes[] = {4, 13, 29, 46};
  for (int n : sizes)
  {
    Eigen::MatrixXd J(3 * n, n), S(n, n);
    for (int i = 0; i < 3 * n; ++i)
      for (int j = 0; j < n; ++j)
        J(i, j) = urand();
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        S(i, j) = urand();
    char nm[64];
    snprintf(nm, sizeof nm, "pd_%d", n);
    emit(nm, J.transpose() * J);                              // symmetric positive definite (well separated spectrum)
    snprintf(nm, sizeof nm, "pd_asym_%d", n);
    emit(nm, J.transpose() * J + 1e-3 * S);                   // + a small asymmetric part (fp32 AtA is symmetric only to rounding)
    snprintf(nm, sizeof nm, "indef_%d", n);
    emit(nm, 0.5 * (S + S.transpose()));                      // symmetric indefinite
  }
  {
    // the shape of a photometric edge system (D = 13 + 16): pose block [[A,-A],[-A,A]] -> 6 exactly dependent columns
    const int n = 29;
    Eigen::MatrixXd J(60, 23);
    for (int i = 0; i < 60; ++i)
      for (int j = 0; j < 23; ++j)
        J(i, j) = urand();
    Eigen::MatrixXd Jf(60, n);
    Jf.leftCols(6) = J.leftCols(6);
    Jf.middleCols(6, 6) = -J.leftCols(6);
    Jf.rightCols(17) = J.rightCols(17);
    const Eigen::MatrixXd M = Jf.transpose() * Jf;
    emit("gauge_29", M);
    Eigen::MatrixXd P(n, n);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        P(i, j) = urand();
    emit("gauge_29_perturbed", M + 1e-15 * M.norm() * 0.5 * (P + P.transpose())); // same matrix to 1e-15: is the output stable?
  }
  printf("]\n");
  return 0;
}
