// host_math.h -- internal host-side dense helpers (double precision).
#pragma once
#include <stdint.h>

#include <vector>

namespace sage
{
void sym_eig(std::vector<double> &A, int n, std::vector<double> &w, std::vector<double> &V);
void rotation_to_angle_axis_as_reference(const float *R, float eps, float *out);

// Envelope (skyline) Cholesky of a symmetric positive definite matrix given by its lower triangle:
// row r stores columns first[r]..r contiguously at data[rowptr[r]...].
struct EnvelopeMatrix
{
  int n = 0;
  std::vector<int> first;      // first stored column of each row
  std::vector<size_t> rowptr;  // offset of column first[r] in data
  std::vector<double> data;
  void init(int n_, const std::vector<int> &first_);
  inline double &at(int r, int c) { return data[rowptr[r] + (size_t)(c - first[r])]; } // first[r] <= c <= r
  // returns false if not positive definite.  block > 1: rows come in aligned groups of `block` rows that share
  // `first` (the window's keyframe blocks) -> the rows of a group are factorised in parallel on `threads` threads.
  bool cholesky_inplace(int block = 1, int threads = 1);
  void solve_inplace(std::vector<double> &b) const; // after cholesky_inplace: b <- A^-1 b
};

// Block-envelope Cholesky solve on the storage the device scatter kernel produces (solve_kernels.hip): block (i,j),
// row_first[i] <= j <= i, lives at T + (row_off[i] + j - row_first[i]) * Bp*Bp and holds the TRANSPOSED block
// ([c][r] = A[i*Bp + r][j*Bp + c]); Bp is 40 or 24.  In place: T becomes L^T blockwise, X (K*Bp*Bp) receives the
// inverses of the diagonal factors, y (K*Bp) the right-hand side on entry and the solution on return.
// Returns 0, or 1 + the block column of the first non-positive pivot.
int block_chol_solve_tr(int K, int Bp, const int32_t *row_first, const int32_t *row_off, double *T, double *X,
                        double *y);
} // namespace sage
