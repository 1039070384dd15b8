// host_math.h -- internal host-side dense helpers (double precision).
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include <utility>
#include <atomic>
#include <sched.h>
#include <vector>

namespace sage
{
// boolean environment switch ("Environment switches"): set and neither empty nor "0"
inline bool env_flag(const char *name)
{
  const char *e = getenv(name);
  return e && e[0] && !(e[0] == '0' && e[1] == 0);
}

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

// Block-envelope Cholesky solve on the storage the device scatter kernel produces (solve_kernels.hip).  Row i keeps
// the blocks of columns B = [row_first[i], i] at T + (row_off[i] + j - row_first[i]) * Bp*Bp and, optionally, a
// second range A = [a_first[i], a_first[i] + a_cnt[i]) (all < row_first[i]) at T + (a_off[i] + j - a_first[i]) * Bp*Bp;
// columns between the two ranges are structurally zero in the factor.  Every block holds the TRANSPOSED block
// ([c][r] = A[i*Bp + r][j*Bp + c]); Bp is 40 or 24.
// n1/n2 > 0 declare that rows [0,n1) and [n1,n1+n2) do not reference each other (two halves of a window split at a
// separator, solve_kernels.hip solver_create): they are factorised concurrently on two cores when a helper thread is
// available (block_chol_arm), otherwise one after the other.
struct BlockEnvelope
{
  int K = 0, Bp = 0;
  const int32_t *row_first = nullptr, *row_off = nullptr;
  const int32_t *a_first = nullptr, *a_cnt = nullptr, *a_off = nullptr; // may be null: no A ranges
  int n1 = 0, n2 = 0;
  // optional: ready[b] == epoch once block b of the storage (and, for a diagonal block, its rows of y) has been
  // delivered by the device; a row is only touched after all its blocks have arrived
  const volatile unsigned *ready = nullptr;
  unsigned epoch = 0;
  // optional, with `ready`: fill[b] != 0 marks a block that is structural fill-in (no link behind it, off the diagonal):
  // the device does not deliver it -- whoever touches it first zeroes it instead of waiting for a ticket (r05: the arrow
  // rows of a loop-closure plan are ~1500 such blocks, 19 MB of zeros that used to cross PCIe behind everything else while
  // the arrow-row tasks waited for them)
  const uint8_t *fill = nullptr;
  // set from block_chol_arm's return value: the halves run WITHOUT their look-ahead stages, whose two cores carry arrow-row
  // chains instead (loop-closure plans with more long chains than the halves' L3 domain has cores left: r05)
  bool no_lookahead = false;
  // optional: called before row i is waited for / touched (the pipelined window solve launches the device work that
  // produces the next rows from here); a non-zero return aborts the factorisation with -2
  int (*before_row)(void *user, int row) = nullptr;
  void (*idle)(void *user) = nullptr; // optional: polled while a thread waits for tickets (e.g. to launch more work)
  void *user = nullptr;
  double *t_ticket_wait = nullptr; // optional: accumulates the seconds spent waiting for tickets (diagnostics)
  // optional (set by block_chol_solve_tr): progress[h] = 1 + the last factorised row of half h (0: rows [0, n1),
  // 1: rows [n1, n1 + n2)), published after the row's forward substitution -- the arrow-row tasks follow it
  std::atomic<int> *progress = nullptr;
  // optional: for every column j the rows i > j that store a block (i, j), ascending (col_rows[col_ptr[j] .. col_ptr[j+1]));
  // the back substitution then visits exactly those instead of scanning all rows below j
  const int32_t *col_ptr = nullptr, *col_rows = nullptr;
  // back substitution: rows m >= bs_skip_from are left out of  sum_m L_mi^T x_m  (their part has been subtracted from y
  // beforehand, in parallel: the arrow rows of a loop-closure plan)
  int bs_skip_from = 0x7fffffff;
  // optional (set by block_chol_solve_tr): two threads per half.  pipe[h] = {rows of half h whose EARLY part is done,
  // rows that are complete}: a look-ahead thread forms, for row i, everything that only needs the rows <= i-2 (all
  // blocks but (i, i-1) and their share of (i, i-1) / the diagonal), the half's own thread follows with the chain that
  // needs row i-1 -- same blocks, same order of the sums, so the factor is the same bit for bit.
  struct RowPipe
  {
    alignas(64) std::atomic<int> early{0};
    alignas(64) std::atomic<int> late{0};
    // r05: the separator rows' blocks against this half's columns, formed by the half's look-ahead thread right behind the
    // rows they depend on (sep_pre): 0 nobody does it (the separator pass forms them itself), 1 pending, 2 done, -1 given up
    alignas(64) std::atomic<int> pre{0};
  };
  RowPipe *pipe = nullptr;
  bool sep_pre = false; // plain split windows: the look-ahead threads pre-form the separator rows' half blocks
};
// In place: T becomes L^T blockwise, X (K*Bp*Bp) receives the inverses of the diagonal factors, y (K*Bp) the
// right-hand side on entry and the solution on return.  Returns 0, or 1 + the block column of the first non-positive
// pivot, -1 for an unsupported Bp, -2 when a block's ticket did not arrive within two seconds.
int block_chol_solve_tr(const BlockEnvelope &env, double *T, double *X, double *y);
// Partial factorisation for domain decomposition (shard_solve.cpp; storage as above, no A ranges): rows [0, nI) are
// factorised and forward-substituted; the separator rows [nI, K) receive L_ij for j < nI, their blocks (i, j >= nI) end
// as the Schur complement C_ij^T = (A_ij - sum_{k<nI} L_ik L_jk^T)^T and y_i as c_i = b_i - sum_{k<nI} L_ik y_k.
// block_chol_partial_back: x of the rows [0, nI) given x of the separators in y[nI..K).  Returns as block_chol_solve_tr.
int block_chol_partial(const BlockEnvelope &env, double *T, double *X, double *y, int nI);
int block_chol_partial_back(const BlockEnvelope &env, double *T, double *X, double *y, int nI);
// Wake the helper thread ahead of a block_chol_solve_tr call with n1 > 0 (it then spins for the job for a few
// milliseconds at most); call it when the system is about to be produced, e.g. before waiting on the D2H copy.
// with_pool: also wake the worker pool that shares the long separator ("arrow") rows of a loop-closure plan.
// Returns true when the solve should run its halves without look-ahead stages (BlockEnvelope::no_lookahead): a plan whose
// long arrow-row chains (block_plan_long_arrow_chains) do not fit the cores the look-ahead stages leave free in the caller's
// L3 domain but do fit with those two cores.
bool block_chol_arm(bool with_pool = false, int long_arrow_chains = 0);
// placement of the solve's threads (host_math.cpp "which CPUs the solve's threads may be placed on")
void placement_set_allowed(const cpu_set_t *allowed); // nullptr: back to the calling thread's affinity mask
std::vector<int> placement_busy_cpus(int ms);
std::vector<int> placement_core_siblings(int cpu);
std::vector<int> placement_l3_domain(int cpu);
int placement_helper_cpus(int *cpus, int n); // CPUs the solve's (up to three) helper threads are pinned to
int placement_monitor_moves();              // helpers moved off crowded cores so far (placement monitor)
// r06: the placement monitor is OPT-IN (SAGE_PLACEMENT_MONITOR=1 or placement_monitor_enable(1)); every thread the solve
// starts (helpers, arrow-row pool, monitor) is joinable: host_threads_shutdown() stops and joins them, the next
// block_chol_arm() starts them again.  host_threads_running() = how many are alive.
void placement_monitor_enable(int on);
int placement_monitor_running();
void host_threads_shutdown();
int host_threads_running();
int block_plan_long_arrow_chains(const BlockEnvelope &env);
// true when the separator rows of the plan reach far into the halves (cover keyframes of loop closures): the
// factorisation then wants the worker pool
bool block_plan_has_arrow_rows(const BlockEnvelope &env);

// Elimination order and block storage plan of a window's normal equations (K keyframe blocks, links (a,b), a < b).
// perm[position] = keyframe, pos[keyframe] = position.  Block b of the storage is (blk_row[b], blk_col[b]) in
// positions; blk_src[b] = link index, | 0x40000000 when the stored (transposed) block is the packed link block read
// row-major (row keyframe == a), or -1 for diagonal / fill-in blocks.
struct BlockPlan
{
  std::vector<int32_t> perm, pos, row_first, row_off, a_first, a_cnt, a_off, blk_row, blk_col, blk_src;
  std::vector<int32_t> col_ptr, col_rows; // BlockEnvelope::col_ptr / col_rows
  int nblk = 0, n1 = 0, n2 = 0;
};
int plan_blocks(int K, const std::vector<std::pair<int, int>> &links, bool allow_split, BlockPlan &out);
} // namespace sage
