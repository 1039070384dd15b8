// shard_solve.cpp -- domain-decomposed damped solve of a link-sharded window (multi-GPU).
//
// No reference counterpart (the reference is one GPU, ISAM2 on the host).  Link ranges are contiguous
// (sage_window_set_shard), so every rank's slice of the block-sparse normal equations is COMPLETE for the keyframes that
// only its own links touch ("interior" keyframes): it eliminates them locally and only the Schur complement on the
// keyframes shared with other ranks ("separators": ~3 per range boundary of a temporal window) crosses xGMI:
//
//   A = H + P + damp*diag(H + P)        H = sum_r H^(r) (per-rank edge sums), P = diagonal priors (owned by one rank each)
//   rank r:   C^(r) = A^(r)_SS - A_SI A_II^-1 A_IS ,  c^(r) = b^(r)_S - A_SI A_II^-1 b_I          (I = its interior)
//   all-reduce (sum, double) of the separator buffer [blocks of C | c | 8 tail doubles]
//   every rank: (sum_r C^(r)) d_S = sum_r c^(r)   (block-envelope Cholesky over the separators, identical on all ranks)
//   rank r:   d_I = A_II^-1 (b_I - A_IS d_S)
//
// Payload at K = 64, 8 ranks, B = 39: 96 blocks + rhs = 1.2 MB instead of the 3.0 MB packed system; per-rank
// factorisation: 8 interior keyframes + the 21-keyframe separator system instead of all 64 keyframes.
// The result equals the single-rank solve of the summed system to rounding (tests/test_shard_schur.py, world 2/4/8;
// tests/test_sharded_reduce_gloo.py over gloo).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "host_math.h"
#include "sage_ba.h"

struct SageShardPlan
{
  int K = 0, B = 0, rank = 0, world = 1, nlinks = 0;
  std::vector<std::pair<int, int>> links;
  std::vector<int> link_lo;        // [world + 1] link range of every rank
  std::vector<int> kf_owner;       // designated owner (lowest rank touching the keyframe; -1: untouched)
  std::vector<int> kf_ntouch;      // number of ranks touching the keyframe
  std::vector<int> sep_all;        // separator keyframes (touched by >= 2 ranks), ascending
  std::vector<int> sep_index;      // keyframe -> index in sep_all, or -1
  std::vector<int> interior;       // this rank's interior keyframes, ascending
  std::vector<int> sep_local;      // indices into sep_all of the separators this rank touches
  std::vector<int> local_links;    // links this rank owns
  // separator buffer = the packed layout of sage_block_solve over the separator keyframes: [diag blocks | blocks of the
  // coupled pairs i < j (rows i, cols j) | rhs | 8 tail doubles]; pair_block maps (i <= j) -> block index
  std::map<std::pair<int, int>, int> pair_block;
  std::vector<int32_t> sep_pairs;  // flat (i, j), i < j, in block order: the "links" of the separator system
  int n_pair_blocks = 0;
  size_t sep_doubles = 0;
  // state between eliminate() and solve()
  sage::EnvelopeMatrix LI;         // Cholesky factor of the damped interior matrix
  std::vector<double> LS;          // [nS_local*B][nI*B]  rows of L^-1 A_IS^T  (i.e. A_SI L^-T)
  std::vector<double> y;           // L^-1 b_I
  std::vector<int> ls_first;       // first non-zero column of every row of LS
  std::vector<int> int_pos;        // keyframe -> position among interior, or -1
  bool have_factor = false;
  bool assembled = false;          // domains of one assembled system (sage_shard_plan_create_domains)
  // fixed-block path (padded block size 40 / 24, the AVX-512 kernels of the window solve): local block system with the
  // interior keyframes at positions [0, nI) (elimination order) and this rank's separators at [nI, nI + nS)
  bool fast = false;
  int Bp = 0, nloc = 0, nblk = 0;
  std::vector<int32_t> row_first, row_off;
  std::vector<double> T, X, yv;
};

namespace
{
__attribute__((target_clones("avx512f", "avx2", "default"))) double sdot(const double *a, const double *b, int len)
{
#pragma clang fp reassociate(on)
  double acc = 0.0;
#pragma clang loop vectorize(enable) interleave_count(4)
  for (int k = 0; k < len; ++k)
    acc += a[k] * b[k];
  return acc;
}

__attribute__((target_clones("avx512f", "avx2", "default"))) void saxpy(double *y, const double *x, double a, int len)
{
#pragma clang loop vectorize(enable) interleave_count(4)
  for (int k = 0; k < len; ++k)
    y[k] -= a * x[k];
}

inline int touch_rank_range(const SageShardPlan &p, int link)
{
  // rank owning `link` (ranges are contiguous and ordered)
  int r = (int)(std::upper_bound(p.link_lo.begin(), p.link_lo.end(), link) - p.link_lo.begin()) - 1;
  return std::min(std::max(r, 0), p.world - 1);
}
} // namespace

// mode-independent part of the plan: given, for every link, the domain (rank) that assembles it, and for every keyframe
// whether it is a separator and who owns its prior
static int finish_plan(SageShardPlan *p, const std::vector<int> &link_dom, const std::vector<char> &is_sep)
{
  const int K = p->K, world = p->world, rank = p->rank;
  // touch[r][k]: domain r couples keyframe k through its interior (Schur fill: all separators a domain touches become
  // mutually coupled).  Assembled mode: a link between two separators is just that one block, not a reason for fill.
  std::vector<std::vector<char>> touch(world, std::vector<char>(K, 0));
  std::vector<std::pair<int, int>> direct_pairs;
  for (int l = 0; l < p->nlinks; ++l)
  {
    const int a = p->links[l].first, b = p->links[l].second;
    if (p->assembled && is_sep[a] && is_sep[b])
      direct_pairs.push_back({a, b});
    else
      touch[link_dom[l]][a] = touch[link_dom[l]][b] = 1;
  }
  p->sep_index.assign(K, -1);
  p->int_pos.assign(K, -1);
  for (int k = 0; k < K; ++k)
  {
    if (is_sep[k])
    {
      p->sep_index[k] = (int)p->sep_all.size();
      p->sep_all.push_back(k);
    }
    else if (p->kf_owner[k] == rank)
    {
      p->int_pos[k] = (int)p->interior.size();
      p->interior.push_back(k);
    }
  }
  for (int l = 0; l < p->nlinks; ++l)
    if (link_dom[l] == rank)
      p->local_links.push_back(l);
  // elimination order of the interior: ascending keyframe index, i.e. a temporal chain keeps its block band of 3.  The
  // rows of the separators BEHIND a piece of chain are short; those of the separators in FRONT of it fill across the
  // piece (3 GEMMs per block column).  Ordering a piece from its middle outwards would keep both short but doubles the
  // band (6 instead of 3): 24 instead of 7.5 block operations per interior row -- measured slower for every piece length.
  for (size_t i = 0; i < p->interior.size(); ++i)
    p->int_pos[p->interior[i]] = (int)i;
  for (size_t s = 0; s < p->sep_all.size(); ++s)
    if (touch[rank][p->sep_all[s]] || (p->assembled && rank == 0))
      p->sep_local.push_back((int)s);
  // coupled separator pairs: every domain couples all the separators it touches with each other (Schur fill through
  // its interior, direct links); identical on every rank
  for (int r = 0; r < world; ++r)
  {
    std::vector<int> sl;
    for (size_t s = 0; s < p->sep_all.size(); ++s)
      if (touch[r][p->sep_all[s]])
        sl.push_back((int)s);
    for (size_t i = 0; i < sl.size(); ++i)
      for (size_t j = i; j < sl.size(); ++j)
        p->pair_block.emplace(std::make_pair(sl[i], sl[j]), 0);
  }
  for (const auto &dp : direct_pairs)
  {
    const int i = p->sep_index[dp.first], j = p->sep_index[dp.second];
    p->pair_block.emplace(std::make_pair(std::min(i, j), std::max(i, j)), 0);
  }
  int nb = (int)p->sep_all.size(); // diagonal blocks first
  for (auto &kv : p->pair_block)
  {
    if (kv.first.first == kv.first.second)
      kv.second = kv.first.first;
    else
    {
      kv.second = nb++;
      p->sep_pairs.push_back(kv.first.first);
      p->sep_pairs.push_back(kv.first.second);
    }
  }
  for (int s = 0; s < (int)p->sep_all.size(); ++s)
    p->pair_block.emplace(std::make_pair(s, s), s);
  p->n_pair_blocks = nb;
  p->sep_doubles = (size_t)nb * p->B * p->B + p->sep_all.size() * (size_t)p->B + 8;
  // ---- block envelope of the local system for the fixed-block path
  p->Bp = (p->B + 7) / 8 * 8;
  p->fast = (p->Bp == 40 || p->Bp == 24);
  if (p->fast)
  {
    const int nI = (int)p->interior.size(), nS = (int)p->sep_local.size();
    p->nloc = nI + nS;
    std::vector<int> lpos(K, -1);
    for (int i = 0; i < nI; ++i)
      lpos[p->interior[i]] = i;
    for (int i = 0; i < nS; ++i)
      lpos[p->sep_all[p->sep_local[i]]] = nI + i;
    p->row_first.resize(p->nloc);
    for (int q = 0; q < p->nloc; ++q)
      p->row_first[q] = q;
    for (int l : p->local_links)
    {
      const int a = lpos[p->links[l].first], b = lpos[p->links[l].second];
      if (a < 0 || b < 0)
        return SAGE_E_STATE;
      p->row_first[std::max(a, b)] = std::min(p->row_first[std::max(a, b)], (int32_t)std::min(a, b));
    }
    // a separator row receives fill from every earlier separator that shares interior columns with it: keep the
    // separator rows' envelopes contiguous down to the first interior column any of them reaches
    p->row_off.resize(p->nloc);
    int nb2 = 0;
    for (int q = 0; q < p->nloc; ++q)
    {
      p->row_off[q] = nb2;
      nb2 += q - p->row_first[q] + 1;
    }
    p->nblk = nb2;
    p->T.assign((size_t)nb2 * p->Bp * p->Bp, 0.0);
    p->X.assign((size_t)p->nloc * p->Bp * p->Bp, 0.0);
    p->yv.assign((size_t)p->nloc * p->Bp, 0.0);
  }
  return SAGE_OK;
}

static SageShardPlan *new_plan(int K, int nlinks, const int32_t *links, int B, int rank, int world)
{
  SageShardPlan *p = new SageShardPlan;
  p->K = K; p->B = B; p->rank = rank; p->world = world; p->nlinks = nlinks;
  p->links.resize(nlinks);
  for (int l = 0; l < nlinks; ++l)
  {
    const int a = links[2 * l], b = links[2 * l + 1];
    if (a < 0 || b <= a || b >= K)
    {
      delete p;
      return nullptr;
    }
    p->links[l] = {a, b};
  }
  return p;
}

extern "C" int sage_shard_plan_create(int K, int nlinks, const int32_t *links, int B, int rank, int world,
                                      SageShardPlan **out)
{
  if (!out || K < 1 || B < 1 || nlinks < 0 || (nlinks > 0 && !links) || world < 1 || rank < 0 || rank >= world)
    return SAGE_E_INVALID;
  SageShardPlan *p = new_plan(K, nlinks, links, B, rank, world);
  if (!p)
    return SAGE_E_INVALID;
  p->link_lo.resize(world + 1);
  for (int r = 0; r <= world; ++r)
    p->link_lo[r] = (int)((long long)nlinks * r / world); // the rule of sage_window_set_shard
  std::vector<int> link_dom(nlinks);
  std::vector<std::vector<char>> touch(world, std::vector<char>(K, 0));
  for (int l = 0; l < nlinks; ++l)
  {
    link_dom[l] = touch_rank_range(*p, l);
    touch[link_dom[l]][p->links[l].first] = touch[link_dom[l]][p->links[l].second] = 1;
  }
  p->kf_owner.assign(K, -1);
  p->kf_ntouch.assign(K, 0);
  std::vector<char> is_sep(K, 0);
  for (int k = 0; k < K; ++k)
  {
    for (int r = 0; r < world; ++r)
      if (touch[r][k])
      {
        if (p->kf_owner[k] < 0)
          p->kf_owner[k] = r;
        p->kf_ntouch[k] += 1;
      }
    if (p->kf_owner[k] < 0)
      p->kf_owner[k] = 0; // a keyframe no link touches: a prior-only interior row of rank 0
    is_sep[k] = p->kf_ntouch[k] >= 2;
  }
  const int rc = finish_plan(p, link_dom, is_sep);
  if (rc)
  {
    delete p;
    return rc;
  }
  *out = p;
  return SAGE_OK;
}

// Domains of ONE assembled system (host threads instead of ranks): keyframe k belongs to domain k * ndomains / K; of
// every link that crosses a domain boundary the newer keyframe becomes a separator (the first 3 keyframes of every
// domain in a temporal window; the far end of a loop closure), so no link joins the interiors of two domains.
extern "C" int sage_shard_plan_create_domains(int K, int nlinks, const int32_t *links, int B, int domain, int ndomains,
                                              SageShardPlan **out)
{
  if (!out || K < 1 || B < 1 || nlinks < 0 || (nlinks > 0 && !links) || ndomains < 1 || domain < 0 ||
      domain >= ndomains)
    return SAGE_E_INVALID;
  SageShardPlan *p = new_plan(K, nlinks, links, B, domain, ndomains);
  if (!p)
    return SAGE_E_INVALID;
  p->assembled = true;
  auto dom = [&](int k) { return std::min(ndomains - 1, (int)((long long)k * ndomains / K)); };
  std::vector<char> is_sep(K, 0);
  // long-range links first (loop closures: keyframes further apart than any temporal back-link): their newer end
  // becomes a separator whatever the domains are, so that no envelope row has to span the window
  int span = 0;
  {
    std::vector<int> d(nlinks);
    for (int l = 0; l < nlinks; ++l)
      d[l] = p->links[l].second - p->links[l].first;
    std::vector<int> sorted(d);
    std::sort(sorted.begin(), sorted.end());
    span = nlinks ? std::max(8, 2 * sorted[(size_t)(0.9 * (nlinks - 1))]) : 8; // twice the 90th-percentile link length
    for (int l = 0; l < nlinks; ++l)
      if (d[l] > span && !is_sep[p->links[l].first] && !is_sep[p->links[l].second])
        is_sep[p->links[l].second] = 1;
  }
  for (int l = 0; l < nlinks; ++l)
  {
    const int a = p->links[l].first, b = p->links[l].second;
    if (dom(a) != dom(b) && !is_sep[a] && !is_sep[b])
      is_sep[b] = 1;
  }
  p->kf_owner.assign(K, 0);
  p->kf_ntouch.assign(K, 1);
  for (int k = 0; k < K; ++k)
    p->kf_owner[k] = is_sep[k] ? 0 : dom(k);
  std::vector<int> link_dom(nlinks);
  for (int l = 0; l < nlinks; ++l)
  {
    const int a = p->links[l].first, b = p->links[l].second;
    link_dom[l] = !is_sep[a] ? dom(a) : (!is_sep[b] ? dom(b) : 0); // separator-separator links: domain 0
  }
  const int rc = finish_plan(p, link_dom, is_sep);
  if (rc)
  {
    delete p;
    return rc;
  }
  *out = p;
  return SAGE_OK;
}

extern "C" void sage_shard_plan_destroy(SageShardPlan *p) { delete p; }
extern "C" size_t sage_shard_sep_count(const SageShardPlan *p) { return p ? p->sep_doubles : 0; }
extern "C" int sage_shard_num_separators(const SageShardPlan *p) { return p ? (int)p->sep_all.size() : 0; }
extern "C" int sage_shard_num_interior(const SageShardPlan *p) { return p ? (int)p->interior.size() : 0; }
extern "C" int sage_shard_keyframe_owner(const SageShardPlan *p, int kf)
{
  return (p && kf >= 0 && kf < p->K) ? p->kf_owner[kf] : -1;
}
extern "C" int sage_shard_keyframe_is_local(const SageShardPlan *p, int kf)
{
  if (!p || kf < 0 || kf >= p->K)
    return 0;
  if (p->int_pos[kf] >= 0)
    return 1;
  const int s = p->sep_index[kf];
  return s >= 0 && std::find(p->sep_local.begin(), p->sep_local.end(), s) != p->sep_local.end();
}

// this rank's damped contribution: element (r, c) of block (kr, kc) of  A^(r) = H^(r) + P^(r) + damp * diag(...)
namespace
{
struct LocalSystem
{
  const SageShardPlan &p;
  const double *diag, *lnk, *g;
  const double *dadd, *gadd;
  double damp;
  int B, BB;
  LocalSystem(const SageShardPlan &pl, const double *packed, double dmp, const double *da, const double *ga)
      : p(pl), dadd(da), gadd(ga), damp(dmp), B(pl.B), BB(pl.B * pl.B)
  {
    diag = packed;
    lnk = diag + (size_t)p.K * BB;
    g = lnk + (size_t)p.nlinks * BB;
  }
  bool owns_prior(int k) const { return p.kf_owner[k] == p.rank; }
  double diag_elem(int k, int r, int c) const // symmetrised like the full solve (0.5 (D + D^T)) + prior + damping
  {
    double v = 0.5 * (diag[(size_t)k * BB + r * B + c] + diag[(size_t)k * BB + c * B + r]);
    if (r == c)
    {
      const double pr = (dadd && owns_prior(k)) ? dadd[(size_t)k * B + r] : 0.0;
      v = (v + pr) * (1.0 + damp);
    }
    return v;
  }
  double rhs(int k, int r) const
  {
    return g[(size_t)k * B + r] + ((gadd && owns_prior(k)) ? gadd[(size_t)k * B + r] : 0.0);
  }
};
} // namespace

extern "C" int sage_shard_eliminate(SageShardPlan *p, const double *packed_local, double damp, const double *diag_add,
                                    const double *g_add, double *sep_out)
{
  if (!p || !packed_local || !sep_out)
    return SAGE_E_INVALID;
  const int B = p->B, BB = B * B, K = p->K;
  const LocalSystem S(*p, packed_local, damp, diag_add, g_add);
  static const bool dbg = sage::env_flag("SAGE_DEBUG_TIMING");
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  const auto t0 = tnow();
  const int nI = (int)p->interior.size(), nS = (int)p->sep_local.size();
  const int NI = nI * B, NS = nS * B;
  std::fill(sep_out, sep_out + p->sep_doubles, 0.0);
  double *sep_rhs = sep_out + (size_t)p->n_pair_blocks * BB;
  // tail: this rank's error / inlier totals at the linearisation point ride along
  const double *tail = packed_local + (size_t)(K + p->nlinks) * BB + (size_t)K * B;
  double *sep_tail = sep_rhs + p->sep_all.size() * (size_t)B;
  for (int i = 0; i < 4; ++i)
    sep_tail[i] = (!p->assembled || p->rank == 0) ? tail[i] : 0.0;
  // local separator position of a keyframe (or -1)
  std::vector<int> spos(K, -1);
  for (int i = 0; i < nS; ++i)
    spos[p->sep_all[p->sep_local[i]]] = i;
  if (p->fast)
  {
    const int Bp = p->Bp, BBp = Bp * Bp;
    std::fill(p->T.begin(), p->T.end(), 0.0);
    std::fill(p->yv.begin(), p->yv.end(), 0.0);
    auto blk = [&](int i, int j) { return p->T.data() + (size_t)(p->row_off[i] + j - p->row_first[i]) * BBp; };
    auto lpos = [&](int k) { return p->int_pos[k] >= 0 ? p->int_pos[k] : (spos[k] >= 0 ? nI + spos[k] : -1); };
    for (int q = 0; q < p->nloc; ++q)
    {
      const int k = q < nI ? p->interior[q] : p->sep_all[p->sep_local[q - nI]];
      const bool contributes = q < nI || !p->assembled || p->rank == 0; // separator diagonals: see the scalar path
      double *D = blk(q, q);
      for (int r = 0; r < Bp; ++r)
        for (int c = 0; c < Bp; ++c)
        {
          double v = 0.0;
          if (r < B && c < B)
            v = contributes ? S.diag_elem(k, r, c) : 0.0;
          else if (r == c)
            v = 1.0; // padding: identity
          D[r * Bp + c] = v;
        }
      if (contributes)
        for (int r = 0; r < B; ++r)
          p->yv[(size_t)q * Bp + r] = S.rhs(k, r);
    }
    for (int l : p->local_links)
    {
      const int a = p->links[l].first, b = p->links[l].second;
      const int qa = lpos(a), qb = lpos(b);
      if (qa < 0 || qb < 0)
        return SAGE_E_STATE;
      const int qi = std::max(qa, qb), qj = std::min(qa, qb);
      double *D = blk(qi, qj); // stored [c in column position][r in row position]; the packed block is [r in a][c in b]
      const bool row_is_a = qi == qa;
      const double *src = S.lnk + (size_t)l * BB;
      for (int r = 0; r < B; ++r)
        for (int c = 0; c < B; ++c)
          D[row_is_a ? c * Bp + r : r * Bp + c] += src[r * B + c];
    }
    sage::BlockEnvelope env;
    env.K = p->nloc; env.Bp = Bp; env.row_first = p->row_first.data(); env.row_off = p->row_off.data();
    const auto t1 = tnow();
    const int rcf = sage::block_chol_partial(env, p->T.data(), p->X.data(), p->yv.data(), nI);
    if (rcf != 0)
      return SAGE_E_NOT_PSD;
    p->have_factor = true;
    // Schur complement blocks and right-hand side of this rank's separators -> separator buffer
    for (int i = 0; i < nS; ++i)
    {
      for (int j = 0; j <= i; ++j)
      {
        if (nI + j < p->row_first[nI + i])
          continue; // outside the envelope: structurally zero
        const auto it = p->pair_block.find({p->sep_local[j], p->sep_local[i]});
        if (it == p->pair_block.end())
          continue; // (inside the envelope but coupled by nothing: zero)
        const double *src = blk(nI + i, nI + j); // [cc][rr] = C[(s_i, rr), (s_j, cc)], valid for rr >= cc when i == j
        double *dst = sep_out + (size_t)it->second * BB; // rows s_j, cols s_i
        for (int r = 0; r < B; ++r)
          for (int c = 0; c < B; ++c)
            dst[r * B + c] = (i != j || c >= r) ? src[r * Bp + c] : src[c * Bp + r];
      }
      for (int r = 0; r < B; ++r)
        sep_rhs[(size_t)p->sep_local[i] * B + r] = p->yv[(size_t)(nI + i) * Bp + r];
    }
    if (dbg)
    {
      auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      fprintf(stderr, "[sage shard eliminate] fixed-block: build %.3f partial factorisation %.3f scatter %.3f ms (%d interior, %d separators)\n",
              ms(t0, t1), ms(t1, tnow()), 0.0, nI, nS);
    }
    return SAGE_OK;
  }
  // ---- interior matrix in envelope form: first[] from the links among interior keyframes
  std::vector<int> first_blk(nI);
  for (int i = 0; i < nI; ++i)
    first_blk[i] = i;
  for (int l : p->local_links)
  {
    const int a = p->int_pos[p->links[l].first], b = p->int_pos[p->links[l].second];
    if (a >= 0 && b >= 0)
      first_blk[std::max(a, b)] = std::min(first_blk[std::max(a, b)], std::min(a, b));
  }
  std::vector<int> first(NI);
  for (int i = 0; i < nI; ++i)
    for (int r = 0; r < B; ++r)
      first[i * B + r] = first_blk[i] * B;
  p->LI.init(NI, first);
  for (int i = 0; i < nI; ++i)
  {
    const int k = p->interior[i];
    for (int r = 0; r < B; ++r)
      for (int c = 0; c <= r; ++c)
        p->LI.at(i * B + r, i * B + c) = S.diag_elem(k, r, c);
  }
  // A_SS (local contribution), A_SI
  std::vector<double> ASS((size_t)NS * NS, 0.0), ASI((size_t)NS * std::max(NI, 1), 0.0), bS(NS, 0.0), bI(NI, 0.0);
  // separator diagonal blocks / right-hand sides: every rank's own contribution (rank mode: its packed buffer only holds
  // its own edge sums); in assembled mode the buffer is complete and domain 0 alone contributes them
  if (!p->assembled || p->rank == 0)
    for (int i = 0; i < nS; ++i)
    {
      const int k = p->sep_all[p->sep_local[i]];
      for (int r = 0; r < B; ++r)
      {
        for (int c = 0; c < B; ++c)
          ASS[(size_t)(i * B + r) * NS + i * B + c] = S.diag_elem(k, r, c);
        bS[i * B + r] = S.rhs(k, r);
      }
    }
  for (int i = 0; i < nI; ++i)
    for (int r = 0; r < B; ++r)
      bI[i * B + r] = S.rhs(p->interior[i], r);
  for (int l : p->local_links)
  {
    const int a = p->links[l].first, b = p->links[l].second; // block rows = a, cols = b
    const double *blk = S.lnk + (size_t)l * BB;
    const int ia = p->int_pos[a], ib = p->int_pos[b], sa = spos[a], sb = spos[b];
    for (int r = 0; r < B; ++r)
      for (int c = 0; c < B; ++c)
      {
        const double v = blk[r * B + c]; // A[a*B + r][b*B + c]
        if (ia >= 0 && ib >= 0)
        {
          if (ib > ia)
            p->LI.at(ib * B + c, ia * B + r) += v;
          else
            p->LI.at(ia * B + r, ib * B + c) += v;
        }
        else if (sa >= 0 && sb >= 0)
        {
          ASS[(size_t)(sa * B + r) * NS + sb * B + c] += v;
          ASS[(size_t)(sb * B + c) * NS + sa * B + r] += v;
        }
        else if (sa >= 0 && ib >= 0)
          ASI[(size_t)(sa * B + r) * NI + ib * B + c] += v;
        else if (ia >= 0 && sb >= 0)
          ASI[(size_t)(sb * B + c) * NI + ia * B + r] += v;
        else
          return SAGE_E_STATE; // a link of this rank touching a keyframe that is neither interior nor its separator
      }
  }
  // ---- eliminate the interior
  p->have_factor = false;
  const auto t1 = tnow();
  auto t2 = t1, t3 = t1;
  if (NI > 0)
  {
    if (!p->LI.cholesky_inplace(B, 1))
      return SAGE_E_NOT_PSD;
    t2 = tnow();
    // forward substitutions: y = L^-1 b_I ; rows of LS = L^-1 (A_SI row)^T
    // v <- L^-1 v, skipping the leading zeros of v (a separator row of A_SI only touches the interior keyframes next
    // to it, which the elimination order puts last); returns the index of the first non-zero
    auto forward = [&](double *v) {
      int z = 0;
      while (z < NI && v[z] == 0.0)
        ++z;
      for (int r = z; r < NI; ++r)
      {
        const int fr = std::max(p->LI.first[r], z);
        const double *Lr = &p->LI.data[p->LI.rowptr[r]] - p->LI.first[r];
        v[r] = (v[r] - sdot(Lr + fr, v + fr, r - fr)) / Lr[r];
      }
      return z;
    };
    p->y = bI;
    forward(p->y.data());
    p->LS = ASI;
    p->ls_first.assign(NS, 0);
    for (int s2 = 0; s2 < NS; ++s2)
      p->ls_first[s2] = forward(&p->LS[(size_t)s2 * NI]);
    t3 = tnow();
    // C = A_SS - LS LS^T ; c = b_S - LS y
    for (int i = 0; i < NS; ++i)
    {
      const double *li = &p->LS[(size_t)i * NI];
      for (int j = i; j < NS; ++j)
      {
        const double *lj = &p->LS[(size_t)j * NI];
        const int z = std::max(p->ls_first[i], p->ls_first[j]);
        const double acc = z < NI ? sdot(li + z, lj + z, NI - z) : 0.0;
        ASS[(size_t)i * NS + j] -= acc;
        if (j != i)
          ASS[(size_t)j * NS + i] -= acc;
      }
      const int z = p->ls_first[i];
      bS[i] -= z < NI ? sdot(li + z, p->y.data() + z, NI - z) : 0.0;
    }
  }
  else
  {
    p->LS.clear();
    p->y.clear();
  }
  p->have_factor = true;
  if (dbg)
  {
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[sage shard eliminate] build %.3f chol(I) %.3f forward(S rows) %.3f schur %.3f ms (NI %d NS %d)\n",
            ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, tnow()), NI, NS);
  }
  // ---- scatter into the separator buffer (upper-triangular block pairs)
  for (int i = 0; i < nS; ++i)
  {
    for (int j = i; j < nS; ++j)
    {
      const auto it = p->pair_block.find({p->sep_local[i], p->sep_local[j]});
      if (it == p->pair_block.end())
      {
        if (p->assembled)
          continue; // domain 0 lists every separator: pairs nothing couples are structurally zero
        return SAGE_E_STATE;
      }
      double *dst = sep_out + (size_t)it->second * BB;
      for (int r = 0; r < B; ++r)
        for (int c = 0; c < B; ++c)
          dst[r * B + c] = ASS[(size_t)(i * B + r) * NS + j * B + c];
    }
    for (int r = 0; r < B; ++r)
      sep_rhs[(size_t)p->sep_local[i] * B + r] = bS[i * B + r];
  }
  return SAGE_OK;
}

extern "C" int sage_shard_solve(SageShardPlan *p, const double *sep_reduced, double *delta)
{
  if (!p || !sep_reduced || !delta || !p->have_factor)
    return SAGE_E_STATE;
  const int B = p->B;
  const int nSa = (int)p->sep_all.size(), NSa = nSa * B;
  std::vector<double> dS(NSa, 0.0);
  if (NSa > 0)
  {
    // the reduced buffer IS a packed block system over the separator keyframes (diag | coupled pairs | rhs): the
    // fixed-block AVX-512 factorisation of the window solve takes it as it is (no damping, no priors: both are inside)
    const int rc = sage_block_solve(sep_reduced, nSa, (int)p->sep_pairs.size() / 2, p->sep_pairs.data(), B, 0.0, nullptr,
                                    nullptr, dS.data());
    if (rc)
      return rc;
  }
  if (p->fast)
  {
    const int Bp = p->Bp, nI2 = (int)p->interior.size(), nS2 = (int)p->sep_local.size();
    for (int s2 = 0; s2 < nS2; ++s2)
    {
      for (int r = 0; r < B; ++r)
        p->yv[(size_t)(nI2 + s2) * Bp + r] = dS[(size_t)p->sep_local[s2] * B + r];
      for (int r = B; r < Bp; ++r)
        p->yv[(size_t)(nI2 + s2) * Bp + r] = 0.0;
    }
    sage::BlockEnvelope env;
    env.K = p->nloc; env.Bp = Bp; env.row_first = p->row_first.data(); env.row_off = p->row_off.data();
    if (sage::block_chol_partial_back(env, p->T.data(), p->X.data(), p->yv.data(), nI2) != 0)
      return SAGE_E_STATE;
    for (int i = 0; i < nI2; ++i)
      for (int r = 0; r < B; ++r)
        delta[(size_t)p->interior[i] * B + r] = p->yv[(size_t)i * Bp + r];
    for (int s2 = 0; s2 < nS2; ++s2)
      for (int r = 0; r < B; ++r)
        delta[(size_t)p->sep_all[p->sep_local[s2]] * B + r] = dS[(size_t)p->sep_local[s2] * B + r];
    return SAGE_OK;
  }
  // ---- back-substitute the interior: d_I = L^-T (y - LS^T d_S(local))
  const int nI = (int)p->interior.size(), nS = (int)p->sep_local.size(), NI = nI * B;
  std::vector<double> v(p->y);
  for (int s = 0; s < nS; ++s)
    for (int r = 0; r < B; ++r)
    {
      const double d = dS[(size_t)p->sep_local[s] * B + r];
      const int row = s * B + r, z = NI > 0 ? p->ls_first[row] : 0;
      if (z < NI)
        saxpy(v.data() + z, &p->LS[(size_t)row * NI] + z, d, NI - z);
    }
  for (int r = NI - 1; r >= 0; --r)
  {
    const int fr = p->LI.first[r];
    const double *Lr = &p->LI.data[p->LI.rowptr[r]];
    const double x = v[r] / Lr[r - fr];
    v[r] = x;
    saxpy(v.data() + fr, Lr, x, r - fr);
  }
  for (int i = 0; i < nI; ++i)
    for (int r = 0; r < B; ++r)
      delta[(size_t)p->interior[i] * B + r] = v[i * B + r];
  for (int s = 0; s < nS; ++s)
    for (int r = 0; r < B; ++r)
      delta[(size_t)p->sep_all[p->sep_local[s]] * B + r] = dS[(size_t)p->sep_local[s] * B + r];
  return SAGE_OK;
}

// ------------------------------------------------------------------------------------------------
// One process, several host threads: the damped solve of an ASSEMBLED window system by domain decomposition.
// Same contract as sage_block_solve; meant for windows the banded factorisation handles badly: long windows (the
// domains are eliminated in parallel) and loop closures (their far ends become separators instead of envelope rows
// that span the whole window).
// ------------------------------------------------------------------------------------------------
#include <thread>

extern "C" int sage_block_solve_domains(const double *packed, int K, int nlinks, const int32_t *links, int B,
                                        double damp, const double *diag_add, const double *g_add, int ndomains,
                                        double *delta)
{
  if (!packed || !delta || K < 1 || B < 1 || ndomains < 1)
    return SAGE_E_INVALID;
  ndomains = std::min(ndomains, std::max(1, K / 4));
  std::vector<SageShardPlan *> plans(ndomains, nullptr);
  int rc = SAGE_OK;
  for (int d = 0; d < ndomains && !rc; ++d)
    rc = sage_shard_plan_create_domains(K, nlinks, links, B, d, ndomains, &plans[d]);
  std::vector<std::vector<double>> sep(ndomains);
  std::vector<int> rcs(ndomains, 0);
  if (!rc)
  {
    const size_t n = sage_shard_sep_count(plans[0]);
    std::vector<std::thread> th;
    for (int d = 0; d < ndomains; ++d)
    {
      sep[d].assign(n, 0.0);
      auto job = [&, d] { rcs[d] = sage_shard_eliminate(plans[d], packed, damp, diag_add, g_add, sep[d].data()); };
      if (d + 1 < ndomains)
        th.emplace_back(job);
      else
        job();
    }
    for (auto &t : th)
      t.join();
    for (int d = 0; d < ndomains; ++d)
      if (rcs[d])
        rc = rcs[d];
    if (!rc)
    {
      for (int d = 1; d < ndomains; ++d)
        for (size_t i = 0; i < n; ++i)
          sep[0][i] += sep[d][i];
      std::fill(delta, delta + (size_t)K * B, 0.0);
      th.clear();
      // every domain solves into a buffer of its own: a separator keyframe next to two domains is written by both (the same
      // values, but two threads storing to one address is still a data race -- found with ThreadSanitizer); merged below
      std::vector<std::vector<double>> dd((size_t)ndomains, std::vector<double>((size_t)K * B, 0.0));
      for (int d = 0; d < ndomains; ++d)
      {
        auto job = [&, d] { rcs[d] = sage_shard_solve(plans[d], sep[0].data(), dd[d].data()); };
        if (d + 1 < ndomains)
          th.emplace_back(job);
        else
          job();
      }
      for (auto &t : th)
        t.join();
      for (int d = 0; d < ndomains; ++d)
        if (rcs[d])
          rc = rcs[d];
      // (an interior keyframe has one owner, a separator's rows are identical in every domain that holds it)
      for (int d = 0; d < ndomains; ++d)
        for (size_t i = 0; i < (size_t)K * B; ++i)
          if (dd[d][i] != 0.0)
            delta[i] = dd[d][i];
    }
  }
  for (SageShardPlan *p : plans)
    sage_shard_plan_destroy(p);
  return rc;
}
