// window.hip -- the batched window engine: edge tables, work lists, deterministic assembly into block-sparse normal
// equations, linearize / error pass / damped solve / LM iteration (no reference counterpart: SURVEY s8 "new").
#include <atomic>
#include "runtime_internal.h"
#include "finalize_bodies.h"

namespace sage
{




// B-index (0..6+CS: pose 6, code CS, scale) -> column of the per-edge system, or -1 if absent
__device__ __forceinline__ int edge_col(int type, int role, int bi, int CS)
{
  if (bi < 6)
    return role * 6 + bi;
  if (type == 0)
  {
    if (role == 1)
      return -1; // a photometric edge does not touch code1 / scale1
    return bi < 6 + CS ? 12 + (bi - 6) : 12 + CS;
  }
  if (bi < 6 + CS)
    return 12 + role * CS + (bi - 6);
  return 12 + 2 * CS + role;
}

// one workgroup per output block; thread per element; contributions summed in a fixed order (deterministic)
__global__ __launch_bounds__(1024) void assemble_kernel(const AssembleParams p)
{
  const int B = 7 + p.CS, BB = B * B;
  const int Dp = 13 + p.CS, Dg = 14 + 2 * p.CS;
  const int split = p.split > 1 ? p.split : 1;
  const int part = (int)blockIdx.x % split, bslot = (int)blockIdx.x / split;
  const int blk = p.blocks ? p.blocks[bslot] : bslot;
  const int tstride = (int)blockDim.x * split, tfirst = part * (int)blockDim.x + (int)threadIdx.x; // element striding
  double *diag = p.packed;
  double *lnk = diag + (size_t)p.K * BB;
  double *g = lnk + (size_t)p.nlinks * BB;
  double *tail = g + (size_t)p.K * B;
  if (blk < p.K)
  {
    // fp64 accumulation of the fp32 per-edge results (the reference widens to double before gtsam sums them:
    // photometric_factor.cpp:305-306).  Adjacency loop outside, the lane's (at most two) outputs inside: the gathers of
    // different adjacency entries are independent, so they overlap instead of forming one chain of ~150 dependent loads
    const int k = blk;
    const int a0 = p.adj_start[k], a1 = p.adj_start[k + 1];
    constexpr int S = 2;
    for (int base = 0; base < BB + B; base += S * tstride) // one pass with 1024 threads (or 4 x 256)
    {
    double acc[S] = {0.0, 0.0};
    int bi[S], bj[S];
    bool isg[S], valid[S];
#pragma unroll
    for (int s = 0; s < S; ++s)
    {
      const int idx = base + tfirst + s * tstride;
      valid[s] = idx < BB + B;
      isg[s] = idx >= BB;
      bi[s] = isg[s] ? idx - BB : idx / B;
      bj[s] = isg[s] ? 0 : idx % B;
    }
#pragma unroll 4
    for (int a = a0; a < a1; ++a)
    {
      const AdjEntry ae = p.adj[a];
      const int D = ae.type == 0 ? Dp : Dg;
      const float *A = ae.type == 0 ? p.AtA_p : p.AtA_g;
      const float *b = ae.type == 0 ? p.Atb_p : p.Atb_g;
#pragma unroll
      for (int s = 0; s < S; ++s)
      {
        if (!valid[s])
          continue;
        const int ci = edge_col(ae.type, ae.role, bi[s], p.CS);
        const int cj = isg[s] ? 0 : edge_col(ae.type, ae.role, bj[s], p.CS);
        if (ci < 0 || cj < 0)
          continue;
        const double *Wd = ae.type == 0 ? p.wide_p : p.wide_g;
        if (Wd)
          acc[s] += Wd[(size_t)ae.edge * (D * D + D) + (isg[s] ? (size_t)D * D + ci : (size_t)ci * D + cj)];
        else
          acc[s] += isg[s] ? (double)b[(size_t)ae.edge * D + ci] : (double)A[(size_t)ae.edge * D * D + (size_t)ci * D + cj];
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
    {
      const int idx = base + tfirst + s * tstride;
      if (!valid[s])
        continue;
      if (isg[s])
        g[(size_t)k * B + bi[s]] = acc[s];
      else
        diag[(size_t)k * BB + idx] = acc[s];
    }
    } // passes
  }
  else if (blk < p.K + p.nlinks)
  {
    const int l = blk - p.K;
    const LinkEdges le = p.links[l];
    for (int idx = tfirst; idx < BB; idx += tstride)
    {
      const int bi = idx / B, bj = idx % B; // bi indexes keyframe a (older), bj keyframe b
      double acc = 0.0;
      {
        for (int type = 0; type < 2; ++type)
        {
          if ((type == 0 && !p.AtA_p) || (type == 1 && !p.AtA_g))
            continue;
          const int D = type == 0 ? Dp : Dg;
          const float *A = type == 0 ? p.AtA_p : p.AtA_g;
          const double *Wd = type == 0 ? p.wide_p : p.wide_g;
          const size_t ws = (size_t)D * D + D;
          // edge a->b : a has role 0, b has role 1
          int ci = edge_col(type, 0, bi, p.CS), cj = edge_col(type, 1, bj, p.CS);
          if (le.e_ab >= 0 && ci >= 0 && cj >= 0) // (each direction on its own: the other one may belong to another rank)
            acc += Wd ? Wd[(size_t)le.e_ab * ws + (size_t)ci * D + cj] : (double)A[(size_t)le.e_ab * D * D + (size_t)ci * D + cj];
          // edge b->a : b has role 0, a has role 1
          ci = edge_col(type, 1, bi, p.CS);
          cj = edge_col(type, 0, bj, p.CS);
          if (le.e_ba >= 0 && ci >= 0 && cj >= 0)
            acc += Wd ? Wd[(size_t)le.e_ba * ws + (size_t)ci * D + cj] : (double)A[(size_t)le.e_ba * D * D + (size_t)ci * D + cj];
        }
      }
      lnk[(size_t)l * BB + idx] = acc;
    }
  }
  else
  {
    // tail: total errors / inlier counts of the local edges; one wave per sum, fixed lane order (deterministic)
    if (part != 0)
      return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool photo = (wave & 1) == 0;
    const int which = wave >> 1; // 0: error, 1: inliers
    const float *st = photo ? p.stats_p : p.stats_g;
    const int n = photo ? p.n_edges_p : p.n_edges_g;
    double acc = 0.0;
    if (st && wave < 4)
      for (int e = lane; e < n; e += 64)
        acc += (double)st[2 * e + which];
    for (int off = 32; off > 0; off >>= 1)
      acc += __shfl_down(acc, off);
    if (lane == 0 && wave < 4)
    {
      tail[which * 2 + (photo ? 0 : 1)] = acc; // [err_photo err_geo n_photo n_geo]
      if (p.tail_mirror)
        p.tail_mirror[which * 2 + (photo ? 0 : 1)] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// per-edge finalize of BOTH factor types in one launch (a workgroup per edge and type): the geometric finalize no longer
// sits between the two big kernels (18 us + a launch gap on the step's critical path, r04 timeline)
// ------------------------------------------------------------------------------------------------
struct WindowFinalizeParams
{
  PhotoFinalizeParams ph;
  GeoFinalizeParams ge;
  int n_p, n_g;
};

template <int CS>
__global__ __launch_bounds__(kFinalizeBlock) void window_finalize_kernel(const WindowFinalizeParams prm)
{
  constexpr int LDS = photo_finalize_lds_doubles(CS) > geo_finalize_lds_doubles(CS) ? photo_finalize_lds_doubles(CS)
                                                                                    : geo_finalize_lds_doubles(CS);
  __shared__ double s[LDS];
  const int bid = (int)blockIdx.x;
  if (bid < prm.n_g) // (the longer finalize first)
    geo_finalize_body<CS>(prm.ge, bid, s);
  else
    photo_finalize_body<CS>(prm.ph, bid - prm.n_g, s);
}

// error pass of a window in ONE tail kernel: per-edge statistics of both factor types from the workgroup partials
// (what stats_finalize_kernel does: photometric_factor_kernels.cpp:1049-1058, geometric :868-878) and their totals
// (a wave-parallel sum in a fixed lane order) -- same summation orders, three launches and their gaps less on the step's critical path.

__global__ __launch_bounds__(1024) void error_totals_kernel(const ErrorTotalsSide ph, const ErrorTotalsSide ge, double *out,
                                                            double *mirror, double epoch)
{
  for (int idx = threadIdx.x; idx < ph.n_edges + ge.n_edges; idx += blockDim.x)
  {
    const bool photo = idx < ph.n_edges;
    const ErrorTotalsSide &sd = photo ? ph : ge;
    const int e = photo ? idx : idx - ph.n_edges;
    const int first = sd.edge_first[e], nt = sd.edge_tiles[e];
    // same order of the sums as stats_finalize_kernel; eight records' loads in flight at a time (a chain of dependent
    // cache misses otherwise: this one-workgroup kernel sits on the step's critical path)
    float se = 0.f, sn = 0.f;
    for (int t0 = 0; t0 < nt; t0 += 8)
    {
      float ve[8], vn[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
      {
        const int t = t0 + u < nt ? t0 + u : nt - 1;
        ve[u] = sd.partials[(size_t)(first + t) * sd.stride + sd.err_off];
        vn[u] = sd.partials[(size_t)(first + t) * sd.stride + sd.cnt_off];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (t0 + u < nt)
        {
          se += ve[u];
          sn += vn[u];
        }
    }
    sd.stats[2 * e + 0] = sn > 0.f ? sd.scale * se / sn : sd.fallback;
    sd.stats[2 * e + 1] = sn;
  }
  __threadfence_block();
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= 4)
    return;
  const bool photo = (wave & 1) == 0;
  const int which = wave >> 1;
  const ErrorTotalsSide &sd = photo ? ph : ge;
  double acc = 0.0;
  for (int e0 = lane; e0 < sd.n_edges; e0 += 64 * 8) // (same order per lane; eight loads in flight)
  {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
    {
      const int e = e0 + 64 * u;
      v[u] = sd.stats[2 * (e < sd.n_edges ? e : lane) + which];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + 64 * u < sd.n_edges)
        acc += (double)v[u];
  }
  for (int off = 32; off > 0; off >>= 1)
    acc += __shfl_down(acc, off);
  if (lane == 0)
  {
    out[which * 2 + (photo ? 0 : 1)] = acc;
    if (mirror)
    {
      // the value, then its ticket: the host spins on the four tickets instead of synchronising the stream
      mirror[which * 2 + (photo ? 0 : 1)] = acc;
      __threadfence_system();
      *reinterpret_cast<volatile double *>(mirror + 4 + which * 2 + (photo ? 0 : 1)) = epoch;
    }
  }
}

__global__ void copy_floats_kernel(const float *__restrict__ src, float *__restrict__ dst, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    dst[i] = src[i];
}

// the (reduced) totals -- tail of the packed buffer, error buffer (may be null) -> pinned host mirror.  One wave; the values,
// a workgroup barrier, then one system-scope release and the four tickets in the kernel's OWN slots (mirror[12..15];
// error_totals_kernel owns mirror[8..11]): the host spins on them instead of synchronising the stream
__global__ void mirror_totals_kernel(const double *__restrict__ tail, const double *__restrict__ err,
                                     double *__restrict__ mirror, double epoch)
{
  const int t = threadIdx.x;
  if (t < 4)
    mirror[t] = tail[t];
  else if (t < 8 && err)
    mirror[t] = err[t - 4];
  __syncthreads();
  if (t == 0)
  {
    __threadfence_system();
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<volatile double *>(mirror + 12 + i) = epoch;
  }
}

// dst += src (peer emulation: the contribution of the ranks that are not there)
__global__ void add_doubles_kernel(const double *__restrict__ src, double *__restrict__ dst, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    dst[i] += src[i];
}

} // namespace sage
static bool ev_get(SageWindow *w, hipEvent_t *e)
{
  if (!w->ev_free.empty())
  {
    *e = w->ev_free.back();
    w->ev_free.pop_back();
    return true;
  }
  return hipEventCreate(e) == hipSuccess;
}
static void ev_put(SageWindow *w, hipEvent_t e)
{
  if (e)
    w->ev_free.push_back(e);
}

static void prof_attach(SageWindow *w, int which, LaunchCommon &lc)
{
  if (!w->profiling || (w->prof_level == 2 && which != 0))
    return;
  hipEvent_t a, b;
  if (!ev_get(w, &a))
    return;
  if (!ev_get(w, &b))
  {
    ev_put(w, a);
    return;
  }
  lc.ev_start = a;
  lc.ev_stop = b;
  w->pending[which].emplace_back(a, b);
}

void window_phase_mark(SageWindow *w, int which)
{
  if (!w->profiling || w->prof_level == 2)
    return;
  if (which == 0) // a new iteration: the previous one's marks are complete
  {
    if (w->phase_cur.ev.size() >= 2)
    {
      if (w->phase_pending.size() >= 1024) // nobody collects them: keep the newest
      {
        for (auto &m : w->phase_pending.front().ev)
          ev_put(w, m.second);
        w->phase_pending.erase(w->phase_pending.begin());
      }
      w->phase_pending.push_back(std::move(w->phase_cur));
    }
    else
      for (auto &m : w->phase_cur.ev)
        ev_put(w, m.second);
    w->phase_cur = SageWindow::PhaseMarks{};
  }
  else if (w->phase_cur.ev.empty())
    return; // (a mark outside an iteration: sage_window_solve / _error called on their own)
  hipEvent_t e;
  if (w->phase_cur.ev.size() >= 64 || !ev_get(w, &e))
    return;
  (void)hipEventRecord(e, w->stream);
  w->phase_cur.ev.emplace_back(which, e);
}

extern "C" int sage_window_get_phase_time(SageWindow *w, double *ms4, int *iterations)
{
  if (!w || !ms4)
    return SAGE_E_INVALID;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  window_phase_mark(w, 0); // flush the iteration in progress
  for (auto &m : w->phase_cur.ev) // (the mark the flush just recorded opens nothing)
    ev_put(w, m.second);
  w->phase_cur = SageWindow::PhaseMarks{};
  for (auto &pm : w->phase_pending)
  {
    // the time between two consecutive marks belongs to the phase the later one closes (1 linearize, 2 all-reduce,
    // 3 solve, 4 error pass); every evaluation of an iteration counts
    double d4[4] = {0, 0, 0, 0};
    bool ok = true;
    for (size_t i = 1; i < pm.ev.size() && ok; ++i)
    {
      float d = 0.f;
      ok = hipEventElapsedTime(&d, pm.ev[i - 1].second, pm.ev[i].second) == hipSuccess;
      const int ph = pm.ev[i].first;
      if (ok && ph >= 1 && ph <= 4)
        d4[ph - 1] += d;
    }
    if (ok)
    {
      for (int i = 0; i < 4; ++i)
        w->phase_ms[i] += d4[i];
      w->phase_n += 1;
    }
    for (auto &m : pm.ev)
      ev_put(w, m.second);
  }
  w->phase_pending.clear();
  for (int i = 0; i < 4; ++i)
  {
    ms4[i] = w->phase_ms[i];
    w->phase_ms[i] = 0;
  }
  if (iterations)
    *iterations = w->phase_n;
  w->phase_n = 0;
  return SAGE_OK;
}

extern "C" int sage_window_set_profiling(SageWindow *w, int on)
{
  if (!w)
    return SAGE_E_INVALID;
  w->profiling = on != 0;
  w->prof_level = on;
  // a stock of events up front: creating them inside the region being profiled costs API time there
  while (w->profiling && w->ev_free.size() < 512)
  {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess)
      break;
    w->ev_free.push_back(e);
  }
  return SAGE_OK;
}

extern "C" int sage_window_get_kernel_time(SageWindow *w, int which, double *total_ms, int *launches)
{
  if (!w || which < 0 || which > 3)
    return SAGE_E_INVALID;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  for (auto &pr : w->pending[which])
  {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
    {
      w->prof_ms[which] += ms;
      w->prof_n[which] += 1;
    }
    ev_put(w, pr.first);
    ev_put(w, pr.second);
  }
  w->pending[which].clear();
  if (total_ms)
    *total_ms = w->prof_ms[which];
  if (launches)
    *launches = w->prof_n[which];
  w->prof_ms[which] = 0;
  w->prof_n[which] = 0;
  return SAGE_OK;
}

static void upload_vars_host(SageWindow *w, int set, std::vector<float> &buf)
{
  buf.assign((size_t)w->K * w->VS, 0.f);
  const int CS = w->cfg.CS;
  for (int k = 0; k < w->K; ++k)
  {
    float *d = &buf[(size_t)k * w->VS];
    std::memcpy(d, &w->pose[set][(size_t)k * 12], 12 * sizeof(float));
    d[12] = w->scale[set][k];
    std::memcpy(d + 13, &w->code[set][(size_t)k * CS], CS * sizeof(float));
  }
}

int window_upload_vars(SageWindow *w, int set)
{
  if (w->dpt_set == set)
    w->dpt_set = -1;
  if (set == 0)
    ++w->vars_epoch; // whatever was linearised is no longer the system at the current variables
  std::vector<float> buf;
  upload_vars_host(w, set, buf);
  SAGE_HIP(hipMemcpyAsync(w->vars[set].p, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice, w->stream));
  SAGE_HIP(hipStreamSynchronize(w->stream)); // buf is a temporary
  return SAGE_OK;
}

// live windows of the process: the last sage_window_destroy stops the solver's host threads (host_math.cpp "life cycle")
static std::atomic<int> g_live_windows{0};

extern "C" int sage_window_create(const SageWindowConfig *cfg, void *hip_stream, SageWindow **out)
{
  if (!cfg || !out || !cfg->mask_dev)
    return SAGE_E_INVALID;
  if (!supported(cfg->CS, cfg->FS) || cfg->pyr.levels < 1 || cfg->pyr.levels > SAGE_MAX_LEVELS)
    return SAGE_E_UNSUPPORTED;
  int ndev = 0;
  SAGE_HIP(hipGetDeviceCount(&ndev));
  if (ndev < 1)
    return (int)hipErrorNoDevice;
  SageWindow *w = new SageWindow();
  w->cfg = *cfg;
  w->stream = reinterpret_cast<hipStream_t>(hip_stream);
  w->B = 7 + cfg->CS;
  w->VS = ((13 + cfg->CS + 3) / 4) * 4;
  *out = w;
  g_live_windows.fetch_add(1, std::memory_order_acq_rel);
  return SAGE_OK;
}

extern "C" void sage_window_destroy(SageWindow *w)
{
  if (!w)
    return;
  // no thread of this library outlives the last window (r06): helpers, arrow-row pool and the opt-in
  // placement monitor are stopped and joined; the next window's first solve starts them again
  struct LastOut
  {
    ~LastOut()
    {
      if (g_live_windows.fetch_sub(1, std::memory_order_acq_rel) == 1)
        sage::host_threads_shutdown();
    }
  } last_out;
  DevBuf *bufs[] = {&w->packed_save, &w->packed_loc, &w->asm_blocks, &w->geo_px, &w->rec_first_p, &w->rec_count_p, &w->wide_p, &w->wide_g, &w->sorted_loc, &w->sorted_homo, &w->vars[0], &w->vars[1], &w->dpt, &w->dgrad, &w->depth_items[0], &w->depth_items[1],
                    &w->pk, &w->f0s, &w->ptab[0], &w->ptab[1], &w->gtab[0], &w->gtab[1], &w->work_p, &w->first_p, &w->tiles_p,
                    &w->work_g, &w->first_g, &w->tiles_g, &w->part_p, &w->part_g, &w->AtA_p, &w->Atb_p,
                    &w->stats_p, &w->AtA_g, &w->Atb_g, &w->stats_g, &w->adj_start, &w->adj, &w->link_edges,
                    &w->packed, &w->errbuf};
  for (DevBuf *b : bufs)
    b->release();
  std::free(w->rccl_hook); // (the communicator itself belongs to the caller)
  sage_shard_plan_destroy(w->shard);
  w->sepbuf.release();
  solver_destroy(w->solver);
  if (w->h_err)
    (void)hipHostFree(w->h_err);
  for (auto &pm : w->phase_pending)
    for (auto &m : pm.ev)
      (void)hipEventDestroy(m.second);
  for (auto &m : w->phase_cur.ev)
    (void)hipEventDestroy(m.second);
  for (auto &pend : w->pending)
    for (auto &pr : pend)
    {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  for (hipEvent_t e : w->ev_free)
    (void)hipEventDestroy(e);
  delete w;
}

extern "C" int sage_window_add_keyframe(SageWindow *w, const SageKeyframeView *v, const float *pose12,
                                        const float *code, float scale)
{
  if (!w || !v || !pose12 || !code || w->finalized)
    return SAGE_E_INVALID;
  if (!v->feat_pyr || !v->grad_pyr || !v->bias || !v->basis || !v->loc1d || !v->homo || v->N < 0)
    return SAGE_E_INVALID;
  w->views.push_back(*v);
  for (int s = 0; s < 2; ++s)
  {
    w->pose[s].insert(w->pose[s].end(), pose12, pose12 + 12);
    w->code[s].insert(w->code[s].end(), code, code + w->cfg.CS);
    w->scale[s].push_back(scale);
  }
  w->pose_init.insert(w->pose_init.end(), pose12, pose12 + 12);
  w->code_added.insert(w->code_added.end(), code, code + w->cfg.CS);
  w->scale_init.push_back(scale);
  return w->K++;
}

extern "C" int sage_window_add_link(SageWindow *w, int a, int b)
{
  if (!w || w->finalized || a == b || a < 0 || b < 0 || a >= w->K || b >= w->K)
    return SAGE_E_INVALID;
  w->links.emplace_back(std::min(a, b), std::max(a, b));
  w->link_geo_loss.push_back(0.f);
  return (int)w->links.size() - 1;
}

extern "C" int sage_window_set_link_geo_loss(SageWindow *w, int link, float loss_param)
{
  if (!w || w->finalized || link < 0 || link >= (int)w->links.size() || !(loss_param >= 0.f))
    return w && w->finalized ? SAGE_E_STATE : SAGE_E_INVALID;
  w->link_geo_loss[link] = loss_param;
  return SAGE_OK;
}

extern "C" int sage_window_num_keyframes(const SageWindow *w) { return w ? w->K : 0; }
extern "C" int sage_window_num_links(const SageWindow *w) { return w ? (int)w->links.size() : 0; }
extern "C" int sage_window_block_size(const SageWindow *w) { return w ? w->B : 0; }
extern "C" size_t sage_window_packed_count(const SageWindow *w)
{
  if (!w)
    return 0;
  const size_t BB = (size_t)w->B * w->B;
  return (size_t)w->K * BB + w->links.size() * BB + (size_t)w->K * w->B + 4;
}
extern "C" double *sage_window_packed_dev(SageWindow *w) { return w ? w->packed.as<double>() : nullptr; }
extern "C" double *sage_window_error_dev(SageWindow *w) { return w ? w->errbuf.as<double>() : nullptr; }
extern "C" double sage_window_residuals_per_linearize(const SageWindow *w) { return w ? w->residuals_per_lin : 0; }
extern "C" double sage_window_bytes_per_linearize(const SageWindow *w) { return w ? w->bytes_per_lin : 0; }


// ---- photometric run plan (work list of the photometric linearize / error pass) ------------------------------------------------
// record cadence of a run length: a partial record every 3-5 sub-tiles (0 = one record per workgroup).  With the second level of
// the noise-critical tiles and their split accumulators in the kernel this puts the K = 64 LM step 7.0-8.2e-5 from the fp32
// oracle's on four windows (r03: tests/tools/delta_probe.py; one record per workgroup: 8.8-9.8e-5) for +2 % of the kernel
static int photo_flush_for_runs(int tpb)
{
  int flush = 0;
  if (tpb >= 6)
    flush = tpb % 4 == 0 ? 4 : (tpb % 5 == 0 ? 5 : (tpb % 3 == 0 ? 3 : 0)); // (r06: run lengths 6, 9, 10, 15)
  if (const char *e = getenv("SAGE_PHOTO_FLUSH"))
    flush = std::max(0, atoi(e));
  return flush;
}

// (re)build the photometric work list for runs of `tpb` sub-tiles and upload it; the partial-record buffer grows to fit
static int window_plan_photo_runs(SageWindow *w, int tpb, int flush)
{
  WorkList wp;
  wp.build(w->Nedge, tpb, nullptr, flush);
  int rc;
  w->n_work_p = (int)wp.work.size();
  w->tpb_p = wp.tiles_per_block;
  w->flush_p = wp.flush;
  w->n_rec_p = wp.n_records;
  if ((rc = upload(w->work_p, wp.work, w->stream)) || (rc = upload(w->first_p, wp.edge_first, w->stream)) ||
      (rc = upload(w->tiles_p, wp.edge_tiles, w->stream)) || (rc = upload(w->rec_first_p, wp.rec_first, w->stream)) ||
      (rc = upload(w->rec_count_p, wp.rec_count, w->stream)))
    return rc;
  SAGE_HIP(hipStreamSynchronize(w->stream)); // (the host vectors go out of scope)
  return w->part_p.reserve(std::max<size_t>(1, std::max(w->n_work_p, w->n_rec_p)) * photo_partial_floats(w->cfg.CS) * sizeof(float));
}

extern "C" int sage_window_finalize(SageWindow *w)
{
  if (!w || w->finalized || w->K < 1)
    return SAGE_E_INVALID;
  const SageWindowConfig &c = w->cfg;
  const int CS = c.CS, FS = c.FS, K = w->K;
  const int H = (int)c.pyr.cam[0].h, W = (int)c.pyr.cam[0].w, HW = H * W;
  int rc;
  // ---- variables & depth buffers
  for (int s = 0; s < 2; ++s)
  {
    if ((rc = w->vars[s].reserve((size_t)K * w->VS * sizeof(float))))
      return rc;
    if ((rc = window_upload_vars(w, s)))
      return rc;
  }
  if ((rc = w->dpt.reserve((size_t)K * HW * sizeof(float))) || (rc = w->dgrad.reserve((size_t)K * 2 * HW * sizeof(float))))
    return rc;
  // ---- local edges.  A link is two directed edges per factor type (a -> b, b -> a: global ids 2l, 2l + 1).  r05: rank r owns
  //      the contiguous range [r*2n/world, (r+1)*2n/world) of the DIRECTED edges -- the two directions of a link may sit on
  //      two ranks (both factor types of a direction stay together: the merged linearize pairs them).  With whole links, 42
  //      links on 8 ranks are 5 or 6 per rank, 20 % imbalance (BASELINE config 4: the 6-link ranks set the job's pace at 4.6x
  //      where the 5-link ranks reach 6x); 84 directed edges are 10 or 11.  Links are added keyframe by keyframe, so a
  //      contiguous range touches ~K/world + (back links) keyframes: only those need depth maps on this rank.  The
  //      domain-decomposed solve (shard_solve.cpp) derives its domains from whole links: windows that will use it
  //      (SAGE_SHARD_SCHUR / K >= 256) keep the link granularity.
  w->local_links.clear();
  w->local_edges.clear();
  {
    const long long nl = (long long)w->links.size();
    bool by_link = sage::env_flag("SAGE_SHARD_BY_LINK");
    if (w->world > 1)
    {
      const char *e = getenv("SAGE_SHARD_SCHUR");
      by_link = by_link || (e ? atoi(e) != 0 : w->K >= 256);
    }
    if (by_link)
    {
      const int lo = (int)(nl * w->rank / w->world), hi = (int)(nl * (w->rank + 1) / w->world);
      for (int l = lo; l < hi; ++l)
      {
        w->local_edges.push_back(2 * l);
        w->local_edges.push_back(2 * l + 1);
      }
    }
    else
    {
      const int lo = (int)(2 * nl * w->rank / w->world), hi = (int)(2 * nl * (w->rank + 1) / w->world);
      for (int ge = lo; ge < hi; ++ge)
        w->local_edges.push_back(ge);
    }
    for (int ge : w->local_edges)
      if (w->local_links.empty() || w->local_links.back() != ge / 2)
        w->local_links.push_back(ge / 2);
  }
  std::vector<char> needed(K, 0);
  for (int l : w->local_links)
    needed[w->links[l].first] = needed[w->links[l].second] = 1;
  w->n_depth = 0;
  for (int k = 0; k < K; ++k)
    w->n_depth += needed[k];
  for (int s = 0; s < 2; ++s)
  {
    std::vector<DepthItem> items;
    for (int k = 0; k < K; ++k)
    {
      if (!needed[k])
        continue;
      const float *vp = w->vars[s].as<float>() + (size_t)k * w->VS;
      items.push_back(DepthItem{w->views[k].bias, w->views[k].basis, vp + 13, vp + 12,
                                w->dpt.as<float>() + (size_t)k * HW, w->dgrad.as<float>() + (size_t)k * 2 * HW});
    }
    if ((rc = upload(w->depth_items[s], items, w->stream)))
      return rc;
    SAGE_HIP(hipStreamSynchronize(w->stream));
  }
  // ---- engine-internal relayout, once per keyframe: [FS][P] -> [FS/4][P][4] for feat, grad-x, grad-y
  const size_t plane_f = (size_t)FS * c.pyr.P;
  if ((rc = w->pk.reserve((size_t)K * 3 * plane_f * sizeof(float))))
    return rc;
  // r05: every texel of level l also carries sqrt(w_l) (feat, fx d/dx, fy d/dy -- and with feat the pre-sampled source
  // features below): the products the kernels sum (h h^T, h r, r^2) then hold the level weight of
  // photometric_factor_kernels.cpp:1143-1149 by themselves -- no per-step weighting in the kernels' sampling loops
  float lvl_scale[SAGE_MAX_LEVELS] = {};
  for (int l = 0; l < c.pyr.levels; ++l)
  {
    if (!(c.photo_weights[l] >= 0.f))
      return SAGE_E_INVALID;
    lvl_scale[l] = std::sqrt(c.photo_weights[l]);
  }
  for (int k = 0; k < K; ++k)
  {
    float *base = w->pk.as<float>() + (size_t)k * 3 * plane_f;
    SAGE_HIP(launch_repack_groups(w->stream, base, w->views[k].feat_pyr, FS, c.pyr.P, 0, &c.pyr, lvl_scale));
    SAGE_HIP(launch_repack_groups(w->stream, base + plane_f, w->views[k].grad_pyr, FS, c.pyr.P, 1, &c.pyr, lvl_scale));
    SAGE_HIP(launch_repack_groups(w->stream, base + 2 * plane_f, w->views[k].grad_pyr + plane_f, FS, c.pyr.P, 2, &c.pyr, lvl_scale));
  }
  // ---- sampled locations: validated (the kernels index depth maps / basis rows with them unchecked) and relaid in
  //      raster order (engine-owned copies; see producers.hip: the sums are order independent, the L1 is not)
  {
    if ((int)w->user_samples.size() != K) // (a retried finalize must not sort the sorted copies onto themselves)
    {
      w->user_samples.resize(K);
      w->user_n.resize(K);
      for (int k = 0; k < K; ++k)
      {
        w->user_samples[k] = {w->views[k].loc1d, w->views[k].homo};
        w->user_n[k] = w->views[k].N;
      }
    }
    for (int k = 0; k < K; ++k)
      w->views[k].N = w->user_n[k];
    // (capacity per keyframe: its samples, or -- tile-padded order -- every 8 x 8 tile of the image with all 64 slots)
    const size_t cap_tiles = (size_t)(((int)c.pyr.cam[0].w + 7) / 8) * (size_t)(((int)c.pyr.cam[0].h + 7) / 8) * 64;
    std::vector<size_t> soff(K + 1, 0);
    int max_n = 0;
    for (int k = 0; k < K; ++k)
    {
      soff[k + 1] = soff[k] + std::max<size_t>(std::max(1, w->views[k].N), cap_tiles);
      max_n = std::max(max_n, w->views[k].N);
    }
    if ((rc = w->sorted_loc.reserve(soff[K] * sizeof(int64_t))) || (rc = w->sorted_homo.reserve(soff[K] * 3 * sizeof(float))))
      return rc;
    std::vector<SortItem> items(K);
    for (int k = 0; k < K; ++k)
      items[k] = SortItem{reinterpret_cast<const long long *>(w->user_samples[k].first), w->user_samples[k].second,
                          w->sorted_loc.as<long long>() + soff[k], w->sorted_homo.as<float>() + 3 * soff[k],
                          w->views[k].N};
    DevBuf d_items, d_mark, d_status, d_tiles, d_pad;
    std::vector<int> status((size_t)2 * K, 0), tiles_nonempty(K, 0), pad_flags(K, 0);
    rc = upload(d_items, items, w->stream);
    if (!rc)
      rc = d_mark.reserve((size_t)K * HW * sizeof(int));
    if (!rc)
      rc = d_status.reserve((size_t)2 * K * sizeof(int));
    if (!rc)
      rc = d_tiles.reserve((size_t)K * sizeof(int));
    hipError_t he = hipSuccess;
    {
      // walk order of the samples: image tiles of 8 x 8 pixels -- a wave's 64 consecutive samples then warp to a compact
      // footprint in every destination keyframe, which is what the LDS-staged sampler of the photometric linearize
      // needs (photo_kernels.hip).  SAGE_SAMPLE_TILE=WxH picks another tile, 0x0 the raster walk.
      static const std::pair<int, int> tile = [] {
        int tw = 8, th = 8;
        if (const char *e = getenv("SAGE_SAMPLE_TILE"))
          if (sscanf(e, "%dx%d", &tw, &th) != 2 || tw < 1 || th < 1)
            tw = th = 0;
        return std::make_pair(tw, th);
      }();
      if (!rc)
        he = launch_sort_locations(w->stream, d_items.as<SortItem>(), K, max_n, HW, d_mark.as<int>(), d_status.as<int>(),
                                   (int)c.pyr.cam[0].w, tile.first, tile.second, nullptr, d_tiles.as<int>());
      if (!rc && he == hipSuccess)
        he = hipMemcpyAsync(status.data(), d_status.p, status.size() * sizeof(int), hipMemcpyDeviceToHost, w->stream);
      if (!rc && he == hipSuccess)
        he = hipMemcpyAsync(tiles_nonempty.data(), d_tiles.p, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, w->stream);
      if (!rc && he == hipSuccess)
        he = hipStreamSynchronize(w->stream);
      // r06 -- tile-padded order (producers.hip, order_locations_kernel): keyframes whose sampled tiles are not all full are relaid
      // with every sampled tile's 64 slots (holes = location -1), so that a wave of the kernels is one tile whatever the mask looks
      // like.  Only where it is cheap: slots <= 1.25 x samples (a dense mask with a ragged edge: a few percent; a sparse random
      // sample set would grow several-fold and keeps the compact order).  SAGE_SAMPLE_PAD=0 turns it off.
      bool any_pad = false;
      if (!rc && he == hipSuccess && tile.first * tile.second == 64 && !(getenv("SAGE_SAMPLE_PAD") && atoi(getenv("SAGE_SAMPLE_PAD")) == 0))
        for (int k = 0; k < K; ++k)
        {
          const long long slots = 64ll * tiles_nonempty[k];
          if (status[2 * k] == 0 && status[2 * k + 1] == w->views[k].N && slots > w->views[k].N && 4 * slots <= 5ll * w->views[k].N)
          {
            pad_flags[k] = 1;
            any_pad = true;
          }
        }
      if (any_pad)
      {
        std::vector<int> status2((size_t)2 * K, 0);
        rc = upload(d_pad, pad_flags, w->stream);
        if (!rc)
          he = launch_sort_locations(w->stream, d_items.as<SortItem>(), K, max_n, HW, d_mark.as<int>(), d_status.as<int>(),
                                     (int)c.pyr.cam[0].w, tile.first, tile.second, d_pad.as<int>(), nullptr, true);
        if (!rc && he == hipSuccess)
          he = hipMemcpyAsync(status2.data(), d_status.p, status2.size() * sizeof(int), hipMemcpyDeviceToHost, w->stream);
        if (!rc && he == hipSuccess)
          he = hipStreamSynchronize(w->stream);
        for (int k = 0; k < K && !rc && he == hipSuccess; ++k)
          if (pad_flags[k])
          {
            if (status2[2 * k + 1] != 64 * tiles_nonempty[k])
              rc = SAGE_E_STATE;
            else
              w->views[k].N = status2[2 * k + 1]; // slots, holes included
          }
      }
    }
    d_items.release();
    d_mark.release();
    d_status.release();
    d_tiles.release();
    d_pad.release();
    if (rc)
      return rc;
    if (he != hipSuccess)
      return (int)he;
    for (int k = 0; k < K; ++k)
    {
      if (status[2 * k] > 0)
        return SAGE_E_INVALID; // a location outside the image
      if (status[2 * k + 1] != w->user_n[k])
        continue; // (a pixel sampled twice: the compaction dropped a sample -> keep the caller's order)
      w->views[k].loc1d = reinterpret_cast<const int64_t *>(items[k].loc_out);
      w->views[k].homo = items[k].homo_out;
    }
  }
  // ---- pose-independent pre-sampled source features, once per keyframe
  std::vector<size_t> f0s_off(K + 1, 0);
  for (int k = 0; k < K; ++k)
    f0s_off[k + 1] = f0s_off[k] + (size_t)c.pyr.levels * FS * std::max(1, w->views[k].N);
  if ((rc = w->f0s.reserve(f0s_off[K] * sizeof(float))))
    return rc;
  for (int k = 0; k < K; ++k)
    SAGE_HIP(launch_presample_source(w->stream, w->f0s.as<float>() + f0s_off[k],
                                     w->pk.as<float>() + (size_t)k * 3 * plane_f, w->views[k].homo, w->views[k].N, FS,
                                     c.pyr));
  // ---- local edges
  w->n_edges = (int)w->local_edges.size();
  // merged linearize (LaunchCommon::merge_geo_weight): the geometric kernel's per-pixel hand-over to the photometric one
  std::vector<size_t> px_off((size_t)w->n_edges + 1, 0);
  for (int e = 0; e < w->n_edges; ++e)
  {
    const int ge = w->local_edges[e], l = ge / 2;
    const int k0 = ge % 2 == 0 ? w->links[l].first : w->links[l].second; // the source keyframe of the direction
    px_off[(size_t)e + 1] = px_off[e] + (size_t)std::max(1, w->views[k0].N);
  }
  w->merge_ok = c.use_photo && c.use_geo && c.geo_weight > 0.f && !sage::env_flag("SAGE_NO_MERGE");
  if (w->merge_ok)
  {
    if ((rc = w->geo_px.reserve(std::max<size_t>(1, px_off[w->n_edges]) * 4 * sizeof(float))))
      return rc;
    SAGE_HIP(hipMemsetAsync(w->geo_px.p, 0, std::max<size_t>(1, px_off[w->n_edges]) * 4 * sizeof(float), w->stream));
  }
  std::vector<LinkEdges> le(w->links.size(), LinkEdges{-1, -1});
  std::vector<int> Nedge(w->n_edges);
  std::vector<std::vector<AdjEntry>> adjv(K);
  double residuals = 0, bytes = 0;
  const double rho = (double)c.pyr.P / (double)HW;
  for (int s = 0; s < 2; ++s)
  {
    std::vector<PhotoEdge> pt(w->n_edges);
    std::vector<GeoEdge> gt(w->n_edges);
    for (int e = 0; e < w->n_edges; ++e)
    {
      const int l = w->local_edges[e] / 2;
      const int ab[2] = {w->links[l].first, w->links[l].second};
      {
        const int dir = w->local_edges[e] % 2;
        const int k0 = ab[dir], k1 = ab[1 - dir];
        const SageKeyframeView &v0 = w->views[k0], &v1 = w->views[k1];
        const float *x0 = w->vars[s].as<float>() + (size_t)k0 * w->VS;
        const float *x1 = w->vars[s].as<float>() + (size_t)k1 * w->VS;
        PhotoEdge pe{};
        pe.feat0 = v0.feat_pyr; pe.feat1 = v1.feat_pyr; pe.grad1 = v1.grad_pyr; pe.bias0 = v0.bias;
        pe.feat0_pk = w->pk.as<float>() + (size_t)k0 * 3 * plane_f;
        pe.feat1_pk = w->pk.as<float>() + (size_t)k1 * 3 * plane_f;
        pe.f0s = w->f0s.as<float>() + f0s_off[k0];
        pe.dpt0 = w->dpt.as<float>() + (size_t)k0 * HW;
        pe.dpt1_geo = (c.use_photo && c.use_geo) ? w->dpt.as<float>() + (size_t)k1 * HW : nullptr;
        pe.geo_loss = w->link_geo_loss[l];
        pe.geo_px = w->merge_ok ? w->geo_px.as<float>() + 4 * px_off[e] : nullptr;
        pe.basis0 = v0.basis; pe.mask1 = c.mask_dev; pe.homo = v0.homo; pe.loc = v0.loc1d; pe.loc_is_i64 = 1;
        pe.R0 = x0; pe.t0 = x0 + 9; pe.R1 = x1; pe.t1 = x1 + 9; pe.R10 = nullptr; pe.t10 = nullptr;
        pe.code0 = x0 + 13; pe.scale0 = x0 + 12; pe.N = v0.N;
        pt[e] = pe;
        GeoEdge ge{};
        ge.dpt0 = w->dpt.as<float>() + (size_t)k0 * HW;
        ge.bias0 = v0.bias; ge.basis0 = v0.basis; ge.dpt1 = w->dpt.as<float>() + (size_t)k1 * HW;
        ge.dgrad1 = w->dgrad.as<float>() + (size_t)k1 * 2 * HW; ge.basis1 = v1.basis; ge.mask1 = c.mask_dev;
        ge.homo = v0.homo; ge.loc = v0.loc1d; ge.loc_is_i64 = 1;
        ge.R0 = x0; ge.t0 = x0 + 9; ge.R1 = x1; ge.t1 = x1 + 9; ge.R10 = nullptr; ge.t10 = nullptr;
        ge.code0 = x0 + 13; ge.scale0 = x0 + 12; ge.scale1 = x1 + 12; ge.N = v0.N;
        ge.loss_param = w->link_geo_loss[l];
        ge.px_out = w->merge_ok ? w->geo_px.as<float>() + 4 * px_off[e] : nullptr;
        gt[e] = ge;
        if (s == 0)
        {
          Nedge[e] = v0.N;
          if (c.use_photo)
          {
            adjv[k0].push_back(AdjEntry{0, e, 0});
            adjv[k1].push_back(AdjEntry{0, e, 1});
            residuals += (double)c.pyr.levels * w->user_n[k0] * FS; // (the caller's samples: holes of a padded order do not count)
            bytes += (double)w->user_n[k0] * 4.0 * (4.0 * FS * rho + CS + 6.0);
          }
          if (c.use_geo)
          {
            adjv[k0].push_back(AdjEntry{1, e, 0});
            adjv[k1].push_back(AdjEntry{1, e, 1});
            residuals += (double)w->user_n[k0];
            bytes += (double)w->user_n[k0] * 4.0 * (2.0 * CS + 9.0);
          }
        }
      }
      if (s == 0)
        (w->local_edges[e] % 2 == 0 ? le[l].e_ab : le[l].e_ba) = e;
    }
    if ((rc = upload(w->ptab[s], pt, w->stream)) || (rc = upload(w->gtab[s], gt, w->stream)))
      return rc;
    SAGE_HIP(hipStreamSynchronize(w->stream));
  }
  w->residuals_per_lin = residuals;
  w->bytes_per_lin = bytes;
  // ---- work lists
  WorkList wl;
  {
    // geometric linearize: the two wave groups of a workgroup alternate over its sub-tiles (geo_kernels.hip), so a
    // workgroup wants an even, longish run of them: the pipeline fill/drain costs one half-step per workgroup
    long long total = 0;
    for (int n : Nedge)
      total += (n + kTile - 1) / kTile;
    // (r05, one rank's shard of the K = 64 window at world 8 = 2.9 k sub-tiles: runs of 8 leave 362 workgroups for 256 CUs --
    //  85 us; runs of 4: 74 us, 2: 77 us; the full window's 23 k sub-tiles keep runs of 16)
    int tpb = total >= 8192 ? 16 : (total >= 4096 ? 8 : (total >= 512 ? 4 : 2));
    if (const char *e = getenv("SAGE_GEO_TPB"))
      tpb = std::max(1, atoi(e));
    wl.build(Nedge, tpb);
  }
  w->n_work_g = (int)wl.work.size();
  w->tpb_g = wl.tiles_per_block;
  if ((rc = upload(w->work_g, wl.work, w->stream)) || (rc = upload(w->first_g, wl.edge_first, w->stream)) ||
      (rc = upload(w->tiles_g, wl.edge_tiles, w->stream)))
    return rc;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  {
    // photometric work list: its own sub-tile run length
    long long total = 0;
    for (int n : Nedge)
      total += (n + kTile - 1) / kTile;
    // run length of a workgroup (sub-tiles it walks: prologue amortisation, vertical L1/L2 reuse between its bands) and,
    // separately, the number of sub-tiles it accumulates in fp32 before a partial record goes out to the double sums
    // (MFMA chains of 64 fmaf per sub-tile and accumulator): the LM step's distance from the exact step grows with the
    // chain length (K = 64 window, tests/tools/tpb_noise_probe.py: 8 -> 2.1e-4, 4 -> 1.2e-4, 2 -> 7.6e-5, 1 -> 4.9e-5 rel-L2;
    // the fp32 oracle itself sits at 5.5e-5).  Records every 2 sub-tiles keep the step inside the 1e-4 parity bar.
    // (r03, one rank's shard of the K = 64 window at world 8 / 4 = 2.9 k / 5.8 k sub-tiles: runs of 4 / 8 are 19 % / 8 % faster
    //  than the 1 / 2 the first heuristic picked; >= ~3 workgroups per CU stay in flight)
    int tpb = total >= 4096 ? 8 : (total >= 1536 ? 4 : (total >= 768 ? 2 : 1));
    // (r05, one rank's shard of BASELINE config 4 at world 8 -- FS = 32, 10 or 11 edges of 252 sub-tiles: with runs of 4 the
    //  11-edge shard's 693 workgroups take 0.231 ms where the 10-edge shard's 630 take 0.163; runs of 6: 0.187 / 0.163, runs
    //  of 7 / 9 / 12 worse for both.  At FS = 16 the same range wants runs of 4 (K = 64 shard: 0.084 ms; 5-8: 0.11-0.12))
    if (FS == 32 && total >= 1536 && total < 4096)
      tpb = 6;
    {
      // even runs: an edge of T sub-tiles is cut into ceil(T / tpb) workgroups of ceil(T / that) sub-tiles each -- with
      // T = 12 (3072 samples: the reference's default) runs of 8 leave a half-length second workgroup per edge and the
      // linearize 25 % slower than runs of 6 (BASELINE config 5: 1.85 -> 1.39 ms, error pass 0.49 -> 0.39 ms)
      std::vector<int> tiles;
      for (int n : Nedge)
        tiles.push_back((n + kTile - 1) / kTile);
      if (!tiles.empty())
      {
        std::nth_element(tiles.begin(), tiles.begin() + tiles.size() / 2, tiles.end());
        const int T = std::max(1, tiles[tiles.size() / 2]); // the typical edge
        const int nwg = (T + tpb - 1) / tpb, rem = T % tpb;
        if (rem != 0 && 4 * rem < 3 * tpb) // (a nearly full last run is left alone: T = 63 stays at runs of 8 -- 7 x 9 and
          tpb = (T + nwg - 1) / nwg;       //  9 x 7 measured 5-7 % slower on the headline window)
      }
    }
    {
      // r06 -- runs per edge a multiple of 8.  Workgroup b runs on XCD b % 8 and every XCD has its own L2: with 8 m runs per edge,
      // run j of EVERY edge lands on XCD j % 8 -- the same band of the image, whose destination texels the XCD's L2 then serves to
      // the next edges that share the keyframe.  The BASELINE sizes have it by luck (63 sub-tiles = 8 runs of 8, config 4: 32
      // runs); on the same window 13 / 11 / 7 runs per edge (SAGE_PHOTO_TPB = 5 / 6 / 10) cost the photometric linearize 20-30 %
      // and the error pass 50 % (profiles/r06_kernel_ab_experiments.txt s11).  Padding an edge to 8 m runs with empty work items
      // is no way out (the XCDs that get the real runs then carry twice the load: +60 %): the run LENGTH is chosen instead,
      // among lengths that leave a record cadence of 3-5 sub-tiles.
      std::vector<int> tiles;
      for (int n : Nedge)
        tiles.push_back((n + kTile - 1) / kTile);
      if (!tiles.empty())
      {
        std::nth_element(tiles.begin(), tiles.begin() + tiles.size() / 2, tiles.end());
        const int T = std::max(1, tiles[tiles.size() / 2]);
        // (r06, later: the edges that share a keyframe -- as destination or as source -- are TWO apart in the launch order, so a run
        //  count of 0 mod 4 already aligns them: taken when no candidate gives 0 mod 8 -- a 192 x 256 window, 165 sub-tiles per
        //  edge: runs of 8 = 21 per edge, runs of 6 = 28: linearize -18 %, error pass -27 %.  sage_window_tune_runs measures.)
        if (T >= 48 && ((T + tpb - 1) / tpb) % 8 != 0)
        {
          int pick = 0;
          for (int mod : {8, 4})
          {
            for (int t : {8, 9, 10, 12, 6, 15, 16, 20})
              if (((T + t - 1) / t) % mod == 0)
              {
                pick = t;
                break;
              }
            if (pick)
              break;
          }
          if (pick)
            tpb = pick;
        }
      }
    }
    if (const char *e = getenv("SAGE_PHOTO_TPB"))
      tpb = std::max(1, atoi(e));
    const int flush = photo_flush_for_runs(tpb);
    // (r06, VERDICT r5 item 4 -- measured and dropped: the runs of an edge dealt to the XCDs in contiguous BANDS of image strips
    //  (workgroup b runs on XCD b % 8; band x = runs [x R / 8, (x + 1) R / 8), all XCDs on the same edge at the same time), so that
    //  vertically adjacent strips share their destination halo in ONE L2: config 4 photometric linearize 1.202 -> 1.191 ms, error
    //  pass 0.638 -> 0.624, K = 64 unchanged (0.622 / 0.623) -- profiles/r06_kernel_ab_experiments.txt)
    // (r05, VERDICT r4 item 6 -- measured and dropped: the linearize's work items in destination-keyframe-major order, the
    //  runs of the <= 6 edges that sample one keyframe interleaved strip by strip, so that the workgroups in flight want ONE
    //  packed pyramid at a time: config 4 photometric linearize 1.195 -> 1.356 ms, K = 64 0.677 -> 0.820 ms, config 2 0.157 ->
    //  0.192 ms.  Consecutive runs of ONE edge share their source streams and overlap in the destination; the link order
    //  (i-1,i) (i-2,i) (i-3,i), both directions adjacent, already keeps keyframe i in three of six consecutive edges.)
    w->Nedge = Nedge;
    w->tpb_heur = tpb;
    if ((rc = window_plan_photo_runs(w, tpb, flush)))
      return rc;
  }
  SAGE_HIP(hipStreamSynchronize(w->stream));
  const size_t Dp = 13 + CS, Dg = 14 + 2 * CS;
  const size_t ne = std::max(1, w->n_edges);
  if ((rc = w->part_p.reserve(std::max<size_t>(1, std::max(w->n_work_p, w->n_rec_p)) * photo_partial_floats(CS) * sizeof(float))) ||
      (rc = w->part_g.reserve(std::max<size_t>(1, w->n_work_g) * geo_partial_floats(CS) * sizeof(float))) ||
      (rc = w->AtA_p.reserve(ne * Dp * Dp * sizeof(float))) || (rc = w->Atb_p.reserve(ne * Dp * sizeof(float))) ||
      (rc = w->stats_p.reserve(ne * 2 * sizeof(float))) || (rc = w->AtA_g.reserve(ne * Dg * Dg * sizeof(float))) ||
      (rc = w->Atb_g.reserve(ne * Dg * sizeof(float))) || (rc = w->stats_g.reserve(ne * 2 * sizeof(float))))
    return rc;
  if ((rc = w->wide_p.reserve(ne * (Dp * Dp + Dp) * sizeof(double))) ||
      (rc = w->wide_g.reserve(ne * (Dg * Dg + Dg) * sizeof(double))))
    return rc;
  // ---- adjacency for the assembly
  std::vector<int32_t> adj_start(K + 1, 0);
  std::vector<AdjEntry> adj;
  for (int k = 0; k < K; ++k)
  {
    adj_start[k] = (int32_t)adj.size();
    adj.insert(adj.end(), adjv[k].begin(), adjv[k].end());
  }
  adj_start[K] = (int32_t)adj.size();
  if ((rc = upload(w->adj_start, adj_start, w->stream)) || (rc = upload(w->adj, adj, w->stream)) ||
      (rc = upload(w->link_edges, le, w->stream)))
    return rc;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  if ((rc = w->packed.reserve(sage_window_packed_count(w) * sizeof(double))) || (rc = w->errbuf.reserve(4 * sizeof(double))))
    return rc;
  {
    // the output blocks this rank's edges contribute to (everything, on a single-rank window)
    std::vector<int32_t> ids;
    for (int k = 0; k < K; ++k)
      if (!adjv[k].empty())
        ids.push_back(k);
    for (size_t l = 0; l < le.size(); ++l)
      if (le[l].e_ab >= 0 || le[l].e_ba >= 0)
        ids.push_back(K + (int32_t)l);
    ids.push_back(K + (int32_t)w->links.size()); // the tail
    w->n_asm_blocks = (int)ids.size();
    if ((rc = upload(w->asm_blocks, ids, w->stream)))
      return rc;
    SAGE_HIP(hipStreamSynchronize(w->stream));
  }
  SAGE_HIP(hipMemsetAsync(w->packed.p, 0, sage_window_packed_count(w) * sizeof(double), w->stream));
  SAGE_HIP(hipMemsetAsync(w->errbuf.p, 0, 4 * sizeof(double), w->stream));
  if (!w->h_err)
  {
    SAGE_HIP(hipHostMalloc(reinterpret_cast<void **>(&w->h_err), 16 * sizeof(double), hipHostMallocDefault));
    std::memset(w->h_err, 0, 16 * sizeof(double));
  }
  w->host_packed.assign(sage_window_packed_count(w), 0.0);
  w->delta.assign((size_t)K * w->B, 0.0);
  rc = solver_create(&w->solver, K, w->B, w->VS, w->links, w->stream);
  if (rc != SAGE_OK && rc != SAGE_E_UNSUPPORTED)
    return rc;
  if (w->world > 1)
  {
    // domain-decomposed solve for sharded windows: on by request (SAGE_SHARD_SCHUR=1) or for long windows, where the
    // replicated factorisation of all K keyframes dominates the iteration (DESIGN s7: K = 512 on 8 ranks: 5x less solve)
    const char *e = getenv("SAGE_SHARD_SCHUR");
    const bool want = e ? atoi(e) != 0 : w->K >= 256;
    if (want)
    {
      std::vector<int32_t> lk(2 * w->links.size());
      for (size_t l = 0; l < w->links.size(); ++l)
      {
        lk[2 * l] = w->links[l].first;
        lk[2 * l + 1] = w->links[l].second;
      }
      if ((rc = sage_shard_plan_create(w->K, (int)w->links.size(), lk.data(), w->B, w->rank, w->world, &w->shard)))
        return rc;
      const size_t ns = sage_shard_sep_count(w->shard);
      w->h_sep.assign(ns, 0.0);
      if ((rc = w->sepbuf.reserve(ns * sizeof(double))))
        return rc;
      w->host_packed.resize(sage_window_packed_count(w));
    }
  }
  w->finalized = true;
  if (const char *e = getenv("SAGE_AUTOTUNE"))
    if (atoi(e) != 0)
    {
      const int rct = sage_window_tune_runs(w, nullptr, nullptr, nullptr, nullptr);
      if (rct)
        return rct;
    }
  return SAGE_OK;
}

int window_local_edge(const SageWindow *w, int global_edge)
{
  const auto it = std::lower_bound(w->local_edges.begin(), w->local_edges.end(), global_edge);
  return it != w->local_edges.end() && *it == global_edge ? (int)(it - w->local_edges.begin()) : -1;
}

static LaunchCommon window_lc(SageWindow *w, bool photo, bool photo_linearize = false)
{
  LaunchCommon lc{};
  lc.work = (photo ? w->work_p : w->work_g).as<WorkItem>();
  lc.edge_first = (photo ? w->first_p : w->first_g).as<int32_t>();
  lc.edge_tiles = (photo ? w->tiles_p : w->tiles_g).as<int32_t>();
  lc.n_work = photo ? w->n_work_p : w->n_work_g;
  lc.n_edges = w->n_edges;
  lc.partials = photo ? w->part_p.as<float>() : w->part_g.as<float>();
  lc.tiles_per_block = photo ? w->tpb_p : w->tpb_g;
  lc.packed = photo;
  if (photo_linearize && w->flush_p > 0)
  {
    // the linearize (and its per-edge finalize) count partial RECORDS, the error pass work items
    lc.edge_first = w->rec_first_p.as<int32_t>();
    lc.edge_tiles = w->rec_count_p.as<int32_t>();
    lc.flush = w->flush_p;
  }
  return lc;
}

static AssembleParams window_assemble_params(SageWindow *w)
{
  const SageWindowConfig &c = w->cfg;
  AssembleParams ap{};
  const bool has = w->n_edges > 0;
  ap.AtA_p = (has && c.use_photo) ? w->AtA_p.as<float>() : nullptr;
  ap.Atb_p = w->Atb_p.as<float>();
  ap.stats_p = (has && c.use_photo) ? w->stats_p.as<float>() : nullptr;
  ap.AtA_g = (has && c.use_geo) ? w->AtA_g.as<float>() : nullptr;
  ap.wide_p = (has && c.use_photo) ? w->wide_p.as<double>() : nullptr;
  ap.wide_g = (has && c.use_geo) ? w->wide_g.as<double>() : nullptr;
  ap.Atb_g = w->Atb_g.as<float>();
  ap.stats_g = (has && c.use_geo) ? w->stats_g.as<float>() : nullptr;
  ap.adj_start = w->adj_start.as<int32_t>();
  ap.adj = w->adj.as<AdjEntry>();
  ap.links = w->link_edges.as<LinkEdges>();
  ap.packed = w->packed.as<double>();
  ap.tail_mirror = w->world == 1 && !w->allreduce ? w->h_err : nullptr; // (a reduced tail is mirrored after the sum)
  ap.K = w->K;
  ap.nlinks = (int)w->links.size();
  ap.CS = c.CS;
  ap.n_edges_p = w->n_edges;
  ap.n_edges_g = w->n_edges;
  ap.split = 1;
  ap.blocks = nullptr;
  return ap;
}

// linearize every local edge at variable set `set` (0 = current estimate, 1 = candidate) and assemble the packed system
// dst: where the packed system is assembled (default: w->packed); local_blocks: only the blocks this rank's edges touch
// (dst then must hold zeros everywhere else: packed_loc)
// merge: the merged linearize of the two factor types (LaunchCommon::merge_geo_weight) -- the per-edge results are then
// mixed (sage_window_get_edge), the assembled system is the same
int window_linearize_set(SageWindow *w, int set, double *dst, bool local_blocks, bool merge)
{
  if (!w || !w->finalized)
    return SAGE_E_STATE;
  const SageWindowConfig &c = w->cfg;
  const int H = (int)c.pyr.cam[0].h, W = (int)c.pyr.cam[0].w;
  if (w->n_edges > 0)
  {
    // depth maps of every keyframe at the current variables: both factor types read their sample depths from them
    // (an accepted candidate's maps from the error pass are still valid: only the gradients are missing then)
    const bool have_depth = w->dpt_set == set;
    SAGE_HIP(launch_depth_batch(w->stream, c.CS, w->depth_items[set].as<DepthItem>(), w->n_depth, H, W, !have_depth,
                                !(have_depth && w->dgrad_valid)));
    w->dpt_set = set;
    w->dgrad_valid = true;
    // main kernels only (stage 1), then ONE finalize launch for both factor types (window_finalize_kernel)
    LaunchCommon lcg = window_lc(w, false), lcp = window_lc(w, true, true);
    lcg.stage = 1;
    lcp.stage = 1;
    merge = merge && w->merge_ok;
    lcg.merge_geo_weight = lcp.merge_geo_weight = merge ? c.geo_weight : 0.f;
    if (c.use_geo)
    {
      EdgeOut out{w->AtA_g.as<float>(), w->Atb_g.as<float>(), w->stats_g.as<float>(), w->wide_g.as<double>()};
      prof_attach(w, 1, lcg);
      SAGE_HIP(launch_geo_linearize(w->stream, c.CS, nullptr, w->gtab[set].as<GeoEdge>(), lcg, c.pyr.cam[0], c.eps,
                                    c.geo_loss_param, c.geo_weight, out));
    }
    if (c.use_photo)
    {
      EdgeOut out{w->AtA_p.as<float>(), w->Atb_p.as<float>(), w->stats_p.as<float>(), w->wide_p.as<double>()};
      prof_attach(w, 0, lcp);
      SAGE_HIP(launch_photo_linearize(w->stream, c.CS, c.FS, nullptr, w->ptab[set].as<PhotoEdge>(), lcp, c.pyr,
                                      c.photo_weights, c.eps, out));
    }
    WindowFinalizeParams fp{};
    fp.n_p = c.use_photo ? w->n_edges : 0;
    fp.n_g = c.use_geo ? w->n_edges : 0;
    fp.ph.table = w->ptab[set].as<PhotoEdge>();
    fp.ph.edge_first = lcp.edge_first; fp.ph.edge_tiles = lcp.edge_tiles; fp.ph.partials = lcp.partials;
    fp.ph.AtA = w->AtA_p.as<float>(); fp.ph.Atb = w->Atb_p.as<float>(); fp.ph.stats = w->stats_p.as<float>();
    fp.ph.wide = w->wide_p.as<double>();
    for (int l = 0; l < c.pyr.levels; ++l)
      fp.ph.wsum += c.photo_weights[l];
    fp.ge.table = w->gtab[set].as<GeoEdge>();
    fp.ge.edge_first = lcg.edge_first; fp.ge.edge_tiles = lcg.edge_tiles; fp.ge.partials = lcg.partials;
    fp.ge.AtA = w->AtA_g.as<float>(); fp.ge.Atb = w->Atb_g.as<float>(); fp.ge.stats = w->stats_g.as<float>();
    fp.ge.wide = w->wide_g.as<double>();
    fp.ge.weight = c.geo_weight;
    if (merge)
    {
      fp.ge.photo_partials = lcp.partials;
      fp.ge.photo_rec_first = lcp.edge_first;
      fp.ge.photo_rec_count = lcp.edge_tiles;
    }
    if (c.CS == 32)
      hipLaunchKernelGGL((window_finalize_kernel<32>), dim3(fp.n_p + fp.n_g), dim3(kFinalizeBlock), 0, w->stream, fp);
    else
      hipLaunchKernelGGL((window_finalize_kernel<16>), dim3(fp.n_p + fp.n_g), dim3(kFinalizeBlock), 0, w->stream, fp);
    SAGE_HIP(hipGetLastError());
  }
  AssembleParams ap = window_assemble_params(w);
  if (dst)
    ap.packed = dst;
  // four workgroups of 512 threads per output block: one element per thread (the kernel is a chain of dependent
  // gathers per element -- 17 us; one 1024-thread workgroup per block with two elements per thread took 27 us)
  ap.split = 4;
  int nblocks = w->K + ap.nlinks + 1;
  if (local_blocks && w->n_asm_blocks > 0)
  {
    ap.blocks = w->asm_blocks.as<int32_t>();
    nblocks = w->n_asm_blocks;
  }
  hipLaunchKernelGGL(assemble_kernel, dim3(nblocks * ap.split), dim3(512), 0, w->stream, ap);
  SAGE_HIP(hipGetLastError());
  window_phase_mark(w, 1);
  w->have_lin = true;
  w->lin_epoch = set == 0 ? w->vars_epoch : 0; // (a candidate's system becomes current only through lm_step's accept)
  w->spec_err_valid = false;
  w->packed_reduced = false;
  return SAGE_OK;
}

extern "C" int sage_window_linearize(SageWindow *w)
{
  if (w)
    window_phase_mark(w, 0); // a caller driving the iteration call by call: it starts here
  return window_linearize_set(w, 0);
}

static int window_error_pass(SageWindow *w, int which, bool speculate_gradients);
extern "C" int sage_window_error(SageWindow *w, int which) { return window_error_pass(w, which, false); }

// the run length a previous sage_window_tune_runs found for this window geometry (an embedder tunes once per image size / mask / window
// length and re-applies the result to the windows it builds afterwards: the tuning costs 20-40 ms, a window lives for a few LM steps)
extern "C" int sage_window_set_runs(SageWindow *w, int tpb)
{
  if (!w || !w->finalized)
    return SAGE_E_STATE;
  if (tpb < 1 || tpb > 64)
    return SAGE_E_INVALID;
  return window_plan_photo_runs(w, tpb, photo_flush_for_runs(tpb));
}

// r06 -- run-length tuning of the photometric kernels, measured on the window itself.  The time of the photometric linearize
// and of the error pass depends on the run length of a workgroup in a way no static rule captured (profiles/
// r06_kernel_ab_experiments.txt s16): which runs share an XCD's L2 with which, how run boundaries fall on image strips, how the
// workgroups fill the 768 slots.  On BASELINE config 4 runs of 12 are 6 % faster than the rule's 8 while 11 and 13 are 14 % slower; on
// a 192 x 256 window the rule's 8 is strip-aligned and 18 % (error pass: 27 %) slower than 6.  This call times the candidates on
// the window's own data -- three timed linearize + error-pass evaluations each at the current estimate, the rule measured first and last -- and keeps the fastest if it beats
// the rule's choice by >= 4 % (so that equal candidates do not flip between calls: results are bit-reproducible for a given run
// length, not across run lengths).  Opt-in: an explicit call, or SAGE_AUTOTUNE=1 at sage_window_finalize; SAGE_PHOTO_TPB pins
// the run length and disables it.  Single-rank windows only (a sharded window's linearize contains a collective).
extern "C" int sage_window_tune_runs(SageWindow *w, int *tpb_out, int *tpb_rule_out, float *ms_rule_out, float *ms_best_out)
{
  if (!w || !w->finalized)
    return SAGE_E_STATE;
  const int rule = w->tpb_heur;
  if (tpb_out)
    *tpb_out = w->tpb_p;
  if (tpb_rule_out)
    *tpb_rule_out = rule;
  if (ms_rule_out)
    *ms_rule_out = 0.f;
  if (ms_best_out)
    *ms_best_out = 0.f;
  if (getenv("SAGE_PHOTO_TPB") || w->world > 1 || w->allreduce || w->n_edges == 0 || !(w->cfg.use_photo))
    return SAGE_OK;
  int typical = 1;
  {
    std::vector<int> tiles;
    for (int n : w->Nedge)
      tiles.push_back((n + kTile - 1) / kTile);
    std::nth_element(tiles.begin(), tiles.begin() + tiles.size() / 2, tiles.end());
    typical = std::max(1, tiles[tiles.size() / 2]);
  }
  std::vector<int> cand{rule};
  for (int t : {4, 6, 8, 9, 10, 12, 16})
    if (t != rule && t <= typical && 2 * t >= std::min(rule, 8)) // (shorter than half the rule's runs: prologue-bound, not tried)
      cand.push_back(t);
  const bool prof_was = w->profiling;
  const int level_was = w->prof_level;
  int rc = sage_window_set_profiling(w, 1);
  double best_ms = 0.0, rule_ms = 0.0;
  int best = rule;
  // one plan's time: an untimed evaluation (its caches), then the fastest of three
  auto measure = [&](int t, double &ms) -> int {
    int r = window_plan_photo_runs(w, t, photo_flush_for_runs(t));
    for (int rep = 0; rep < 4 && !r; ++rep)
    {
      if ((r = sage_window_linearize(w)) || (r = sage_window_error(w, 1)))
        break;
      double lin = 0.0, err = 0.0;
      int nl = 0, ne = 0;
      if ((r = sage_window_get_kernel_time(w, 0, &lin, &nl)) || (r = sage_window_get_kernel_time(w, 2, &err, &ne)))
        break;
      const double tt = lin / std::max(1, nl) + err / std::max(1, ne);
      if (rep > 0)
        ms = rep == 1 ? tt : std::min(ms, tt);
    }
    return r;
  };
  // the device's clocks and caches settle over the first ~25 evaluations of a process (profiles/r04_step_series.txt): the rule's plan
  // is measured first AND last, so that whoever comes first is not charged for the warm-up
  for (int i = 0; i < 8 && !rc; ++i)
    if ((rc = sage_window_linearize(w)) || (rc = sage_window_error(w, 1)))
      break;
  (void)sage_window_get_kernel_time(w, 0, nullptr, nullptr);
  (void)sage_window_get_kernel_time(w, 2, nullptr, nullptr);
  if (!rc)
    rc = measure(rule, rule_ms);
  best_ms = rule_ms;
  for (size_t c = 1; c < cand.size() && !rc; ++c)
  {
    double ms = 0.0;
    if ((rc = measure(cand[c], ms)))
      break;
    if (ms < best_ms)
    {
      best_ms = ms;
      best = cand[c];
    }
  }
  if (!rc && best != rule)
  {
    double again = 0.0;
    if (!(rc = measure(rule, again)))
      rule_ms = std::min(rule_ms, again);
  }
  (void)sage_window_get_kernel_time(w, 1, nullptr, nullptr); // (drop the geometric kernels' records of these evaluations)
  (void)sage_window_get_kernel_time(w, 3, nullptr, nullptr);
  (void)sage_window_set_profiling(w, prof_was ? level_was : 0);
  if (!rc && !(best_ms < 0.96 * rule_ms))
    best = rule;
  const int rc2 = window_plan_photo_runs(w, rc ? rule : best, photo_flush_for_runs(rc ? rule : best));
  if (tpb_out)
    *tpb_out = w->tpb_p;
  if (ms_rule_out)
    *ms_rule_out = (float)rule_ms;
  if (ms_best_out)
    *ms_best_out = (float)(best == rule ? rule_ms : best_ms);
  return rc ? rc : rc2;
}

// speculate_gradients (the LM iteration's candidate evaluation, one GPU): the depth-map gradients of the evaluated set are
// launched right behind the totals -- the stream is idle while the host takes the accept / reject decision, and an accepted
// candidate's next linearize then finds maps AND gradients in place (one launch and 13 us off the accepted iteration; a
// rejected candidate's gradients are never read: the next evaluation rebuilds the maps)
static int window_error_pass(SageWindow *w, int which, bool speculate_gradients)
{
  if (!w || !w->finalized || which < 0 || which > 1)
    return SAGE_E_STATE;
  const SageWindowConfig &c = w->cfg;
  const int H = (int)c.pyr.cam[0].h, W = (int)c.pyr.cam[0].w;
  const bool has = w->n_edges > 0;
  if (has && w->dpt_set != which)
  {
    SAGE_HIP(launch_depth_batch(w->stream, c.CS, w->depth_items[which].as<DepthItem>(), w->n_depth, H, W, true, false));
    w->dpt_set = which;
    w->dgrad_valid = false;
  }
  ErrorTotalsSide ph{}, ge{};
  // both factor types: ONE kernel -- the photometric error kernel also evaluates the geometric edge at the same warp
  // (PhotoEdge::dpt1_geo), which saves the geometric launch (39 us + a gap) of the error pass
  const bool fused = has && c.use_photo && c.use_geo;
  if (has && c.use_photo)
  {
    LaunchCommon lc = window_lc(w, true);
    prof_attach(w, 2, lc);
    lc.stage = 1; // main kernel only: the per-edge statistics are formed by error_totals_kernel below
    lc.fused_geo_loss_param = fused ? c.geo_loss_param : 0.f;
    SAGE_HIP(launch_photo_error(w->stream, c.CS, c.FS, nullptr, w->ptab[which].as<PhotoEdge>(), lc, c.pyr,
                                c.photo_weights, c.eps, w->stats_p.as<float>()));
    float wsum = 0.f;
    for (int l = 0; l < c.pyr.levels; ++l)
      wsum += c.photo_weights[l];
    ph = ErrorTotalsSide{lc.edge_first, lc.edge_tiles, lc.partials, w->stats_p.as<float>(), 10.0f * wsum, 1.0f, w->n_edges,
                         fused ? 4 : 2, 0, 1};
    if (fused)
      ge = ErrorTotalsSide{lc.edge_first, lc.edge_tiles, lc.partials, w->stats_g.as<float>(), 10.0f * c.geo_weight,
                           c.geo_weight, w->n_edges, 4, 2, 3};
  }
  if (has && c.use_geo && !fused)
  {
    LaunchCommon lc = window_lc(w, false);
    prof_attach(w, 3, lc);
    lc.stage = 1;
    SAGE_HIP(launch_geo_error(w->stream, c.CS, nullptr, w->gtab[which].as<GeoEdge>(), lc, c.pyr.cam[0], c.eps,
                              c.geo_loss_param, c.geo_weight, w->stats_g.as<float>()));
    ge = ErrorTotalsSide{lc.edge_first, lc.edge_tiles, lc.partials, w->stats_g.as<float>(), 10.0f * c.geo_weight,
                         c.geo_weight, w->n_edges, 2, 0, 1};
  }
  w->err_epoch += 1;
  hipLaunchKernelGGL(error_totals_kernel, dim3(1), dim3(1024), 0, w->stream, ph, ge, w->errbuf.as<double>(),
                     w->world == 1 && !w->allreduce && w->h_err ? w->h_err + 4 : nullptr, (double)w->err_epoch);
  SAGE_HIP(hipGetLastError());
  window_phase_mark(w, 4);
  if (speculate_gradients && has && c.use_geo && w->dpt_set == which && !w->dgrad_valid)
  {
    SAGE_HIP(launch_depth_batch(w->stream, c.CS, w->depth_items[which].as<DepthItem>(), w->n_depth, H, W, false, true));
    w->dgrad_valid = true;
  }
  return SAGE_OK;
}

// After a device solve the candidate variables / delta live in the solver's pinned buffers until the stream has
// drained: refresh the host mirrors (set 1) here.  Returns SAGE_E_NOT_PSD when the factorisation hit a non-positive
// pivot (the candidate is then meaningless).
static int window_total_error(SageWindow *w, int from_linearize, double *err, bool stream_idle);

// Single-rank windows: wait for the error pass by spinning on the tickets its totals kernel writes into the pinned mirror
// (sage_window_error is the last thing in the stream then, and everything enqueued before it has landed too): a host
// thread blocked in hipStreamSynchronize for more than a few dozen microseconds wakes up through an interrupt, 20-30 us
// after the kernel has finished -- on the LM iteration's critical path.  false: no mirror / timed out (the caller
// synchronises the stream as before).
static bool window_spin_totals(SageWindow *w, bool reduced_mirror = false)
{
  // (reduced_mirror: the totals have just been mirrored by mirror_totals_kernel -- its own tickets and epoch)
  if (!w->h_err)
    return false;
  if (!reduced_mirror && (w->world != 1 || w->allreduce || w->err_epoch == 0))
    return false;
  const volatile double *t = w->h_err + (reduced_mirror ? 12 : 8);
  const double want = (double)(reduced_mirror ? w->mirror_epoch : w->err_epoch);
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (!(t[0] == want && t[1] == want && t[2] == want && t[3] == want))
  {
    __builtin_ia32_pause();
    if ((++spins & 0x3ff) == 0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05)
      return false;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return true;
}

// the window's all-reduce of n doubles + (peer emulation) the absent ranks' share from table entry `iterate`
static int window_allreduce(SageWindow *w, double *buf, size_t n, int iterate)
{
  if (w->allreduce && w->allreduce(buf, n, w->allreduce_user))
    return SAGE_E_STATE;
  if (w->emu_rest && w->emu_n > 0)
  {
    const size_t np = sage_window_packed_count(w);
    const double *rest = w->emu_rest + (size_t)(((iterate % w->emu_n) + w->emu_n) % w->emu_n) * np;
    if (n == np)
      hipLaunchKernelGGL(add_doubles_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, w->stream, rest, buf, np);
    else if (n == 4) // error totals of a candidate: the tail of the absent ranks' system at that iterate
      hipLaunchKernelGGL(add_doubles_kernel, dim3(1), dim3(64), 0, w->stream, rest + np - 4, buf, (size_t)4);
    else
      return SAGE_E_UNSUPPORTED;
    SAGE_HIP(hipGetLastError());
  }
  return SAGE_OK;
}

// out of place: recv = sum over the ranks of send (n = the packed system).  Native RCCL reduces send -> recv directly; a
// plain in-place hook gets a device copy first
__global__ void copy_doubles_kernel(const double *__restrict__ src, double *__restrict__ dst, size_t n);
static int window_allreduce_into(SageWindow *w, const double *send, double *recv, size_t n, int iterate)
{
  if (w->allreduce2)
  {
    if (w->allreduce2(send, recv, n, w->allreduce_user))
      return SAGE_E_STATE;
    SageAllReduceFn keep = w->allreduce;
    w->allreduce = nullptr; // (the sum is done: window_allreduce below only adds the emulated peers' share)
    const int rc = window_allreduce(w, recv, n, iterate);
    w->allreduce = keep;
    return rc;
  }
  hipLaunchKernelGGL(copy_doubles_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, w->stream, send, recv, n);
  SAGE_HIP(hipGetLastError());
  return window_allreduce(w, recv, n, iterate);
}

// enqueue the mirror of the (reduced) totals and their tickets
static int window_mirror_totals(SageWindow *w, bool with_err, const double *system = nullptr)
{
  w->mirror_epoch += 1;
  hipLaunchKernelGGL(mirror_totals_kernel, dim3(1), dim3(64), 0, w->stream,
                     (system ? system : w->packed.as<double>()) + sage_window_packed_count(w) - 4,
                     with_err ? w->errbuf.as<double>() : nullptr,
                     w->h_err, (double)w->mirror_epoch);
  SAGE_HIP(hipGetLastError());
  return SAGE_OK;
}

int window_sync_candidate(SageWindow *w, bool stream_idle)
{
  if (!w->cand_pending)
    return SAGE_OK;
  if (!stream_idle)
    SAGE_HIP(hipStreamSynchronize(w->stream));
  w->cand_pending = false;
  const DeviceSolver *S = w->last_solver ? w->last_solver : w->solver;
  if (solver_host_status(S) != 0)
    return SAGE_E_NOT_PSD;
  const int K = w->K, CS = w->cfg.CS, VS = w->VS;
  const float *v = solver_host_vars(S);
  for (int k = 0; k < K; ++k)
  {
    std::memcpy(&w->pose[1][(size_t)k * 12], v + (size_t)k * VS, 12 * sizeof(float));
    w->scale[1][k] = v[(size_t)k * VS + 12];
    std::memcpy(&w->code[1][(size_t)k * CS], v + (size_t)k * VS + 13, CS * sizeof(float));
  }
  std::memcpy(w->delta.data(), solver_host_delta(S), w->delta.size() * sizeof(double));
  return SAGE_OK;
}

// prior error terms at a variable set (a9): code prior w*||c||^2/CS per keyframe (code_factor.cpp:99-104, zero
// prior code), scale prior on keyframe 0 w*(ln s0 - ln s)^2 (scale_factor.cpp:102-129), pose prior on kf 0.
static void pose_local(const float *origin, const float *other, double out[6])
{
  // gtsam_traits.h:78-89 : [t1 - R1 R0^T t0, log(R1 R0^T)]
  double Rr[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      Rr[i * 3 + j] = (double)other[i * 3 + 0] * origin[j * 3 + 0] + (double)other[i * 3 + 1] * origin[j * 3 + 1] +
                      (double)other[i * 3 + 2] * origin[j * 3 + 2];
  for (int i = 0; i < 3; ++i)
    out[i] = other[9 + i] - (Rr[i * 3 + 0] * origin[9] + Rr[i * 3 + 1] * origin[10] + Rr[i * 3 + 2] * origin[11]);
  const double tr = Rr[0] + Rr[4] + Rr[8];
  const double cs = std::min(1.0, std::max(-1.0, 0.5 * (tr - 1.0)));
  const double th = std::acos(cs);
  const double k = th < 1e-8 ? 0.5 : th / (2.0 * std::sin(th));
  out[3] = k * (Rr[7] - Rr[5]);
  out[4] = k * (Rr[2] - Rr[6]);
  out[5] = k * (Rr[3] - Rr[1]);
}

static double prior_error(const SageWindow *w, int set)
{
  const SageWindowConfig &c = w->cfg;
  double e = 0;
  for (int k = 0; k < w->K; ++k)
  {
    double s = 0;
    for (int i = 0; i < c.CS; ++i)
      s += (double)w->code[set][(size_t)k * c.CS + i] * w->code[set][(size_t)k * c.CS + i];
    e += c.code_prior_weight * s / c.CS;
  }
  if (c.scale_prior_weight > 0)
  {
    const double d = std::log((double)w->scale_init[0]) - std::log((double)w->scale[set][0]);
    e += c.scale_prior_weight * d * d;
  }
  if (c.pose_prior_weight > 0)
  {
    double loc[6];
    pose_local(&w->pose[set][0], &w->pose_init[0], loc);
    for (int i = 0; i < 6; ++i)
      e += c.pose_prior_weight * loc[i] * loc[i];
  }
  return e;
}

extern "C" int sage_window_total_error(SageWindow *w, int from_linearize, double *err)
{
  return window_total_error(w, from_linearize, err, false);
}

// stream_idle: the caller has seen the error pass's tickets (window_spin_totals) -- nothing to synchronise
static int window_total_error(SageWindow *w, int from_linearize, double *err, bool stream_idle)
{
  if (!w || !w->finalized || !err)
    return SAGE_E_STATE;
  double t[4];
  int rcs = window_sync_candidate(w, stream_idle);
  if (rcs && rcs != SAGE_E_NOT_PSD)
    return rcs;
  // a failed factorisation only invalidates the CANDIDATE: the error at the linearisation point is still served
  const int rc_out = from_linearize ? SAGE_OK : rcs;
  if (w->world == 1 && !w->allreduce && w->h_err)
  {
    // single-rank window WITHOUT an all-reduce hook: the kernels mirrored the totals into pinned host memory (the same
    // condition the writers use -- ap.tail_mirror, the error_totals mirror; a one-rank RCCL communicator or
    // sage_window_set_allreduce leaves the mirror unwritten and takes the copies below: ADVICE r5)
    if (!stream_idle)
      SAGE_HIP(hipStreamSynchronize(w->stream));
    const double *m = w->h_err + (from_linearize ? 0 : 4);
    *err = m[0] + m[1] + prior_error(w, from_linearize ? 0 : 1);
    return rc_out;
  }
  if (from_linearize)
  {
    const size_t off = sage_window_packed_count(w) - 4;
    SAGE_HIP(hipMemcpyAsync(t, w->packed.as<double>() + off, 4 * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  }
  else
    SAGE_HIP(hipMemcpyAsync(t, w->errbuf.p, 4 * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  SAGE_HIP(hipStreamSynchronize(w->stream));
  *err = t[0] + t[1] + prior_error(w, from_linearize ? 0 : 1);
  return rc_out;
}

extern "C" int sage_window_solve(SageWindow *w, double damp, double *step_norm)
{
  if (!w || !w->finalized || !w->have_lin)
    return SAGE_E_STATE;
  const SageWindowConfig &c = w->cfg;
  const int K = w->K, B = w->B, CS = c.CS, BB = B * B, n = K * B;
  if (w->solver)
  {
    // device path: nothing leaves HBM but the candidate's host mirror (pinned, async); no synchronisation here
    // unless the caller asks for the step norm
    int rc = window_sync_candidate(w); // an unconsumed earlier candidate (a re-solve with another damping)
    if (rc && rc != SAGE_E_NOT_PSD)
      return rc;
    if (w->dpt_set == 1)
      w->dpt_set = -1; // the solve rewrites the candidate set
    rc = solver_run(w->solver, w->stream, w->packed.as<double>(), w->vars[0].as<float>(), w->vars[1].as<float>(), CS,
                    damp, c.code_prior_weight, c.scale_prior_weight, c.pose_prior_weight, w->scale_init[0],
                    &w->pose_init[0]);
    if (rc)
      return rc;
    window_phase_mark(w, 3);
    w->cand_pending = true;
    w->last_solver = w->solver;
    if (step_norm)
    {
      if ((rc = window_sync_candidate(w)))
        return rc;
      *step_norm = std::sqrt(solver_host_step_norm2(w->last_solver ? w->last_solver : w->solver));
    }
    return SAGE_OK;
  }
  const size_t np = sage_window_packed_count(w);
  static const bool dbg = sage::env_flag("SAGE_DEBUG_TIMING");
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_a = tnow();
  SAGE_HIP(hipStreamSynchronize(w->stream));
  auto t_b = tnow();
  SAGE_HIP(hipMemcpyAsync(w->host_packed.data(), w->packed.p, np * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  SAGE_HIP(hipStreamSynchronize(w->stream));
  auto t_c = tnow();
  // diagonal priors (a9): code prior on every keyframe, scale / pose priors on keyframe 0
  std::vector<double> dadd((size_t)n, 0.0), gadd((size_t)n, 0.0);
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < CS; ++i)
    {
      dadd[k * B + 6 + i] += c.code_prior_weight;
      gadd[k * B + 6 + i] += c.code_prior_weight * (0.0 - (double)w->code[0][(size_t)k * CS + i]);
    }
  if (c.scale_prior_weight > 0)
  {
    const double s = w->scale[0][0];
    dadd[6 + CS] += c.scale_prior_weight / (s * s);
    gadd[6 + CS] += c.scale_prior_weight / s * (std::log((double)w->scale_init[0]) - std::log(s));
  }
  if (c.pose_prior_weight > 0)
  {
    double loc[6];
    pose_local(&w->pose[0][0], &w->pose_init[0], loc);
    for (int i = 0; i < 6; ++i)
    {
      dadd[i] += c.pose_prior_weight;
      gadd[i] += c.pose_prior_weight * loc[i];
    }
  }
  std::vector<int32_t> lk(2 * w->links.size());
  for (size_t l = 0; l < w->links.size(); ++l)
  {
    lk[2 * l] = w->links[l].first;
    lk[2 * l + 1] = w->links[l].second;
  }
  std::vector<double> rhs((size_t)n);
  (void)BB;
  int rcs = sage_block_solve(w->host_packed.data(), K, (int)w->links.size(), lk.data(), B, damp, dadd.data(),
                             gadd.data(), rhs.data());
  if (rcs)
    return rcs;
  auto t_d = tnow();
  w->delta = rhs;
  double nrm = 0;
  for (double v : rhs)
    nrm += v * v;
  if (step_norm)
    *step_norm = std::sqrt(nrm);
  // candidate = retract(current, delta)
  for (int k = 0; k < K; ++k)
  {
    float d6[6];
    for (int i = 0; i < 6; ++i)
      d6[i] = (float)rhs[k * B + i];
    sage_pose_retract(&w->pose[0][(size_t)k * 12], d6, &w->pose[1][(size_t)k * 12]);
    for (int i = 0; i < CS; ++i)
      w->code[1][(size_t)k * CS + i] = w->code[0][(size_t)k * CS + i] + (float)rhs[k * B + 6 + i];
    w->scale[1][k] = w->scale[0][k] + (float)rhs[k * B + 6 + CS];
  }
  const int rcu = window_upload_vars(w, 1);
  if (dbg)
  {
    auto t_e = tnow();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[sage solve] wait-kernels %.3f d2h %.3f block_solve %.3f retract+h2d %.3f ms\n", ms(t_a, t_b),
            ms(t_b, t_c), ms(t_c, t_d), ms(t_d, t_e));
  }
  return rcu;
}

extern "C" int sage_window_accept(SageWindow *w)
{
  if (!w || !w->finalized)
    return SAGE_E_STATE;
  int rcs = window_sync_candidate(w);
  if (rcs)
    return rcs;
  w->pose[0] = w->pose[1];
  w->code[0] = w->code[1];
  w->scale[0] = w->scale[1];
  ++w->vars_epoch;
  ++w->emu_cur;
  w->dpt_set = w->dpt_set == 1 ? 0 : -1; // depth maps evaluated at the candidate now belong to the current set
  // (a kernel, not hipMemcpyAsync: a device-to-device copy of 11 KB costs ~10 us of API time on the step's critical path)
  const int nv = w->K * w->VS;
  hipLaunchKernelGGL(copy_floats_kernel, dim3((nv + 255) / 256), dim3(256), 0, w->stream, w->vars[1].as<float>(),
                     w->vars[0].as<float>(), nv);
  SAGE_HIP(hipGetLastError());
  return SAGE_OK;
}

extern "C" int sage_window_reset(SageWindow *w)
{
  if (!w || !w->finalized)
    return SAGE_E_STATE;
  (void)window_sync_candidate(w);
  for (int s = 0; s < 2; ++s)
  {
    w->pose[s] = w->pose_init;
    w->code[s] = w->code_added;
    w->scale[s] = w->scale_init;
  }
  int rc;
  if ((rc = window_upload_vars(w, 0)) || (rc = window_upload_vars(w, 1)))
    return rc;
  w->have_lin = false;
  w->emu_cur = 0;
  return SAGE_OK;
}

extern "C" int sage_window_get_keyframe(const SageWindow *w, int kf, float *pose12, float *code, float *scale)
{
  if (!w || kf < 0 || kf >= w->K)
    return SAGE_E_INVALID;
  if (pose12)
    std::memcpy(pose12, &w->pose[0][(size_t)kf * 12], 12 * sizeof(float));
  if (code)
    std::memcpy(code, &w->code[0][(size_t)kf * w->cfg.CS], w->cfg.CS * sizeof(float));
  if (scale)
    *scale = w->scale[0][kf];
  return SAGE_OK;
}

extern "C" int sage_window_set_keyframe(SageWindow *w, int kf, const float *pose12, const float *code, float scale)
{
  if (!w || kf < 0 || kf >= w->K || !pose12 || !code)
    return SAGE_E_INVALID;
  (void)window_sync_candidate(w);
  for (int s = 0; s < 2; ++s)
  {
    std::memcpy(&w->pose[s][(size_t)kf * 12], pose12, 12 * sizeof(float));
    std::memcpy(&w->code[s][(size_t)kf * w->cfg.CS], code, w->cfg.CS * sizeof(float));
    w->scale[s][kf] = scale;
  }
  if (w->finalized)
  {
    int rc;
    if ((rc = window_upload_vars(w, 0)) || (rc = window_upload_vars(w, 1)))
      return rc;
  }
  return SAGE_OK;
}

extern "C" int sage_window_get_delta(const SageWindow *w, double *delta)
{
  if (!w || !delta)
    return SAGE_E_INVALID;
  int rcs = window_sync_candidate(const_cast<SageWindow *>(w));
  if (rcs)
    return rcs;
  std::memcpy(delta, w->delta.data(), w->delta.size() * sizeof(double));
  return SAGE_OK;
}

static void window_priors(const SageWindow *w, std::vector<double> &dadd, std::vector<double> &gadd)
{
  const SageWindowConfig &c = w->cfg;
  const int K = w->K, B = w->B, CS = c.CS;
  dadd.assign((size_t)K * B, 0.0);
  gadd.assign((size_t)K * B, 0.0);
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < CS; ++i)
    {
      dadd[k * B + 6 + i] += c.code_prior_weight;
      gadd[k * B + 6 + i] += c.code_prior_weight * (0.0 - (double)w->code[0][(size_t)k * CS + i]);
    }
  if (c.scale_prior_weight > 0)
  {
    const double s = w->scale[0][0];
    dadd[6 + CS] += c.scale_prior_weight / (s * s);
    gadd[6 + CS] += c.scale_prior_weight / s * (std::log((double)w->scale_init[0]) - std::log(s));
  }
  if (c.pose_prior_weight > 0)
  {
    double loc[6];
    pose_local(&w->pose[0][0], &w->pose_init[0], loc);
    for (int i = 0; i < 6; ++i)
    {
      dadd[i] += c.pose_prior_weight;
      gadd[i] += c.pose_prior_weight * loc[i];
    }
  }
}

// prior error terms of the keyframes THIS rank owns (the other ranks' copies of their variables are stale here)
static double prior_error_owned(const SageWindow *w, int set)
{
  const SageWindowConfig &c = w->cfg;
  double e = 0;
  for (int k = 0; k < w->K; ++k)
  {
    if (sage_shard_keyframe_owner(w->shard, k) != w->rank)
      continue;
    double s2 = 0;
    for (int i = 0; i < c.CS; ++i)
      s2 += (double)w->code[set][(size_t)k * c.CS + i] * w->code[set][(size_t)k * c.CS + i];
    e += c.code_prior_weight * s2 / c.CS;
    if (k == 0 && c.scale_prior_weight > 0)
    {
      const double d = std::log((double)w->scale_init[0]) - std::log((double)w->scale[set][0]);
      e += c.scale_prior_weight * d * d;
    }
    if (k == 0 && c.pose_prior_weight > 0)
    {
      double loc[6];
      pose_local(&w->pose[set][0], &w->pose_init[0], loc);
      for (int i = 0; i < 6; ++i)
        e += c.pose_prior_weight * loc[i] * loc[i];
    }
  }
  return e;
}

__global__ void add_to_double_kernel(double *p, double v) { p[0] += v; }

// local elimination -> all-reduce of the separator system -> separator solve + back substitution of this rank's
// keyframes -> candidate variables of those keyframes.  *lin_error (optional) receives the total error at the
// linearisation point (edge totals ride in the payload tail, prior terms are contributed by their owners).
// Returns SAGE_E_NOT_PSD consistently on every rank (a rank whose local elimination fails poisons the payload).
static int schur_solve(SageWindow *w, double damp, double *lin_error)
{
  const SageWindowConfig &c = w->cfg;
  const int K = w->K, B = w->B, CS = c.CS;
  const size_t np = sage_window_packed_count(w), ns = w->h_sep.size();
  SAGE_HIP(hipMemcpyAsync(w->host_packed.data(), w->packed.p, np * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  SAGE_HIP(hipStreamSynchronize(w->stream));
  std::vector<double> dadd, gadd;
  window_priors(w, dadd, gadd);
  int rc = sage_shard_eliminate(w->shard, w->host_packed.data(), damp, dadd.data(), gadd.data(), w->h_sep.data());
  if (rc && rc != SAGE_E_NOT_PSD)
    return rc;
  if (rc == SAGE_E_NOT_PSD)
  {
    // a failed local elimination is flagged in the spare tail slot [ns-3] (a positive count after the sum: every rank
    // sees it); the separator blocks of this rank are void, the error totals at the linearisation point (tail[0..4],
    // written by sage_shard_eliminate before it factorises) stay finite so that st->error is valid on every rank
    std::fill(w->h_sep.begin(), w->h_sep.end() - 8, 0.0);
    w->h_sep[ns - 3] = 1.0;
  }
  w->h_sep[ns - 4] = prior_error_owned(w, 0);
  SAGE_HIP(hipMemcpyAsync(w->sepbuf.p, w->h_sep.data(), ns * sizeof(double), hipMemcpyHostToDevice, w->stream));
  if (w->allreduce(w->sepbuf.as<double>(), ns, w->allreduce_user))
    return SAGE_E_STATE;
  SAGE_HIP(hipMemcpyAsync(w->h_sep.data(), w->sepbuf.p, ns * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  SAGE_HIP(hipStreamSynchronize(w->stream));
  if (lin_error)
    *lin_error = w->h_sep[ns - 8] + w->h_sep[ns - 7] + w->h_sep[ns - 4];
  if (w->h_sep[ns - 3] > 0.0 || std::isnan(w->h_sep[0]))
    return SAGE_E_NOT_PSD;
  w->delta.assign((size_t)K * B, 0.0);
  rc = sage_shard_solve(w->shard, w->h_sep.data(), w->delta.data());
  if (rc)
    return rc; // SAGE_E_NOT_PSD of the separator system: identical on every rank
  // candidate = retract(current, delta) for the keyframes this rank touches; the others keep their (stale) values
  w->pose[1] = w->pose[0];
  w->code[1] = w->code[0];
  w->scale[1] = w->scale[0];
  for (int k = 0; k < K; ++k)
  {
    if (!sage_shard_keyframe_is_local(w->shard, k))
      continue;
    float d6[6];
    for (int i = 0; i < 6; ++i)
      d6[i] = (float)w->delta[(size_t)k * B + i];
    sage_pose_retract(&w->pose[0][(size_t)k * 12], d6, &w->pose[1][(size_t)k * 12]);
    for (int i = 0; i < CS; ++i)
      w->code[1][(size_t)k * CS + i] = w->code[0][(size_t)k * CS + i] + (float)w->delta[(size_t)k * B + 6 + i];
    w->scale[1][k] = w->scale[0][k] + (float)w->delta[(size_t)k * B + 6 + CS];
  }
  w->cand_pending = false;
  return window_upload_vars(w, 1);
}

// after a Schur-mode run every rank holds current variables only for the keyframes it touches: sum the owners' copies
extern "C" int sage_window_sync_variables(SageWindow *w)
{
  if (!w || !w->finalized)
    return SAGE_E_STATE;
  if (!w->shard)
    return SAGE_OK; // every rank solves the whole system: nothing to exchange
  if (!w->allreduce)
    return SAGE_E_STATE;
  const int K = w->K, CS = w->cfg.CS, VS = 13 + CS;
  std::vector<double> buf((size_t)K * VS, 0.0);
  for (int k = 0; k < K; ++k)
    if (sage_shard_keyframe_owner(w->shard, k) == w->rank)
    {
      double *b = &buf[(size_t)k * VS];
      for (int i = 0; i < 12; ++i)
        b[i] = w->pose[0][(size_t)k * 12 + i];
      b[12] = w->scale[0][k];
      for (int i = 0; i < CS; ++i)
        b[13 + i] = w->code[0][(size_t)k * CS + i];
    }
  DevBuf d;
  int rc = d.reserve(buf.size() * sizeof(double));
  if (rc)
    return rc;
  SAGE_HIP(hipMemcpyAsync(d.p, buf.data(), buf.size() * sizeof(double), hipMemcpyHostToDevice, w->stream));
  if (w->allreduce(d.as<double>(), buf.size(), w->allreduce_user))
  {
    d.release();
    return SAGE_E_STATE;
  }
  SAGE_HIP(hipMemcpyAsync(buf.data(), d.p, buf.size() * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  SAGE_HIP(hipStreamSynchronize(w->stream));
  d.release();
  for (int s = 0; s < 2; ++s)
    for (int k = 0; k < K; ++k)
    {
      const double *b = &buf[(size_t)k * VS];
      for (int i = 0; i < 12; ++i)
        w->pose[s][(size_t)k * 12 + i] = (float)b[i];
      w->scale[s][k] = (float)b[12];
      for (int i = 0; i < CS; ++i)
        w->code[s][(size_t)k * CS + i] = (float)b[13 + i];
    }
  if ((rc = window_upload_vars(w, 0)) || (rc = window_upload_vars(w, 1)))
    return rc;
  return SAGE_OK;
}

__global__ void copy_doubles_kernel(const double *__restrict__ src, double *__restrict__ dst, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    dst[i] = src[i];
}

// SageLmConfig::linearize_at_candidate: one LM iteration in which the candidate is evaluated by the linearize kernels.
// `packed` holds the system at the current estimate (kept from the previous accepted iteration); per evaluation: damped
// solve -> candidate; linearize at the candidate into a second buffer (its finalize kernels deliver the error); accepted:
// the two buffers swap -- the candidate's system IS the next iteration's, nothing is re-evaluated or copied; rejected:
// `packed` never left and the damping goes up.  Decisions, damping schedule and iterates are those of
// the default sequence; sage_window_get_edge afterwards returns the per-edge results of the LAST evaluation.
static int lm_step_at_candidate(SageWindow *w, SageLmState *st, const SageLmConfig *cfg, bool sharded)
{
  int rc;
  auto clampd = [&](double d) { return std::min(std::max((double)cfg->min_damp, d), (double)cfg->max_damp); };
  const size_t np = sage_window_packed_count(w);
  if ((rc = w->packed_save.reserve(np * sizeof(double))))
    return rc;
  if (sharded && !w->packed_loc.p)
  {
    if ((rc = w->packed_loc.reserve(np * sizeof(double))))
      return rc;
    SAGE_HIP(hipMemsetAsync(w->packed_loc.p, 0, np * sizeof(double), w->stream)); // blocks of other ranks: zero for good
  }
  // one evaluation: linearize at variable set `set`, the (reduced) system into `dst`, its totals into the pinned mirror with
  // their tickets -- the host never blocks in a stream synchronise on the iteration's critical path.  A reduced window
  // assembles only the blocks its own edges touch (into packed_loc) and sums out of place into dst; the emulated peers'
  // share of iterate `it` is added behind the sum.
  auto evaluate = [&](int set, double *dst, int it) -> int {
    int r;
    if (!sharded)
      return (r = window_linearize_set(w, set, dst, false, true)) ? r : window_mirror_totals(w, false, dst);
    if ((r = window_linearize_set(w, set, w->packed_loc.as<double>(), true, true)) ||
        (r = window_allreduce_into(w, w->packed_loc.as<double>(), dst, np, it)))
      return r;
    window_phase_mark(w, 2);
    return window_mirror_totals(w, false, dst);
  };
  // the system at the current estimate is reused only if it is the GLOBAL one: sage_window_linearize / _prepass leave a
  // rank-local `packed` behind on a sharded window (every rank sees the same flags: same call sequence on all ranks)
  bool mirrored_now = false;
  if (!(w->have_lin && w->lin_epoch == w->vars_epoch && (!sharded || w->packed_reduced)))
  {
    if ((rc = evaluate(0, w->packed.as<double>(), w->emu_cur)))
      return rc;
    w->packed_reduced = true;
    w->spec_err_valid = false;
    mirrored_now = true;
  }
  if (!w->spec_err_valid)
  {
    // the totals at the current estimate: mirrored by the evaluation above -- or the system was left by
    // sage_window_linearize (unsharded: mirror its tail now)
    if (!mirrored_now && (rc = window_mirror_totals(w, false)))
      return rc;
    if (!window_spin_totals(w, true))
      SAGE_HIP(hipStreamSynchronize(w->stream));
    w->spec_error = w->h_err[0] + w->h_err[1] + prior_error(w, 0);
    w->spec_err_valid = true;
  }
  st->error = w->spec_error;
  int evals = 0;
  st->accepted = 0;
  for (;;)
  {
    rc = sage_window_solve(w, st->damp, nullptr);
    bool not_psd = rc == SAGE_E_NOT_PSD;
    if (rc && !not_psd)
      return rc;
    double cur_tot[4] = {0, 0, 0, 0};
    bool have_cur_tot = false; // (a synchronous NOT_PSD of the solve leaves the mirror alone: nothing to put back)
    st->candidate_error = INFINITY;
    if (!not_psd)
    {
      const uint64_t lin_epoch = w->lin_epoch;
      const bool was_reduced = w->packed_reduced;
      std::memcpy(cur_tot, w->h_err, sizeof(cur_tot)); // (mirrored and seen at the end of the previous evaluation)
      have_cur_tot = true;
      if ((rc = evaluate(1, w->packed_save.as<double>(), w->emu_cur + 1)))
        return rc;
      w->lin_epoch = lin_epoch; // (`packed` still is the current estimate's system; the candidate's sits in packed_save)
      w->packed_reduced = was_reduced;
      w->spec_err_valid = true; // spec_error still is the error at the current estimate
      const bool idle = window_spin_totals(w, true);
      rc = window_sync_candidate(w, idle); // a non-positive pivot of the factorisation shows up here
      if (rc && rc != SAGE_E_NOT_PSD)
        return rc;
      not_psd = rc == SAGE_E_NOT_PSD;
      if (!idle)
        SAGE_HIP(hipStreamSynchronize(w->stream));
      if (!not_psd)
        st->candidate_error = w->h_err[0] + w->h_err[1] + prior_error(w, 1);
    }
    ++evals;
    if (st->candidate_error < st->error)
    {
      st->accepted = 1;
      if ((rc = sage_window_accept(w)))
        return rc;
      std::swap(w->packed.p, w->packed_save.p); // the candidate's system IS the next iteration's
      std::swap(w->packed.cap, w->packed_save.cap);
      w->lin_epoch = w->vars_epoch;
      w->packed_reduced = true; // (summed over the ranks by evaluate(), or nothing to sum)
      w->spec_error = st->candidate_error;
      st->damp = clampd(st->damp / cfg->damp_dec_factor);
      break;
    }
    // rejected: `packed` never left; the mirror goes back to the current estimate's totals
    if (have_cur_tot)
      std::memcpy(w->h_err, cur_tot, sizeof(cur_tot));
    const bool give_up = st->damp >= cfg->max_damp || (cfg->max_inner_evals > 0 && evals >= cfg->max_inner_evals);
    st->damp = clampd(st->damp * cfg->damp_inc_factor);
    if (give_up)
      break;
  }
  st->iters += 1;
  return SAGE_OK;
}

extern "C" int sage_window_lm_step(SageWindow *w, SageLmState *st, const SageLmConfig *cfg)
{
  if (!w || !st || !cfg)
    return SAGE_E_INVALID;
  int rc;
  if (w->world > 1 && !w->allreduce)
    return SAGE_E_STATE;
  const bool sharded = w->allreduce != nullptr; // (a hook on a single-rank window is honoured too)
  if (st->iters == 0 && st->damp <= 0)
    st->damp = cfg->init_damp;
  auto clampd = [&](double d) { return std::min(std::max((double)cfg->min_damp, d), (double)cfg->max_damp); };
  const bool schur = sharded && w->shard != nullptr;
  window_phase_mark(w, 0);
  // (rank-independent decision: the window's link count, not this rank's share of it -- a rank without links must
  //  issue the same collectives as the others).  linearize_at_candidate 0 = automatic: the sequence with one collective
  //  and no separate error pass per iteration whenever the window is reduced over ranks (the shard's kernels are short
  //  there, the second collective and its host round trip are not), the classic sequence on a single rank
  const bool at_candidate = cfg->linearize_at_candidate > 0 || (cfg->linearize_at_candidate == 0 && sharded);
  if (at_candidate && !schur && !w->links.empty())
    return lm_step_at_candidate(w, st, cfg, sharded);
  if ((rc = window_linearize_set(w, 0, nullptr, false, true)))
    return rc;
  if (sharded && !schur && (rc = window_allreduce(w, w->packed.as<double>(), sage_window_packed_count(w), w->emu_cur)))
    return rc;
  window_phase_mark(w, 2);
  int evals = 0;
  st->accepted = 0;
  while (schur)
  {
    // domain-decomposed iteration: the collectives are the separator system and the 4-double error totals
    double lin_error = 0;
    rc = schur_solve(w, st->damp, &lin_error);
    if (rc && rc != SAGE_E_NOT_PSD)
      return rc;
    window_phase_mark(w, 3);
    if (evals == 0)
      st->error = lin_error;
    if (rc == SAGE_E_NOT_PSD)
      st->candidate_error = INFINITY;
    else
    {
      if ((rc = sage_window_error(w, 1)))
        return rc;
      hipLaunchKernelGGL(add_to_double_kernel, dim3(1), dim3(1), 0, w->stream, w->errbuf.as<double>(),
                         prior_error_owned(w, 1));
      if (w->allreduce(w->errbuf.as<double>(), 4, w->allreduce_user))
        return SAGE_E_STATE;
      double t4[4];
      SAGE_HIP(hipMemcpyAsync(t4, w->errbuf.p, sizeof(t4), hipMemcpyDeviceToHost, w->stream));
      SAGE_HIP(hipStreamSynchronize(w->stream));
      st->candidate_error = t4[0] + t4[1];
    }
    ++evals;
    if (st->candidate_error < st->error)
    {
      st->accepted = 1;
      break;
    }
    const bool give_up = st->damp >= cfg->max_damp || (cfg->max_inner_evals > 0 && evals >= cfg->max_inner_evals);
    st->damp = clampd(st->damp * cfg->damp_inc_factor);
    if (give_up)
      break;
  }
  while (!schur)
  {
    // everything of one evaluation is enqueued before the host looks at a number: the error at the linearisation
    // point (tail of the packed buffer) is read together with the candidate's
    rc = sage_window_solve(w, st->damp, nullptr);
    if (rc == SAGE_E_NOT_PSD)
    {
      // the damped system has a non-positive pivot: a rejected evaluation (every rank factors the same system and
      // takes this branch together; no error pass, no collective)
      if (evals == 0 && (rc = sage_window_total_error(w, 1, &st->error)))
        return rc;
      st->candidate_error = INFINITY;
      ++evals;
      if (st->damp >= cfg->max_damp || (cfg->max_inner_evals > 0 && evals >= cfg->max_inner_evals))
      {
        st->damp = clampd(st->damp * cfg->damp_inc_factor);
        break;
      }
      st->damp = clampd(st->damp * cfg->damp_inc_factor);
      continue;
    }
    if (rc)
      return rc;
    if ((rc = window_error_pass(w, 1, !sharded)))
      return rc;
    if (sharded)
    {
      if ((rc = window_allreduce(w, w->errbuf.as<double>(), 4, w->emu_cur + 1)) || (rc = window_mirror_totals(w, true)))
        return rc;
      // (the host spins on the mirror's tickets instead of blocking in a stream synchronise: with the kernels of a shard
      //  8x shorter, the 20-30 us wake-up of a blocked thread would be 4 % of an iteration)
      const bool idle = window_spin_totals(w, true);
      // a non-positive pivot of the damped system is a REJECTED evaluation (raise the damping), not a hard error; every
      // rank factors the same reduced system, so all of them take this branch together and the number of collectives
      // per iteration stays the same on every rank
      rc = window_sync_candidate(w, idle);
      if (rc && rc != SAGE_E_NOT_PSD)
        return rc;
      const bool not_psd = rc == SAGE_E_NOT_PSD;
      if (!idle)
        SAGE_HIP(hipStreamSynchronize(w->stream));
      if (evals == 0)
        st->error = w->h_err[0] + w->h_err[1] + prior_error(w, 0);
      st->candidate_error = not_psd ? INFINITY : w->h_err[4] + w->h_err[5] + prior_error(w, 1);
    }
    else
    {
      const bool idle = window_spin_totals(w);
      if (evals == 0 && (rc = window_total_error(w, 1, &st->error, idle)))
        return rc;
      rc = window_total_error(w, 0, &st->candidate_error, idle);
      if (rc == SAGE_E_NOT_PSD)
        st->candidate_error = INFINITY;
      else if (rc)
        return rc;
    }
    ++evals;
    if (st->candidate_error < st->error)
    {
      st->accepted = 1;
      break;
    }
    if (st->damp >= cfg->max_damp || (cfg->max_inner_evals > 0 && evals >= cfg->max_inner_evals))
    {
      st->damp = clampd(st->damp * cfg->damp_inc_factor);
      break;
    }
    st->damp = clampd(st->damp * cfg->damp_inc_factor);
  }
  if (st->accepted)
  {
    if ((rc = sage_window_accept(w)))
      return rc;
    st->damp = clampd(st->damp / cfg->damp_dec_factor);
  }
  st->iters += 1;
  return SAGE_OK;
}

// n LM iterations in one call (the loop a C++ caller writes around sage_window_lm_step; bench.py uses it so that no Python
// runs between the iterations it times).  trace (optional): n x {error, candidate_error, accepted, damp after the step}.
// Stops early on an error code; *done (optional) = iterations completed.
static int window_lm_run(SageWindow *w, SageLmState *st, const SageLmConfig *cfg, int n, double *trace, int *done,
                         double *step_seconds)
{
  if (!w || !st || !cfg || n < 0)
    return SAGE_E_INVALID;
  int i = 0, rc = SAGE_OK;
  auto t_prev = std::chrono::steady_clock::now();
  for (; i < n; ++i)
  {
    if ((rc = sage_window_lm_step(w, st, cfg)))
      break;
    if (trace)
    {
      trace[4 * i + 0] = st->error;
      trace[4 * i + 1] = st->candidate_error;
      trace[4 * i + 2] = (double)st->accepted;
      trace[4 * i + 3] = st->damp;
    }
    if (step_seconds)
    {
      const auto t = std::chrono::steady_clock::now();
      step_seconds[i] = std::chrono::duration<double>(t - t_prev).count();
      t_prev = t;
    }
  }
  if (done)
    *done = i;
  return rc;
}

extern "C" int sage_window_lm_run(SageWindow *w, SageLmState *st, const SageLmConfig *cfg, int n, double *trace, int *done)
{
  return window_lm_run(w, st, cfg, n, trace, done, nullptr);
}

// the same with the host wall time of every iteration (step_seconds[n]: from the return of the previous iteration -- the
// call's entry for the first -- to this one's; an iteration returns once its accept / reject decision is taken)
extern "C" int sage_window_lm_run_timed(SageWindow *w, SageLmState *st, const SageLmConfig *cfg, int n, double *trace,
                                        int *done, double *step_seconds)
{
  return window_lm_run(w, st, cfg, n, trace, done, step_seconds);
}
