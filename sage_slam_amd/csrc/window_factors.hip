// window_factors.hip -- f2: per-edge results and the per-Values factor cache behind the gtsam adapter
// (photometric_factor.cpp:72-219, geometric_factor.cpp:41-233): batched prepass, factor blocks, NearestPsd on host threads.
#include "runtime_internal.h"

extern "C" int sage_window_get_edge(const SageWindow *w, int type, int e, float *AtA, float *Atb, float *err,
                                    float *n_in)
{
  if (!w || !w->finalized || (type != 0 && type != 1))
    return SAGE_E_INVALID;
  // e is the global directed-edge index: link e/2, direction e%2 ; map to the local index
  const int le = window_local_edge(w, e);
  if (le < 0)
    return SAGE_E_INVALID;
  const size_t D = type == 0 ? 13 + w->cfg.CS : 14 + 2 * w->cfg.CS;
  const DevBuf &A = type == 0 ? w->AtA_p : w->AtA_g, &b = type == 0 ? w->Atb_p : w->Atb_g,
               &st = type == 0 ? w->stats_p : w->stats_g;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  if (AtA)
    SAGE_HIP(hipMemcpy(AtA, A.as<float>() + (size_t)le * D * D, D * D * sizeof(float), hipMemcpyDeviceToHost));
  if (Atb)
    SAGE_HIP(hipMemcpy(Atb, b.as<float>() + (size_t)le * D, D * sizeof(float), hipMemcpyDeviceToHost));
  float s2[2];
  SAGE_HIP(hipMemcpy(s2, st.as<float>() + (size_t)le * 2, 2 * sizeof(float), hipMemcpyDeviceToHost));
  if (err)
    *err = s2[0];
  if (n_in)
    *n_in = s2[1];
  return SAGE_OK;
}

// ------------------------------------------------------------------------------------------------
// f2: the batched per-Values prepass behind the gtsam factors.  ISAM2 asks every factor of the window for
// linearize(values) / error(values) one at a time with the SAME Values; the reference answers each with its own kernel
// launches, .item() syncs and a NearestPsd (photometric_factor.cpp:72-219, geometric_factor.cpp:41-233).  Here the first
// factor that sees new values triggers ONE sage_window_linearize (or sage_window_error) for the whole window and one
// device-to-host copy of the per-edge results; every other factor is served from the host cache.
static int local_edge_index(const SageWindow *w, int e) { return window_local_edge(w, e); }

extern "C" int sage_window_prepass(SageWindow *w, const float *pose12, const float *codes, const float *scales,
                                   int jacobians, int *recomputed)
{
  if (!w || !w->finalized || !pose12 || !codes || !scales)
    return SAGE_E_INVALID;
  const int K = w->K, CS = w->cfg.CS;
  SageWindow::FactorCache &fc = w->fc;
  const size_t np = (size_t)K * 12, nc = (size_t)K * CS;
  const bool same = fc.pose.size() == np && std::memcmp(fc.pose.data(), pose12, np * sizeof(float)) == 0 &&
                    std::memcmp(fc.code.data(), codes, nc * sizeof(float)) == 0 &&
                    std::memcmp(fc.scale.data(), scales, (size_t)K * sizeof(float)) == 0;
  if (recomputed)
    *recomputed = 0;
  if (same && (fc.lin || (!jacobians && fc.err)))
    return SAGE_OK; // a cached linearisation also carries the errors (a1 returns the same error as a2)
  int rc;
  if (!same)
  {
    (void)window_sync_candidate(w);
    fc.lin = fc.err = false;
    fc.psd_mode = -1;
    fc.pose.assign(pose12, pose12 + np);
    fc.code.assign(codes, codes + nc);
    fc.scale.assign(scales, scales + K);
  }
  // the window's CURRENT variables become the requested values (both sets: a later solve starts from them)
  if (std::memcmp(w->pose[0].data(), pose12, np * sizeof(float)) != 0 ||
      std::memcmp(w->code[0].data(), codes, nc * sizeof(float)) != 0 ||
      std::memcmp(w->scale[0].data(), scales, (size_t)K * sizeof(float)) != 0)
  {
    (void)window_sync_candidate(w);
    for (int s = 0; s < 2; ++s)
    {
      w->pose[s].assign(pose12, pose12 + np);
      w->code[s].assign(codes, codes + nc);
      w->scale[s].assign(scales, scales + K);
    }
    if ((rc = window_upload_vars(w, 0)) || (rc = window_upload_vars(w, 1)))
      return rc;
    w->have_lin = false;
  }
  const size_t ne = (size_t)w->n_edges, Dp = 13 + CS, Dg = 14 + 2 * CS;
  if (jacobians)
  {
    if ((rc = sage_window_linearize(w)))
      return rc;
  }
  else if ((rc = sage_window_error(w, 0)))
    return rc;
  SAGE_HIP(hipStreamSynchronize(w->stream));
  auto pull = [&](std::vector<float> &dst, const DevBuf &src, size_t n) -> hipError_t {
    dst.resize(n);
    return n ? hipMemcpy(dst.data(), src.p, n * sizeof(float), hipMemcpyDeviceToHost) : hipSuccess;
  };
  if (w->cfg.use_photo)
  {
    if (jacobians)
    {
      SAGE_HIP(pull(fc.Ap, w->AtA_p, ne * Dp * Dp));
      SAGE_HIP(pull(fc.bp, w->Atb_p, ne * Dp));
    }
    SAGE_HIP(pull(fc.sp, w->stats_p, ne * 2));
  }
  if (w->cfg.use_geo)
  {
    if (jacobians)
    {
      SAGE_HIP(pull(fc.Ag, w->AtA_g, ne * Dg * Dg));
      SAGE_HIP(pull(fc.bg, w->Atb_g, ne * Dg));
    }
    SAGE_HIP(pull(fc.sg, w->stats_g, ne * 2));
  }
  fc.lin = jacobians != 0;
  fc.err = true;
  if (recomputed)
    *recomputed = 1;
  return SAGE_OK;
}

extern "C" int sage_window_factor_error(const SageWindow *w, int type, int e, double *err_out)
{
  if (!w || !w->finalized || (type != 0 && type != 1) || !err_out)
    return SAGE_E_INVALID;
  const SageWindow::FactorCache &fc = w->fc;
  if (!fc.err)
    return SAGE_E_STATE;
  const int le = local_edge_index(w, e);
  const std::vector<float> &st = type == 0 ? fc.sp : fc.sg;
  if (le < 0 || (size_t)le * 2 + 1 >= st.size())
    return SAGE_E_INVALID;
  *err_out = (double)st[(size_t)le * 2];
  return SAGE_OK;
}

extern "C" int sage_window_factor(const SageWindow *w, int type, int e, int psd_mode, double *G_out, double *g_out,
                                  double *f_out, int *dims_out, int *nkeys_out)
{
  if (!w || !w->finalized || (type != 0 && type != 1))
    return SAGE_E_INVALID;
  const SageWindow::FactorCache &fc = w->fc;
  if (!fc.lin)
    return SAGE_E_STATE;
  const int le = local_edge_index(w, e);
  const int CS = w->cfg.CS;
  const size_t D = type == 0 ? 13 + CS : 14 + 2 * CS;
  const std::vector<float> &A = type == 0 ? fc.Ap : fc.Ag, &b = type == 0 ? fc.bp : fc.bg, &st = type == 0 ? fc.sp : fc.sg;
  if (le < 0 || ((size_t)le + 1) * D * D > A.size())
    return SAGE_E_INVALID;
  if (f_out)
    *f_out = (double)st[(size_t)le * 2];
  const std::vector<double> &Cc = type == 0 ? fc.Cp : fc.Cg;
  if (fc.psd_mode == psd_mode && ((size_t)le + 1) * D * D <= Cc.size()) // prepared on the host threads already
    return sage_factor_cut_blocks(type, CS, Cc.data() + (size_t)le * D * D, b.data() + (size_t)le * D, G_out, g_out,
                                  dims_out, nkeys_out);
  return sage_factor_hessian_blocks(type, CS, A.data() + (size_t)le * D * D, b.data() + (size_t)le * D, psd_mode, G_out,
                                    g_out, dims_out, nkeys_out);
}

// NearestPsd of EVERY cached factor on `n_threads` host threads (0 = a quarter of the host's hardware threads, at most 64): the per-factor
// projection is the host cost of the gtsam path (an SVD / eigen-decomposition of a 45 x 45 and a 78 x 78 matrix per link
// direction, photometric_factor.cpp:142-149) -- ISAM2 pays it factor by factor, here it is paid once per Values in
// parallel and sage_window_factor only cuts blocks afterwards.
extern "C" int sage_window_prepare_factors(SageWindow *w, int psd_mode, int n_threads)
{
  if (!w || !w->finalized || psd_mode < 0 || psd_mode > 2)
    return SAGE_E_INVALID;
  SageWindow::FactorCache &fc = w->fc;
  if (!fc.lin)
    return SAGE_E_STATE;
  if (fc.psd_mode == psd_mode)
    return SAGE_OK;
  const int CS = w->cfg.CS;
  const size_t Dp = 13 + CS, Dg = 14 + 2 * CS;
  const size_t nep = fc.Ap.size() / (Dp * Dp), neg = fc.Ag.size() / (Dg * Dg);
  fc.psd_mode = -1; // the projected matrices are being rewritten: whatever they held is gone until this call succeeds
  fc.Cp.assign(nep * Dp * Dp, 0.0);
  fc.Cg.assign(neg * Dg * Dg, 0.0);
  const size_t total = nep + neg;
  if (n_threads <= 0)
    n_threads = (int)std::min<unsigned>(64u, std::max(1u, std::thread::hardware_concurrency() / 4)); // a quarter of the host, <= 64
  n_threads = (int)std::min<size_t>((size_t)n_threads, std::max<size_t>(1, total));
  std::atomic<size_t> next{0};
  std::atomic<int> bad{0};
  auto work = [&]() {
    for (;;)
    {
      // geometric factors first: they are the long jobs (78 x 78)
      const size_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= total)
        return;
      int rc;
      if (i < neg)
        rc = sage_factor_psd(1, CS, fc.Ag.data() + i * Dg * Dg, psd_mode, fc.Cg.data() + i * Dg * Dg);
      else
        rc = sage_factor_psd(0, CS, fc.Ap.data() + (i - neg) * Dp * Dp, psd_mode, fc.Cp.data() + (i - neg) * Dp * Dp);
      if (rc)
        bad.store(rc, std::memory_order_relaxed);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < n_threads; ++t)
    th.emplace_back(work);
  work();
  for (auto &t : th)
    t.join();
  if (bad.load())
    return bad.load();
  fc.psd_mode = psd_mode;
  return SAGE_OK;
}

