"""f2: the per-Values batched prepass that serves the gtsam factors' linearize() / error() from one window
evaluation (core/gtsam/photometric_factor.cpp:72-219, geometric_factor.cpp:41-233, mapper.cpp:544-551).  Product =
sage_window_prepass / sage_window_factor / sage_window_factor_error through the C ABI; checker = the CPU oracle edge by
edge + the block cutting of sage_factor_hessian_blocks (itself pinned in tests/test_host_logic.py)."""
import copy

import numpy as np
import pytest

from sage_slam_amd import synth
from tests.helpers import oracle_geo, oracle_photo, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from sage_slam_amd import capi as c
    c.lib()
    return c


def window_values(w):
    poses = np.stack([np.concatenate([np.asarray(k.R, np.float32).reshape(-1), np.asarray(k.t, np.float32).reshape(-1)])
                      for k in w.keyframes])
    codes = np.stack([np.asarray(k.code, np.float32).reshape(-1) for k in w.keyframes])
    scales = np.array([k.scale for k in w.keyframes], np.float32)
    return poses, codes, scales


@pytest.mark.parametrize("CS", [16, 32])
def test_prepass_serves_every_factor_from_one_evaluation(capi, orc, CS):
    w = synth.make_window(K=4, H=64, W=80, FS=16, CS=CS, L=4, seed=5, back_links=2)
    win = capi.Window(w)
    poses, codes, scales = window_values(w)
    assert win.prepass(poses, codes, scales, jacobians=True) is True
    assert win.prepass(poses, codes, scales, jacobians=True) is False      # same Values: nothing launched
    assert win.prepass(poses, codes, scales, jacobians=False) is False     # the linearisation carries the errors
    nk = {0: 4, 1: 6}
    for l, (a, b) in enumerate(w.links):
        for d, (k0, k1) in enumerate(((a, b), (b, a))):
            for t, ofn in ((0, oracle_photo), (1, oracle_geo)):
                ref = ofn(orc, w, k0, k1)
                for psd in (0, 1, 2):
                    blocks, gs, f, dims = win.factor(t, 2 * l + d, psd_mode=psd)
                    assert len(dims) == nk[t] and dims[:2] == [6, 6] and dims[2] == CS
                    rb, rg, rdims = capi.factor_hessian_blocks(t, CS, ref["AtA"], ref["Atb"], psd_mode=psd)
                    assert rdims == dims
                    # psd 2 (the reference's as-written NearestPsd) amplifies input noise on gauge-deficient edges
                    # (DESIGN s6): compared through the engine's own fp32 result instead of the oracle's there
                    if psd == 2:
                        he = win.get_edge(t, 2 * l + d)
                        rb, rg, _ = capi.factor_hessian_blocks(t, CS, he["AtA"], he["Atb"], psd_mode=2)
                        tol = 1e-12
                    else:
                        tol = 2e-5
                    num = sum(np.sum((blocks[k] - rb[k]) ** 2) for k in rb)
                    den = sum(np.sum(rb[k] ** 2) for k in rb)
                    assert np.sqrt(num / den) < tol, (t, l, d, psd)
                    assert rel(np.concatenate(gs), np.concatenate(rg)) < (1e-12 if psd == 2 else 2e-4)
                    assert f == pytest.approx(ref["error"], rel=1e-5)
                assert win.factor_error(t, 2 * l + d) == pytest.approx(ref["error"], rel=1e-5)
    win.close()


def test_prepass_recomputes_on_new_values_and_error_only_path(capi, orc):
    CS = 32
    w = synth.make_window(K=4, H=64, W=80, FS=16, CS=CS, L=4, seed=6, back_links=2)
    win = capi.Window(w)
    poses, codes, scales = window_values(w)
    assert win.prepass(poses, codes, scales, jacobians=True)
    # new Values (what a Dogleg trial point looks like): a moved pose, another code, another scale
    rng = np.random.default_rng(0)
    w2 = copy.deepcopy(w)
    k1 = w2.keyframes[1]
    k1.t = (np.asarray(k1.t, np.float32) + np.float32(0.004) * rng.standard_normal(3).astype(np.float32)).astype(np.float32)
    w2.keyframes[2].code = (np.asarray(w2.keyframes[2].code, np.float32) +
                            np.float32(0.02) * rng.standard_normal(CS).astype(np.float32)).astype(np.float32)
    w2.keyframes[3].scale = float(np.float32(w2.keyframes[3].scale * 1.01))
    p2, c2, s2 = window_values(w2)
    assert win.prepass(p2, c2, s2, jacobians=False) is True                 # one error pass for the whole window
    assert win.prepass(p2, c2, s2, jacobians=False) is False
    with pytest.raises(capi.SageError):                                     # no linearisation at these values yet
        win.factor(0, 0)
    errs = {}
    for l, (a, b) in enumerate(w2.links):
        for d, (k0, k1_) in enumerate(((a, b), (b, a))):
            for t, ofn in ((0, oracle_photo), (1, oracle_geo)):
                ref = ofn(orc, w2, k0, k1_, jac=False)
                errs[(t, l, d)] = win.factor_error(t, 2 * l + d)
                assert errs[(t, l, d)] == pytest.approx(ref["error"], rel=2e-5), (t, l, d)
    # the window's current variables are the requested values
    for k, kf in enumerate(w2.keyframes):
        pose, code, s = win.get_keyframe(k)
        assert np.array_equal(pose, p2[k]) and np.array_equal(code, c2[k]) and s == s2[k]
    # linearize() at the same values: one more evaluation, same errors (a1 and a2 agree), blocks of the moved system
    assert win.prepass(p2, c2, s2, jacobians=True) is True
    for (t, l, d), e in errs.items():
        assert win.factor_error(t, 2 * l + d) == pytest.approx(e, rel=1e-5)
    ref = oracle_photo(orc, w2, *w2.links[0])
    blocks, gs, f, dims = win.factor(0, 0, psd_mode=1)
    rb, rg, _ = capi.factor_hessian_blocks(0, CS, ref["AtA"], ref["Atb"], psd_mode=1)
    assert rel(blocks[(2, 2)], rb[(2, 2)]) < 2e-5 and rel(np.concatenate(gs), np.concatenate(rg)) < 2e-4
    # back to the first values: recomputed again (the cache holds one Values)
    assert win.prepass(poses, codes, scales, jacobians=True) is True
    win.close()


def test_per_link_geometric_loss_parameter(capi, orc):
    """mapper.cpp:367-373: geo_loss_param = factor * avg_squared_dpt_bias of the link's newer keyframe -- one value per
    link, not per window.  sage_window_set_link_geo_loss: both geometric edges of the link (linearize, and the error pass
    that is fused into the photometric error kernel) use the link's value, the other links the window's."""
    CS = 32
    w = synth.make_window(K=3, H=64, W=80, FS=16, CS=CS, L=4, seed=8, back_links=2)
    w.link_geo_loss = [0.0, 2.5 * w.geo_loss_param, 0.4 * w.geo_loss_param][:len(w.links)]
    win = capi.Window(w)
    win.linearize()
    tot_lin = 0.0
    for l, (a, b) in enumerate(w.links):
        wl = copy.copy(w)
        if w.link_geo_loss[l] > 0:
            wl.geo_loss_param = w.link_geo_loss[l]
        for d, (k0, k1) in enumerate(((a, b), (b, a))):
            ref = oracle_geo(orc, wl, k0, k1)
            he = win.get_edge(1, 2 * l + d)
            assert rel(he["AtA"], ref["AtA"]) < 2e-5 and rel(he["Atb"], ref["Atb"]) < 2e-4, (l, d)
            assert he["error"] == pytest.approx(ref["error"], rel=1e-5)
            if w.link_geo_loss[l] > 0:   # ... and it is not the window's value
                other = oracle_geo(orc, w, k0, k1)
                assert abs(other["error"] - ref["error"]) > 1e-3 * abs(ref["error"])
            tot_lin += ref["error"]
    # error pass (geometric error evaluated inside the photometric error kernel)
    win.error(0)
    import torch
    torch.cuda.synchronize()
    tot = win.error_tensor().cpu().numpy()      # [err_photo err_geo n_photo n_geo]
    assert tot[1] == pytest.approx(tot_lin, rel=1e-5)
    win.close()


def test_prepare_factors_on_host_threads_matches_lazy(capi):
    """sage_window_prepare_factors: NearestPsd of every cached factor on host threads right after the prepass; the blocks
    sage_window_factor then hands out are bit-identical to the ones it computes factor by factor, for every psd_mode; a
    prepass at new Values invalidates the prepared set."""
    import time
    CS = 32
    w = synth.make_window(K=6, H=64, W=80, FS=16, CS=CS, L=4, seed=9)
    win = capi.Window(w)
    poses, codes, scales = window_values(w)
    assert win.prepass(poses, codes, scales, jacobians=True)
    ne = 2 * len(w.links)
    for psd in (1, 2, 0):
        t0 = time.perf_counter()
        lazy = {(t, e): win.factor(t, e, psd_mode=psd) for t in (0, 1) for e in range(ne)}
        t_lazy = time.perf_counter() - t0
        t0 = time.perf_counter()
        win.prepare_factors(psd, 0)
        t_prep = time.perf_counter() - t0
        t0 = time.perf_counter()
        for (t, e), (blocks, gs, f, dims) in lazy.items():
            b2, g2, f2, d2 = win.factor(t, e, psd_mode=psd)
            assert d2 == dims and f2 == f
            assert all(np.array_equal(b2[k], blocks[k]) for k in blocks)
            assert all(np.array_equal(x, y) for x, y in zip(g2, gs))
        t_cut = time.perf_counter() - t0
        print(f"psd_mode {psd}: {2 * ne} factors lazily {1e3 * t_lazy:.1f} ms; prepared on host threads {1e3 * t_prep:.1f} ms "
              f"+ block cutting {1e3 * t_cut:.1f} ms")
        # a different mode is served lazily again (and still right)
        other = 1 if psd != 1 else 2
        bo, _, _, _ = win.factor(0, 0, psd_mode=other)
        assert all(np.array_equal(bo[k], v) for k, v in
                   capi.factor_hessian_blocks(0, CS, win.get_edge(0, 0)["AtA"], win.get_edge(0, 0)["Atb"], psd_mode=other)[0].items())
    # new Values: the prepared set is dropped with the cache
    poses2 = poses.copy(); poses2[1, 9:] += np.float32(0.003)
    assert win.prepass(poses2, codes, scales, jacobians=True)
    b_new, _, _, _ = win.factor(0, 0, psd_mode=1)
    he = win.get_edge(0, 0)
    ref = capi.factor_hessian_blocks(0, CS, he["AtA"], he["Atb"], psd_mode=1)[0]
    assert all(np.array_equal(b_new[k], ref[k]) for k in ref)
    win.close()
