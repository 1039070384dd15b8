"""Switches of the photometric linearize that change HOW the work is laid out, not what is computed: the walk order of the
samples (SAGE_SAMPLE_TILE=WxH; the default 8x8 tiles feed the LDS-staged sampler, the raster walk 0x0 and wide tiles send
most waves through the texture-path sampler) and the partial-record cadence (SAGE_PHOTO_FLUSH).  The switches are read once per process, so every
variant runs in its own subprocess on the same window; the packed normal equations must agree with the default build's to
fp32 accumulation-order noise, inlier totals exactly."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = """
import sys, numpy as np
sys.path.insert(0, {root!r})
from sage_slam_amd import capi, synth
w = synth.make_window(K=24, H=128, W=160, FS=16, CS=32, L=4, seed=4)     # 8.3 k sub-tiles: runs of 8 per workgroup
win = capi.Window(w)
win.linearize()
p = win.packed_host().astype(np.float64)
win.solve(1e-3)
np.savez({out!r}, packed=p, delta=win.delta())
"""


def _run(tmp_path, name, env):
    out = str(tmp_path / (name + ".npz"))
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", SNIPPET.format(root=ROOT, out=out)], env=e, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def test_layout_switches_do_not_change_the_result(tmp_path):
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    base = _run(tmp_path, "base", {})
    for name, env in (("raster", {"SAGE_SAMPLE_TILE": "0x0"}), ("tile16x4", {"SAGE_SAMPLE_TILE": "16x4"}),
                      ("tile4x16", {"SAGE_SAMPLE_TILE": "4x16"}), ("flush0", {"SAGE_PHOTO_FLUSH": "0"}),
                      ("flush2", {"SAGE_PHOTO_FLUSH": "2"})):
        v = _run(tmp_path, name, env)
        assert np.array_equal(v["packed"][-2:], base["packed"][-2:]), name          # inlier totals: exact
        assert v["packed"][-4:-2] == pytest.approx(base["packed"][-4:-2], rel=1e-6), name
        r = rel(v["packed"][:-4], base["packed"][:-4])
        rd = rel(v["delta"], base["delta"])
        print(f"{name}: packed rel {r:.2e}  LM delta rel {rd:.2e}")
        assert r < 1e-6, (name, r)                                                    # fp32 accumulation order only
        assert rd < 2e-4, (name, rd)                                                  # (cond ~1e9 amplifies it)


# ---------------------------------------------------------------------------------------------------------------
# r06: image-border taps with mask = 1.  synth.make_window zeroes a 2-px mask border by default, so a
# level-0 tap that leaves the image never carried weight in the suite.  Here border = 0, erode = 0: every pixel is sampled,
# the mask is 1 up to the edge, inliers whose bilinear footprint straddles the edge exist at every level -- the per-tap zero
# padding of photometric_factor_kernels.cpp:168-222 on the LDS-staged sampler (bounding-box clamps, repeated border texels
# at weight 0) and on the texture-path sampler (SAGE_SAMPLE_TILE=0x0), per edge against the oracle.
# ---------------------------------------------------------------------------------------------------------------
BORDER_SNIPPET = """
import sys, numpy as np
sys.path.insert(0, {root!r})
from sage_slam_amd import capi, synth
w = synth.make_window(K=4, H={H}, W={W}, FS=16, CS=32, L=4, seed={seed}, border={border}, erode={erode}, pose_noise=2.0, back_links=3)
for k, dz in enumerate({dz!r}):
    w.keyframes[k].t[2] += np.float32(dz)
win = capi.Window(w)
win.linearize()
out = dict(packed=win.packed_host().astype(np.float64))
for t in (0, 1):
    for e in range(2 * len(w.links)):
        r = win.get_edge(t, e)
        out[f"AtA_{{t}}_{{e}}"] = r["AtA"]; out[f"Atb_{{t}}_{{e}}"] = r["Atb"]
        out[f"st_{{t}}_{{e}}"] = np.array([r["error"], r["num_inliers"]], np.float64)
# error pass (candidate = current variables after a zero-damping-free reset): per-window totals
win.error(0)
out["err_tot"] = np.array([win.total_error(False)])
# merged linearize of the LM iteration at the same point
st = capi.SageLmState(); cfg = capi.lm_config_default(); cfg.max_inner_evals = 1
win.lm_step(st, cfg)
out["merged"] = win.packed_host().astype(np.float64)
out["lm"] = np.array([st.error, st.candidate_error, st.accepted])
np.savez({out!r}, **out)
"""


def _edge_crossers(w, k0, k1):
    """inliers of edge k0 -> k1 whose level-0 bilinear footprint leaves the image (numpy, fp64: a count, not a reference)"""
    from sage_slam_amd import synth
    a, b = w.keyframes[k0], w.keyframes[k1]
    R10, t10 = synth.relative_pose(a.R, a.t, b.R, b.t)
    d = (np.float64(a.scale) * (a.bias.astype(np.float64) + a.basis.astype(np.float64) @ a.code.astype(np.float64)))[a.loc1d]
    X = (a.homo.astype(np.float64) * d[:, None]) @ np.asarray(R10, np.float64).reshape(3, 3).T + np.asarray(t10, np.float64)
    c = w.cams[0]
    p = X[:, 0] / X[:, 2] * float(c.fx) + float(c.cx); q = X[:, 1] / X[:, 2] * float(c.fy) + float(c.cy)
    rp, rq = np.rint(p), np.rint(q)
    inl = (X[:, 2] > w.eps) & (rp >= 0) & (rp < w.W) & (rq >= 0) & (rq < w.H)
    fx, fy = np.floor(p), np.floor(q)
    cross = inl & ((fx < 0) | (fx + 1 >= w.W) | (fy < 0) | (fy + 1 >= w.H))
    return int(cross.sum()), int(inl.sum())


@pytest.mark.parametrize("H,W,seed", [(64, 80, 51), (128, 160, 52)])
def test_border_taps_with_full_mask_match_oracle(tmp_path, orc, H, W, seed):
    from sage_slam_amd import synth
    from tests.helpers import oracle_geo, oracle_photo, rel
    w = synth.make_window(K=4, H=H, W=W, FS=16, CS=32, L=4, seed=seed, border=0, erode=0, pose_noise=2.0, back_links=3)
    assert w.mask.min() == 1.0 and w.keyframes[0].homo.shape[0] == H * W
    edges = [(k0, k1) for (a, b) in w.links for (k0, k1) in ((a, b), (b, a))]
    crossers = [_edge_crossers(w, k0, k1) for k0, k1 in edges]
    assert sum(c for c, _ in crossers) >= 50 and min(c for c, _ in crossers) >= 1, crossers   # the case under test exists on every edge
    oracle = {}
    for e, (k0, k1) in enumerate(edges):
        oracle[(0, e)] = oracle_photo(orc, w, k0, k1)
        oracle[(1, e)] = oracle_geo(orc, w, k0, k1)
    runs = {}
    for name, env in (("staged", {}), ("texture_path", {"SAGE_SAMPLE_TILE": "0x0"})):
        out = str(tmp_path / f"border_{name}.npz")
        e_ = dict(os.environ); e_.update(env)
        r = subprocess.run([sys.executable, "-c", BORDER_SNIPPET.format(root=ROOT, out=out, H=H, W=W, seed=seed, border=0, erode=0, dz=[])], env=e_,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        v = runs[name] = np.load(out)
        worst = [0.0, 0.0]
        for (t, e), o in oracle.items():
            st = v[f"st_{t}_{e}"]
            assert int(st[1]) == o["num_inliers"] > 0, (name, t, e)
            assert st[0] == pytest.approx(o["error"], rel=2e-5), (name, t, e)
            ra, rb = rel(v[f"AtA_{t}_{e}"], o["AtA"]), rel(v[f"Atb_{t}_{e}"], o["Atb"])
            worst = [max(worst[0], ra), max(worst[1], rb)]
            assert ra < 2e-5 and rb < 2e-5, (name, t, e, ra, rb)
        # error pass == the linearize's own error totals; merged linearize == separate kernels' system
        tot = sum(o["error"] for o in oracle.values())
        assert float(v["packed"][-4] + v["packed"][-3]) == pytest.approx(tot, rel=2e-5)
        assert rel(v["merged"][:-4], v["packed"][:-4]) < 2e-6 and np.array_equal(v["merged"][-2:], v["packed"][-2:])
        assert v["lm"][2] == 1 and v["lm"][1] < v["lm"][0]
        from tests.conftest import summary_line
        summary_line(f"[border=0 {H}x{W} {name}] {sum(c for c, _ in crossers)} inliers with a level-0 tap outside the image over "
                     f"{len(edges)} edges; worst per-edge rel-L2 vs the oracle: AtA {worst[0]:.1e} Atb {worst[1]:.1e}")
    assert np.array_equal(runs["staged"]["packed"][-2:], runs["texture_path"]["packed"][-2:])


# ---------------------------------------------------------------------------------------------------------------
# r06: the four waves of a photometric workgroup meet at one s_barrier per staging fill (photo_kernels.hip, SAGE_PHOTO_LOCKSTEP).
# A wave whose footprint does not fit the staging regions samples through the texture path, a wave without inliers skips
# the sampling altogether -- both must still execute the SAME NUMBER of barriers as their staged neighbours, or the
# workgroup hangs (or, worse, pairs up barriers of different fills).  Here keyframes 1 and 3 are pushed towards the scene:
# the zoom between the keyframes runs through the staging limit (level-0 box of 128 texels: ~1.2-1.4 x) inside the
# image, so workgroups with staged AND texture-path waves exist, the ragged sample rectangle (erode = 5) adds dead lanes
# (tile-padded order), and the far keyframes' edges have tiles without inliers.  Per edge against the oracle.
# ---------------------------------------------------------------------------------------------------------------
def _level0_box_fits(w, k0, k1):
    """per 8 x 8 source tile of edge k0 -> k1: does the inliers' level-0 bounding box fit 128 texels?  (numpy, fp64: an
    existence check for the case under test, not a reference) -> {(tile_y, tile_x): fits}"""
    from sage_slam_amd import synth
    a, b = w.keyframes[k0], w.keyframes[k1]
    R10, t10 = synth.relative_pose(a.R, a.t, b.R, b.t)
    d = (np.float64(a.scale) * (a.bias.astype(np.float64) + a.basis.astype(np.float64) @ a.code.astype(np.float64)))[a.loc1d]
    X = (a.homo.astype(np.float64) * d[:, None]) @ np.asarray(R10, np.float64).reshape(3, 3).T + np.asarray(t10, np.float64)
    c = w.cams[0]
    p = X[:, 0] / X[:, 2] * float(c.fx) + float(c.cx); q = X[:, 1] / X[:, 2] * float(c.fy) + float(c.cy)
    rp, rq = np.rint(p), np.rint(q)
    inl = (X[:, 2] > w.eps) & (rp >= 0) & (rp < w.W) & (rq >= 0) & (rq < w.H)
    inl &= w.mask[np.clip(rq, 0, w.H - 1).astype(int), np.clip(rp, 0, w.W - 1).astype(int)] > 0
    ty, tx = (a.loc1d // w.W) // 8, (a.loc1d % w.W) // 8
    fits = {}
    for key in set(zip(ty.tolist(), tx.tolist())):
        m = inl & (ty == key[0]) & (tx == key[1])
        if not m.any():
            fits[key] = None     # a dead slice
            continue
        bw = np.floor(p[m].max()) + 1 - np.floor(p[m].min()) + 1
        bh = np.floor(q[m].max()) + 1 - np.floor(q[m].min()) + 1
        fits[key] = bool(bw * bh <= 128)
    return fits


@pytest.mark.timeout(900)
def test_mixed_sampler_paths_inside_a_workgroup_match_oracle(tmp_path, orc):
    from sage_slam_amd import synth
    from tests.helpers import oracle_geo, oracle_photo, rel
    H, W, seed, dz = 128, 160, 61, [0.0, 0.2, 0.03, 0.27]
    w = synth.make_window(K=4, H=H, W=W, FS=16, CS=32, L=4, seed=seed, border=2, erode=5, pose_noise=2.0, back_links=3)
    for k, z in enumerate(dz):
        w.keyframes[k].t[2] += np.float32(z)
    edges = [(k0, k1) for (a, b) in w.links for (k0, k1) in ((a, b), (b, a))]
    # the case under test exists: groups of four x-adjacent tiles (= the four waves of a workgroup's sub-tile) with both kinds
    mixed = dead = 0
    for k0, k1 in edges:
        f = _level0_box_fits(w, k0, k1)
        for (ty, tx), v in f.items():
            grp = [f.get((ty, (tx // 4) * 4 + i)) for i in range(4)]
            mixed += (tx % 4 == 0) and (True in grp) and (False in grp)
            dead += v is None
    assert mixed >= 20, mixed
    oracle = {}
    for e, (k0, k1) in enumerate(edges):
        oracle[(0, e)] = oracle_photo(orc, w, k0, k1)
        oracle[(1, e)] = oracle_geo(orc, w, k0, k1)
    out = str(tmp_path / "mixed.npz")
    r = subprocess.run([sys.executable, "-c", BORDER_SNIPPET.format(root=ROOT, out=out, H=H, W=W, seed=seed, border=2, erode=5, dz=dz)],
                       env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    v = np.load(out)
    worst = [0.0, 0.0]
    for (t, e), o in oracle.items():
        st = v[f"st_{t}_{e}"]
        assert int(st[1]) == o["num_inliers"], (t, e)
        if o["num_inliers"] == 0:
            continue
        assert st[0] == pytest.approx(o["error"], rel=2e-5), (t, e)
        ra, rb = rel(v[f"AtA_{t}_{e}"], o["AtA"]), rel(v[f"Atb_{t}_{e}"], o["Atb"])
        worst = [max(worst[0], ra), max(worst[1], rb)]
        assert ra < 2e-5 and rb < 2e-5, (t, e, ra, rb)
    assert rel(v["merged"][:-4], v["packed"][:-4]) < 2e-6 and np.array_equal(v["merged"][-2:], v["packed"][-2:])
    from tests.conftest import summary_line
    summary_line(f"[mixed sampler paths] {mixed} sub-tiles with staged and texture-path waves side by side, {dead} tiles without "
                 f"inliers over {len(edges)} edges; worst per-edge rel-L2 vs the oracle: AtA {worst[0]:.1e} Atb {worst[1]:.1e}")
