"""sage_window_tune_runs (r06): the run length of the photometric workgroups measured on the window itself.  The tuned plan
must give the same normal equations (to fp32 summation order) and the same LM behaviour as the rule's plan; a pinned run
length (SAGE_PHOTO_TPB) and a second call are no-ops."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from sage_slam_amd import capi as c
    c.lib()
    return c


@pytest.mark.parametrize("K,H,W,FS", [(8, 192, 256, 16), (6, 128, 160, 32)])
def test_tuned_runs_give_the_same_system(capi, K, H, W, FS):
    from sage_slam_amd import synth
    w = synth.make_window(K=K, H=H, W=W, FS=FS, CS=32, L=4, seed=71)
    win = capi.Window(w)
    win.linearize()
    p0 = win.packed_host().astype(np.float64)
    r = win.tune_runs()
    assert r["tpb"] in (4, 6, 8, 9, 10, 12, 16, r["tpb_rule"]) and r["ms_rule"] > 0
    assert r["ms_best"] <= r["ms_rule"] and (r["tpb"] == r["tpb_rule"] or r["ms_best"] < 0.96 * r["ms_rule"])
    win.linearize()
    p1 = win.packed_host().astype(np.float64)
    assert rel(p1[:-4], p0[:-4]) < 2e-6 and np.array_equal(p1[-2:], p0[-2:])      # inlier counts exact
    assert p1[-4] == pytest.approx(p0[-4], rel=1e-6) and p1[-3] == pytest.approx(p0[-3], rel=1e-6)
    # the LM iteration runs on the tuned plan (merged kernels, error pass) and descends
    st = capi.SageLmState(); cfg = capi.lm_config_default(); cfg.max_inner_evals = 1
    win.lm_step(st, cfg)
    assert st.accepted == 1 and st.candidate_error < st.error
    # an embedder re-applies a tuned run length to another window of the same geometry
    win2 = capi.Window(w)
    win2.set_runs(r["tpb"])
    win2.linearize()
    assert np.array_equal(win2.packed_host(), p1)                                   # same plan, same bits
    # a second call measures again and must not flip between near-equal candidates by more than the 3 % rule allows
    r2 = win.tune_runs()
    assert r2["tpb_rule"] == r["tpb_rule"]
    from tests.conftest import summary_line
    summary_line(f"[tune_runs K={K} {H}x{W}x{FS}] rule {r['tpb_rule']} ({r['ms_rule']:.3f} ms linearize + error pass) -> "
                 f"{r['tpb']} ({r['ms_best']:.3f} ms); second call -> {r2['tpb']}")


def test_pinned_run_length_is_not_tuned(tmp_path):
    code = f"""
import sys
sys.path.insert(0, {ROOT!r})
from sage_slam_amd import capi, synth
w = synth.make_window(K=4, H=128, W=160, FS=16, CS=32, L=4, seed=3)
win = capi.Window(w)
r = win.tune_runs()
assert r["tpb"] == 5 and r["ms_rule"] == 0.0, r
print("ok")
"""
    env = dict(os.environ); env["SAGE_PHOTO_TPB"] = "5"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-1500:]
