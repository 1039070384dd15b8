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
