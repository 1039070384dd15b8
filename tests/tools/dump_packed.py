"""Dump the engine's packed normal equations of the K = 5 noise-floor windows (tests/test_gpu_parity.py) for offline
block-by-block analysis against the oracle (dev tool, GPU).  usage: python tests/tools/dump_packed.py out.npz [seeds...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sage_slam_amd import capi, synth          # noqa: E402

out = sys.argv[1]
seeds = [int(a) for a in sys.argv[2:]] or [22, 23, 24]
res = {}
for seed in seeds:
    w = synth.make_window(K=5, H=64, W=80, FS=16, CS=32, L=4, seed=seed, back_links=2)
    win = capi.Window(w)
    win.linearize()
    res[f"packed_{seed}"] = win.packed_host().astype(np.float64)
    win.close()
np.savez(out, **res)
print("saved", out, list(res))
