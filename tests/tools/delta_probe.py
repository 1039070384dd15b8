"""LM-delta distance of the engine from the committed golden deltas (tests/golden/window_delta_k*_seed*.npz), all seeds of a
window size in seconds (GPU; dev tool for accumulation-noise experiments: SAGE_PHOTO_FLUSH, SAGE_BA_LIB variants ...).
usage: python tests/tools/delta_probe.py [K ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sage_slam_amd import capi, synth          # noqa: E402
from tests.helpers import rel                  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden")
for K in [int(a) for a in sys.argv[1:]] or [16, 64]:
    out = []
    for seed in range(4):
        path = os.path.join(GOLD, f"window_delta_k{K}_seed{seed}.npz")
        if not os.path.exists(path):
            continue
        g = np.load(path)
        w = synth.make_window(K=K, H=128, W=160, FS=16, CS=32, L=4, seed=seed)
        win = capi.Window(w)
        win.linearize()
        win.solve(float(g["damp"]))
        dh = win.delta()
        out.append((seed, rel(dh, g["d32"]), rel(dh, g["d64"]), rel(g["d32"], g["d64"])))
        win.close()
    print(f"K={K}: " + " | ".join(f"seed {s}: hip-fp32oracle {a:.2e} hip-exact {b:.2e} (fp32oracle-exact {c:.2e})" for s, a, b, c in out),
          f"| worst hip-fp32oracle {max(o[1] for o in out):.2e}", flush=True)
