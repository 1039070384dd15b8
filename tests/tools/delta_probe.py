"""LM-delta distance of the engine from the committed golden deltas (tests/golden/window_delta_k*_seed*.npz), all seeds of a
window size in seconds (GPU; dev tool for accumulation-noise experiments: SAGE_PHOTO_FLUSH, SAGE_GEO_TPB, SAGE_BA_LIB variants).
Both linearize paths: the separate kernels (sage_window_linearize) and the merged ones of the LM iteration (r05).
usage: python tests/tools/delta_probe.py [K ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sage_slam_amd import capi, synth          # noqa: E402
from tests.helpers import rel, damped_delta    # noqa: E402
from tests.test_gpu_configs import add_priors  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden")
for K in [int(a) for a in sys.argv[1:]] or [16, 64]:
    out = []
    for seed in range(8):
        path = os.path.join(GOLD, f"window_delta_k{K}_seed{seed}.npz")
        if not os.path.exists(path):
            continue
        g = np.load(path)
        w = synth.make_window(K=K, H=128, W=160, FS=16, CS=32, L=4, seed=seed)
        win = capi.Window(w)
        win.linearize()
        win.solve(float(g["damp"]))
        dh = win.delta()
        cfg = capi.lm_config_default(); cfg.max_inner_evals = 1; cfg.linearize_at_candidate = -1
        win.reset()
        win.lm_step(capi.SageLmState(), cfg)
        Hm, gm = add_priors(*capi.unpack_dense(win.packed_host().astype(np.float64), K, w.links, w.CS)[:2], w, w.CS)
        dm = damped_delta(Hm, gm, float(g["damp"]))
        out.append((seed, rel(dh, g["d32"]), rel(dh, g["d64"]), rel(g["d32"], g["d64"]), rel(dm, g["d32"]), rel(dm, g["d64"])))
        win.close()
    print(f"K={K} separate: " + " ".join(f"{a:.2e}/{b:.2e}" for s, a, b, c, d, e in out) + f" | worst vs fp32 oracle {max(o[1] for o in out):.2e}")
    print(f"K={K} merged:   " + " ".join(f"{d:.2e}/{e:.2e}" for s, a, b, c, d, e in out) + f" | worst vs fp32 oracle {max(o[4] for o in out):.2e}"
          f"   (fp32 oracle vs exact: " + " ".join(f"{c:.1e}" for s, a, b, c, d, e in out) + ")", flush=True)
