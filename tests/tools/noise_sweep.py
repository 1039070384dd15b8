"""Seed sweep of the K = 5 noise-floor window (tests/test_gpu_parity.py::test_window_step_noise_floor): LM-step distance of
the engine and of the fp32 oracle from the exact (fp64-oracle) step (dev tool, GPU).
usage: python tests/tools/noise_sweep.py first_seed n_seeds"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sage_slam_amd import capi, synth                                      # noqa: E402
from tests.helpers import rel, oracle_photo, oracle_geo, damped_delta    # noqa: E402
from oracle import oracle as orc                                           # noqa: E402

orc.build()
CS = 32
s0, ns = int(sys.argv[1]), int(sys.argv[2])
rows = []
for seed in range(s0, s0 + ns):
    w = synth.make_window(K=5, H=64, W=80, FS=16, CS=CS, L=4, seed=seed, back_links=2)
    K, B = len(w.keyframes), 7 + CS
    win = capi.Window(w); win.linearize(); ph = win.packed_host(); win.close()
    sysm = {}
    for prec in ("f32", "f64"):
        res = {}
        for l, (a, b) in enumerate(w.links):
            for d, (k0, k1) in enumerate(((a, b), (b, a))):
                res[(0, l, d)] = oracle_photo(orc, w, k0, k1, prec=prec)
                res[(1, l, d)] = oracle_geo(orc, w, k0, k1, prec=prec)
        sysm[prec] = capi.assemble_packed(K, w.links, CS, res)

    def system(p):
        H, g, _ = capi.unpack_dense(p, K, w.links, CS)
        for k, kf in enumerate(w.keyframes):
            idx = np.arange(k * B + 6, k * B + 6 + CS)
            H[idx, idx] += 1e-3
            g[idx] += 1e-3 * (0 - kf.code.astype(np.float64))
        s = w.keyframes[0].scale
        H[6 + CS, 6 + CS] += 1e4 / (s * s)
        H[np.arange(6), np.arange(6)] += 1e4
        return H, g
    (Hh, gh), (Ho, go), (He, ge) = system(ph), system(sysm["f32"]), system(sysm["f64"])
    dh, do, de = (damped_delta(H, g, 1e-3) for H, g in ((Hh, gh), (Ho, go), (He, ge)))
    rows.append((seed, rel(dh, de), rel(do, de), rel(dh, do), rel(damped_delta(He, gh, 1e-3), de), rel(damped_delta(Hh, ge, 1e-3), de)))
    print("seed %d: hip-exact %.2e  oracle-exact %.2e  hip-oracle %.2e   (g only %.2e, H only %.2e)" % rows[-1], flush=True)
a = np.array(rows)[:, 1:]
print("mean hip-exact %.2e  oracle-exact %.2e  hip-oracle %.2e | max hip-oracle %.2e | rms hip-exact %.2e oracle-exact %.2e" %
      (a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 2].max(), np.sqrt((a[:, 0] ** 2).mean()), np.sqrt((a[:, 1] ** 2).mean())))
