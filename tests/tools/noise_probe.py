"""Noise-floor probe (GPU): distance of the HIP window system / LM step from the exact (fp64-oracle) one,
next to the fp32 oracle's own distance.  Dev tool; prints one line per seed."""
import sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sage_slam_amd import synth, capi
from oracle import oracle as orc
from tests.helpers import oracle_photo, oracle_geo, damped_delta, rel

CS = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for seed in range(21, 27):
    w = synth.make_window(K=5, H=64, W=80, FS=16, CS=CS, L=4, seed=seed, back_links=2)
    win = capi.Window(w); win.linearize()
    ph = win.packed_host().astype(np.float64)
    K = len(w.keyframes); B = 7 + CS
    out = {}
    for prec in ("f32", "f64"):
        res = {}
        for l, (a, b) in enumerate(w.links):
            for d, (k0, k1) in enumerate(((a, b), (b, a))):
                res[(0, l, d)] = oracle_photo(orc, w, k0, k1, prec=prec)
                res[(1, l, d)] = oracle_geo(orc, w, k0, k1, prec=prec)
        out[prec] = capi.assemble_packed(K, w.links, CS, res)
    def sys_of(p):
        H, g, _ = capi.unpack_dense(p, K, w.links, CS)
        for k, kf in enumerate(w.keyframes):
            idx = np.arange(k * B + 6, k * B + 6 + CS)
            H[idx, idx] += 1e-3; g[idx] += 1e-3 * (0 - kf.code.astype(np.float64))
        s = w.keyframes[0].scale
        H[6 + CS, 6 + CS] += 1e4 / (s * s); H[np.arange(6), np.arange(6)] += 1e4
        return H, g
    Hh, gh = sys_of(ph); Ho, go = sys_of(out["f32"]); He, ge = sys_of(out["f64"])
    damp = 1e-3
    dh, do, de = damped_delta(Hh, gh, damp), damped_delta(Ho, go, damp), damped_delta(He, ge, damp)
    pm = np.zeros(K * B, bool)
    for k in range(K):
        pm[k * B:k * B + 6] = True
    cond = np.linalg.cond(He + damp * np.diag(np.diag(He)))
    print(f"seed {seed} CS {CS}: delta hip-exact {rel(dh, de):.2e} orc32-exact {rel(do, de):.2e} hip-orc32 {rel(dh, do):.2e} | "
          f"H hip {rel(Hh, He):.1e} orc {rel(Ho, He):.1e} | Hpp hip {rel(Hh[np.ix_(pm, pm)], He[np.ix_(pm, pm)]):.1e} orc {rel(Ho[np.ix_(pm, pm)], He[np.ix_(pm, pm)]):.1e} | "
          f"Hcc hip {rel(Hh[np.ix_(~pm, ~pm)], He[np.ix_(~pm, ~pm)]):.1e} orc {rel(Ho[np.ix_(~pm, ~pm)], He[np.ix_(~pm, ~pm)]):.1e} | "
          f"g hip {rel(gh, ge):.1e} orc {rel(go, ge):.1e} | cond {cond:.1e}")
    # block-wise: p = pose, c = code, s = scale
    idx = {"p": [], "c": [], "s": []}
    for k in range(K):
        idx["p"] += list(range(k * B, k * B + 6)); idx["c"] += list(range(k * B + 6, k * B + 6 + CS)); idx["s"].append(k * B + 6 + CS)
    line = "   blocks hip/orc:"
    for a in "pcs":
        for b in "pcs":
            if a > b and False:
                continue
            ia, ib = np.array(idx[a]), np.array(idx[b])
            line += f" H{a}{b} {rel(Hh[np.ix_(ia, ib)], He[np.ix_(ia, ib)]):.0e}/{rel(Ho[np.ix_(ia, ib)], He[np.ix_(ia, ib)]):.0e}"
    for a in "pcs":
        ia = np.array(idx[a])
        line += f" g{a} {rel(gh[ia], ge[ia]):.0e}/{rel(go[ia], ge[ia]):.0e}"
    # which perturbation drives the step error: swap in exact H or exact g
    line += f" | step err with exact H: {rel(damped_delta(He, gh, damp), de):.1e}, with exact g: {rel(damped_delta(Hh, ge, damp), de):.1e}"
    print(line)
    win.close()
