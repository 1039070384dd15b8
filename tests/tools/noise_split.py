"""Where the engine's LM-step noise comes from (dev tool, GPU; r06): per seed of the K = 5 noise-floor family, the step of MIXED
systems against the exact (fp64-oracle) step --
    photo-only   engine's photometric edges + exact geometric edges        geo-only   exact photometric + engine's geometric
and, for the photometric edges, block families substituted one at a time into the exact system:
    pp   pose x pose (incl. pose gradient)      pc   pose x code0 / scale0 x code0      cc   code0 x code0      gcode  code0 gradient
The same splits for the fp32 oracle next to them: which family makes the engine the noisier evaluation.
usage: python tests/tools/noise_split.py first_seed n_seeds"""
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
Dp = 13 + CS
POSE = np.arange(12); CODE = np.arange(12, 12 + CS); SC = np.array([12 + CS])


def family_mask(name):
    M = np.zeros((Dp, Dp), bool); v = np.zeros(Dp, bool)
    ps = np.concatenate([POSE, SC])
    if name == "pp":
        M[np.ix_(ps, ps)] = True; v[ps] = True
    elif name == "pc":
        M[np.ix_(ps, CODE)] = True; M[np.ix_(CODE, ps)] = True
    elif name == "cc":
        M[np.ix_(CODE, CODE)] = True
    elif name == "gcode":
        v[CODE] = True
    return M, v


acc = {}
for seed in range(s0, s0 + ns):
    w = synth.make_window(K=5, H=64, W=80, FS=16, CS=CS, L=4, seed=seed, back_links=2)
    K, B = len(w.keyframes), 7 + CS
    win = capi.Window(w); win.linearize()
    res = {"hip": {}, "f32": {}, "f64": {}}
    for l, (a, b) in enumerate(w.links):
        for d, (k0, k1) in enumerate(((a, b), (b, a))):
            for t, fn in ((0, oracle_photo), (1, oracle_geo)):
                res["hip"][(t, l, d)] = win.get_edge(t, 2 * l + d)
                for prec in ("f32", "f64"):
                    res[prec][(t, l, d)] = fn(orc, w, k0, k1, prec=prec)
    win.close()

    def step(edges):
        p = capi.assemble_packed(K, w.links, CS, edges)
        H, g, _ = capi.unpack_dense(p, K, w.links, CS)
        for k, kf in enumerate(w.keyframes):
            idx = np.arange(k * B + 6, k * B + 6 + CS)
            H[idx, idx] += 1e-3
            g[idx] += 1e-3 * (0 - kf.code.astype(np.float64))
        s = w.keyframes[0].scale
        H[6 + CS, 6 + CS] += 1e4 / (s * s)
        H[np.arange(6), np.arange(6)] += 1e4
        return damped_delta(H, g, 1e-3)

    de = step(res["f64"])
    out = {}
    for src in ("hip", "f32"):
        out[(src, "all")] = rel(step(res[src]), de)
        out[(src, "photo-only")] = rel(step({k: (res[src][k] if k[0] == 0 else res["f64"][k]) for k in res["f64"]}), de)
        out[(src, "geo-only")] = rel(step({k: (res[src][k] if k[0] == 1 else res["f64"][k]) for k in res["f64"]}), de)
        for fam in ("pp", "pc", "cc", "gcode"):
            M, v = family_mask(fam)
            mixed = {}
            for k, e64 in res["f64"].items():
                if k[0] != 0:
                    mixed[k] = e64
                    continue
                A = np.array(e64["AtA"], np.float64); bb = np.array(e64["Atb"], np.float64).reshape(-1)
                As = np.asarray(res[src][k]["AtA"], np.float64); bs = np.asarray(res[src][k]["Atb"], np.float64).reshape(-1)
                A[M] = As[M]; bb[v] = bs[v]
                mixed[k] = dict(AtA=A, Atb=bb, error=e64["error"], num_inliers=e64["num_inliers"])
            out[(src, "photo:" + fam)] = rel(step(mixed), de)
    print(f"seed {seed}: " + "  ".join(f"{k[1]} {out[('hip', k[1])]:.1e}/{out[('f32', k[1])]:.1e}" for k in out if k[0] == "hip"), flush=True)
    for k, v in out.items():
        acc.setdefault(k, []).append(v)
print("rms (engine / fp32 oracle):")
for k in [k for k in acc if k[0] == "hip"]:
    print(f"  {k[1]:12s} {np.sqrt(np.mean(np.square(acc[k]))):.2e} / {np.sqrt(np.mean(np.square(acc[('f32', k[1])]))):.2e}")
