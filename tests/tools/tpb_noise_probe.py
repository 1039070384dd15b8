#!/usr/bin/env python3
"""LM-delta noise of the K-keyframe window against the exact (fp64-oracle) step as a function of the fp32 accumulation
run length of the two linearize kernels (SAGE_PHOTO_TPB / SAGE_GEO_TPB = sub-tiles a workgroup sums before it writes a
partial record; the partials are summed in double).  (SAGE_PHOTO_TPB = sub-tiles a workgroup walks, SAGE_PHOTO_FLUSH = sub-tiles per partial record).
usage: python tests/tools/tpb_noise_probe.py [K]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc
from sage_slam_amd import capi, synth
from tests.helpers import damped_delta, oracle_geo, oracle_photo, rel
from tests.test_gpu_configs import add_priors

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 0
orc.build()
w = synth.make_window(K=K, H=128, W=160, FS=16, CS=32, L=4, seed=SEED)
CS = 32
t0 = time.time()
res = {"f32": {}, "f64": {}}
for l, (a, b) in enumerate(w.links):
    for d, (k0, k1) in enumerate(((a, b), (b, a))):
        for prec in res:
            res[prec][(0, l, d)] = oracle_photo(orc, w, k0, k1, prec=prec)
            res[prec][(1, l, d)] = oracle_geo(orc, w, k0, k1, prec=prec)
print(f"oracle {time.time() - t0:.0f} s", flush=True)
D = {}
for prec in res:
    H, g = add_priors(*capi.unpack_dense(capi.assemble_packed(K, w.links, CS, res[prec]), K, w.links, CS)[:2], w, CS)
    D[prec] = damped_delta(H, g, 1e-3)
print(f"fp32 oracle vs exact: {rel(D['f32'], D['f64']):.2e}")
CASES = [(8, 8), (8, 4), (8, 2), (8, 1)] if SEED else [(8, 8), (8, 4), (8, 2), (8, 1), (4, 4), (16, 16)]
for pt, fl in CASES:   # (run length, sub-tiles per partial record); the LDS second level of the cross / pose tiles is always on
    os.environ["SAGE_PHOTO_TPB"] = str(pt); os.environ["SAGE_PHOTO_FLUSH"] = str(fl); gt = 16
    win = capi.Window(w)
    win.set_profiling(True)
    for _ in range(5):
        win.linearize()
    win.solve(1e-3)
    dh = win.delta()
    kt = [win.kernel_time(i) for i in range(2)]
    print(f"photo run {pt:2d} sub-tiles, record every {fl:2d}: hip-exact {rel(dh, D['f64']):.2e}  hip-fp32oracle {rel(dh, D['f32']):.2e}   "
          f"photo lin {kt[0][0] / max(1, kt[0][1]):.3f} ms  geo lin {kt[1][0] / max(1, kt[1][1]):.3f} ms", flush=True)
    win.close()
