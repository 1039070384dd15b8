#!/usr/bin/env python3
"""Golden LM deltas of the BASELINE windows from the CPU oracle (CPU only; run anywhere the oracle builds).

For window (K, seed) of sage_slam_amd/synth.py at 128x160x16 / CS 32 / L 4 (BASELINE configs 2 and 3): every directed edge
(2 photometric + 2 geometric per link) goes through oracle/sage_oracle.c in fp32 and in fp64, the per-edge results are
assembled into the packed block system (capi.assemble_packed -- numpy, no device), the engine's default priors are added
and the damped system (H + 1e-3 diag H) d = g is solved in double:

    d32  LM delta of the fp32-oracle system (the reference's arithmetic)      d64  of the fp64-oracle system ("exact")
    atb64 / atb32_floor (seed 0 only): per-edge fp64 Atb and rel(Atb32, Atb64), for the per-edge floor rule of
    tests/test_gpu_configs.py

The GPU tests compare the engine's delta with d32 / d64 in seconds instead of re-running 2 x 744 dense oracle edges per
seed (the K = 64 double pass took 3.5-9 min of host time per seed).  Seed 0 still runs the fp32
oracle live on every edge in the GPU test; this file only caches what is a pure function of (K, seed).

usage: python tests/golden/make_window_delta_golden.py [K ...]      (default: 16 64; seeds 0-7)
"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as orc                                    # noqa: E402
from sage_slam_amd import capi, synth                              # noqa: E402
from tests.helpers import damped_delta, oracle_geo, oracle_photo, rel   # noqa: E402

DAMP = 1e-3
SEEDS = tuple(range(8))


def add_priors(H, g, w, CS):
    B = 7 + CS
    for k, kf in enumerate(w.keyframes):
        idx = np.arange(k * B + 6, k * B + 6 + CS)
        H[idx, idx] += 1e-3
        g[idx] += 1e-3 * (0 - kf.code.astype(np.float64))
    s = float(w.keyframes[0].scale)
    H[6 + CS, 6 + CS] += 1e4 / (s * s)
    H[np.arange(6), np.arange(6)] += 1e4
    return H, g


def oracle_all_edges(w, precs=("f32", "f64")):
    """{prec: {(type, link, dir): result}}; edges in parallel host threads (ctypes releases the GIL), OpenMP inside"""
    cores = os.cpu_count() or 1
    workers = max(1, min(16, cores // 8))
    omp = max(1, cores // workers)
    jobs = [(t, l, d, k0, k1) for l, (a, b) in enumerate(w.links) for d, (k0, k1) in enumerate(((a, b), (b, a)))
            for t in (0, 1)]

    def run(job):
        t, l, d, k0, k1 = job
        orc.set_threads(omp)                                        # per calling thread (libgomp ICV)
        fn = oracle_photo if t == 0 else oracle_geo
        return job, {p: fn(orc, w, k0, k1, prec=p) for p in precs}

    out = {p: {} for p in precs}
    with ThreadPoolExecutor(workers) as ex:
        for (t, l, d, _, _), r in ex.map(run, jobs):
            for p in precs:
                out[p][(t, l, d)] = r[p]
    return out


def main():
    orc.build()
    Ks = [int(a) for a in sys.argv[1:]] or [16, 64]
    for K in Ks:
        for seed in SEEDS:
            path = os.path.join(HERE, f"window_delta_k{K}_seed{seed}.npz")
            if os.path.exists(path):
                print("exists:", path)
                continue
            t0 = time.time()
            w = synth.make_window(K=K, H=128, W=160, FS=16, CS=32, L=4, seed=seed)
            CS = w.CS
            res = oracle_all_edges(w)
            d = {}
            for p in ("f32", "f64"):
                packed = capi.assemble_packed(K, w.links, CS, res[p])
                H, g = add_priors(*capi.unpack_dense(packed, K, w.links, CS)[:2], w, CS)
                d[p] = damped_delta(H, g, DAMP)
                if p == "f64":
                    cond = np.linalg.cond(H + DAMP * np.diag(np.diag(H)))
            arrs = dict(d32=d["f32"], d64=d["f64"], damp=np.float64(DAMP), n_links=np.int64(len(w.links)),
                        N=np.int64(w.keyframes[0].homo.shape[0]), cond=np.float64(cond))
            if seed == 0:
                keys = sorted(res["f64"])
                arrs["atb64"] = np.concatenate([res["f64"][k]["Atb"].astype(np.float64).reshape(-1) for k in keys])
                arrs["atb32_floor"] = np.array([rel(res["f32"][k]["Atb"], res["f64"][k]["Atb"]) for k in keys])
                arrs["edge_keys"] = np.array(keys, np.int32)
            np.savez_compressed(path, **arrs)
            print(f"K={K} seed={seed}: {4 * len(w.links)} edges x 2 precisions in {time.time() - t0:.0f} s; "
                  f"fp32oracle-exact {rel(d['f32'], d['f64']):.2e}  cond {cond:.1e} -> {os.path.basename(path)}", flush=True)


if __name__ == "__main__":
    main()
