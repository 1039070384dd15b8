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

r06: BASELINE configs 4 and 5 at their own sizes --
    cfg4   K = 16, 256x320x32, CS 32, seed 41, dense (168 edges of 76 k pixels)      -> window_delta_cfg4_k16_seed41.npz
    cfg5   K = 512, 64x80x16, N = 3072, seed 7, loop_radius 0.12: the 1 530 temporal links alone and with the five
           loop-closure links of tests/test_gpu_configs.py (6 120 / 6 140 edges; the temporal edges are evaluated once)
                                                                                    -> window_delta_cfg5_k512_seed7_{noloops,loops}.npz
    the 19 968-unknown systems of cfg5 are solved with a sparse LU (scipy splu, fp64) -- nothing of the engine's block solver.

usage: python tests/golden/make_window_delta_golden.py [K ... | cfg4 | cfg5]      (default: 16 64; seeds 0-7)
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


CFG5_LOOPS = [(0, 511), (2, 509), (1, 510), (0, 256), (100, 130)]     # tests/test_gpu_configs.py::test_config5_loop_closure_k512


def sparse_damped_delta(packed, K, links, CS, w):
    """(H + damp diag H) d = g of a packed block system with the engine's default priors, by sparse LU in double"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    B = 7 + CS
    BB = B * B
    diag = packed[:K * BB].reshape(K, B, B)
    lnk = packed[K * BB:(K + len(links)) * BB].reshape(len(links), B, B)
    g = packed[(K + len(links)) * BB:(K + len(links)) * BB + K * B].astype(np.float64).copy()
    rows, cols, blocks = [], [], []
    acc = {}
    for k in range(K):
        acc[(k, k)] = 0.5 * (diag[k] + diag[k].T)
    for l, (a, b) in enumerate(links):
        acc[(a, b)] = acc.get((a, b), 0) + lnk[l]
        acc[(b, a)] = acc.get((b, a), 0) + lnk[l].T
    # priors (add_priors above) on the diagonal blocks
    for k, kf in enumerate(w.keyframes):
        idx = np.arange(6, 6 + CS)
        acc[(k, k)][idx, idx] += 1e-3
        g[k * B + idx] += 1e-3 * (0 - kf.code.astype(np.float64))
    s0 = float(w.keyframes[0].scale)
    acc[(0, 0)][6 + CS, 6 + CS] += 1e4 / (s0 * s0)
    acc[(0, 0)][np.arange(6), np.arange(6)] += 1e4
    for k in range(K):
        d = np.diag(acc[(k, k)]).copy()
        acc[(k, k)][np.arange(B), np.arange(B)] += DAMP * d
    keys = sorted(acc)
    indptr = [0]
    indices = []
    data = []
    r = -1
    for (i, j) in keys:
        while r < i:
            r += 1
            if r > 0:
                indptr.append(len(indices))
        indices.append(j)
        data.append(acc[(i, j)])
    indptr.append(len(indices))
    M = sp.bsr_matrix((np.array(data), np.array(indices), np.array(indptr)), shape=(K * B, K * B)).tocsc()
    return spl.splu(M).solve(g)


def run_edges(w, jobs, precs=("f32", "f64")):
    """like oracle_all_edges for an explicit job list [(t, l, d, k0, k1)]"""
    cores = os.cpu_count() or 1
    workers = max(1, min(16, cores // 2))
    omp = max(1, cores // workers)

    def run(job):
        t, l, d, k0, k1 = job
        orc.set_threads(omp)
        fn = oracle_photo if t == 0 else oracle_geo
        return job, {p: fn(orc, w, k0, k1, prec=p) for p in precs}

    out = {p: {} for p in precs}
    with ThreadPoolExecutor(workers) as ex:
        for (t, l, d, _, _), r in ex.map(run, jobs):
            for p in precs:
                out[p][(t, l, d)] = r[p]
    return out


def make_cfg4():
    path = os.path.join(HERE, "window_delta_cfg4_k16_seed41.npz")
    if os.path.exists(path):
        print("exists:", path)
        return
    t0 = time.time()
    K, CS = 16, 32
    w = synth.make_window(K=K, H=256, W=320, FS=32, CS=CS, L=4, seed=41)
    res = oracle_all_edges(w)
    d = {}
    for p in ("f32", "f64"):
        packed = capi.assemble_packed(K, w.links, CS, res[p])
        H, g = add_priors(*capi.unpack_dense(packed, K, w.links, CS)[:2], w, CS)
        d[p] = damped_delta(H, g, DAMP)
        if p == "f64":
            cond = np.linalg.cond(H + DAMP * np.diag(np.diag(H)))
    np.savez_compressed(path, d32=d["f32"], d64=d["f64"], damp=np.float64(DAMP), n_links=np.int64(len(w.links)),
                        N=np.int64(w.keyframes[0].homo.shape[0]), cond=np.float64(cond))
    print(f"cfg4: {4 * len(w.links)} edges x 2 precisions in {time.time() - t0:.0f} s; fp32oracle-exact "
          f"{rel(d['f32'], d['f64']):.2e}  cond {cond:.1e} -> {os.path.basename(path)}", flush=True)


def make_cfg5():
    paths = {name: os.path.join(HERE, f"window_delta_cfg5_k512_seed7_{name}.npz") for name in ("noloops", "loops")}
    if all(os.path.exists(p) for p in paths.values()):
        print("exist:", list(paths.values()))
        return
    t0 = time.time()
    K, CS = 512, 32
    w = synth.make_window(K=K, H=64, W=80, FS=16, CS=CS, L=4, n_samples=3072, seed=7, loop_radius=0.12)
    n_temporal = len(w.links)
    links_all = list(w.links) + CFG5_LOOPS
    jobs = [(t, l, d, k0, k1) for l, (a, b) in enumerate(links_all) for d, (k0, k1) in enumerate(((a, b), (b, a)))
            for t in (0, 1)]
    res = run_edges(w, jobs)
    print(f"cfg5: {len(jobs)} edges x 2 precisions in {time.time() - t0:.0f} s", flush=True)
    for name, links in (("noloops", links_all[:n_temporal]), ("loops", links_all)):
        d = {}
        for p in ("f32", "f64"):
            sub = {k: v for k, v in res[p].items() if k[1] < len(links)}
            packed = capi.assemble_packed(K, links, CS, sub)
            d[p] = sparse_damped_delta(packed, K, links, CS, w)
        np.savez_compressed(paths[name], d32=d["f32"], d64=d["f64"], damp=np.float64(DAMP), n_links=np.int64(len(links)),
                            N=np.int64(w.keyframes[0].homo.shape[0]), links=np.array(links, np.int32))
        print(f"cfg5/{name}: {len(links)} links; fp32oracle-exact {rel(d['f32'], d['f64']):.2e} -> "
              f"{os.path.basename(paths[name])} ({time.time() - t0:.0f} s)", flush=True)


def main():
    orc.build()
    if "cfg4" in sys.argv[1:]:
        make_cfg4()
    if "cfg5" in sys.argv[1:]:
        make_cfg5()
    Ks = [int(a) for a in sys.argv[1:] if a.isdigit()] or ([] if any(a.startswith("cfg") for a in sys.argv[1:]) else [16, 64])
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
