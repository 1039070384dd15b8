#!/usr/bin/env python3
"""bench.py -- LM-iteration throughput of the MI355X dense-BA engine on the headline workload.

Workload (BASELINE.json metric: "LM iters/sec + M residuals/sec, 64-keyframe feature-metric BA @128x160"):
  K = 64 keyframes, 128x160x16 feature pyramids (L = 4), 32-dim depth code, dense sampling of the eroded mask,
  every keyframe linked to its 3 predecessors -> 186 links = 372 photometric + 372 geometric directed edges.
  Synthetic, geometrically consistent scene (sage_slam_amd/synth.py).

A "step" is one LM iteration = one pass of the hot path over the whole window:
  linearize every edge (photometric + geometric) -> assemble block normal equations -> [all-reduce] ->
  damped solve on the host -> retract -> total error at the candidate [all-reduce] -> accept / reject.
Inputs are resident in HBM before the timed region.  Residuals per step = E_photo*L*N*FS + E_geo*N
(linearisation residuals only; the error-evaluation pass is not double counted).

Multi-GPU: factor-graph links are sharded over ranks (rank r owns the contiguous range [r*2n/world, (r+1)*2n/world) of the directed edges),
keyframes replicated; a reduced window runs the one-collective LM sequence (linearize at the candidate: ONE all-reduce of
the packed normal equations per step, the error totals in its tail; strong scaling).  Single-GPU runs also measure one rank
of an 8-rank job on this device (`shard_emulation`: one-rank RCCL communicator, the peers' share from a table).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable

# Counter constants of the K = 64 headline window (rocprofv3 --pmc, separate passes, scripts/pmc_run.sh ->
# profiles/r06_v5_pmc_summary.txt; bench.py cannot collect PMC counters itself, null for any other workload).  They are
# per-launch INSTRUCTION / BYTE counts of one build on one workload -- fixed by the code, not by the box -- and are combined
# below with the launch durations measured live in this run.  They belong to ONE build of the kernel sources: the summary
# records sage_slam_amd.build.kernel_source_sha16() of the tree it was taken from, and the constants are quoted only while
# the tree still hashes to it (a stale constant used to go out silently).
#   kernels: the LM iteration's merged pair -- photo_kernel<32,16,true,2> and geo_kernel<32,true,true>
#   traffic: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane reads on gfx950, + WRITE_SIZE
PMC_K64 = {
    "source": "profiles/r06_v5_pmc_summary.txt",
    "kernel_source_sha16": "3f7e88daa2f13aac",
    # grbm_gui_active / pmc_pass_avg_us: the engine clock of the counter pass itself (r06: issue fractions at the measured
    # clock -- 1.04469e7 / 8 / 577.46 us = 2.26 GHz -- not at the 2.4 GHz spec clock)
    "photo": {"fetch_size_kb": 863517.0, "write_size_kb": 44167.9, "insts_vmem_rd": 7.41147e6, "insts_valu": 1.50417e8,
              "insts_mfma": 8.96938e6, "mfma_busy_cycles": 2.8702e8, "lds_idx_active": 1.36861e8, "lds_bank_conflict": 3.54242e7,
              "grbm_gui_active": 1.04469e7, "pmc_pass_avg_us": 577.459},
    "geo": {"insts_vmem_rd": 9.07898e6, "insts_valu": 6.70424e7, "insts_mfma": 1.4949e7, "mfma_busy_cycles": 4.78367e8,
            "lds_idx_active": 3.31144e7, "grbm_gui_active": 6.95112e6, "pmc_pass_avg_us": 373.603},
}
PMC_TRAFFIC_BYTES_K64 = {"hbm_bytes_per_launch": (2 * PMC_K64["photo"]["fetch_size_kb"] + PMC_K64["photo"]["write_size_kb"]) * 1024.0}


def pmc_constants_current():
    """True while the kernel sources still are the ones the committed counters were taken from."""
    try:
        from sage_slam_amd.build import kernel_source_sha16
        return kernel_source_sha16() == PMC_K64["kernel_source_sha16"]
    except Exception:
        return False


N_CU, N_SIMD, CLK_HZ = 256, 1024, 2.4e9          # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock
L1_PEAK_GBS = N_CU * 64 * CLK_HZ / 1e9         # CU texture path: 64 B / clk / CU  (39.3 TB/s)
MFMA_F32_PEAK_TFLOPS = 157.3                    # dense f32 matrix peak (= the f32 vector peak on this part)


def measured_clock_hz(pmc):
    """engine clock of the counter pass itself: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the kernel's average duration
    in the same rocprofv3 pass (MI355X_MICROARCH.md "DVFS give-back")"""
    if pmc.get("grbm_gui_active") and pmc.get("pmc_pass_avg_us"):
        return pmc["grbm_gui_active"] / 8.0 / (pmc["pmc_pass_avg_us"] * 1e-6)
    return CLK_HZ


def roofs(pmc, ms, ach_gbs):
    clk = measured_clock_hz(pmc)
    t = ms * 1e-3
    valu_s = pmc["insts_valu"] * 4.0 / N_SIMD / clk          # one quad-cycle issue slot per wave instruction (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU)
    mfma_s = pmc["mfma_busy_cycles"] / N_SIMD / clk if pmc.get("mfma_busy_cycles") else pmc["insts_mfma"] * 32.0 / N_SIMD / clk
    lds_s = pmc["lds_idx_active"] / N_CU / clk
    return {"hbm_algorithmic": {"achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS},
            "issue_serialised": {"valu_ms": 1e3 * valu_s, "mfma_ms": 1e3 * mfma_s, "lds_ms": 1e3 * lds_s, "launch_ms": ms,
                                 "frac": (valu_s + mfma_s + lds_s) / t, "clock_ghz": clk / 1e9,
                                 "note": "busy-sum / wall of the VALU issue port, the MFMA pipe (per SIMD) and the LDS array (per CU)"}}


def issue_roofline(pmc, ms):
    """Per-pipe utilisation of one launch from its instruction counts (committed PMC constants) and its duration measured
    in this run.  l1: wave-loads x 1 KiB (16 B / lane, the dwordx4 taps / staging loads) against the CUs' texture-path
    rate; lds: LDS-array busy cycles per CU; valu: one issue slot (4 cycles per SIMD) per wave instruction; mfma: FLOP of
    the 16x16x4 f32 instructions against the dense f32 matrix peak.  r06: peaks at the clock the counter pass measured
    (GRBM_GUI_ACTIVE / 8 / kernel duration), not at the 2.4 GHz spec clock."""
    if not pmc or ms <= 0:
        return None
    t = ms * 1e-3
    clk = measured_clock_hz(pmc)   # r06: the clock of the counter pass (GRBM_GUI_ACTIVE / 8 / duration), not the 2.4 GHz peak
    out = {"l1": {"achieved": pmc["insts_vmem_rd"] * 1024.0 / t / 1e9, "peak": N_CU * 64 * clk / 1e9, "unit": "GB/s"},
           "valu": {"achieved": pmc["insts_valu"] * 4.0 / N_SIMD / t / 1e9, "peak": clk / 1e9, "unit": "Gcycles/s per SIMD"},
           "mfma": {"achieved": pmc["insts_mfma"] * 2048.0 / t / 1e12, "peak": MFMA_F32_PEAK_TFLOPS * clk / CLK_HZ, "unit": "TFLOP/s"},
           "lds": {"achieved": pmc["lds_idx_active"] / N_CU / t / 1e9, "peak": clk / 1e9, "unit": "Gcycles/s per CU"}}
    for v in out.values():
        v["frac"] = v["achieved"] / v["peak"]
    return out


def cpu_baseline(win, budget_s: float = 20.0):
    """Reference-style CPU path (oracle/, kind = "port": materialise J per residual, then reduce) timed on a bounded
    sample of the same workload: both directed edges of a few links, linearize + error evaluation.  The OpenMP thread
    count is the best of a short sweep (the port's reduction stage does not scale past a few dozen threads: 256 threads
    are 8x SLOWER than 32 on the 2x EPYC host); the single-thread rate is reported next to it."""
    from oracle import oracle as orc
    from sage_slam_amd import synth
    orc.build()
    cores = os.cpu_count() or 1
    w = win

    def edge_pair(k0, k1):
        A, Bk = w.keyframes[k0], w.keyframes[k1]
        R10, t10 = synth.relative_pose(A.R, A.t, Bk.R, Bk.t)
        D1, g1 = synth.depth_and_grad(Bk, w.H, w.W)           # producer output, not timed (resident inputs)
        t0 = time.perf_counter()
        orc.photo_jac_error(R10, t10, A.R, A.t, Bk.R, Bk.t, A.bias, A.basis, A.code, w.mask, A.loc1d, A.homo,
                            A.feat_pyr, Bk.feat_pyr, Bk.grad_pyr, w.level_offsets, A.scale, w.cams, w.eps,
                            w.photo_weights)
        orc.geo_jac_error(R10, t10, A.R, A.t, Bk.R, Bk.t, A.bias, A.basis, A.code, D1, g1,
                          Bk.basis.reshape(w.H, w.W, w.CS), w.mask, A.loc1d, A.homo, A.scale, Bk.scale,
                          w.cams[0], w.eps, w.geo_loss_param, w.geo_weight)
        orc.photo_error(R10, t10, A.bias, A.basis, A.code, w.mask, A.loc1d, A.homo, A.feat_pyr, Bk.feat_pyr,
                        w.level_offsets, A.scale, w.cams, w.eps, w.photo_weights)
        orc.geo_error(R10, t10, A.bias, A.basis, A.code, D1, w.mask, A.loc1d, A.homo, A.scale, w.cams[0],
                      w.eps, w.geo_loss_param, w.geo_weight)
        return time.perf_counter() - t0, w.L * A.homo.shape[0] * w.FS + A.homo.shape[0]

    a0, b0 = w.links[0]
    sweep = {}
    for th in sorted({1, min(16, cores), min(32, cores), min(64, cores), cores}):
        orc.set_threads(th)
        if th > 1:
            edge_pair(a0, b0)                                  # thread pool warm-up
        sweep[th] = edge_pair(a0, b0)[0]
    best = min(sweep, key=sweep.get)
    orc.set_threads(best)
    t_used, n_edges, residuals = 0.0, 0, 0.0
    for l, (a, b) in enumerate(w.links):
        for k0, k1 in ((a, b), (b, a)):
            t, r = edge_pair(k0, k1)
            t_used += t
            n_edges += 1
            residuals += r
        if t_used > budget_s or n_edges >= 16:
            break
    r1 = w.L * w.keyframes[a0].homo.shape[0] * w.FS + w.keyframes[a0].homo.shape[0]
    return dict(value=residuals / t_used / 1e6, unit="Mresiduals/s", cores=best, kind="port",
                sample=f"{n_edges} directed edge pairs (photometric+geometric linearize and error pass) of the "
                       f"same window, {t_used:.1f} s of oracle time at the best OpenMP thread count of the sweep "
                       f"{ {k: round(v * 1e3, 1) for k, v in sweep.items()} } ms/pair; host has {cores} hardware threads",
                value_1thread=r1 / sweep[1] / 1e6,
                lm_iters_per_sec=(1.0 / (t_used / n_edges * 2 * len(w.links))))


def edge_mode(capi, synth, torch):
    """Latency of the drop-in path: what ISAM2 / the tracker call one factor at a time -- reference
    defaults N = 3072 samples of a 64x80x16 keyframe, CS 16 and 32, samples in the reference's shuffled order
    (mapper.cpp:1326-1340) and raster-sorted (sage_sort_locations) -- and one full BASELINE config-1 tracker frame
    (<= 40 LM iterations, TrackNewFrame = photometric + reprojection, TrackFrame = photometric + match geometry with
    scale).  Every call ends with the host reading `error` (stream synchronise), like the reference's .item<float>().
    The CPU port (oracle) runs the same calls on the host cores beside it."""
    import ctypes as C
    from oracle import oracle as orc
    from tests.helpers import presample_source
    orc.build()
    cpu_threads = min(16, os.cpu_count() or 1)   # the port's reduction stage stops scaling there (see cpu_baseline)
    orc.set_threads(cpu_threads)
    L = capi.lib()
    H, W, FS, NS, REP = 64, 80, 16, 3072, 200
    ws = capi.Workspace()
    res = {"cpu_port_threads": cpu_threads}
    f = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda()

    def timeit(fn, rep=REP):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(rep):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / rep

    # ctypes overhead of a call of this arity (NULL workspace -> SAGE_E_INVALID before anything runs)
    noop = lambda: L.sage_photometric_jac_error_calculate(None, *([None] * 19), C.c_float(1), None, C.c_float(0), None, 0, 16, 32)
    res["ctypes_noop_us"] = timeit(noop, 2000)
    for CS in (16, 32):
        w = synth.make_window(K=2, H=H, W=W, FS=FS, CS=CS, L=4, n_samples=NS, seed=0)
        a, b = w.keyframes
        pyr = capi.make_pyramid(w.cams[0], w.L)
        mask = f(w.mask)
        R10, t10 = synth.relative_pose(a.R, a.t, b.R, b.t)
        small = f(np.concatenate([np.asarray(x, np.float32).reshape(-1) for x in (R10, t10, a.R, a.t, b.R, b.t, a.code)]))
        base = small.data_ptr()
        sp = lambda off: C.c_void_p(base + 4 * off)
        kb = capi.DeviceKeyframe(b, H, W)
        dpt1, dgrad1 = capi.depth_and_grad(ws, kb.bias, kb.basis, b.code, b.scale, H, W, CS)
        wts = np.ascontiguousarray(w.photo_weights, np.float32)
        wp = wts.ctypes.data_as(C.POINTER(C.c_float))
        for order in ("shuffled", "sorted"):
            ka = capi.DeviceKeyframe(a, H, W)
            if order == "sorted":
                lo, ho, ok = capi.sort_locations(ws, ka.loc1d, ka.homo, H, W)
                ka.loc1d, ka.homo = lo, ho
                ka.loc1d_i32 = lo.to(torch.int32)
            Dp, Dg = 13 + CS, 14 + 2 * CS
            AtA = torch.empty(Dg * Dg, device="cuda"); Atb = torch.empty(Dg, device="cuda")
            err = C.c_float(); nin = C.c_float()
            d = capi.dptr
            calls = {
                "photometric_jac_error": lambda: L.sage_photometric_jac_error_calculate(
                    ws.h, d(AtA), d(Atb), C.byref(err), C.byref(nin), sp(0), sp(9), sp(12), sp(21), sp(24), sp(33),
                    d(ka.bias), d(ka.basis), sp(36), d(mask), d(ka.loc1d), d(ka.homo), d(ka.feat_pyr), d(kb.feat_pyr),
                    d(kb.grad_pyr), C.c_float(a.scale), C.byref(pyr), C.c_float(w.eps), wp, ka.N, FS, CS),
                "photometric_error": lambda: L.sage_photometric_error_calculate(
                    ws.h, C.byref(err), C.byref(nin), sp(0), sp(9), d(ka.bias), d(ka.basis), sp(36), d(mask),
                    d(ka.loc1d), d(ka.homo), d(ka.feat_pyr), d(kb.feat_pyr), C.c_float(a.scale), C.byref(pyr),
                    C.c_float(w.eps), wp, ka.N, FS, CS),
                "geometric_jac_error": lambda: L.sage_geometric_jac_error_calculate(
                    ws.h, d(AtA), d(Atb), C.byref(err), C.byref(nin), sp(0), sp(9), sp(12), sp(21), sp(24), sp(33),
                    d(ka.bias), d(ka.basis), sp(36), d(dpt1), d(dgrad1), d(kb.basis), d(mask), d(ka.loc1d_i32),
                    d(ka.homo), C.c_float(a.scale), C.c_float(b.scale), C.byref(pyr.cam[0]), C.c_float(w.eps),
                    C.c_float(w.geo_loss_param), C.c_float(w.geo_weight), ka.N, CS),
                "geometric_error": lambda: L.sage_geometric_error_calculate(
                    ws.h, C.byref(err), C.byref(nin), sp(0), sp(9), d(ka.bias), d(ka.basis), sp(36), d(dpt1), d(mask),
                    d(ka.loc1d_i32), d(ka.homo), C.c_float(a.scale), C.byref(pyr.cam[0]), C.c_float(w.eps),
                    C.c_float(w.geo_loss_param), C.c_float(w.geo_weight), ka.N, CS),
            }
            for name, fn in calls.items():
                assert fn() == 0, name
                res[f"{name}_CS{CS}_{order}_us"] = timeit(fn)
        # the same four calls through the CPU port (default OpenMP threads), CS as above, once per CS
        t = {}
        for name, fn in {
            "photometric_jac_error": lambda: orc.photo_jac_error(R10, t10, a.R, a.t, b.R, b.t, a.bias, a.basis, a.code, w.mask, a.loc1d, a.homo, a.feat_pyr, b.feat_pyr, b.grad_pyr, w.level_offsets, a.scale, w.cams, w.eps, w.photo_weights),
            "geometric_jac_error": lambda: orc.geo_jac_error(R10, t10, a.R, a.t, b.R, b.t, a.bias, a.basis, a.code, *synth.depth_and_grad(b, H, W), b.basis.reshape(H, W, CS), w.mask, a.loc1d, a.homo, a.scale, b.scale, w.cams[0], w.eps, w.geo_loss_param, w.geo_weight),
        }.items():
            fn()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            res[f"cpu_port_{name}_CS{CS}_us"] = 1e6 * (time.perf_counter() - t0) / 5
    # ---- one tracker frame (config 1: CS 32) ----
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests.test_gpu_tracker import Scene
    sc = Scene(capi, orc)
    cfg = capi.lm_config_default()
    for name, dof, up, uk in (("TrackNewFrame_photo+reproj", 6, True, True), ("TrackNewFrame_photo", 6, True, False),
                              ("TrackFrame_photo+matchgeom", 7, True, True)):
        prob = sc.problem(dof, up, uk)
        s0 = float(sc.s_true) * (0.96 if dof == 7 else 1.0)
        rc, _, _, _, iters, tr = capi.track_frame(cfg, dof, prob, sc.start_pose(), s0)
        assert rc == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            capi.track_frame(cfg, dof, prob, sc.start_pose(), s0)
        ms = 1e3 * (time.perf_counter() - t0) / 20
        lin, errf = sc.oracle_callbacks(dof, up, uk)
        t0 = time.perf_counter()
        capi.track_lm(cfg, dof, lin, errf, sc.start_pose(), s0)
        cpu_ms = 1e3 * (time.perf_counter() - t0)
        res[name] = dict(ms_per_frame=ms, lm_iterations=iters, candidate_evaluations=len(tr), cpu_port_ms_per_frame=cpu_ms)
    sc.close()
    ws.close()
    key = "photometric_jac_error_CS32_sorted_us"
    return {"metric": "us per drop-in operator call (photometric linearize, N=3072, 64x80x16, CS 32, raster-sorted samples)",
            "value": res[key], "unit": "us/call", "n_gpus": 1, "steps": REP, "warmup": 10, "ms_per_step": res[key] / 1e3,
            "higher_is_better": False, "scaling": "n/a", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 1: 2-keyframe tracker BA / per-edge operator API at the reference defaults "
                                   "(64x80x16 feature maps, N = 3072 samples, CS 16 and 32)"},
            "edge": res}


def shard_emulation(capi, torch, win, win_h, rank, world, steps, restart, ms_one_gpu_classic, compare_one_gpu=True):
    """One rank of an N-rank job measured on a one-GPU box: a window holding rank `rank`'s links of
    `world` (sage_window_set_shard), a REAL one-rank RCCL communicator on the window's stream (the launch cost of
    ncclAllReduce, not its xGMI transfer), and the other ranks' share of every reduced system added from a table
    (sage_window_emulate_peers) that is computed beforehand with the full window at the iterates of the job's own
    trajectory -- so the rank solves the job's real reduced systems and takes its real accept / reject decisions.  What is
    measured is this rank's critical path per LM iteration: its shard's kernels, the collective's launch, the replicated
    host solve (uncontended: the other ranks' host threads are not there), the decision round trip.  Not in it: the
    collective's transfer time over xGMI."""
    K = len(win_h.keyframes)
    classic = capi.lm_config_default(); classic.max_inner_evals = 1; classic.linearize_at_candidate = -1
    st = capi.SageLmState()
    win.set_profiling(0)
    win.reset()
    full, xs = [], []
    for i in range(restart + 1):
        win.linearize()
        torch.cuda.synchronize()
        full.append(win.packed_tensor().clone())
        xs.append([win.get_keyframe(k) for k in range(K)])
        if i < restart:
            if i == 0:
                st.iters = 0; st.damp = float(classic.init_damp)
            win.lm_step(st, classic)
            if not st.accepted:
                return {"error": f"iterate {i} of the reference trajectory was rejected: no table to emulate with"}
    comm = capi.rccl_comm_create(capi.rccl_unique_id(), 0, 1)
    # (the replicated solve: the domain-decomposed one -- default from K >= 256 -- all-reduces a separator system whose peer
    #  contributions cannot be tabulated beforehand)
    os.environ["SAGE_SHARD_SCHUR"] = "0"
    sw = capi.Window(win_h, rank=rank, world=world)
    rest = torch.empty(restart + 1, sw.packed_count, dtype=torch.float64, device="cuda")
    for i in range(restart + 1):
        for k in range(K):
            sw.set_keyframe(k, *xs[i][k])
        sw.linearize()
        torch.cuda.synchronize()
        rest[i] = full[i] - sw.packed_tensor()
    del full
    sw.reset()
    sw.use_rccl(comm)
    sw.emulate_peers(rest)
    cfg = capi.lm_config_default(); cfg.max_inner_evals = 1      # linearize_at_candidate 0 = automatic: reduced windows
    state = capi.SageLmState()                                   # evaluate the candidate with the linearize kernels

    def cycles(w, c, n_cycles, per_step):
        acc = 0
        for _ in range(n_cycles):
            w.reset()
            state.iters = 0; state.damp = float(c.init_damp)
            tr, sec = w.lm_run_timed(state, c, restart)     # (C++ loop: no Python between the iterations)
            for j in range(len(sec)):
                per_step[j].append(float(sec[j]))
                acc += int(tr[j][2])
        return acc

    n_cyc = max(2, (steps + restart - 1) // restart)
    sw.set_profiling(1)
    cycles(sw, cfg, 2, [[] for _ in range(restart)])
    for which in range(4):
        sw.kernel_time(which)
    sw.phase_time()
    cycles(sw, cfg, n_cyc, [[] for _ in range(restart)])
    kt = [sw.kernel_time(which) for which in range(4)]
    phase, phase_n = sw.phase_time()
    sw.set_profiling(0)
    cycles(sw, cfg, 3, [[] for _ in range(restart)])                       # warm-up without the event records
    torch.cuda.synchronize()
    ps = [[] for _ in range(restart)]
    t0 = time.perf_counter()
    acc = cycles(sw, cfg, n_cyc, ps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    steady = [x for j in range(1, restart) for x in ps[j]] or ps[0]
    # median of the steady iterations (a host thread that is descheduled for a few hundred microseconds once in a run would
    # otherwise be the result); the mean and the spread are reported next to it
    ms_steady = 1e3 * float(np.median(steady))
    # the same sequence (linearize-at-candidate) on the whole window: the one-GPU step the shard's is compared with
    ms_one_same = float("nan")
    if compare_one_gpu:
        one = capi.lm_config_default(); one.max_inner_evals = 1; one.linearize_at_candidate = 1
        cycles(win, one, 2, [[] for _ in range(restart)])
        torch.cuda.synchronize()
        p1 = [[] for _ in range(restart)]
        cycles(win, one, n_cyc, p1)
        torch.cuda.synchronize()
        ms_one_same = 1e3 * float(np.median([x for j in range(1, restart) for x in p1[j]] or p1[0]))
    local_edges = len(capi.shard_edges(len(win_h.links), rank, world))
    out = {"world": world, "rank": rank, "local_directed_edges": local_edges, "directed_edges": 2 * len(win_h.links),
           "ms_per_step": ms_steady,
           "ms_per_step_mean": 1e3 * float(np.mean(steady)),
           "ms_per_step_p10_p90": [1e3 * float(np.percentile(steady, 10)), 1e3 * float(np.percentile(steady, 90))],
           "ms_first_step_after_restart": 1e3 * float(np.median(ps[0])),
           "ms_per_step_incl_restarts": 1e3 * wall / (n_cyc * restart),
           "steps_timed": len(steady), "accepted_steps": acc, "steps": n_cyc * restart,
           "lm": "linearize-at-candidate (automatic for reduced windows): per iteration 1 solve + 1 linearize of the shard + ONE "
                 "all-reduce (packed system, error totals in its tail); `ms_per_step` = iterations 2.." + str(restart) +
                 " after a restart (the first one also linearizes the initial estimate)",
           "collective": "one-rank ncclAllReduce (RCCL) on the window's stream + the peers' share added from a precomputed "
                         "table (sage_window_emulate_peers): launch cost measured, xGMI transfer time NOT included",
           "host": "replicated host solve, uncontended (the other ranks' processes are not running)",
           "one_gpu_ms_per_step_classic": ms_one_gpu_classic, "one_gpu_ms_per_step_same_sequence": ms_one_same,
           "speedup_vs_one_gpu_classic": ms_one_gpu_classic / ms_steady,
           "speedup_vs_one_gpu_same_sequence": ms_one_same / ms_steady,
           "kernel_ms": {"photo_linearize": kt[0][0] / max(1, kt[0][1]), "geo_linearize": kt[1][0] / max(1, kt[1][1])}}
    if phase_n > 0:
        out["phase_ms"] = {k: v / phase_n for k, v in phase.items()}
        out["phase_ms"]["iterations_sampled"] = phase_n
    sw.close()
    capi.rccl_comm_destroy(comm)
    win.reset()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--keyframes", type=int, default=64)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--fs", type=int, default=16)
    ap.add_argument("--cs", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tune", action="store_true", help="keep the static rule's photometric run length (no sage_window_tune_runs)")
    ap.add_argument("--config", type=int, default=0,
                    help="BASELINE.json configuration 1..5 (1 = tracker frame -> --mode edge; 3 = the headline window = default)")
    ap.add_argument("--lm-variant", choices=["auto", "classic", "candidate"], default="auto",
                    help="classic: linearize, solve, error pass at the candidate (SURVEY s8d's definition of an LM iteration; "
                         "what `value` is quoted on at --gpus 1); candidate: SageLmConfig.linearize_at_candidate -- the "
                         "candidate is evaluated by the linearize kernels, an accepted iteration has no separate error pass; "
                         "auto (default) = the engine's own choice: classic on one rank, candidate for windows reduced over "
                         "ranks (one collective per iteration instead of two)")
    ap.add_argument("--emulate-shard", default="0/8", metavar="R/N",
                    help="single-GPU runs also measure rank R of an N-rank job on this one device (shard_emulation key of the "
                         "line: one-rank RCCL communicator, the peers' share from a table); 'off' skips it")
    ap.add_argument("--mode", choices=["window", "edge"], default="window",
                    help="edge: latency of the drop-in per-edge operator API and of a full tracker frame (config 1)")
    args = ap.parse_args()
    loops = []
    synth_kw = {}
    if args.config == 1:
        args.mode = "edge"
    elif args.config == 2:
        args.keyframes, args.height, args.width, args.fs, args.cs = 16, 128, 160, 16, 32
    elif args.config == 4:
        args.keyframes, args.height, args.width, args.fs, args.cs = 16, 256, 320, 32, 32
    elif args.config == 5:
        args.keyframes, args.height, args.width, args.fs, args.cs = 512, 64, 80, 16, 32
        synth_kw = dict(n_samples=3072, loop_radius=0.12)
        loops = [] if os.environ.get("SAGE_BENCH_NO_LOOPS") == "1" else [(0, 511), (2, 509), (1, 510), (0, 256), (100, 130)]

    # ---- multi-GPU entry point.  Two ways in: (a) the driver's `python -m torch.distributed.run --nproc-per-node N
    # bench.py --gpus N` (WORLD_SIZE set: this process is one rank); (b) plain `python bench.py --gpus N`: this process
    # becomes the launcher and re-executes itself under torch.distributed.run with N ranks, one GPU per rank.  A box
    # with fewer than N GPUs is refused (rc 2) instead of silently measuring one rank.
    one_dev = os.environ.get("SAGE_BENCH_ONE_DEVICE") == "1"       # dev knob: all ranks on cuda:0, gloo collective
    dry_run = os.environ.get("SAGE_BENCH_DRY_RUN") == "1"          # dev knob: launcher / rendezvous check without a GPU
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']} in the environment\n")
        raise SystemExit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not (one_dev or dry_run):
            import torch
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if have < args.gpus:
                sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but this box has {have} visible HIP device(s); "
                                 "refusing to run fewer ranks than asked for (SAGE_BENCH_ONE_DEVICE=1 runs all ranks "
                                 "on cuda:0 over gloo as a functional check)\n")
                raise SystemExit(2)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        raise SystemExit(subprocess.call(cmd, env=env))

    # the driver reads ONE JSON line from stdout: libraries that print banners to fd 1 (RCCL prints its version block at
    # communicator teardown) are sent to stderr; the JSON line goes out through the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if dry_run:
        # launcher check (CPU test): N ranks rendezvous over gloo, agree on the world size, rank 0 prints the line
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pids = [None] * world
        dist.all_gather_object(pids, (rank, os.getpid()))
        if rank == 0:
            os.write(json_fd, (json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": len(set(pids)),
                                           "pids": [p for _, p in sorted(pids)]}) + "\n").encode())
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    # SAGE_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 with the gloo backend -- a functional check of the sharded path
    # on a 1-GPU box (the driver's multi-GPU runs use one GPU per rank and RCCL)
    if not one_dev and torch.cuda.device_count() < world:
        sys.stderr.write(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible HIP device(s)\n")
        raise SystemExit(2)
    dev_index = 0 if one_dev else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    # SAGE_BENCH_FORCE_DIST=1 (dev knob): take the sharded code path (process group, all-reduces) with one rank too
    if world > 1 or os.environ.get("SAGE_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))

    from sage_slam_amd import capi, synth
    capi.lib()
    # one process per GPU: stay on the NUMA node the GPU hangs off (the window solve reads freshly DMA'd pinned memory)
    full_affinity = os.sched_getaffinity(0)
    if os.environ.get("SAGE_BENCH_NO_BIND") != "1":
        capi.bind_thread_to_device(dev_index)
    if args.mode == "edge":
        os.sched_setaffinity(0, full_affinity)     # (no host solve in this mode; its CPU leg wants its 16 threads on 16 cores)
        out = edge_mode(capi, synth, torch)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        return
    win_h = synth.make_window(K=args.keyframes, H=args.height, W=args.width, FS=args.fs, CS=args.cs, L=4, seed=0,
                              **synth_kw)
    for lk in loops:
        win_h.links.append(lk)
    win = capi.Window(win_h, rank=rank, world=world)
    # r06: the run length of the photometric workgroups measured on the window itself (sage_window_tune_runs: opt-in set-up step of
    # the engine, ~20-40 ms, single-rank windows; --no-tune / SAGE_PHOTO_TPB keep the static rule's choice).  The headline window
    # keeps the rule's 8; BASELINE config 4 moves 9 -> 12 (-7 % linearize + error pass)
    photo_runs = {"tuned": False}
    if world == 1 and not args.no_tune and os.environ.get("SAGE_BENCH_FORCE_DIST") != "1":
        t_tune = time.perf_counter()
        photo_runs = dict(win.tune_runs(), tuned=True)
        photo_runs["tune_ms"] = round(1e3 * (time.perf_counter() - t_tune), 1)
        win.reset()
    packed = win.packed_tensor()
    errt = win.error_tensor()
    cfg = capi.lm_config_default()
    cfg.linearize_at_candidate = {"auto": 0, "classic": -1, "candidate": 1}[args.lm_variant]
    classic_seq = args.lm_variant == "classic" or (args.lm_variant == "auto" and world == 1 and
                                                   os.environ.get("SAGE_BENCH_FORCE_DIST") != "1")
    damp = float(cfg.init_damp)

    # totals over the whole job (all ranks): every link has 2 photometric + 2 geometric directed edges
    N = win_h.keyframes[0].homo.shape[0]
    n_dir = 2 * len(win_h.links)
    residuals_per_step = n_dir * (win_h.L * N * win_h.FS + N)
    rho = win_h.P / float(win_h.H * win_h.W)
    bytes_photo_px = 4.0 * (4.0 * win_h.FS * rho + win_h.CS + 6.0)
    bytes_geo_px = 4.0 * (2.0 * win_h.CS + 9.0)

    def clamp(d):
        return min(max(float(cfg.min_damp), d), float(cfg.max_damp))

    state = capi.SageLmState()
    cfg.max_inner_evals = 1          # one linearize + one solve + one error pass per step, as the multi-GPU path

    # SAGE_BENCH_PY_STEPS=1 (dev knob): drive the sharded window from Python call by call instead of through
    # sage_window_lm_step + the all-reduce hook
    py_steps = os.environ.get("SAGE_BENCH_PY_STEPS") == "1"
    rccl_comm = None
    collective = "none"
    if dist is not None and not py_steps:
        if one_dev or os.environ.get("SAGE_BENCH_TORCH_ALLREDUCE") == "1":
            win.set_allreduce(dist)                      # torch.distributed hook (gloo on one device / dev knob)
            collective = "torch.distributed.all_reduce hook"
        else:
            # native RCCL: rank 0 draws the unique id, torch.distributed only ferries its 128 bytes; from here on the
            # two all-reduces of an LM iteration are ncclAllReduce calls issued by the C++ host on the window's stream.
            # Every rank first probes that the engine found an RCCL to bind; the ranks agree (MIN) before committing.
            try:
                uid = capi.rccl_unique_id()
                have = 1
            except Exception:
                uid, have = bytes(128), 0
            flag = torch.tensor([have], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                t = torch.tensor(list(uid), dtype=torch.uint8, device="cuda")
                dist.broadcast(t, 0)
                rccl_comm = capi.rccl_comm_create(bytes(t.cpu().tolist()), rank, world)
                win.use_rccl(rccl_comm)
                collective = "native ncclAllReduce (RCCL) on the window's stream"
            else:
                win.set_allreduce(dist)
                collective = "torch.distributed.all_reduce hook (no librccl found by the engine)"

    def lm_step():
        nonlocal damp
        if dist is None or not py_steps:
            # the engine's own LM iteration (sage_window_lm_step): no Python between the launches; a sharded window
            # enters Python only for its two all-reduces (packed normal equations, 4-double error totals)
            state.damp = damp
            win.lm_step(state, cfg)
            damp = state.damp
            return state.error, state.candidate_error, bool(state.accepted)
        # multi-GPU: the same sequence with the two all-reduces in between; nothing is read back before the candidate's
        # error pass has been enqueued (the error at the linearisation point is the tail of the reduced packed buffer)
        win.linearize()
        dist.all_reduce(packed)
        win.solve(damp, want_norm=False)
        win.error(1)
        dist.all_reduce(errt)
        e0 = win.total_error(True)
        e1 = win.total_error(False)
        if e1 < e0:
            win.accept()
            damp = clamp(damp / float(cfg.damp_dec_factor))
            return e0, e1, True
        damp = clamp(damp * float(cfg.damp_inc_factor))
        return e0, e1, False

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    hist = []
    RESTART = 3   # the LM converges in ~3 steps on this scene: restart from the initial estimate every RESTART steps so
                  # the timed steps are live descent steps (a restart is one small H2D of the variables, inside the timing)
    def run_steps(n_steps):
        nonlocal damp
        if dist is None or not py_steps:
            # the engine's own loop: sage_window_lm_run drives the RESTART iterations between two restarts from C++ (no
            # Python between the iterations being timed; a sharded window still enters Python for a gloo hook, never for RCCL)
            i = 0
            while i < n_steps:
                win.reset()
                damp = float(cfg.init_damp)
                state.iters = 0
                state.damp = damp
                n = min(RESTART, n_steps - i)
                for e0, e1, acc, d in win.lm_run(state, cfg, n):
                    hist.append((float(e0), float(e1), bool(acc)))
                damp = state.damp
                i += n
        else:
            for i in range(n_steps):
                if i % RESTART == 0:
                    win.reset()
                    damp = float(cfg.init_damp)
                    state.iters = 0
                hist.append(lm_step())

    # instrumented pass FIRST (every hot kernel + the phase marks: eleven event records per step, 30-50 us of the
    # stream's time): the other kernels' durations and phase_ms come from it.  It also brings the GPU / host clocks up:
    # the step time of this loop settles only after ~25 steps (profiles/r04_step_series.txt: 2.2, 1.9, 1.88, 1.84 ...
    # 1.78 ms), more than --warmup 5 covers.
    win.set_profiling(1)
    n_instr = int(os.environ.get("SAGE_BENCH_INSTR_STEPS", min(max(args.steps, 1), 9 * RESTART)))   # (dev knob)
    run_steps(RESTART)
    for which in range(4):
        win.kernel_time(which)
    win.phase_time()
    barrier()
    t1 = time.perf_counter()
    run_steps(n_instr)
    barrier()
    ms_instr = 1e3 * (time.perf_counter() - t1) / max(1, n_instr)
    ktime_full = [win.kernel_time(which) for which in range(4)]
    phase, phase_n = win.phase_time()
    del hist[:]
    # W warm-up steps, then the timed region: HIP events around the dominant kernel only (the roofline's live launch
    # duration: two event records per step)
    win.set_profiling(2)
    win.reset()
    damp = float(cfg.init_damp)
    state.iters = 0
    state.damp = damp
    for i in range(args.warmup):
        hist.append(lm_step())
    for which in range(4):
        win.kernel_time(which)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ktime = [win.kernel_time(0)] + ktime_full[1:]
    win.set_profiling(False)
    per_rank_ms = None
    if dist is not None:
        # average launch duration of the three hot kernels on every rank (HIP events on the window's stream)
        mine = torch.tensor([ktime[i][0] / max(1, ktime[i][1]) for i in range(3)] + [1e3 * (time.perf_counter() - t0)],
                            dtype=torch.float64, device="cpu" if one_dev else "cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [{"rank": r, "photo_linearize": float(v[0]), "geo_linearize": float(v[1]),
                        "error_pass": float(v[2])} for r, v in enumerate(allr)]
        # r06: the first SCALE run verifies itself -- what each rank's COMMUNICATOR says about the job
        # (ncclCommCount / ncclCommUserRank) and how many directed edges the rank's window actually linearizes
        seen = (-1, -1)
        if rccl_comm is not None:
            try:
                seen = capi.rccl_comm_info(rccl_comm)
            except Exception:
                pass
        schur_r = os.environ.get("SAGE_SHARD_SCHUR", "1" if args.keyframes >= 256 else "0") not in ("0", "")
        n_loc = (2 * len(capi.shard_links(len(win_h.links), rank, world)) if schur_r else
                 len(capi.shard_edges(len(win_h.links), rank, world)))
        mine2 = torch.tensor([seen[0], seen[1], n_loc], dtype=torch.int64, device="cpu" if one_dev else "cuda")
        all2 = [torch.zeros_like(mine2) for _ in range(world)]
        dist.all_gather(all2, mine2)
        for r, v in enumerate(all2):
            per_rank_ms[r].update({"rccl_ranks_seen": int(v[0]), "rccl_rank": int(v[1]), "local_directed_edges": int(v[2])})

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        # dominant kernel: the fused photometric linearize; algorithmic bytes of ONE launch on this rank
        # (windows that take the domain-decomposed solve shard by whole links, the others by directed edge)
        schur = world > 1 and (os.environ.get("SAGE_SHARD_SCHUR", "1" if args.keyframes >= 256 else "0") not in ("0", ""))
        px_launch = (2 * len(capi.shard_links(len(win_h.links), rank, world)) if schur else
                     len(capi.shard_edges(len(win_h.links), rank, world))) * N
        ms_photo = ktime[0][0] / max(1, ktime[0][1])
        ms_geo = ktime[1][0] / max(1, ktime[1][1])
        ach = px_launch * bytes_photo_px / (ms_photo * 1e-3) / 1e9 if ms_photo > 0 else 0.0
        ach_geo = px_launch * bytes_geo_px / (ms_geo * 1e-3) / 1e9 if ms_geo > 0 else 0.0
        headline = world == 1 and args.keyframes == 64 and args.height == 128 and args.fs == 16 and args.cs == 32
        pmc_ok = headline and pmc_constants_current()
        n_acc = int(sum(1 for h in hist[args.warmup:] if h[2]))
        # phases of an iteration on rank 0's stream timeline (HIP events); "host_idle" = the rest of the step: accept /
        # reject decision, the error totals' all-reduce, launch gaps
        phase_ms = None
        if phase_n > 0:
            phase_ms = {k: v / phase_n for k, v in phase.items()}
            phase_ms["host_idle"] = max(0.0, ms_instr - sum(phase_ms.values()))
            phase_ms["iterations_sampled"] = phase_n
            phase_ms["ms_per_step_instrumented"] = ms_instr  # the pass the marks were taken in (BEFORE the warm-up and the timed region)
        out = {
            "metric": f"M residuals/sec (+ LM iters/sec), {args.keyframes}-keyframe feature-metric BA @{args.height}x{args.width}",
            "value": residuals_per_step * args.steps / elapsed / 1e6,
            "unit": "Mresiduals/s",
            "lm_iters_per_sec": args.steps / elapsed,
            "accepted_iters_per_sec": n_acc / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # untimed steps this process ran before the W warm-up steps (instrumented pass: kernel events + phase marks)
            "pre_timed_steps": RESTART + n_instr,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.keyframes}-keyframe window, {args.height}x{args.width}x{args.fs} feature "
                                   f"pyramids (L=4), {args.cs}-dim depth code, dense sampling N={N}, "
                                   f"{len(win_h.links)} links = {n_dir} photometric + {n_dir} geometric directed edges",
                       "residuals_per_step": residuals_per_step,
                       "photo_runs": photo_runs,
                       "parallelism": f"edge-shard x{world}" if world > 1 else "single GPU",
                       "collective": collective,
                       **({"per_rank_kernel_ms": per_rank_ms} if per_rank_ms else {}),
                       "lm": ("1 linearize + 1 solve (device scatter/retract, host block Cholesky) + 1 error pass per step"
                              if classic_seq else
                              "linearize-at-candidate: per evaluation 1 solve + 1 linearize at the candidate (error and "
                              "system from one pass); no separate error pass; +1 linearize after every restart"),
                       "accepted_steps": n_acc,
                       "error_first_last": [hist[0][0], hist[-1][1]]},
            "roofline": {"bound": "hbm",
                         # r06: BOTH roofs in the parsed line.  hbm_algorithmic = SURVEY s8(d)'s bytes over the
                         # launch time (the contract's `frac`); issue_serialised = (VALU + MFMA issue time per SIMD + LDS-array
                         # time per CU) / launch time at the MEASURED engine clock -- ~1 means the three pipes run one after the
                         # other, which is what binds this kernel (profiles/r06_photo_wave_timeline.txt), not HBM
                         "roofs": roofs(PMC_K64["photo"], ms_photo, ach) if pmc_ok else
                                  {"hbm_algorithmic": {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}},
                         "kernel": "photo_kernel<CS,FS,true,2> (fused photometric linearize of the LM iteration; since r05 it also "
                                   "contracts the geometric edge's code0 blocks of the same pair -- the merged linearize)",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "bound_note": "the HBM figure is SURVEY s8(d)'s algorithmic-byte roofline of the PHOTOMETRIC factor alone "
                                       "(492 B per source pixel at FS 16 / CS 32) over this kernel's launch time, although the kernel "
                                       "now does a third of the geometric factor's contractions as well; `linearize_pair` prices "
                                       "both factor types' algorithmic bytes against both kernels' time.  What binds the kernel is "
                                       "VALU + MFMA issue and the LDS pipe (see issue), not HBM",
                         # both linearize kernels of a step as one unit: (B_photo + B_geo) per source pixel over t_photo + t_geo
                         # (r04, separate kernels: 4.70 GB / (0.639 + 0.494 ms) = 4.15 TB/s = 0.519)
                         "linearize_pair": {"achieved": px_launch * (bytes_photo_px + bytes_geo_px) / ((ms_photo + ms_geo) * 1e-3) / 1e9
                                            if ms_photo + ms_geo > 0 else 0.0,
                                            "frac": px_launch * (bytes_photo_px + bytes_geo_px) / ((ms_photo + ms_geo) * 1e-3) / 1e9 / HBM_PEAK_GBS
                                            if ms_photo + ms_geo > 0 else 0.0,
                                            "bytes_per_launch_pair": px_launch * (bytes_photo_px + bytes_geo_px),
                                            "ms": ms_photo + ms_geo},
                         "traffic": PMC_TRAFFIC_BYTES_K64["hbm_bytes_per_launch"] if pmc_ok else None,
                         "traffic_source": ("rocprofv3 --pmc FETCH_SIZE(x2 gfx950 correction)+WRITE_SIZE, " + PMC_K64["source"]
                                            if pmc_ok else "none: " + ("the kernel sources no longer hash to the committed counter "
                                            "summary's build (" + PMC_K64["source"] + ")" if headline else "counters are committed "
                                            "for the headline workload only")),
                         "issue": issue_roofline(PMC_K64["photo"], ms_photo) if pmc_ok else None,
                         "bytes_per_launch": px_launch * bytes_photo_px, "avg_launch_ms": ms_photo,
                         "geo_kernel": {"achieved": ach_geo, "frac": ach_geo / HBM_PEAK_GBS, "avg_launch_ms": ms_geo,
                                        "bytes_per_launch": px_launch * bytes_geo_px,
                                        "issue": issue_roofline(PMC_K64["geo"], ms_geo) if pmc_ok else None},
                         # (the window's error pass evaluates both factor types in the photometric error kernel: no
                         #  separate geometric launch)
                         "error_pass_ms": {"photo+geo" if ktime[3][1] == 0 else "photo": ktime[2][0] / max(1, ktime[2][1]),
                                           **({"geo": ktime[3][0] / ktime[3][1]} if ktime[3][1] else {})}},
        }
        if phase_ms:
            out["phase_ms"] = phase_ms
        if world == 1 and args.emulate_shard != "off" and len(win_h.links) >= 8:
            er, ew = (int(v) for v in args.emulate_shard.split("/"))
            try:
                out["shard_emulation"] = shard_emulation(capi, torch, win, win_h, er, ew, args.steps, RESTART, ms_per_step)
            except Exception as exc:            # the headline line must not depend on the emulation
                out["shard_emulation"] = {"error": repr(exc)}
            se = out["shard_emulation"]
            if isinstance(se, dict) and "speedup_vs_one_gpu_classic" in se:
                # r06: the ceiling belongs in the line, not only in DESIGN.md
                sp = se["speedup_vs_one_gpu_classic"]
                se["ceiling_note"] = (
                    f"one rank of {ew} measured on this device runs an LM iteration {sp:.1f}x faster than the one-GPU step BEFORE any "
                    "xGMI transfer: the >= 6x strong-scaling target is out of reach at this window size by design -- the "
                    "replicated host factorisation (a chain of K/2 block rows per half, ~0.2 ms) does not shard.  The four-chain "
                    "elimination was costed again in r06 (DESIGN s7): two of its four chains sit between two separators and drag "
                    "3-row arrow chains of the same length behind them -- ~2x the multiply-adds, ~14 host threads per rank for "
                    "~30 us of critical path; not built")
            if classic_seq and se.get("one_gpu_ms_per_step_same_sequence", 0) == se.get("one_gpu_ms_per_step_same_sequence", float("nan")):
                # the same window with the engine's other LM sequence (SageLmConfig.linearize_at_candidate = 1: the candidate is
                # evaluated by the linearize kernels, an accepted iteration has no separate error pass; identical iterates and
                # decisions) -- reported next to `value`, which stays on the classic sequence of SURVEY s8d
                ms_c = se["one_gpu_ms_per_step_same_sequence"]
                out["linearize_at_candidate"] = {"ms_per_step": ms_c, "lm_iters_per_sec": 1e3 / ms_c,
                                                 "value": residuals_per_step / (ms_c * 1e-3) / 1e6, "unit": "Mresiduals/s",
                                                 "note": "median of the iterations 2..3 after each restart (every timed iteration "
                                                         "accepted); a rejected iteration costs a linearize here instead of an error pass"}
        # where the host side ran (the replicated solve is a fifth of the step and lives on host cores that a 1-GPU box
        # shares with the node's other tenants: DESIGN s7 "what the slow processes are")
        try:
            import ctypes
            out["host_placement"] = {"lm_thread_cpu": int(ctypes.CDLL(None).sched_getcpu()),
                                     "lm_thread_mask_cpus": len(os.sched_getaffinity(0)),
                                     "solver_helper_cpus": capi.solver_helper_cpus(),
                                     "placement_monitor_moves": capi.solver_placement_moves(),
                                     "placement_monitor": "opt-in (SAGE_PLACEMENT_MONITOR=1); off" if os.environ.get("SAGE_PLACEMENT_MONITOR", "0") in ("", "0") else "on",
                                     "host_threads_running": capi.host_threads_running(),
                                     "loadavg_1min": float(open("/proc/loadavg").read().split()[0])}
        except Exception as e:                                  # diagnostics only
            out["host_placement"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            # the baseline's OpenMP team inherits the caller's mask: give it back every CPU the process started with (the
            # binding above narrows the LM thread to one L3 domain)
            os.sched_setaffinity(0, full_affinity)
            out["cpu_baseline"] = cpu_baseline(win_h)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    win.close()
    if rccl_comm is not None:
        capi.rccl_comm_destroy(rccl_comm)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
