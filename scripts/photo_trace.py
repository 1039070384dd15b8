"""Wave timeline of the merged photometric linearize (dev tool, VERDICT r5 item 1: "per-slot timeline").

Runs the headline window's LM iteration on a -DSAGE_PHOTO_TRACE variant build (scripts/build_variant.sh; SAGE_BA_LIB
selects it), reads the per-wave phase stamps back and writes them to gpurun_out/trace/<name>.npz.
``python scripts/photo_trace.py analyse <npz>`` prints the per-phase wall-clock shares of a wave and how the co-resident
waves of a SIMD overlap (runs anywhere).

  usage on the GPU box:  SAGE_BA_LIB=.../_variants/libsage_trace.so python scripts/photo_trace.py run <name> [K H W FS CS]
"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MAXWG, SUBS, MARKS = 3072, 9, 8
PH = ["A warp", "B set-up", "B taps", "C rows", "D contract", "E l2/flush"]


def run(name, K=64, H=128, W=160, FS=16, CS=32):
    import torch
    from sage_slam_amd import capi, synth
    w = synth.make_window(K=K, H=H, W=W, FS=FS, CS=CS, L=4, seed=0)
    win = capi.Window(w)
    cfg = capi.lm_config_default(); cfg.max_inner_evals = 1; cfg.linearize_at_candidate = -1
    L = capi.lib()
    L.sage_debug_photo_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    for _ in range(3):
        win.reset(); win.lm_step(capi.SageLmState(), cfg)
    torch.cuda.synchronize()
    L.sage_debug_photo_trace_clear()
    win.set_profiling(True)
    win.reset(); win.lm_step(capi.SageLmState(), cfg)
    torch.cuda.synchronize()
    ms, c = win.kernel_time(0)
    buf = np.zeros(MAXWG * 4 * SUBS * MARKS, dtype=np.uint64)
    rc = L.sage_debug_photo_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
    assert rc == 0, rc
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "trace")
    os.makedirs(out, exist_ok=True)
    np.savez_compressed(os.path.join(out, name + ".npz"), trace=buf.reshape(MAXWG, 4, SUBS, MARKS), kernel_ms=ms / max(1, c))
    print(json.dumps({"name": name, "photo_kernel_ms": round(ms / max(1, c), 4)}))
    analyse(os.path.join(out, name + ".npz"))


def analyse(path, verbose=True):
    z = np.load(path)
    tr = z["trace"].astype(np.int64)
    hdr = tr[:, :, SUBS - 1, :]
    live = hdr[:, 0, 3] != 0
    nwg = int(live.sum())
    tr = tr[live]; hdr = hdr[live]
    hw = hdr[:, :, 0]
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = hdr[:, :, 1] & 15
    waveslot = hw & 15
    t0 = tr[:, :, :SUBS - 1, :7]                       # [wg][wave][sub][mark]
    res = {"workgroups": nwg, "kernel_ms": float(z["kernel_ms"])}
    # a sub-tile that ran: mark 0 and mark 6 stamped
    ran = (t0[..., 0] > 0) & (t0[..., 6] > 0)
    staged = ran & (t0[..., 2] > 0) & (t0[..., 3] > 0) & (t0[..., 4] > 0) & (t0[..., 5] > 0)
    d = np.diff(t0, axis=-1)                           # phase durations [.., 6]
    per = d[staged]
    mean = per.mean(axis=0)
    res["subtiles"] = int(ran.sum()); res["subtiles_staged"] = int(staged.sum())
    res["phase_cycles_mean"] = {PH[i]: round(float(mean[i]), 0) for i in range(6)}
    res["phase_cycles_p10_p50_p90"] = {PH[i]: [int(np.percentile(per[:, i], q)) for q in (10, 50, 90)] for i in range(6)}
    tot = (t0[..., 6] - t0[..., 0])[staged]
    res["subtile_cycles_mean"] = round(float(tot.mean()), 0)
    res["phase_share"] = {PH[i]: round(float(mean[i] / tot.mean()), 3) for i in range(6)}
    # gap between consecutive sub-tiles of a wave (flush / barriers)
    gap = (t0[:, :, 1:, 0] - t0[:, :, :-1, 6])[ran[:, :, 1:] & ran[:, :, :-1]]
    res["between_subtiles_cycles_mean"] = round(float(gap.mean()), 0)
    # workgroup lifetime and the launch's span (clock of the first stamp .. last stamp per XCD)
    wg_start = hdr[:, :, 3].min(axis=1)
    last = np.where(ran, t0[..., 6], 0).max(axis=(1, 2))
    res["wg_cycles_mean"] = round(float((last - wg_start).mean()), 0)
    # prologue (kernel entry -> first sub-tile: work item, edge table, poses, s_red barrier) per wave, and -- per CU slot -- the
    # time between a workgroup's last stamp and its successor's ENTRY (the hardware's dispatch) when the entry stamp exists
    entry = hdr[:, :, 6]
    if (entry > 0).all():
        res["prologue_cycles_p10_p50_p90"] = [int(np.percentile((hdr[:, :, 3] - entry).reshape(-1), q)) for q in (10, 50, 90)]
        cu_key = ((xcc[:, 0] * 8 + se[:, 0]) * 2 + sh[:, 0]) * 16 + cu[:, 0]
        gaps_d, gaps_t = [], []
        for k in np.unique(cu_key):
            m = np.nonzero(cu_key == k)[0]
            for slot in np.unique(waveslot[m, 0]):
                mm = m[waveslot[m, 0] == slot]
                o = mm[np.argsort(entry[mm].min(axis=1))]
                for a_, b_ in zip(o[:-1], o[1:]):
                    gaps_d.append(int(entry[b_].min() - last[a_]))
                    gaps_t.append(int(hdr[b_, :, 3].min() - last[a_]))
        res["dispatch_gap_cycles_p10_p50_p90"] = [int(np.percentile(gaps_d, q)) for q in (10, 50, 90)]
        res["end_to_first_subtile_cycles_p10_p50_p90"] = [int(np.percentile(gaps_t, q)) for q in (10, 50, 90)]
    spans = []
    for x in range(8):
        m = xcc[:, 0] == x
        if m.any():
            spans.append(int(last[m].max() - wg_start[m].min()))
    res["xcd_span_cycles"] = spans
    # co-phase statistics per SIMD: at sampled instants, how many of the SIMD's resident waves are in each phase class
    # classes: mem = A + B set-up (global-latency bound), lds = B taps, valu = C, mfma = D, other = E / between
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu          # [wg][wave] CU key
    key = key * 4 + simd
    ivs = {}                                             # simd key -> list of (start, end, class)
    cls_of = [0, 0, 1, 2, 3, 4]
    W_, V_, S_ = np.nonzero(staged)
    for a, b, c in zip(W_, V_, S_):
        k = int(key[a, b])
        lst = ivs.setdefault(k, [])
        for i in range(6):
            lst.append((int(t0[a, b, c, i]), int(t0[a, b, c, i + 1]), cls_of[i]))
    rng = np.random.default_rng(0)
    joint = np.zeros((5, 5), dtype=np.float64)          # time-weighted: P(wave x in class i AND another wave of the SIMD in class j)
    occ_hist = np.zeros(8)
    same_cls = np.zeros(5); cls_time = np.zeros(5)
    nsimd = 0
    for k, lst in ivs.items():
        arr = np.array(lst, dtype=np.int64)
        lo, hi = arr[:, 0].min(), arr[:, 1].max()
        ts = rng.integers(lo, hi, size=400)
        nsimd += 1
        for t in ts:
            m = (arr[:, 0] <= t) & (arr[:, 1] > t)
            cl = arr[m, 2]
            occ_hist[min(len(cl), 7)] += 1
            cnt = np.bincount(cl, minlength=5)
            for i in range(5):
                if cnt[i]:
                    cls_time[i] += cnt[i]
                    same_cls[i] += cnt[i] * (cnt[i] - 1)
                    for j in range(5):
                        joint[i, j] += cnt[i] * (cnt[j] - (1 if i == j else 0))
    names = ["mem(A+Bsetup)", "lds(Btaps)", "valu(C)", "mfma(D)", "E"]
    res["simds_seen"] = nsimd
    res["waves_in_a_phase_per_simd_hist"] = {str(i): round(float(occ_hist[i] / occ_hist.sum()), 3) for i in range(8)}
    res["class_time_share"] = {names[i]: round(float(cls_time[i] / cls_time.sum()), 3) for i in range(5)}
    # P(a given co-resident wave is in class j | this wave in class i), vs the unconditional share: > share = in phase
    cond = {}
    for i in range(5):
        row = joint[i] / max(1.0, joint[i].sum())
        cond[names[i]] = {names[j]: round(float(row[j]), 3) for j in range(5)}
    res["partner_class_given_mine"] = cond
    if verbose:
        print(json.dumps(res, indent=1))
    return res


if __name__ == "__main__":
    if sys.argv[1] == "run":
        a = [int(x) for x in sys.argv[3:8]]
        run(sys.argv[2], *a)
    else:
        analyse(sys.argv[2])
