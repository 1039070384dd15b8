"""Re-entrancy of the drop-in boundary under the reference's threading contract (SURVEY s8b "threading"): the reference
calls the df::*_calculate operators concurrently from up to 4 host threads (frame thread, mapping_thread_,
local_loop_detect_thread_, global_loop_detect_thread_; core/deepfactors.cpp:1497-1505), with no locks, every call
synchronous.  Here: 4 threads, each with its OWN SageWorkspace (two on the legacy default stream like the reference, two
on their own streams), hammer sage_photometric_jac_error_calculate / sage_photometric_error_calculate /
sage_geometric_jac_error_calculate / sage_track_frame on different edges while a fifth thread drives
sage_window_lm_step on a window (its helper thread of the host factorisation included, CholHelper::busy) and a sixth a
second window.  ctypes releases the GIL around every call.  Every result must be BIT-IDENTICAL to the single-threaded
one: the kernels have no atomics and no shared scratch, so any difference is a shared-state bug."""
import threading
import time

import numpy as np
import pytest

from sage_slam_amd import synth

pytestmark = pytest.mark.gpu

HAMMER_SECONDS = 2.5


def _edge_calls(capi, torch, w, dk, mask, pyr, ws, k0, k1):
    """the three operator calls of one directed edge through workspace `ws` -> dict of results"""
    a, b = w.keyframes[k0], w.keyframes[k1]
    A, B = dk[k0], dk[k1]
    R10, t10 = synth.relative_pose(a.R, a.t, b.R, b.t)
    out = {}
    out["pj"] = capi.photometric_jac_error(ws, R10, t10, a.R, a.t, b.R, b.t, A.bias, A.basis, a.code, mask, A.loc1d,
                                           A.homo, A.feat_pyr, B.feat_pyr, B.grad_pyr, a.scale, pyr, w.eps,
                                           w.photo_weights, w.FS, w.CS)
    out["pe"] = capi.photometric_error(ws, R10, t10, A.bias, A.basis, a.code, mask, A.loc1d, A.homo, A.feat_pyr,
                                       B.feat_pyr, a.scale, pyr, w.eps, w.photo_weights, w.FS, w.CS)
    d1, g1 = capi.depth_and_grad(ws, B.bias, B.basis, b.code, b.scale, w.H, w.W, w.CS)
    out["gj"] = capi.geometric_jac_error(ws, R10, t10, a.R, a.t, b.R, b.t, A.bias, A.basis, a.code, d1, g1, B.basis,
                                         mask, A.loc1d_i32, A.homo, a.scale, b.scale, pyr.cam[0], w.eps,
                                         w.geo_loss_param, w.geo_weight, w.CS)
    return out


def _same(x, y, path=""):
    """None when bit-identical, else a description of the first difference"""
    if isinstance(x, dict):
        for k in x:
            d = _same(x[k], y[k], path + "/" + str(k))
            if d:
                return d
        return None
    if isinstance(x, tuple):
        for i, (a, b) in enumerate(zip(x, y)):
            d = _same(a, b, path + "/" + str(i))
            if d:
                return d
        return None
    a, b = np.asarray(x), np.asarray(y)
    if np.array_equal(a, b):
        return None
    return f"{path}: max abs diff {np.abs(a.astype(np.float64) - b.astype(np.float64)).max():.3e} of {np.abs(b).max():.3e}, " \
           f"{int((a != b).sum())}/{a.size} entries"


def _lm_trace(capi, win, steps):
    cfg = capi.lm_config_default()
    cfg.max_inner_evals = 1
    st = capi.SageLmState()
    tr = []
    for i in range(steps):
        if i % 4 == 0:
            win.reset()
            st = capi.SageLmState()
        win.lm_step(st, cfg)
        tr.append((st.error, st.candidate_error, int(st.accepted), st.damp))
    return np.array(tr)


def test_operators_and_windows_are_reentrant_across_host_threads():
    import torch
    assert torch.cuda.is_available()
    from sage_slam_amd import capi
    from tests.test_gpu_tracker import Scene
    capi.lib()
    NT = 4
    w = synth.make_window(K=5, H=64, W=80, FS=16, CS=32, L=4, n_samples=3072, seed=12)
    dk = [capi.DeviceKeyframe(k, w.H, w.W) for k in w.keyframes]
    mask = torch.from_numpy(w.mask).cuda()
    pyr = capi.make_pyramid(w.cams[0], w.L)
    edges = [(a, b) for a, b in w.links] + [(b, a) for a, b in w.links]
    my_edges = [edges[t::NT] for t in range(NT)]
    # tracker problems: one scene per thread (own workspace inside), dof 6 photo+reproj and dof 7 photo+match geometry
    scenes = [Scene(capi, None, seed=31 + t) for t in range(NT)]
    cfg = capi.lm_config_default()

    def track(sc, dof):
        prob = sc.problem(dof, True, True)
        s0 = float(sc.s_true) * (0.97 if dof == 7 else 1.0)
        rc, pose, scale, err, iters, tr = capi.track_frame(cfg, dof, prob, sc.start_pose(), s0)
        assert rc == 0
        return (pose, np.float64(scale), np.float64(err), iters, len(tr))

    # windows for the LM threads
    w_lm = [synth.make_window(K=12, H=64, W=80, FS=16, CS=32, L=4, seed=21),
            synth.make_window(K=7, H=48, W=64, FS=16, CS=16, L=3, n_samples=1500, seed=22)]
    wins = [capi.Window(x) for x in w_lm]

    # ---- single-threaded references
    ws0 = capi.Workspace()
    ref_edges = {e: _edge_calls(capi, torch, w, dk, mask, pyr, ws0, *e) for e in edges}
    ref_track = [{dof: track(sc, dof) for dof in (6, 7)} for sc in scenes]
    ref_lm = [_lm_trace(capi, win, 12) for win in wins]
    ws0.close()
    torch.cuda.synchronize()

    # ---- the same calls from 4 + 2 threads at once
    streams = [None, None, torch.cuda.Stream(), torch.cuda.Stream()]
    failures, counts = [], [0] * (NT + 2)
    t_end = time.perf_counter() + HAMMER_SECONDS
    go = threading.Barrier(NT + 2)

    def operator_thread(t):
        try:
            torch.cuda.set_device(0)
            ws = capi.Workspace(streams[t])
            scenes[t].ws.close()
            scenes[t].ws = capi.Workspace(streams[t])            # tracker on this thread's stream too
            go.wait()
            while time.perf_counter() < t_end or counts[t] == 0:
                for e in my_edges[t]:
                    d = _same(_edge_calls(capi, torch, w, dk, mask, pyr, ws, *e), ref_edges[e])
                    if d:
                        failures.append(("edge", t, e, counts[t], d))
                for dof in (6, 7):
                    d = _same(track(scenes[t], dof), ref_track[t][dof])
                    if d:
                        failures.append(("track", t, dof, counts[t], d))
                counts[t] += 1
            ws.close()
        except Exception as ex:                                   # noqa: BLE001 -- reported by the main thread
            failures.append(("exception", t, repr(ex)))

    def lm_thread(i):
        try:
            torch.cuda.set_device(0)
            go.wait()
            while time.perf_counter() < t_end or counts[NT + i] == 0:
                if not np.array_equal(_lm_trace(capi, wins[i], 12), ref_lm[i]):
                    failures.append(("lm", i))
                counts[NT + i] += 1
        except Exception as ex:                                   # noqa: BLE001
            failures.append(("exception", NT + i, repr(ex)))

    threads = [threading.Thread(target=operator_thread, args=(t,)) for t in range(NT)] + \
              [threading.Thread(target=lm_thread, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not any(th.is_alive() for th in threads), "a thread hung"
    assert not failures, failures[:10]
    assert all(c >= 1 for c in counts), counts
    print("rounds per thread:", counts)
    for win in wins:
        win.close()
    for sc in scenes:
        sc.close()


_LIFECYCLE_SNIPPET = r"""
import os, sys
sys.path.insert(0, %(root)r)
from sage_slam_amd import capi, synth
def n_tasks():
    return len(os.listdir("/proc/self/task"))
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()          # runtime threads exist before the baseline is taken
w = synth.make_window(K=24, H=32, W=40, FS=16, CS=32, L=3, seed=3)     # long enough for the two-ended elimination order
a = capi.Window(w); b = capi.Window(w)
st = capi.SageLmState(); cfg = capi.lm_config_default()
base = n_tasks() - capi.host_threads_running()
a.lm_step(st, cfg); b.lm_step(capi.SageLmState(), cfg)
up = capi.host_threads_running()
a.close()
mid = capi.host_threads_running()                                 # one window is still alive: the helpers stay
b.close()
down, tasks = capi.host_threads_running(), n_tasks()
c = capi.Window(w); c.lm_step(capi.SageLmState(), cfg)            # ... and come back for the next window
again = capi.host_threads_running()
e0 = st.candidate_error
c.close()
sys.stdout.write("%%d %%d %%d %%d %%d %%d %%d" %% (base, up, mid, down, tasks, again, capi.host_threads_running()))
"""


def test_no_library_thread_outlives_the_last_window():
    """r06: the solve's helper threads are started by the first solve, survive while any window is alive and are
    stopped AND joined by the last sage_window_destroy of the process (no detached thread, no monitor by default); the next window
    starts them again."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "SAGE_PLACEMENT_MONITOR"}
    r = subprocess.run([sys.executable, "-c", _LIFECYCLE_SNIPPET % {"root": root}], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    base, up, mid, down, tasks, again, end = (int(v) for v in r.stdout.split())
    assert up >= 1 and mid == up, (up, mid)
    assert down == 0 and end == 0 and again == up, (down, again, end)
    assert tasks == base, (tasks, base)          # joined: the process is back at its thread count from before the first solve
