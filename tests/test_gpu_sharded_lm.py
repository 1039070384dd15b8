"""Sharded LM iteration on the GPU: two ranks (gloo, both on cuda:0) drive sage_window_lm_step through the all-reduce
hook (sage_window_set_allreduce) and must walk the same LM trajectory as the single-rank window."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make():
    from sage_slam_amd import synth
    return synth.make_window(K=6, H=48, W=64, FS=16, CS=32, L=3, n_samples=900, seed=5)


def _run(win, capi, steps, at_candidate=False):
    cfg = capi.lm_config_default()
    cfg.max_inner_evals = 1
    cfg.linearize_at_candidate = {True: 1, False: -1, None: 0}[at_candidate]   # None: the engine's automatic choice
    st = capi.SageLmState()
    trace = []
    for _ in range(steps):
        win.lm_step(st, cfg)
        trace.append((st.error, st.candidate_error, int(st.accepted), st.damp))
    return np.array(trace)


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sage_slam_amd import capi
    w = _make()
    win = capi.Window(w, rank=rank, world=world)
    st = capi.SageLmState()
    cfg = capi.lm_config_default()
    try:                                           # a sharded window refuses to step without the hook
        win.lm_step(st, cfg)
        raise AssertionError("lm_step ran a sharded window without an all-reduce hook")
    except capi.SageError:
        pass
    win.set_allreduce(dist)
    np.save(os.path.join(out_dir, f"trace_{rank}.npy"), _run(win, capi, 4))
    np.save(os.path.join(out_dir, f"vars_{rank}.npy"), win.delta())
    # the linearize-at-candidate variant through the same hook (reduced candidate systems, restored on a rejection)
    win.reset()
    np.save(os.path.join(out_dir, f"trace_c_{rank}.npy"), _run(win, capi, 6, at_candidate=True))
    # automatic (linearize_at_candidate = 0): a reduced window takes the one-collective sequence by itself
    win.reset()
    np.save(os.path.join(out_dir, f"trace_a_{rank}.npy"), _run(win, capi, 6, at_candidate=None))
    dist.destroy_process_group()


def test_sharded_lm_step_matches_single_rank(tmp_path):
    import torch.multiprocessing as mp
    from sage_slam_amd import capi
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    t0, t1 = (np.load(tmp_path / f"trace_{r}.npy") for r in range(world))
    assert np.array_equal(t0, t1)                  # every rank solves the same reduced system
    assert np.array_equal(np.load(tmp_path / "vars_0.npy"), np.load(tmp_path / "vars_1.npy"))
    single = _run(capi.Window(_make()), capi, 4)
    assert np.array_equal(single[:, 2], t0[:, 2])  # same accept / reject decisions
    # fp32 per-edge sums are reduced in double; only the order of the double additions differs between 1 and 2 ranks
    np.testing.assert_allclose(t0[:, :2], single[:, :2], rtol=1e-6)
    assert t0[-1, 1] < t0[0, 0]
    c0, c1 = (np.load(tmp_path / f"trace_c_{r}.npy") for r in range(world))
    assert np.array_equal(c0, c1)
    single6 = _run(capi.Window(_make()), capi, 6)
    assert np.array_equal(single6[:, 2], c0[:, 2])
    fin = np.isfinite(single6[:, 1])
    np.testing.assert_allclose(c0[:, 0], single6[:, 0], rtol=2e-6)
    np.testing.assert_allclose(c0[fin, 1], single6[fin, 1], rtol=2e-6)
    a0, a1 = (np.load(tmp_path / f"trace_a_{r}.npy") for r in range(world))
    assert np.array_equal(a0, a1) and np.array_equal(a0, c0)      # automatic == forced, on every rank


# ---------------------------------------------------------------------------------------------------------------
# domain-decomposed solve wired into sage_window_lm_step (SAGE_SHARD_SCHUR=1): the all-reduced payload is the separator
# system, a rank only updates the keyframes it touches; same trajectory as the single-rank window, and after
# sage_window_sync_variables every rank holds the single-rank variables
# ---------------------------------------------------------------------------------------------------------------
def _make_long():
    from sage_slam_amd import synth
    return synth.make_window(K=14, H=32, W=40, FS=16, CS=16, L=2, n_samples=500, seed=8)


def _all_vars(win, K):
    out = []
    for k in range(K):
        pose, code, scale = win.get_keyframe(k)
        out.append(np.concatenate([pose, code, [scale]]))
    return np.array(out)


def _schur_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SAGE_SHARD_SCHUR"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sage_slam_amd import capi
    w = _make_long()
    win = capi.Window(w, rank=rank, world=world)
    win.set_allreduce(dist)
    np.save(os.path.join(out_dir, f"strace_{rank}.npy"), _run(win, capi, 4))
    win.sync_variables()
    np.save(os.path.join(out_dir, f"svars_{rank}.npy"), _all_vars(win, len(w.keyframes)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_schur_sharded_lm_step_matches_single_rank(tmp_path, world):
    import torch.multiprocessing as mp
    from sage_slam_amd import capi
    mp.spawn(_schur_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    traces = [np.load(tmp_path / f"strace_{r}.npy") for r in range(world)]
    for t in traces[1:]:
        assert np.array_equal(t, traces[0])        # the reduced totals are identical on every rank
    w = _make_long()
    ref = capi.Window(w)
    single = _run(ref, capi, 4)
    assert np.array_equal(single[:, 2], traces[0][:, 2])
    np.testing.assert_allclose(traces[0][:, :2], single[:, :2], rtol=1e-6)
    assert traces[0][0, 2] == 1 and traces[0][-1, 1] < traces[0][0, 0]
    v_ref = _all_vars(ref, len(w.keyframes))
    for r in range(world):
        v = np.load(tmp_path / f"svars_{r}.npy")
        assert np.abs(v - v_ref).max() < 2e-5 * max(1.0, np.abs(v_ref).max())


# ---------------------------------------------------------------------------------------------------------------
# r05: peer emulation (sage_window_emulate_peers; bench.py's shard_emulation): ONE rank of a 3-rank job on one device, a
# one-rank RCCL communicator on its stream, the other ranks' share of every reduced system from a table that was computed
# beforehand at the iterates of the job's own trajectory -- the rank must walk the single-rank window's trajectory
# (same accept / reject decisions, errors and iterates to the reduction's rounding)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rank", [0, 2])
def test_emulated_peers_walk_the_real_trajectory(rank):
    import torch
    from sage_slam_amd import capi
    w = _make()
    K, world, steps = len(w.keyframes), 3, 3
    classic = capi.lm_config_default(); classic.max_inner_evals = 1; classic.linearize_at_candidate = -1
    full = capi.Window(w)
    st = capi.SageLmState()
    table, xs, ref = [], [], []
    for i in range(steps + 1):
        full.linearize()
        torch.cuda.synchronize()
        table.append(full.packed_tensor().clone())
        xs.append([full.get_keyframe(k) for k in range(K)])
        if i < steps:
            full.lm_step(st, classic)
            assert st.accepted == 1
            ref.append((st.error, st.candidate_error, int(st.accepted), st.damp))
    v_ref = _all_vars(full, K)
    full.close()
    comm = capi.rccl_comm_create(capi.rccl_unique_id(), 0, 1)
    sh = capi.Window(w, rank=rank, world=world)
    rest = torch.empty(steps + 1, sh.packed_count, dtype=torch.float64, device="cuda")
    for i in range(steps + 1):
        for k in range(K):
            sh.set_keyframe(k, *xs[i][k])
        sh.linearize()
        torch.cuda.synchronize()
        rest[i] = table[i] - sh.packed_tensor()
    sh.reset()
    sh.use_rccl(comm)
    sh.emulate_peers(rest)
    got = _run(sh, capi, steps, at_candidate=None)            # automatic: the one-collective sequence of a reduced window
    ref = np.array(ref)
    assert np.array_equal(got[:, 2], ref[:, 2]) and np.array_equal(got[:, 3], ref[:, 3])
    np.testing.assert_allclose(got[:, :2], ref[:, :2], rtol=2e-6)
    v = _all_vars(sh, K)
    assert np.abs(v - v_ref).max() < 2e-5 * max(1.0, np.abs(v_ref).max())
    sh.close()
    capi.rccl_comm_destroy(comm)
