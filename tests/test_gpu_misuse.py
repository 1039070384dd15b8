"""Window-API misuse on a live device: every wrong call answers with a status code (SAGE_E_INVALID -1, SAGE_E_UNSUPPORTED -2,
SAGE_E_STATE -4), never with a fault, and leaves the window usable.  (The argument checks that need no device are probed for
every entry point in tests/test_host_logic.py::test_every_entry_point_survives_null_and_zero_arguments.)"""
import ctypes as C

import numpy as np
import pytest

from sage_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from sage_slam_amd import capi as c
    c.lib()
    return c


def test_window_misuse_returns_status_codes(capi):
    L = capi.lib()
    fp = capi._fp
    w = synth.make_window(K=4, H=48, W=64, FS=16, CS=32, L=3, n_samples=1500, seed=3)
    win = capi.Window(w)
    h = win.h
    A = np.zeros((200, 200), np.float32); b = np.zeros(200, np.float32); e = C.c_float(); n = C.c_float()
    pose = np.zeros(12, np.float32); code = np.zeros(32, np.float32); s = C.c_float()
    # a finalized window: structure is frozen, indices are checked
    assert L.sage_window_add_link(h, 0, 1) < 0 and L.sage_window_add_link(h, 0, 99) < 0
    assert L.sage_window_set_shard(h, 0, 2) < 0
    assert L.sage_window_finalize(h) < 0                                               # twice
    for t, ed in ((0, 10 ** 6), (0, -1), (7, 0)):
        assert L.sage_window_get_edge(h, t, ed, fp(A), fp(b), C.byref(e), C.byref(n)) < 0
    for k in (-1, 4):
        assert L.sage_window_get_keyframe(h, k, fp(pose), fp(code), C.byref(s)) < 0
    assert L.sage_window_set_keyframe(h, 99, fp(pose), fp(code), C.c_float(1.0)) < 0
    assert L.sage_window_set_runs(h, 0) < 0 and L.sage_window_set_runs(h, 1000) < 0
    assert L.sage_window_error(h, 2) < 0 and L.sage_window_error(h, -1) < 0
    assert L.sage_window_set_link_geo_loss(h, 99, C.c_float(1.0)) < 0
    assert L.sage_window_solve(h, C.c_double(1e-3), None) == -4                       # nothing linearized yet: SAGE_E_STATE
    # ... and the window still works
    st = capi.SageLmState(); cfg = capi.lm_config_default(); cfg.max_inner_evals = 1
    win.lm_step(st, cfg)
    assert st.accepted == 1 and st.candidate_error < st.error
    # a window that was never finalized: every evaluation entry point refuses (SAGE_E_STATE)
    h2 = C.c_void_p()
    assert L.sage_window_create(C.byref(win.cfg), C.c_void_p(0), C.byref(h2)) == 0
    st2 = capi.SageLmState()
    for rc in (L.sage_window_linearize(h2), L.sage_window_error(h2, 1), L.sage_window_lm_step(h2, C.byref(st2), C.byref(cfg)),
               L.sage_window_solve(h2, C.c_double(1e-3), None), L.sage_window_tune_runs(h2, None, None, None, None)):
        assert rc == -4
    assert L.sage_window_finalize(h2) < 0                                              # no keyframes
    L.sage_window_destroy(h2)
    # configurations the kernels are not instantiated for / a missing mask
    bad = capi.SageWindowConfig()
    h3 = C.c_void_p()
    for field, value, want in (("CS", 17, -2), ("FS", 20, -2), ("mask_dev", 0, -1)):
        C.memmove(C.byref(bad), C.byref(win.cfg), C.sizeof(bad))
        setattr(bad, field, value)
        assert L.sage_window_create(C.byref(bad), C.c_void_p(0), C.byref(h3)) == want, field
    win.close()
