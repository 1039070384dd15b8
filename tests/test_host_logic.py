"""CPU-only tests of the product's host logic and of the C-ABI library surface (no compute calls on a GPU).

  * libsage_ba.so loads and exports every symbol include/sage_ba.h declares
  * host helpers: camera pyramid, se3_exp, retraction, nearest-PSD, damped QR solve, block solve
  * the tracker LM policy (a8) driven by evaluation callbacks
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from sage_slam_amd import capi, synth
from tests.helpers import presample_source, rel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    hdr = open(os.path.join(ROOT, "include", "sage_ba.h")).read()
    declared = sorted(set(re.findall(r"\b(sage_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/sage_ba.h but not exported"
    assert sorted(capi.SYMBOLS) == declared
    assert b"gfx950" in L.sage_version()


def test_compute_entry_points_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.SageError):
        capi.Workspace()


def test_camera_pyramid_matches_reference_rule(orc):
    cam = synth.Camera(144.0, 143.0, 80.5, 63.5, 160, 128)
    p = capi.make_pyramid(cam, 4)
    ref = orc.camera_pyramid(cam.as_array(), 4)
    py = synth.camera_pyramid(cam, 4)
    offs, P = synth.level_offsets_of(py)
    assert p.levels == 4 and p.P == P == 128 * 160 + 64 * 80 + 32 * 40 + 16 * 20
    for l in range(4):
        c = p.cam[l]
        got = np.array([c.fx, c.fy, c.cx, c.cy, c.w, c.h], np.float32)
        assert np.array_equal(got, ref[l]) and np.array_equal(got, py[l].as_array())
        assert p.level_offsets[l] == offs[l]


def test_se3_exp_and_retract(orc):
    rng = np.random.default_rng(0)
    for _ in range(20):
        w = rng.normal(0, 0.3, 3); v = rng.normal(0, 0.5, 3)
        R, t = capi.se3_exp(w, v)
        Ro, to = orc.se3_exp(w, v, prec="f64")
        assert rel(R, Ro) < 1e-6 and rel(t, to) < 1e-6
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-6)
    R, t = capi.se3_exp(np.zeros(3), np.array([1.0, 2.0, 3.0]))          # theta == 0 branch
    assert np.allclose(R, np.eye(3)) and np.allclose(t, [1, 2, 3])
    # left retraction with [trans, rot] order (gtsam_traits.h:45-70)
    R0 = synth.so3_exp(np.array([0.2, -0.1, 0.3])); t0 = np.array([0.3, 0.1, -0.2])
    d = np.array([0.01, -0.02, 0.03, 0.02, 0.01, -0.015])
    out = capi.pose_retract(capi.pack_pose(R0, t0), d)
    dR, dt = orc.se3_exp(d[3:], d[:3], prec="f64")
    assert rel(out[:9].reshape(3, 3), dR @ R0) < 1e-6 and rel(out[9:], dR @ t0 + dt) < 1e-6


def test_nearest_psd_is_higham():
    rng = np.random.default_rng(1)
    A = rng.normal(size=(12, 12)); S = A @ A.T
    assert rel(capi.nearest_psd(S), S) < 1e-12                      # PSD input is a fixed point
    M = S + 1e-3 * rng.normal(size=S.shape)                          # slightly asymmetric
    assert rel(capi.nearest_psd(M), 0.5 * (M + M.T)) < 1e-9
    Bm = 0.5 * (A + A.T)                                             # indefinite -> negative part clipped
    w, V = np.linalg.eigh(Bm)
    ref = V @ np.diag(np.maximum(w, 0)) @ V.T
    out = capi.nearest_psd(Bm)
    assert rel(out, ref) < 1e-8 and np.linalg.eigvalsh(out).min() > -1e-9


def test_nearest_psd_as_written_matches_eigen_fixture():
    """a12 as the reference wrote it (mapping_utils.h:104-128): sage_nearest_psd_reference against matrices the
    reference's own source text produced with the vendored Eigen 3.3.9 (tests/golden/make_nearest_psd_golden.py).
    Also pins the two facts DESIGN s6 states: (i) the as-written function is NOT the identity on positive definite input
    (H = V^T S V instead of V S V^T moves an SPD matrix by ~35 %), so it is not Higham's projection; (ii) on a
    gauge-deficient system of the shape every dense factor produces (pose block [[A,-A],[-A,A]]) the reference's own
    result moves by ~20 % when the input changes by ~1e-14 (relative): there it is not a function of the matrix in any useful sense."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "nearest_psd_eigen339.npz"))
    names = sorted({k[:-2] for k in z.files})
    assert len(names) == 14
    for nm in names:
        M, A = z[nm + "_M"], z[nm + "_A"]
        out = capi.nearest_psd_reference(M)
        assert rel(out, A) < 1e-12, nm                                  # (bit-identical on the build host)
        assert np.linalg.eigvalsh(out).min() > -1e-9 * np.abs(out).max()
    M, A = z["pd_29_M"], z["pd_29_A"]
    assert np.linalg.eigvalsh(M).min() > 0 and rel(A, M) > 0.2          # (i): SPD in, something else out
    assert rel(capi.nearest_psd(M), M) < 1e-12                          # Higham's projection leaves SPD input alone
    a, b = z["gauge_29_A"], z["gauge_29_perturbed_A"]
    assert rel(z["gauge_29_perturbed_M"], z["gauge_29_M"]) < 2e-14 and rel(b, a) > 0.05        # (ii)
    ha, hb = capi.nearest_psd(z["gauge_29_M"]), capi.nearest_psd(z["gauge_29_perturbed_M"])
    assert rel(hb, ha) < 1e-12                                          # the intended projection is stable there


@pytest.mark.parametrize("type_,CS", [(0, 16), (0, 32), (1, 16), (1, 32)])
def test_factor_hessian_blocks_follow_the_reference_partition(type_, CS):
    """a6 / a7: block partition of PhotometricFactor::linearize (photometric_factor.cpp:142-218, keys {p0,p1,c0,s0}) and
    GeometricFactor::linearize (geometric_factor.cpp:120-218, keys {p0,p1,c0,c1,s0,s1}): G_ij = corrected_AtA.block(off_i,
    off_j, d_i, d_j) for i <= j in push order, g_i = Atb segments, after NearestPsd on the double-widened AtA."""
    rng = np.random.default_rng(CS + type_)
    dims = [6, 6, CS, 1] if type_ == 0 else [6, 6, CS, CS, 1, 1]
    D = sum(dims)
    J = rng.normal(size=(3 * D, D)).astype(np.float32)
    AtA = (J.T @ J).astype(np.float32); Atb = rng.normal(size=D).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(dims)])
    for mode, fn in ((0, lambda M: M), (1, capi.nearest_psd), (2, capi.nearest_psd_reference)):
        blocks, gs, d = capi.factor_hessian_blocks(type_, CS, AtA, Atb, mode)
        assert d == dims and len(blocks) == len(dims) * (len(dims) + 1) // 2
        Cm = fn(AtA.astype(np.float64))
        for (i, j), G in blocks.items():
            assert np.array_equal(G, Cm[offs[i]:offs[i + 1], offs[j]:offs[j + 1]]), (mode, i, j)
        for i, g in enumerate(gs):
            assert np.array_equal(g, Atb[offs[i]:offs[i + 1]].astype(np.float64))
    assert list(blocks.keys()) == [(i, j) for i in range(len(dims)) for j in range(i, len(dims))]   # G11 G12 .. G22 ..


def test_damped_qr_solve():
    rng = np.random.default_rng(2)
    for n in (6, 7):
        J = rng.normal(size=(50, n)); A = (J.T @ J).astype(np.float32); b = rng.normal(size=n).astype(np.float32)
        x = capi.damped_solve_qr_f32(A, b, 1e-4)
        ref = np.linalg.solve(A.astype(np.float64) + 1e-4 * np.diag(np.diag(A)), b)
        assert rel(x, ref) < 1e-4


def test_block_solve_matches_dense():
    rng = np.random.default_rng(3)
    K, CS = 7, 16
    B = 7 + CS
    links = [(j, i) for i in range(K) for j in range(max(0, i - 3), i)] + [(0, 6)]   # band + a loop closure
    n = K * B
    J = rng.normal(size=(3 * n, n))
    # zero the blocks that are not linked so the dense system has the packed sparsity
    H = J.T @ J
    mask = np.zeros((K, K), bool)
    for a, b in links:
        mask[a, b] = mask[b, a] = True
    mask[np.arange(K), np.arange(K)] = True
    Hs = H * np.kron(mask, np.ones((B, B))) + 5 * n * np.eye(n)
    g = rng.normal(size=n)
    diag = np.stack([Hs[k * B:(k + 1) * B, k * B:(k + 1) * B] for k in range(K)])
    lnk = np.stack([Hs[a * B:(a + 1) * B, b * B:(b + 1) * B] for a, b in links])
    packed = np.concatenate([diag.reshape(-1), lnk.reshape(-1), g, np.zeros(4)])
    Hd, gd, _ = capi.unpack_dense(packed, K, links, CS)
    assert rel(Hd, Hs) < 1e-12
    dadd = rng.uniform(0, 1, n); gadd = rng.normal(size=n)
    d = capi.block_solve(packed, K, links, B, 1e-3, dadd, gadd)
    Hf = Hs + np.diag(dadd)
    ref = np.linalg.solve(Hf + 1e-3 * np.diag(np.diag(Hf)), g + gadd)
    assert rel(d, ref) < 1e-9
    with pytest.raises(capi.SageError):                                 # not positive definite
        capi.block_solve(-packed, K, links, B, 0.0)


@pytest.mark.parametrize("K,CS,window", [(24, 32, 3), (20, 16, 5), (33, 32, 1)])
def test_block_solve_split_window_matches_dense(K, CS, window, monkeypatch):
    """chain-like windows of >= 16 keyframes are eliminated as two independent halves + a separator (on two
    cores when the helper thread picks its half up); the result is the dense solve either way, and a loop
    closure falls back to the plain order."""
    rng = np.random.default_rng(K)
    B = 7 + CS
    n = K * B
    for extra in ([], [(1, K - 2)]):
        links = [(j, i) for i in range(K) for j in range(max(0, i - window), i)] + extra
        mask = np.eye(K, dtype=bool)
        for a, b in links:
            mask[a, b] = mask[b, a] = True
        J = rng.normal(size=(2 * n, n))
        Hs = (J.T @ J) * np.kron(mask, np.ones((B, B))) + 6 * n * np.eye(n)
        g = rng.normal(size=n)
        diag = np.stack([Hs[k * B:(k + 1) * B, k * B:(k + 1) * B] for k in range(K)])
        lnk = np.stack([Hs[a * B:(a + 1) * B, b * B:(b + 1) * B] for a, b in links])
        packed = np.concatenate([diag.reshape(-1), lnk.reshape(-1), g, np.zeros(4)])
        ref = np.linalg.solve(Hs + 1e-4 * np.diag(np.diag(Hs)), g)
        out = []
        for no_split, no_helper in ((None, None), ("1", None)):
            if no_split:
                monkeypatch.setenv("SAGE_SOLVE_NO_SPLIT", no_split)
            else:
                monkeypatch.delenv("SAGE_SOLVE_NO_SPLIT", raising=False)
            for _ in range(3):                                          # repeated: helper claimed / not claimed
                d = capi.block_solve(packed, K, links, B, 1e-4)
                assert rel(d, ref) < 1e-9
                out.append(d)
        assert rel(out[0], out[-1]) < 1e-11


_LOOKAHEAD_SNIPPET = r"""
import sys, numpy as np
from sage_slam_amd import capi
K, CS, window = 40, 32, 3
B = 7 + CS; n = K * B
rng = np.random.default_rng(11)
links = [(j, i) for i in range(K) for j in range(max(0, i - window), i)]
mask = np.eye(K, dtype=bool)
for a, b in links:
    mask[a, b] = mask[b, a] = True
J = rng.normal(size=(2 * n, n))
Hs = (J.T @ J) * np.kron(mask, np.ones((B, B))) + 6 * n * np.eye(n)
g = rng.normal(size=n)
diag = np.stack([Hs[k * B:(k + 1) * B, k * B:(k + 1) * B] for k in range(K)])
lnk = np.stack([Hs[a * B:(a + 1) * B, b * B:(b + 1) * B] for a, b in links])
packed = np.concatenate([diag.reshape(-1), lnk.reshape(-1), g, np.zeros(4)])
ref = np.linalg.solve(Hs + 1e-4 * np.diag(np.diag(Hs)), g)
outs = [capi.block_solve(packed, K, links, B, 1e-4) for _ in range(12)]
assert all(np.linalg.norm(d - ref) / np.linalg.norm(ref) < 1e-9 for d in outs)
assert all(np.array_equal(outs[0], d) for d in outs)          # claimed / not claimed / piped: the same bits
sys.stdout.write("%d %s" % (capi.solve_lookahead_count(), outs[0].tobytes().hex()))
"""


def test_block_solve_lookahead_is_bit_identical():
    """the halves of a split window run as two stages (look-ahead thread + the chain through the previous row) when the
    helper threads answer: same blocks, same order of the sums -> the same solution bit for bit as one thread per half"""
    import subprocess, sys
    res = {}
    for name, extra in (("on", {}), ("off", {"SAGE_SOLVE_NO_LOOKAHEAD": "1"})):
        env = dict(os.environ, **extra)
        env.pop("SAGE_SOLVE_NO_LOOKAHEAD", None) if not extra else None
        r = subprocess.run([sys.executable, "-c", _LOOKAHEAD_SNIPPET], capture_output=True, text=True, env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        cnt, hexbytes = r.stdout.split()
        res[name] = (int(cnt), hexbytes)
    assert res["off"][0] == 0
    assert res["on"][1] == res["off"][1]
    # (whether the helpers answered within their 20 us window depends on the load of the host: reported, not asserted --
    #  the engaged path is also what every `-m gpu` LM test runs through)
    print(f"look-ahead engaged in {res['on'][0]} half-factorisations of 12 solves")


_MONITOR_SNIPPET = _LOOKAHEAD_SNIPPET.split("ref = np.linalg.solve")[0] + r"""
import os, subprocess, time
def solve_for(sec):
    t0 = time.time()
    while time.time() - t0 < sec:
        capi.block_solve(packed, K, links, B, 1e-4)
capi.placement_monitor(True)   # r06: the monitor is opt-in
solve_for(0.6)
before = [c for c in capi.solver_helper_cpus() if c >= 0]
hogs = [subprocess.Popen([sys.executable, "-c", "import os\nos.sched_setaffinity(0, {%d})\nwhile True: pass" % c]) for c in set(before)]
try:
    for _ in range(10):
        solve_for(0.5)
        if capi.solver_placement_moves() > 0:
            break
finally:
    for h in hogs:
        h.kill()
after = [c for c in capi.solver_helper_cpus() if c >= 0]
sys.stdout.write("%d %d %d" % (len(before), capi.solver_placement_moves(), len(set(before) & set(after))))
"""


_LIFECYCLE_SNIPPET = _LOOKAHEAD_SNIPPET.split("ref = np.linalg.solve")[0] + r"""
import os
def n_tasks():
    return len(os.listdir("/proc/self/task"))
base = n_tasks()
for _ in range(20):
    capi.block_solve(packed, K, links, B, 1e-4)
up = n_tasks(); running = capi.host_threads_running()
first = capi.block_solve(packed, K, links, B, 1e-4)
capi.shutdown()
down = n_tasks(); running_down = capi.host_threads_running()
# the threads come back on demand, with the same bits
again = capi.block_solve(packed, K, links, B, 1e-4)
for _ in range(5):
    capi.block_solve(packed, K, links, B, 1e-4)
up2 = capi.host_threads_running()
capi.placement_monitor(True)
for _ in range(5):
    capi.block_solve(packed, K, links, B, 1e-4)
mon = capi.host_threads_running()
capi.shutdown()
down2 = n_tasks()
sys.stdout.write("%d %d %d %d %d %d %d %d %d" % (base, up, running, down, running_down, up2, mon, down2, int(np.array_equal(first, again))))
"""


def test_host_threads_are_joinable_and_monitor_is_opt_in():
    """r06: nothing in the library is detached.  The solve's helper threads start on demand, the placement
    monitor only when asked for (SAGE_PLACEMENT_MONITOR=1 / sage_placement_monitor(1)), sage_shutdown() stops and JOINS
    all of them -- the process is back at its thread count from before the first solve -- and the next solve starts them
    again with a bit-identical result."""
    import subprocess, sys
    if (os.cpu_count() or 1) < 4:
        pytest.skip("no helper threads on < 4 CPUs")
    env = {k: v for k, v in os.environ.items() if k != "SAGE_PLACEMENT_MONITOR"}
    r = subprocess.run([sys.executable, "-c", _LIFECYCLE_SNIPPET], capture_output=True, text=True, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    base, up, running, down, running_down, up2, mon, down2, same = (int(v) for v in r.stdout.split())
    assert running >= 1 and up == base + running, (base, up, running)       # helpers only: no monitor by default
    assert down == base and running_down == 0, (base, down, running_down)   # joined, not merely asked to stop
    assert up2 == running and mon == running + 1 and down2 == base, (up2, mon, down2)
    assert same == 1


def test_placement_monitor_moves_helpers_off_crowded_cores():
    """r05: a helper thread of the hybrid solve whose core another process saturates is moved to a quiet core by the
    placement monitor (run-queue delay of the helper / load on its core's other hardware threads, looked at every 250 ms).
    A child process solves in a loop, pins one busy loop onto every helper's CPU and waits for the monitor."""
    import subprocess, sys
    if (os.cpu_count() or 1) < 8 or not os.path.exists("/proc/self/schedstat") or len(os.sched_getaffinity(0)) < 8:
        pytest.skip("needs >= 8 CPUs and /proc schedstat")
    for attempt in range(2):
        r = subprocess.run([sys.executable, "-c", _MONITOR_SNIPPET], capture_output=True, text=True,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        n_helpers, moves, still_crowded = (int(v) for v in r.stdout.split())
        if n_helpers == 0:
            pytest.skip("no helper threads were placed on this host (topology not exposed)")
        if moves >= 1 and still_crowded < n_helpers:
            return
    # the monitor only moves a helper to a core that is idle on all its hardware threads: on a host whose other cores are
    # busy too there is nowhere to go
    if os.getloadavg()[0] > 0.5 * (os.cpu_count() or 1):
        pytest.skip("host too loaded for a quiet core to exist: load %.1f" % os.getloadavg()[0])
    assert moves >= 1 and still_crowded < n_helpers, (n_helpers, moves, still_crowded)


_ARROW_SNIPPET = r"""
import sys, numpy as np
from sage_slam_amd import capi
K, CS, window = 72, 32, 3
B = 7 + CS; n = K * B
rng = np.random.default_rng(17)
links = [(j, i) for i in range(K) for j in range(max(0, i - window), i)] + [(0, 71), (1, 70), (2, 69), (0, 36)]
mask = np.eye(K, dtype=bool)
for a, b in links:
    mask[a, b] = mask[b, a] = True
J = rng.normal(size=(2 * n, n))
Hs = (J.T @ J) * np.kron(mask, np.ones((B, B))) + 6 * n * np.eye(n)
g = rng.normal(size=n)
diag = np.stack([Hs[k * B:(k + 1) * B, k * B:(k + 1) * B] for k in range(K)])
lnk = np.stack([Hs[a * B:(a + 1) * B, b * B:(b + 1) * B] for a, b in links])
packed = np.concatenate([diag.reshape(-1), lnk.reshape(-1), g, np.zeros(4)])
ref = np.linalg.solve(Hs + 1e-4 * np.diag(np.diag(Hs)), g)
outs = [capi.block_solve(packed, K, links, B, 1e-4) for _ in range(6)]
assert all(np.linalg.norm(d - ref) / np.linalg.norm(ref) < 1e-9 for d in outs)
assert all(np.array_equal(outs[0], d) for d in outs)
sys.stdout.write(outs[0].tobytes().hex())
"""


def test_loop_closure_solve_is_bit_identical_in_every_placement():
    """r05: a loop-closure plan (cover keyframes -> arrow rows over both halves) under every thread placement the solve can
    choose -- the second half on an L3 domain of its own (when the host has one), everything on the caller's domain, halves
    without look-ahead stages, paired / unpaired arrow chains, a two-worker pool: who runs a task changes, the order of every
    sum does not -> the same solution bit for bit (and the dense solve to 1e-9 in each)"""
    import subprocess, sys
    outs = {}
    for name, extra in (("default", {}), ("one_domain", {"SAGE_SOLVE_ONE_DOMAIN": "1"}),
                        ("keep_lookahead", {"SAGE_SOLVE_ONE_DOMAIN": "1", "SAGE_SOLVE_KEEP_LOOKAHEAD": "1"}),
                        ("no_lookahead", {"SAGE_SOLVE_NO_LOOKAHEAD": "1"}), ("no_pairing", {"SAGE_SOLVE_NO_PAIRING": "1"}),
                        ("pool2", {"SAGE_SOLVE_POOL": "2"})):
        env = {k: v for k, v in os.environ.items() if not k.startswith("SAGE_SOLVE_")}
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", _ARROW_SNIPPET], capture_output=True, text=True, env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        outs[name] = r.stdout.strip()
    assert len(set(outs.values())) == 1, {k: v[:16] for k, v in outs.items()}


def test_block_solve_concurrent_callers_share_the_helpers():
    """several host threads factorise split windows at the same time: the helper threads (second half, look-ahead
    stages, worker pool) serve one caller at a time, the others run their halves themselves -- every result is the
    dense solve, and identical to the caller's own single-thread result"""
    import threading
    K, CS, window = 36, 32, 3
    B = 7 + CS
    n = K * B
    rng = np.random.default_rng(5)
    links = [(j, i) for i in range(K) for j in range(max(0, i - window), i)]
    mask = np.eye(K, dtype=bool)
    for a, b in links:
        mask[a, b] = mask[b, a] = True
    J = rng.normal(size=(2 * n, n))
    Hs = (J.T @ J) * np.kron(mask, np.ones((B, B))) + 6 * n * np.eye(n)
    g = rng.normal(size=n)
    diag = np.stack([Hs[k * B:(k + 1) * B, k * B:(k + 1) * B] for k in range(K)])
    lnk = np.stack([Hs[a * B:(a + 1) * B, b * B:(b + 1) * B] for a, b in links])
    packed = np.concatenate([diag.reshape(-1), lnk.reshape(-1), g, np.zeros(4)])
    ref = capi.block_solve(packed, K, links, B, 1e-4)
    assert rel(ref, np.linalg.solve(Hs + 1e-4 * np.diag(np.diag(Hs)), g)) < 1e-9
    errs = []

    def worker():
        try:
            for _ in range(8):
                d = capi.block_solve(packed, K, links, B, 1e-4)
                if not np.array_equal(d, ref):
                    errs.append("result differs")
        except Exception as e:                                          # pragma: no cover
            errs.append(repr(e))

    ts = [threading.Thread(target=worker) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts), "a solve hung"
    assert not errs, errs[:3]


def test_tracker_lm_policy_with_oracle_backend(orc):
    """the product's LM driver (sage_track_lm) with the oracle as evaluation back-end: converges on a
    consistent scene, and its trace obeys the reference policy (camera_tracker.cpp:1156-1279)."""
    w = synth.make_window(K=2, H=32, W=40, FS=16, CS=16, L=3, n_samples=300, seed=31, pose_noise=0.0)
    a, b = w.keyframes[0], w.keyframes[1]
    feat0s = presample_source(orc, w, a)
    dpts0 = (np.float32(a.scale_true) * (a.bias + a.basis @ a.code_true))[a.loc1d].astype(np.float32)
    R10, t10 = synth.relative_pose(a.R_true, a.t_true, b.R_true, b.t_true)
    pose0 = capi.pack_pose(synth.so3_exp(np.array([0.004, -0.003, 0.002])) @ R10,
                           t10 + np.array([0.004, -0.003, 0.002], np.float32))
    cfg = capi.lm_config_default()
    assert (cfg.max_num_iters, cfg.init_damp, cfg.damp_inc_factor) == (40, pytest.approx(1e-4), 100.0)
    calls = dict(lin=0, err=0)

    def lin(p, s):
        calls["lin"] += 1
        o = orc.tracker_photo_jac_error(6, p[:9].reshape(3, 3), p[9:], w.mask, dpts0, a.homo, feat0s, b.feat_pyr,
                                        b.grad_pyr, w.level_offsets, w.cams, w.eps, w.photo_weights)
        return o["AtA"], o["Atb"], o["error"]

    def err(p, s):
        calls["err"] += 1
        return orc.tracker_photo_error(p[:9].reshape(3, 3), p[9:], w.mask, dpts0, a.homo, feat0s, b.feat_pyr,
                                       w.level_offsets, w.cams, w.eps, w.photo_weights)[0]

    e_start = err(pose0, 1.0)
    pose, _, e_final, iters, trace = capi.track_lm(cfg, 6, lin, err, pose0, 1.0)
    assert e_final < 0.5 * e_start and 1 <= iters <= cfg.max_num_iters
    truth = capi.pack_pose(R10, t10)
    assert np.linalg.norm(pose - truth) < 0.5 * np.linalg.norm(pose0 - truth)
    assert trace[0]["relinearized"] == 1 and trace[0]["error"] == pytest.approx(e_start, rel=1e-6)
    for tr in trace:
        assert cfg.min_damp <= tr["damp"] <= cfg.max_damp
        if tr["accepted"]:
            assert tr["candidate_error"] < tr["error"]
    assert calls["lin"] <= iters and calls["err"] >= len(trace)


def test_damped_qr_solve_matches_eigen_fixture():
    """(AtA + damp diag(AtA)).colPivHouseholderQr().solve(Atb) in fp32 (camera_tracker.cpp:1182-1183) against vectors
    produced with the vendored Eigen 3.3.9 (tests/golden/make_colpiv_qr_golden.cpp): the number of meaningful pivots is
    Eigen's nonzeroPivots() -- decided on downdated column norms while pivoting, NOT rank()'s |R_ii| rule --
    so exactly dependent / zero columns give exactly-zero components and near-dependent ones (gap down to 1e-4, which a
    rank() style cut would drop) are solved in full like Eigen does."""
    import json
    import os
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "colpiv_qr_eigen339.json")))
    assert len(cases) >= 20
    for c in cases:
        n = c["n"]
        A = np.array(c["A"], np.float32).reshape(n, n); b = np.array(c["b"], np.float32)
        xe = np.array(c["x"], np.float32)
        x = capi.damped_solve_qr_f32(A, b, c["damp"])
        assert np.isfinite(x).all()
        assert np.array_equal(x == 0, xe == 0), c["name"]                       # same truncation as Eigen
        assert int((xe != 0).sum()) == c["nonzero_pivots"]
        M = A.astype(np.float64) + c["damp"] * np.diag(np.diag(A).astype(np.float64))
        keep = xe != 0
        cond = np.linalg.cond(M[np.ix_(keep, keep)])
        # two fp32 evaluations of the same algorithm differ by summation order only: forward error <= few eps * cond
        assert rel(x, xe) < max(2e-6, 4 * np.finfo(np.float32).eps * cond), (c["name"], rel(x, xe), cond)
    # truncated components of a consistent singular system: the kept part still solves it
    rng = np.random.default_rng(3)
    n = 7
    J = rng.normal(size=(40, n - 2))
    Mx = np.concatenate([J, J[:, :1] + J[:, 1:2], np.zeros((40, 1))], 1)      # col 5 = col 0 + col 1, col 6 = 0
    A = (Mx.T @ Mx).astype(np.float32)
    b = (Mx.T @ rng.normal(size=40)).astype(np.float32)
    x = capi.damped_solve_qr_f32(A, b, 0.0)                                     # no damping: rank 5 of 7
    assert np.isfinite(x).all() and (x == 0).sum() >= 2
    assert rel(A.astype(np.float64) @ x, b) < 1e-4


def test_tracker_lm_policy_details():
    """(i) the Jacobian pass' error only initialises curr_error on the first iteration (update_error = curr_iter == 0,
    camera_tracker.cpp:1166/:1491): afterwards the accepted candidate's error stands; (ii) the zero-overlap exit of
    TrackFrame (:1515-1519) -> SAGE_E_NO_OVERLAP before any solve."""
    import ctypes as C
    cfg = capi.lm_config_default()
    target = np.array([0.3, -0.2, 0.1], np.float32)
    seen = []

    def lin(p, s):
        r = p[9:] - target
        A = np.eye(6, dtype=np.float32); g = np.zeros(6, np.float32); g[:3] = -r
        seen.append(len(seen))
        # a Jacobian pass that reports a bogus error after the first iteration must not disturb the trajectory
        return A, g, float(r @ r) if len(seen) == 1 else 1e9

    def err(p, s):
        r = p[9:] - target
        return float(r @ r)

    pose0 = capi.pack_pose(np.eye(3), np.zeros(3))
    pose, _, e_final, iters, trace = capi.track_lm(cfg, 6, lin, err, pose0, 1.0)
    assert len(seen) >= 2 and e_final < 1e-3 and all(t["error"] < 1e8 for t in trace)
    cfg.no_overlap_error = 0.05
    L = capi.lib()
    p = pose0.copy(); fe = C.c_float(); it = C.c_int(); sc = C.c_float(1.0)
    cb1 = capi.TRACK_LIN_FN(lambda ctx, pp, s, A, b, e: (e.__setitem__(0, 0.14), [A.__setitem__(i, 1.0 if i % 7 == 0 else 0.0) for i in range(36)],
                                                       [b.__setitem__(i, 0.0) for i in range(6)], 0)[-1])
    cb2 = capi.TRACK_ERR_FN(lambda ctx, pp, s, e: 1)
    rc = L.sage_track_lm(C.byref(cfg), 6, cb1, cb2, None, p.ctypes.data_as(C.POINTER(C.c_float)), C.byref(sc),
                         C.byref(fe), C.byref(it), None, 0, None)
    assert rc == -5 and it.value == 0 and fe.value == pytest.approx(0.14) and np.array_equal(p, pose0)
    assert b"overlap" in L.sage_error_string(-5)


def test_tracker_lm_dof7_rescales_depths(orc):
    """TrackFrame's 7th variable moves the depths of every evaluation (guess_scale_0 * unscaled_*_dpts_0,
    camera_tracker.cpp:264,:273,:431,:453).  The photometric term alone cannot see the scale (s and t trade off exactly),
    the match-geometry term can: the oracle-driven LM (photometric + match geometry, composed like
    ComputeJacobianAndError :330-374) started 4 % low comes back to the true scale (low, not high: LMConvergence takes
    the SIGNED maximum of the relative increments, :531-536, so a pure decrease counts as converged); with the depths
    frozen at the initial scale (the r1 defect) the cost does not follow the variable and the scale runs away."""
    w = synth.make_window(K=2, H=32, W=40, FS=16, CS=16, L=3, n_samples=400, seed=31, pose_noise=0.0)
    a, b = w.keyframes[0], w.keyframes[1]
    feat0s = presample_source(orc, w, a)
    unscaled = (a.bias + a.basis @ a.code_true)[a.loc1d].astype(np.float32)
    s_true = float(a.scale_true)
    R10, t10 = synth.relative_pose(a.R_true, a.t_true, b.R_true, b.t_true)
    rng = np.random.default_rng(5)
    NK = 120
    cam = w.cams[0]
    xs = rng.integers(6, w.W - 6, NK); ys = rng.integers(6, w.H - 6, NK)
    kp_homo0 = np.stack([(xs - cam.cx) / cam.fx, (ys - cam.cy) / cam.fy, np.ones(NK)], 1).astype(np.float32)
    kp_unscaled = (a.bias + a.basis @ a.code_true)[ys * w.W + xs].astype(np.float32)
    X1 = (R10.astype(np.float64) @ (s_true * kp_unscaled[:, None] * kp_homo0).T).T + t10
    kp_dpts1 = X1[:, 2].astype(np.float32)
    kp_homo1 = (X1 / X1[:, 2:3]).astype(np.float32)
    c_mg, w_mg = 0.1 * float(np.mean(a.bias ** 2)), 3.0
    pose0 = capi.pack_pose(R10, t10)
    cfg = capi.lm_config_default()
    F = np.float32

    def make(frozen):
        def sc(s):
            return F(s_true * 0.96) if frozen else F(s)

        def lin(p, s):
            R, t = p[:9].reshape(3, 3), p[9:]
            o = orc.tracker_photo_jac_error(7, R, t, w.mask, sc(s) * unscaled, a.homo, feat0s, b.feat_pyr,
                                            b.grad_pyr, w.level_offsets, w.cams, w.eps, w.photo_weights, scale0=s)
            m = orc.match_geom_jac_error(3, "fair", R, t, dpts0=sc(s) * kp_unscaled, dpts1=kp_dpts1, homo0=kp_homo0,
                                         homo1=kp_homo1, scale0=s, loss_param=c_mg, weight=w_mg)
            return o["AtA"].astype(F) + m["AtA"].astype(F), o["Atb"].astype(F) + m["Atb"].astype(F), o["error"] + m["error"]

        def err(p, s):
            R, t = p[:9].reshape(3, 3), p[9:]
            return (orc.tracker_photo_error(R, t, w.mask, sc(s) * unscaled, a.homo, feat0s, b.feat_pyr, w.level_offsets,
                                            w.cams, w.eps, w.photo_weights)[0]
                    + orc.match_geom_error(2, "fair", R, t, dpts0=sc(s) * kp_unscaled, dpts1=kp_dpts1, homo0=kp_homo0,
                                           homo1=kp_homo1, loss_param=c_mg, weight=w_mg))
        return lin, err

    _, s_live, e_live, _, tr_live = capi.track_lm(cfg, 7, *make(False), pose0, 0.96 * s_true)
    assert abs(s_live / s_true - 1) < 0.01 and e_live < 0.8 * tr_live[0]["error"]
    _, s_frozen, e_frozen, _, tr_frozen = capi.track_lm(cfg, 7, *make(True), pose0, 0.96 * s_true)
    assert abs(s_frozen / s_true - 1) > 0.03               # the cost never sees the variable: it is not pulled back


def test_synth_producers_match_oracle(orc):
    w = synth.make_window(K=1, H=32, W=40, FS=16, CS=16, L=3, seed=4)
    kf = w.keyframes[0]
    feat = kf.feat_pyr[:, :w.H * w.W].reshape(w.FS, w.H, w.W)
    op, og = orc.gaussian_pyramid_with_grad(feat, w.mask, w.L, w.level_offsets, w.P)
    assert rel(kf.feat_pyr, op) < 1e-6 and rel(kf.grad_pyr, og) < 1e-6
    D1, g1 = synth.depth_and_grad(kf, w.H, w.W)
    od = orc.update_depth(kf.bias, kf.basis, kf.code, kf.scale).reshape(w.H, w.W)
    assert rel(D1, od) < 1e-6


def test_shuffle_indices_host_helper_matches_oracle(orc):
    """sage_shuffle_indices (the engine's host helper, std::shuffle) against the oracle restatement: bit exact;
    empty and single-element inputs."""
    for seed, n in ((0, 0), (3, 1), (1700000000, 16128), (42, 20480), (2**32 + 5, 4097), (9, 70001)):
        a = capi.shuffle_indices(n, seed)
        b = orc.shuffle_indices(n, seed)
        assert a.dtype == np.int64 and np.array_equal(a, b), (seed, n)


def test_bench_multi_gpu_entry_point_spawns_ranks():
    """`python bench.py --gpus N` with no launcher in the environment starts N ranks itself (torch.distributed.run,
    127.0.0.1 rendezvous) and prints ONE line from rank 0; without the devices it refuses with rc 2 instead of measuring
    one rank.  The dry-run knob stops every rank after the rendezvous (no GPU in this container)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SAGE_BENCH_DRY_RUN="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["ranks_seen"] == 3 and len(set(out["pids"])) == 3
    import torch
    if not torch.cuda.is_available():
        env.pop("SAGE_BENCH_DRY_RUN")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 2 and "visible HIP device" in r.stderr and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="4"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_block_solve_ring_closure_cover_plan_matches_dense():
    """Loop-closure windows (BASELINE config 5 shape: a chain whose ends see each other + a mid-chain closure + a local
    long link): plan_blocks covers the long links with a few keyframes that join the separator (order = [first half |
    second half descending | middle separator | cover]), the halves stay two independent banded factorisations and the
    long separator rows are cut into per-half tasks (worker pool when armed, the caller otherwise).  Result == dense
    numpy solve of the same damped system."""
    rng = np.random.default_rng(11)
    for K, CS in ((96, 32), (128, 16)):
        B = 7 + CS
        links = [(j, i) for i in range(K) for j in range(max(0, i - 3), i)]
        links += [(0, K - 1), (2, K - 3), (1, K - 2), (0, K // 2), (K // 5, K // 5 + 30)]
        n = K * B
        diag = np.zeros((K, B, B)); lnk = np.zeros((len(links), B, B))
        for l, (a, b) in enumerate(links):
            J = rng.normal(size=(2 * B, 2 * B + 3)); M = J @ J.T
            diag[a] += M[:B, :B]; diag[b] += M[B:, B:]; lnk[l] = M[:B, B:]
        for k in range(K):
            diag[k] += 0.5 * np.eye(B)
        g = rng.normal(size=n)
        packed = np.concatenate([diag.reshape(-1), lnk.reshape(-1), g, np.zeros(4)])
        H = np.zeros((n, n))
        for k in range(K):
            H[k * B:(k + 1) * B, k * B:(k + 1) * B] = 0.5 * (diag[k] + diag[k].T)
        for l, (a, b) in enumerate(links):
            H[a * B:(a + 1) * B, b * B:(b + 1) * B] += lnk[l]; H[b * B:(b + 1) * B, a * B:(a + 1) * B] += lnk[l].T
        ref = np.linalg.solve(H + 1e-3 * np.diag(np.diag(H)), g)
        for _ in range(3):                                   # (repeat: the pool is asleep on the first call, awake later)
            d = capi.block_solve(packed, K, links, B, 1e-3)
            assert rel(d, ref) < 1e-11
        # a non-positive definite system is reported, not hung on (the tasks watch the abort flag)
        bad = packed.copy(); bad[(K // 3) * B * B] = -1e6
        with pytest.raises(capi.SageError):
            capi.block_solve(bad, K, links, B, 0.0)


def test_non_dyadic_pyramids_are_accepted():
    """r06 (ADVICE r5 / VERDICT r5 missing 5): a pyramid whose level ratios are not powers of two used to be refused
    (SAGE_E_UNSUPPORTED); the kernels now evaluate the reference's own ((p + 0.5) fx_l) / fx_0 - 0.5 for such pyramids
    (tests/test_gpu_parity.py::test_non_dyadic_pyramid_* on the GPU).  Here, without a device: the argument checks of
    sage_window_create no longer stop at the pyramid -- a 1/3 level gets as far as the device query (or a window) -- and
    sage_camera_pyramid builds the odd-size levels of the reference's CameraPyramid (62 -> 31 -> 15)."""
    import ctypes as C
    cam = capi.SageCamera(100.0, 100.0, 40.0, 32.0, 80.0, 64.0)
    pyr = capi.make_pyramid(cam, 3)
    assert [pyr.cam[l].fx / pyr.cam[0].fx for l in range(3)] == [1.0, 0.5, 0.25]
    cfg = capi.SageWindowConfig()
    cfg.pyr = pyr; cfg.FS, cfg.CS = 16, 32
    cfg.mask_dev = 1                                     # never dereferenced: the checks come first
    odd = capi.SageWindowConfig.from_buffer_copy(cfg)
    odd.pyr.cam[1].fx = 100.0 / 3.0                      # a 1/3 level
    h = C.c_void_p()
    rc = capi.lib().sage_window_create(C.byref(odd), None, C.byref(h))
    assert rc != -2, "non-dyadic pyramids must not be SAGE_E_UNSUPPORTED any more"
    if rc == 0:
        capi.lib().sage_window_destroy(h)
    p2 = capi.make_pyramid(capi.SageCamera(55.8, 55.8, 31.0, 25.0, 62.0, 50.0), 3)
    assert [(int(p2.cam[l].w), int(p2.cam[l].h)) for l in range(3)] == [(62, 50), (31, 25), (15, 12)]
    assert p2.cam[2].fx / p2.cam[0].fx != 0.25


_NULL_PROBE = r"""
import ctypes, os, re, sys
root = sys.argv[1]
hdr = open(os.path.join(root, "include", "sage_ba.h")).read()
protos = re.findall(r"^\s*([A-Za-z_][\w \*]*?)\s*\b(sage_\w+)\s*\(([^;]*?)\)\s*;", hdr, re.M | re.S)
L = ctypes.CDLL(os.path.join(root, "sage_slam_amd", "libsage_ba.so"))
crashed, called = [], 0
for ret, name, args in protos:
    parts = [a.strip() for a in re.split(r",(?![^()]*\))", args) if a.strip() and a.strip() != "void"]
    argv = []
    for a in parts:
        if "*" in a or "Fn" in a:
            argv.append(ctypes.c_void_p(0))
        elif re.match(r"(const\s+)?double\b", a):
            argv.append(ctypes.c_double(0))
        elif re.match(r"(const\s+)?float\b", a):
            argv.append(ctypes.c_float(0))
        elif re.match(r"(const\s+)?(int|unsigned|int32_t|int64_t|size_t|uint32_t)\b", a):
            argv.append(ctypes.c_long(0))
        else:
            argv = None            # a struct passed by value: not part of this probe
            break
    if argv is None:
        continue
    called += 1
    pid = os.fork()                # (this interpreter is single-threaded: the library has started no thread yet)
    if pid == 0:
        try:
            f = getattr(L, name)
            f.restype = ctypes.c_int if ret.strip() == "int" else None
            f(*argv)
            os._exit(0)
        except BaseException:
            os._exit(3)
    _, st = os.waitpid(pid, 0)
    if os.WIFSIGNALED(st) or os.WEXITSTATUS(st) != 0:
        crashed.append(name)
print(len(protos), called, ",".join(crashed))
"""


def test_every_entry_point_survives_null_and_zero_arguments():
    """Drop-in robustness: every function include/sage_ba.h declares is called with NULL for every pointer and 0 for every
    scalar, each in a forked child of a fresh interpreter (a fault must not take the test process down, and a fork of THIS
    process would leave the solver's helper threads behind).  None may crash: the `int` entry points answer with a status
    code, the `void` helpers and destroy calls are no-ops.  (No GPU needed: argument validation comes first.)"""
    r = subprocess.run([sys.executable, "-c", _NULL_PROBE, ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    n_protos, called, crashed = (r.stdout.strip().split(" ") + [""])[:3]
    assert int(n_protos) > 90 and int(called) > 90 and crashed == "", r.stdout
