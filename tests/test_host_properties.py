"""Property tests of the host side of the engine (no GPU): random window graphs through the block solvers, NearestPsd
and SE(3) properties.  hypothesis draws the cases; the checker is numpy in double precision."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from sage_slam_amd import capi
from tests.helpers import rel

# derandomize: the CPU suite runs the same cases on every box; no example database in the tree
SET = settings(max_examples=30, deadline=None, derandomize=True, database=None,
               suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


def random_window_system(rng, K, CS, links):
    """SPD block system with the packed sparsity of a window: H = sum over links of (J J^T on the two keyframes) + ridge."""
    B = 7 + CS
    n = K * B
    H = np.zeros((n, n))
    for a, b in links:
        idx = np.concatenate([np.arange(a * B, (a + 1) * B), np.arange(b * B, (b + 1) * B)])
        J = rng.normal(size=(2 * B, 2 * B + 3))
        H[np.ix_(idx, idx)] += J @ J.T
    H += 0.5 * np.eye(n)
    g = rng.normal(size=n)
    diag = np.stack([H[k * B:(k + 1) * B, k * B:(k + 1) * B] for k in range(K)])
    lnk = np.stack([H[a * B:(a + 1) * B, b * B:(b + 1) * B] for a, b in links])
    packed = np.concatenate([diag.reshape(-1), lnk.reshape(-1), g, np.zeros(4)])
    return H, g, packed


@st.composite
def window_graphs(draw):
    K = draw(st.integers(3, 22))
    back = draw(st.integers(1, 3))
    links = [(j, i) for i in range(K) for j in range(max(0, i - back), i)]
    n_loops = draw(st.integers(0, 3))
    for _ in range(n_loops):                                   # loop closures anywhere in the window
        a = draw(st.integers(0, K - 1)); b = draw(st.integers(0, K - 1))
        lo, hi = min(a, b), max(a, b)
        if hi - lo > back and (lo, hi) not in links:
            links.append((lo, hi))
    return K, links, draw(st.sampled_from([16, 32])), draw(st.integers(0, 2 ** 31 - 1))


@SET
@given(window_graphs(), st.floats(0.0, 1e-2))
def test_block_solve_on_random_window_graphs(graph, damp):
    """sage_block_solve (two-halves split, plain order with loop closures, both block sizes) == dense solve"""
    K, links, CS, seed = graph
    rng = np.random.default_rng(seed)
    H, g, packed = random_window_system(rng, K, CS, links)
    d = capi.block_solve(packed, K, links, 7 + CS, damp)
    ref = np.linalg.solve(H + damp * np.diag(np.diag(H)), g)
    assert rel(d, ref) < 1e-9


@SET
@given(window_graphs(), st.integers(2, 5))
def test_domain_decomposed_solve_on_random_window_graphs(graph, ndomains):
    """sage_block_solve_domains (keyframe-range domains on host threads, far ends of long links as separators) and the
    rank-mode plan (link-range shards, sage_shard_*) agree with sage_block_solve on any window graph"""
    K, links, CS, seed = graph
    rng = np.random.default_rng(seed)
    H, g, packed = random_window_system(rng, K, CS, links)
    B = 7 + CS
    ref = np.linalg.solve(H, g)
    d = capi.block_solve_domains(packed, K, links, B, 0.0, min(ndomains, K))
    assert rel(d, ref) < 1e-9


@SET
@given(st.integers(2, 40), st.integers(0, 2 ** 31 - 1), st.floats(0.0, 3.0))
def test_nearest_psd_properties(n, seed, neg):
    """Higham projection: symmetric PSD output, identity on PSD input, idempotent, never farther from the input than the
    clipped-eigenvalue matrix it is defined by"""
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, n)); A = 0.5 * (A + A.T) + (1.0 - neg) * np.eye(n)
    P = capi.nearest_psd(A)
    assert np.allclose(P, P.T, atol=1e-12)
    assert np.linalg.eigvalsh(P).min() > -1e-9 * max(1.0, np.abs(A).max())
    scale = max(1.0, np.abs(A).max())                           # absolute bars: P may be numerically zero (A negative definite)
    assert np.linalg.norm(capi.nearest_psd(P) - P) <= 1e-9 * scale * n
    w, V = np.linalg.eigh(A)
    clipped = (V * np.maximum(w, 0.0)) @ V.T
    assert np.linalg.norm(P - A) <= np.linalg.norm(clipped - A) * (1 + 1e-6) + 1e-9
    S = A @ A.T + 1e-3 * np.eye(n)                              # positive definite -> unchanged
    assert rel(capi.nearest_psd(S), S) < 1e-10


@SET
@given(st.integers(0, 2 ** 31 - 1), st.floats(1e-6, 3.0))
def test_se3_exp_and_retract_properties(seed, angle):
    """R is a rotation by |omega| about omega; exp(0, v) is a pure translation; retract composes from the left
    (gtsam_traits.h:45-70: T_new = exp(delta) T)"""
    rng = np.random.default_rng(seed)
    axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
    omega = (angle * axis).astype(np.float32); v = rng.normal(size=3).astype(np.float32)
    R, t = capi.se3_exp(omega, v)
    R = np.asarray(R, np.float64).reshape(3, 3)
    assert np.allclose(R @ R.T, np.eye(3), atol=5e-6) and abs(np.linalg.det(R) - 1) < 5e-6
    assert np.allclose(R @ axis, axis, atol=5e-6)
    assert abs(np.trace(R) - (1 + 2 * np.cos(angle))) < 2e-5
    R0, t0 = capi.se3_exp(np.zeros(3, np.float32), v)
    assert np.allclose(np.asarray(R0).reshape(3, 3), np.eye(3), atol=1e-7) and np.allclose(t0, v, atol=1e-7)
    pose = np.concatenate([R.reshape(-1), rng.normal(size=3)]).astype(np.float32)
    d6 = np.concatenate([v, omega]).astype(np.float32)          # [trans(3), rot(3)]
    out = np.asarray(capi.pose_retract(pose, d6), np.float64)
    Rn, tn = out[:9].reshape(3, 3), out[9:]
    assert np.allclose(Rn, R @ pose[:9].reshape(3, 3).astype(np.float64), atol=2e-5)
    assert np.allclose(tn, R @ pose[9:].astype(np.float64) + np.asarray(t, np.float64), atol=2e-5)


@SET
@given(st.integers(1, 5000), st.integers(0, 2 ** 31 - 1))
def test_shuffle_indices_is_a_seeded_permutation(n, seed):
    a = capi.shuffle_indices(n, seed); b = capi.shuffle_indices(n, seed)
    assert np.array_equal(a, b) and np.array_equal(np.sort(a), np.arange(n))
