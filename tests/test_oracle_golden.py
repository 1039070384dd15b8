"""Pin the CPU oracle against the reference's own Python implementation of the
photometric / geometric terms (fixtures: tests/golden/diffba_*.npz, generated from
/root/reference/representation/models/diff_ba.py by tests/golden/make_diffba_golden.py).

What this pins (SURVEY.md s8c "Python secondary oracle"): sampling conventions
(zero-padded bilinear == grid_sample(align_corners=False)), the relative-pose
projection Jacobian (a3), the depth/code/scale Jacobians (a1/a3/a4), the geometric
residual, its Jacobian wrt pose0/code0/scale0 and the Cauchy weighting (a4).
Column order differs: diff_ba = [rot3, trans3, scale, code], C++ = [trans3, rot3, ...].
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["diffba_allvalid", "diffba_invalid"]


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: z[k] for k in z.files}


def setup(c):
    H, W, N, FS, CS = (int(c[k]) for k in ("H", "W", "N", "FS", "CS"))
    fx, fy, cx, cy = (float(v) for v in c["intr"])
    cams = np.array([[fx, fy, cx, cy, W, H]], dtype=np.float32)
    lo = np.array([0], dtype=np.int32)
    feat1 = c["feat1"].reshape(FS, H * W).astype(np.float32)
    grad1 = np.stack([c["gx"].reshape(FS, -1), c["gy"].reshape(FS, -1)], 0).astype(np.float32)
    homo = np.ascontiguousarray(c["homo"].T, dtype=np.float32)
    return H, W, N, FS, CS, cams, lo, feat1, grad1, homo


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_tracker_photo_rows_match_diffba(orc, name, prec):
    c = load(name)
    H, W, N, FS, CS, cams, lo, feat1, grad1, homo = setup(c)
    scale = float(c["scale"])
    dpts0 = (np.float32(scale) * (c["bias"] + c["basis"] @ c["code"])).astype(np.float32)
    feat0s = np.ascontiguousarray(c["src_feats"].T[None], dtype=np.float32)      # [1,N,FS]
    o = orc.tracker_photo_jac_error(7, c["R"], c["t"], c["mask"], dpts0, homo, feat0s, feat1, grad1,
                                    lo, cams, float(c["depth_eps"]), np.ones(1, np.float32),
                                    scale0=scale, prec=prec, want_rows=True)
    A = c["photo_A"].reshape(N, FS, 7 + CS)
    valid = c["photo_valid"].reshape(N) > 0.5
    J = o["J"][0]                                                                # [N,FS,7]
    tol = 2e-5 if prec == "f32" else 5e-6
    assert rel(J[valid][..., 0:3], A[valid][..., 3:6]) < tol       # translation columns
    assert rel(J[valid][..., 3:6], A[valid][..., 0:3]) < tol       # rotation columns
    assert rel(J[valid][..., 6], A[valid][..., 6]) < tol           # scale column
    assert np.all(J[~valid] == 0)                                  # C++ masks the Jacobian rows
    assert rel(o["r"][0], c["photo_diff"].T) < tol                 # residual m*(f0-f1)
    assert o["num_inliers"] == pytest.approx(valid.sum())


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_photo_factor_rows_match_diffba(orc, name, prec):
    """a1 with T1 = identity: world-frame pose0 block == relative-pose block; code/scale columns."""
    c = load(name)
    H, W, N, FS, CS, cams, lo, feat1, grad1, homo = setup(c)
    loc = c["loc"].astype(np.int64)
    bias = np.zeros(H * W, np.float32); bias[loc] = c["bias"]
    basis = np.zeros((H * W, CS), np.float32); basis[loc] = c["basis"]
    feat0 = np.zeros((FS, H * W), np.float32); feat0[:, loc] = c["src_feats"]
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    o = orc.photo_jac_error(c["R"], c["t"], c["R"], c["t"], I3, z3, bias, basis, c["code"], c["mask"],
                            loc, homo, feat0, feat1, grad1, lo, float(c["scale"]), cams,
                            float(c["depth_eps"]), np.ones(1, np.float32), prec=prec, want_rows=True)
    A = c["photo_A"].reshape(N, FS, 7 + CS)
    valid = c["photo_valid"].reshape(N) > 0.5
    J = o["J"][0]
    tol = 5e-5 if prec == "f32" else 2e-5      # feat0 re-sampled through homo -> 1e-6-level tap leakage
    assert rel(J[valid][..., 0:3], A[valid][..., 3:6]) < tol
    assert rel(J[valid][..., 3:6], A[valid][..., 0:3]) < tol
    assert np.array_equal(J[..., 6:12], -J[..., 0:6])              # P_pose1 = -P_pose0
    assert rel(J[valid][..., 12:12 + CS], A[valid][..., 7:]) < tol  # code columns
    assert rel(J[valid][..., 12 + CS], A[valid][..., 6]) < tol      # scale column
    assert rel(o["r"][0], c["photo_diff"].T) < 2e-4
    # reduction: AtA = (1/n_in) J^T J over the valid rows the reference would also keep
    Jv = o["J"].reshape(-1, 13 + CS).astype(np.float64)
    AtA = Jv.T @ Jv / valid.sum()
    assert rel(o["AtA"], AtA) < (1e-5 if prec == "f32" else 1e-12)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_geometric_rows_match_diffba(orc, name, prec):
    c = load(name)
    H, W, N, FS, CS, cams, lo, feat1, grad1, homo = setup(c)
    loc = c["loc"].astype(np.int64)
    bias = np.zeros(H * W, np.float32); bias[loc] = c["bias"]
    basis = np.zeros((H * W, CS), np.float32); basis[loc] = c["basis"]
    rng = np.random.default_rng(5)
    basis1 = (0.05 * rng.standard_normal((H, W, CS))).astype(np.float32)   # code1 columns: not in diff_ba
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    loss = float(c["geo_cauchy_factor"]) * float(c["geo_mean_sq"])
    dgrad = np.stack([c["dgx"], c["dgy"]], 0).astype(np.float32)
    o = orc.geo_jac_error(c["R"], c["t"], c["R"], c["t"], I3, z3, bias, basis, c["code"],
                          c["dmap"], dgrad, basis1, c["mask"], loc, homo, float(c["scale"]), 1.0,
                          cams[0], float(c["depth_eps"]), loss, 1.0, prec=prec, want_rows=True)
    A = c["geo_A"]                                 # [N, 7+CS] = sqrt_w * valid * [rot3, trans3, scale, code]
    valid = c["geo_valid"].reshape(N) > 0.5
    J = o["J"]
    tol = 2e-5 if prec == "f32" else 5e-6
    assert rel(J[:, 0:3], A[:, 3:6]) < tol
    assert rel(J[:, 3:6], A[:, 0:3]) < tol
    assert np.array_equal(J[:, 6:12], -J[:, 0:6])
    assert rel(J[:, 12:12 + CS], A[:, 7:]) < tol
    assert rel(J[:, 12 + 2 * CS], A[:, 6]) < tol
    assert rel(o["r"], c["geo_diff"].reshape(-1)) < tol
    assert np.all(J[~valid] == 0)
    assert o["num_inliers"] == pytest.approx(valid.sum())
    assert o["error"] == pytest.approx(float(c["geo_err"].sum()) / valid.sum(), rel=1e-5)


def test_shuffle_restatement_matches_std_shuffle_golden(orc):
    """orc_shuffle_indices (MT19937 + libstdc++ 11 std::shuffle restated in C) against the permutations the literal
    reference call sequence (std::iota / std::mt19937::seed / std::shuffle, mapper.cpp:1326-1333) produced
    (tests/golden/make_shuffle_golden.cpp -> shuffle_golden.json): bit exact, incl. seeds >= 2^32 and n > 65535
    (where libstdc++ switches from two-positions-per-draw to one)."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "shuffle_golden.json")))
    assert len(g["cases"]) == 55
    for c in g["cases"]:
        idx = orc.shuffle_indices(c["n"], c["seed"])
        assert sorted(idx.tolist()) == list(range(c["n"]))          # a permutation
        h = 1469598103934665603
        for v in idx.tolist():
            h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        assert str(h) == c["fnv1a"], (c["seed"], c["n"])
        assert idx[:16].tolist() == c["head"]
