"""Pin the CPU oracle against the reference's own Python implementation of the
photometric / geometric terms (fixtures: tests/golden/diffba_*.npz, generated from
/root/reference/representation/models/diff_ba.py by tests/golden/make_diffba_golden.py).

Keypoint terms (fixtures diffba_keypoints*.npz): the match-geometry factor and the projection Jacobians of the
reprojection factor.

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
    # a5, the error-only operator, against compute_geometry_error (:1995-2063; weight * mean Cauchy error over the inliers)
    wg = float(c["geo_term_weight"])
    e_only, n_only = orc.geo_error(c["R"], c["t"], bias, basis, c["code"], c["dmap"], c["mask"], loc, homo,
                                   float(c["scale"]), cams[0], float(c["depth_eps"]), loss, wg, prec=prec)
    assert e_only == pytest.approx(float(c["geo_error_only"]), rel=2e-5)
    assert n_only == pytest.approx(valid.sum())


KP_CASES = ["diffba_keypoints", "diffba_keypoints32"]


def _kp(c):
    from types import SimpleNamespace
    N, CS = int(c["N"]), int(c["CS"])
    fx, fy, cx, cy = (float(v) for v in c["intr"])
    cam = SimpleNamespace(fx=fx, fy=fy, cx=cx, cy=cy, w=int(c["W"]), h=int(c["H"]))
    homo = np.ascontiguousarray(c["homo"].T)
    return N, CS, cam, homo, np.arange(N)


@pytest.mark.parametrize("name", KP_CASES)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_match_geometry_rows_match_diffba(orc, name, prec):
    """f3 match-geometry factor against diff_ba.match_geometry_term (:891-951): residual match - keypoint, fair-loss
    weights and error, translation / code0 / scale0 columns, and the rotation columns except the one entry where the
    Python reference has a sign typo (compute_loc_3d_diff_jac_rel_pose :881 writes row 2 as [-Y, -X, 0]; the C++ kernel,
    match_geometry_factor_kernels.cpp:196-198, and this oracle have the skew-symmetric [Y, -X, 0]).  Mapper variant with
    T1 = identity (world pose0 == relative pose), matched depths as keyframe 1's bias; tracker variant with scale."""
    c = load(name)
    N, CS, cam, homo, loc = _kp(c)
    I3, z3 = np.eye(3), np.zeros(3)
    lp = float(c["mg_param_factor"]) * float(c["mean_sq"])
    A = c["mg_A"].reshape(N, 3, 7 + CS); diff = c["mg_diff"].reshape(N, 3)
    tol = 2e-5 if prec == "f32" else 5e-6           # (the fixture itself is fp32 arithmetic)
    o = orc.match_geom_jac_error(0, "fair", c["R"], c["t"], c["R"], c["t"], I3, z3, bias0=c["bias"],
                                 bias1=c["match_depths"], basis0=c["basis"], basis1=np.zeros((N, CS)), code0=c["code"],
                                 code1=np.zeros(CS), homo0=homo, homo1=np.ascontiguousarray(c["match_homo"].T), loc0=loc,
                                 loc1=loc, scale0=float(c["scale"]), scale1=1.0, loss_param=lp, weight=1.0, prec=prec,
                                 want_rows=True)
    J = o["J"]
    keep = np.ones((3, 3), bool); keep[2, 0] = False
    assert rel(o["r"], diff) < tol
    assert rel(J[..., 0:3], A[..., 3:6]) < tol                                  # translation
    assert rel(J[..., 3:6][:, keep], A[..., 0:3][:, keep]) < tol                # rotation (but the typo entry)
    assert np.allclose(J[:, 2, 3], -A[:, 2, 0], rtol=1e-4, atol=1e-7)           # ... which differs exactly by its sign
    assert np.array_equal(J[..., 6:12], -J[..., 0:6]) or rel(J[..., 6:12], -J[..., 0:6]) < 1e-6   # T1 = I
    assert rel(J[..., 12:12 + CS], A[..., 7:]) < tol                            # code0
    assert rel(J[..., 12 + 2 * CS], A[..., 6]) < tol                            # scale0
    assert o["error"] == pytest.approx(float(c["mg_err"].mean()), rel=1e-5)     # weight * mean fair error
    # tracker variant with scale (pose 6 + scale), depths given
    d0 = float(c["scale"]) * (c["bias"] + c["basis"] @ c["code"])
    ot = orc.match_geom_jac_error(3, "fair", c["R"], c["t"], dpts0=d0, dpts1=c["match_depths"], homo0=homo,
                                  homo1=np.ascontiguousarray(c["match_homo"].T), scale0=float(c["scale"]), loss_param=lp,
                                  weight=1.0, prec=prec, want_rows=True)
    assert rel(ot["J"][..., 0:3], A[..., 3:6]) < tol and rel(ot["J"][..., 6], A[..., 6]) < tol
    assert rel(ot["r"], diff) < tol
    # error-only operator against compute_match_geom_error (:1822-1850): weight * mean fair error
    e_only = orc.match_geom_error(2, "fair", c["R"], c["t"], dpts0=d0, dpts1=c["match_depths"], homo0=homo,
                                  homo1=np.ascontiguousarray(c["match_homo"].T), loss_param=lp,
                                  weight=float(c["mg_term_weight"]), prec=prec)
    assert e_only == pytest.approx(float(c["mg_error_only"]), rel=2e-5)


@pytest.mark.parametrize("name", KP_CASES)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_reprojection_jacobians_match_diffba(orc, name, prec):
    """f3 reprojection factor: the unweighted projection Jacobians wrt the relative pose, the code and the scale against
    diff_ba's jacobian_projected_2d_location_wrt_camera_pose / _wrt_src_depth (:322-386), residual = matched - projected
    (diff_ba.reproj_term itself cannot run: it reads attributes its constructor never sets)."""
    c = load(name)
    N, CS, cam, homo, loc = _kp(c)
    I3, z3 = np.eye(3), np.zeros(3)
    rng = np.random.default_rng(1)
    d0 = float(c["scale"]) * (c["bias"] + c["basis"] @ c["code"])
    X = d0 * (c["R"].astype(np.float64) @ c["homo"]) + c["t"][:, None]
    proj = np.stack([X[0] / X[2] * cam.fx + cam.cx, X[1] / X[2] * cam.fy + cam.cy], 1)
    matched = proj + rng.standard_normal((N, 2))
    o = orc.reproj_jac_error(c["R"], c["t"], c["R"], c["t"], I3, z3, c["bias"], c["basis"], c["code"], loc, homo,
                             matched, float(c["scale"]), cam, 1e-4, 2.0, 1.0, prec=prec, want_rows=True)
    Ju = o["J"] / o["sw"][:, :, None]
    Jp, Jd = c["proj_J_pose"], c["proj_J_depth"]                                # [N,2,6] = [rot, trans], [N,2,1]
    tol = 2e-5 if prec == "f32" else 5e-6
    assert rel(Ju[..., 0:3], Jp[..., 3:6]) < tol
    assert rel(Ju[..., 3:6], Jp[..., 0:3]) < tol
    assert rel(Ju[..., 12:12 + CS], Jd * (float(c["scale"]) * c["basis"])[:, None, :]) < tol
    assert rel(Ju[..., 12 + CS], Jd[..., 0] * (d0 / float(c["scale"]))[:, None]) < tol
    assert rel(o["r"] / o["sw"], matched - proj) < (1e-4 if prec == "f32" else 1e-6)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_se3_exp_and_retraction_match_reference_python(orc, prec):
    """a13 / a8 UpdateVariables: se3_exp (mapping_utils.h:316-346 <-> representation/utils/processing.py:596-633) on
    twists from 1e-4 to ~4 rad, and the left-multiplicative pose update + additive scale / code of
    DiffBundleAdjustment.update_variables (:830-842) -- oracle and the product's host helpers (sage_se3_exp,
    sage_pose_retract).  Order: the Python twist is [omega, v], the C++ delta [v, omega]."""
    from sage_slam_amd import capi
    c = load("diffba_retract")
    tol = 3e-6 if prec == "f32" else 1e-6                                       # the fixture is fp32 arithmetic
    for xi, E in zip(c["xi"], c["exp"]):
        R, t = orc.se3_exp(xi[:3], xi[3:], prec=prec)
        # (translation: the fp32 reference forms (1 - cos a)/a and (a - sin a)/a by cancellation: its own error is
        # ~eps/a relative to |t| -- 1e-4 at a = 1e-4 rad, 1e-6 above 0.1 rad)
        tol_t = 3e-6 + 2e-8 / float(np.linalg.norm(xi[:3]))
        assert rel(R.reshape(3, 3), E[:, :3]) < tol and rel(t, E[:, 3]) < tol_t
        Rp, tp = capi.se3_exp(xi[:3], xi[3:])
        assert rel(Rp.reshape(3, 3), E[:, :3]) < 3e-6 and rel(tp, E[:, 3]) < tol_t
    pose0 = np.concatenate([c["R0"].reshape(-1), c["t0"]]).astype(np.float32)
    for sol, upd in zip(c["sol"], c["updated"]):
        out = capi.pose_retract(pose0, np.concatenate([sol[3:6], sol[0:3]]))    # [v, omega]
        assert rel(out[:9], upd[:9]) < 3e-6 and rel(out[9:], upd[9:12]) < 1e-5
        assert upd[12] == pytest.approx(float(c["scale0"]) + sol[6], rel=1e-6)  # s <- s + ds (camera_tracker.cpp:504)
        assert np.allclose(upd[13:], c["code0"] + sol[7:], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_multilevel_photometric_error_matches_reference_python(orc, prec):
    """The multi-level coordinate rule and level weighting (a2/a3): compute_photo_error (diff_ba.py:1853-1939) over a
    3-level pyramid -- each level projected with its own intrinsics and sampled with grid_sample(align_corners=False)
    -- against the tracker error operator, whose levels all derive from the level-0 coordinates
    (u_l = (u_0 + .5) fx_l/fx_0 - .5, photometric_factor_kernels.cpp:142-160).  The fixture's level intrinsics are the
    ones that rule implies; masks all valid, so the per-level inlier counts (Python) equal the level-0 count (C++)."""
    c = load("diffba_photo_levels")
    L, N, FS, H, W = (int(c[k]) for k in ("L", "N", "FS", "H", "W"))
    fx, fy, cx, cy = (float(v) for v in c["intr0"])
    cams = orc.camera_pyramid([fx, fy, cx, cy, W, H], L, prec=prec)
    offs = [0]
    for cam in cams:
        offs.append(offs[-1] + int(cam[4]) * int(cam[5]))
    feat1 = np.concatenate([c[f"level{l}"].reshape(FS, -1) for l in range(L)], 1)
    feat0s = np.ascontiguousarray(c["src"].transpose(0, 2, 1))                   # [L, N, FS]
    e, n = orc.tracker_photo_error(c["R"], c["t"], np.ones((H, W)), c["depths"], np.ascontiguousarray(c["homo"].T),
                                   feat0s, feat1, np.array(offs[:-1], np.int32), cams, float(c["depth_eps"]),
                                   c["weights"], prec=prec)
    assert n == N
    assert e == pytest.approx(float(c["error"]), rel=3e-5 if prec == "f32" else 1e-5)


def test_gaussian_pyramid_matches_reference_python(orc):
    """f1 producer: the masked Gaussian pyramid (mapper.cpp:1384-1426 / mapping_utils.h) against
    DiffBundleAdjustment.generate_gaussian_pyramid (diff_ba.py:44-71) on a feature map with masked bands: same 3x3
    binomial weights, stride 2, renormalisation by the blurred mask."""
    c = load("diffba_pyramid")
    feat, mask, L = c["feat"], c["mask"], int(c["L"])
    FS, H, W = feat.shape
    cams = orc.camera_pyramid([0.9 * W, 0.9 * W, W / 2, H / 2, W, H], L)
    offs = [0]
    for cam in cams:
        offs.append(offs[-1] + int(cam[4]) * int(cam[5]))
    pyr, _ = orc.gaussian_pyramid_with_grad(feat, mask, L, np.array(offs[:-1], np.int32), offs[-1])
    for l in range(L):
        ref = c[f"level{l}"]
        got = pyr[:, offs[l]:offs[l + 1]].reshape(ref.shape)
        assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), l


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


# ---------------------------------------------------------------------------------------------------------------
# second batch (tests/golden/make_autograd_golden.py): world-frame pose blocks with T1 != I, code1 / scale1 columns,
# multi-level Jacobian weighting, level-0-inlier normalisation with partially masked frames, f4 matching core
# ---------------------------------------------------------------------------------------------------------------
def _rot(w):
    w = np.asarray(w, np.float64)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def _hat(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]], np.float64)


def _adjoint_inv(R1, t1):
    """Ad of T1^-1 = (R1^T, -R1^T t1) for twists ordered [v, omega]: a left perturbation delta0 of the WORLD pose T0 is
    the left perturbation Ad(T1^-1) delta0 of T10 = T1^-1 T0 (and -Ad(T1^-1) delta1 for T1)."""
    R = R1.T; t = -R1.T @ t1
    A = np.zeros((6, 6))
    A[:3, :3] = R; A[:3, 3:] = _hat(t) @ R; A[3:, 3:] = R
    return A


@pytest.mark.parametrize("name", CASES)
def test_world_frame_pose_blocks_for_general_T1_via_adjoint(orc, name):
    """a1 / a4 with T1 != I (photometric_factor_kernels.cpp:258-297, geometric_factor_kernels.cpp:671-679): the
    reference's Python rows are relative-pose rows; the C++ world-frame blocks must be those rows times Ad(T1^-1)
    (pose0) and its negative (pose1), for any decomposition T0 = T1 T10."""
    c = load(name)
    H, W, N, FS, CS, cams, lo, feat1, grad1, homo = setup(c)
    loc = c["loc"].astype(np.int64)
    bias = np.zeros(H * W, np.float32); bias[loc] = c["bias"]
    basis = np.zeros((H * W, CS), np.float32); basis[loc] = c["basis"]
    feat0 = np.zeros((FS, H * W), np.float32); feat0[:, loc] = c["src_feats"]
    R10, t10 = c["R"].astype(np.float64), c["t"].astype(np.float64)
    for tw, t1 in (((0.3, -0.2, 0.25), (0.4, -0.3, 0.2)), ((-1.1, 0.7, 0.4), (-2.0, 1.0, 0.5))):
        R1 = _rot(tw); t1 = np.array(t1)
        R0 = R1 @ R10; t0 = R1 @ t10 + t1
        Ad = _adjoint_inv(R1, t1)
        # photometric
        o = orc.photo_jac_error(R10, t10, R0, t0, R1, t1, bias, basis, c["code"], c["mask"], loc, homo, feat0, feat1,
                                grad1, lo, float(c["scale"]), cams, float(c["depth_eps"]), np.ones(1, np.float32),
                                prec="f64", want_rows=True)
        A = c["photo_A"].reshape(N, FS, 7 + CS).astype(np.float64)
        valid = c["photo_valid"].reshape(N) > 0.5
        Jrel = np.concatenate([A[..., 3:6], A[..., 0:3]], -1)                   # [v, omega]
        J = o["J"][0]
        assert rel(J[valid][..., 0:6], Jrel[valid] @ Ad) < 2e-5
        assert rel(J[valid][..., 6:12], -(Jrel[valid] @ Ad)) < 2e-5
        assert rel(J[valid][..., 12:12 + CS], A[valid][..., 7:]) < 2e-5          # code / scale columns do not see T1
        # geometric
        loss = float(c["geo_cauchy_factor"]) * float(c["geo_mean_sq"])
        dgrad = np.stack([c["dgx"], c["dgy"]], 0)
        basis1 = np.zeros((H, W, CS))
        g = orc.geo_jac_error(R10, t10, R0, t0, R1, t1, bias, basis, c["code"], c["dmap"], dgrad, basis1, c["mask"],
                              loc, homo, float(c["scale"]), 1.0, cams[0], float(c["depth_eps"]), loss, 1.0,
                              prec="f64", want_rows=True)
        Ag = c["geo_A"].astype(np.float64)
        Grel = np.concatenate([Ag[:, 3:6], Ag[:, 0:3]], -1)
        assert rel(g["J"][:, 0:6], Grel @ Ad) < 5e-6
        assert rel(g["J"][:, 6:12], -(Grel @ Ad)) < 5e-6


def _central_grad_hw(img):
    p = np.pad(img, ((1, 1), (1, 1)), mode="edge")
    return np.stack([0.5 * (p[1:-1, 2:] - p[1:-1, :-2]), 0.5 * (p[2:, 1:-1] - p[:-2, 1:-1])], 0)


@pytest.mark.parametrize("name", ["diffba_autograd_geo_allvalid", "diffba_autograd_geo_band"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_geometric_gradient_matches_reference_autograd(orc, name, prec):
    """dE/dx of the reference's compute_geometry_error (autograd, fp64) == -2 Atb of the geometric factor for ALL column
    blocks: both world poses (T1 != I), code0, code1 (-s1 * bilinear(basis1), geometric_factor_kernels.cpp:696),
    scale0, scale1 (-D/s1, :688), with the caller-side D1 / grad D1 precompute of geometric_factor.cpp:317-347 and the
    weight / n_inliers normalisation (:931-947) on a partially masked frame."""
    c = load(name)
    H, W, N, CS = (int(c[k]) for k in ("H", "W", "N", "CS"))
    fx, fy, cx, cy = (float(v) for v in c["intr"])
    cam = np.array([[fx, fy, cx, cy, W, H]])
    loc = c["loc"].astype(np.int64)
    bias0 = np.zeros(H * W); bias0[loc] = c["bias0"]
    basis0 = np.zeros((H * W, CS)); basis0[loc] = c["basis0"]
    R0, t0, R1, t1 = c["R0"], c["t0"], c["R1"], c["t1"]
    R10, t10 = R1.T @ R0, R1.T @ (t0 - t1)
    s0, s1 = float(c["s0"]), float(c["s1"])
    unscaled1 = c["bias1"] + c["basis1"] @ c["code1"]
    D1 = s1 * unscaled1
    gD1 = s1 * _central_grad_hw(unscaled1)
    loss = float(c["cauchy_factor"]) * float(c["mean_sq"])
    o = orc.geo_jac_error(R10, t10, R0, t0, R1, t1, bias0, basis0, c["code0"], D1, gD1, c["basis1"], c["mask"], loc,
                          np.ascontiguousarray(c["homo"].T), s0, s1, cam[0], float(c["depth_eps"]), loss,
                          float(c["weight"]), prec=prec)
    tol = 1e-6 if prec == "f64" else 2e-4
    assert o["error"] == pytest.approx(float(c["E"]), rel=1e-7 if prec == "f64" else 2e-5)
    g = -2.0 * o["Atb"].astype(np.float64)
    assert rel(g[0:6], c["g_pose0"]) < tol and rel(g[6:12], c["g_pose1"]) < tol
    assert rel(g[12:12 + CS], c["g_code0"]) < tol
    assert rel(g[12 + CS:12 + 2 * CS], c["g_code1"]) < tol                       # code1 columns
    assert g[12 + 2 * CS] == pytest.approx(float(c["g_s0"]), rel=tol)
    assert g[13 + 2 * CS] == pytest.approx(float(c["g_s1"]), rel=tol)            # scale1 column
    if "band" in name:
        assert 0 < o["num_inliers"] < N                                          # the band really masks samples


@pytest.mark.parametrize("name", ["diffba_autograd_photo_allvalid", "diffba_autograd_photo_band"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_multilevel_photometric_gradient_matches_reference_autograd(orc, name, prec):
    """dE/dx of the reference's 3-level compute_photo_error (autograd, fp64) == -2 Atb of a1: the level weights inside
    the Jacobian reduction (photometric_factor_kernels.cpp:1143-1150), the 1/n_inliers(level 0) normalisation (:1139)
    with a masked band, the level coordinate rule in the Jacobian (fx_l, :241-245), both world-pose blocks with
    T1 != I (:258-297), code and scale columns (:324-335)."""
    c = load(name)
    H, W, N, CS, FS, L = (int(c[k]) for k in ("H", "W", "N", "CS", "FS", "L"))
    fx, fy, cx, cy = (float(v) for v in c["intr0"])
    cams = orc.camera_pyramid([fx, fy, cx, cy, W, H], L, prec=prec)
    offs = [0]
    for cam in cams:
        offs.append(offs[-1] + int(cam[4]) * int(cam[5]))
    feat0 = np.concatenate([c[f"feat0_level{l}"].reshape(FS, -1) for l in range(L)], 1)
    feat1 = np.concatenate([c[f"feat1_level{l}"].reshape(FS, -1) for l in range(L)], 1)
    grads = []
    for l in range(L):
        lv = c[f"feat1_level{l}"]
        grads.append(np.stack([_central_grad_hw(lv[ch]) for ch in range(FS)], 1).reshape(2, FS, -1))   # [2,FS,HW_l]
    grad1 = np.concatenate(grads, 2)
    loc = c["loc"].astype(np.int64)
    bias0 = np.zeros(H * W); bias0[loc] = c["bias0"]
    basis0 = np.zeros((H * W, CS)); basis0[loc] = c["basis0"]
    R0, t0, R1, t1 = c["R0"], c["t0"], c["R1"], c["t1"]
    R10, t10 = R1.T @ R0, R1.T @ (t0 - t1)
    o = orc.photo_jac_error(R10, t10, R0, t0, R1, t1, bias0, basis0, c["code0"], c["mask"], loc,
                            np.ascontiguousarray(c["homo"].T), feat0, feat1, grad1, np.array(offs[:-1], np.int32),
                            float(c["s0"]), cams, float(c["depth_eps"]), c["weights"], prec=prec)
    tol = 2e-6 if prec == "f64" else 3e-4
    assert o["error"] == pytest.approx(float(c["E"]), rel=1e-6 if prec == "f64" else 3e-5)
    g = -2.0 * o["Atb"].astype(np.float64)
    assert rel(g[0:6], c["g_pose0"]) < tol and rel(g[6:12], c["g_pose1"]) < tol
    # (the code / scale gradients are 3e-4 of the pose gradient's magnitude here: sums that cancel -> fp32 noise 1e-3)
    assert rel(g[12:12 + CS], c["g_code0"]) < (tol if prec == "f64" else 2e-3)
    assert g[12 + CS] == pytest.approx(float(c["g_s0"]), rel=tol if prec == "f64" else 2e-3)
    if "band" in name:
        assert 0 < o["num_inliers"] < N


def test_cycle_match_matches_the_reference_tensor_expression(orc):
    """f4: the oracle's matching core against the literal torch expression of match_geometry_factor.cpp:62-97 /
    camera_tracker.cpp:608-633 (evaluated by the same ATen ops; torch_cycle_match.npz), incl. exactly tied descriptors:
    torch::max returns the FIRST maximal index."""
    z = np.load(os.path.join(GOLD, "torch_cycle_match.npz"))
    for k in ("c0", "c1", "c2"):
        d0, d1, kp = z[f"{k}_desc0"], z[f"{k}_desc1"], z[f"{k}_kp"]
        C_, H, W = d0.shape
        raw, cyc, flags = orc.cycle_match(d0, d1, kp, H, W, float(z[f"{k}_thresh"]))
        assert np.array_equal(raw, z[f"{k}_raw"]), k
        assert np.array_equal(cyc, z[f"{k}_cyc"]), k
        assert np.array_equal(flags, z[f"{k}_inlier"]), k
        assert np.array_equal(np.nonzero(flags)[0], z[f"{k}_inlier_idx"]), k
