"""Finite-difference checks of the oracle's analytic Jacobians (fp64 build).

These cover the parts the diff_ba fixtures cannot pin: the world-frame
two-pose blocks dX/dT0 = R1^T E(Xw), dX/dT1 = -R1^T E(Xw) with non-trivial R0, R1
(photometric_factor_kernels.cpp:258-297), the geometric code1 / scale1 columns
(geometric_factor_kernels.cpp:688,696) and the per-level half-pixel coordinate mapping
(:142-144).  This is the relationship the reference's own commented-out checks use
(core/gtsam/photometric_factor.cpp:124-143): r(x (+) d) = r(x) - J d + O(d^2).

The Jacobians use *sampled gradient images*, not the derivative of the bilinear
interpolant, so the check uses images that are linear ramps per level (central-difference
gradient == true gradient, bilinear interpolation exact).
"""
import numpy as np
import pytest

from sage_slam_amd import synth


def rot(w):
    return synth.so3_exp(np.asarray(w, dtype=np.float64))


def retract(R, t, d):
    """T <- exp([v, w]) * T  (left multiplication, [trans, rot] order; gtsam_traits.h:45-70)."""
    from oracle import oracle as orc
    dR, dt = orc.se3_exp(d[3:6], d[0:3], prec="f64")
    return dR @ R, dR @ t + dt


def ramp_pyramid(rng, cams, FS):
    """per level: f_c(x,y) = a*x + b*y + c ; grad = (a, b) constant."""
    pyr, gx, gy = [], [], []
    for cam in cams:
        Wl, Hl = int(cam.w), int(cam.h)
        yy, xx = np.meshgrid(np.arange(Hl, dtype=np.float64), np.arange(Wl, dtype=np.float64), indexing="ij")
        a = rng.uniform(-0.05, 0.05, (FS, 1, 1)); b = rng.uniform(-0.05, 0.05, (FS, 1, 1))
        c = rng.uniform(-0.5, 0.5, (FS, 1, 1))
        pyr.append((a * xx + b * yy + c).reshape(FS, -1))
        gx.append(np.broadcast_to(a, (FS, Hl, Wl)).reshape(FS, -1))
        gy.append(np.broadcast_to(b, (FS, Hl, Wl)).reshape(FS, -1))
    return np.concatenate(pyr, 1), np.stack([np.concatenate(gx, 1), np.concatenate(gy, 1)], 0)


@pytest.fixture(scope="module")
def scene():
    rng = np.random.default_rng(3)
    H, W, L, FS, CS, N = 32, 40, 3, 4, 8, 40
    cams = synth.camera_pyramid(synth.Camera(36.0, 34.0, 19.6, 16.3, W, H), L)
    lo, P = synth.level_offsets_of(cams)
    feat0, _ = ramp_pyramid(rng, cams, FS)
    feat1, grad1 = ramp_pyramid(rng, cams, FS)
    # central region samples so every tap at every level stays interior
    ys = rng.integers(10, H - 10, N); xs = rng.integers(10, W - 10, N)
    loc = (ys * W + xs).astype(np.int64)
    homo = np.stack([(xs - cams[0].cx) / cams[0].fx, (ys - cams[0].cy) / cams[0].fy, np.ones(N)], 1)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    def lin():
        return (rng.uniform(-2e-3, 2e-3) * xx + rng.uniform(-2e-3, 2e-3) * yy).reshape(-1)
    bias0 = 1.0 + lin(); basis0 = np.stack([0.05 * rng.standard_normal() + lin() for _ in range(CS)], 1)
    bias1 = 1.0 + lin(); basis1 = np.stack([0.05 * rng.standard_normal() + lin() for _ in range(CS)], 1)
    x = dict(R0=rot([0.3, -0.2, 0.1]), t0=np.array([0.4, -0.3, 0.2]),
             code0=0.1 * rng.standard_normal(CS), code1=0.1 * rng.standard_normal(CS), s0=1.3, s1=0.9)
    # T1 = T0 * small relative motion, so the projections stay interior
    dR = rot([0.01, -0.015, 0.02]); dt = np.array([0.02, -0.01, 0.015])
    x["R1"] = x["R0"] @ dR; x["t1"] = x["R0"] @ dt + x["t0"]
    return dict(H=H, W=W, L=L, FS=FS, CS=CS, N=N, cams=cams, lo=lo, P=P, feat0=feat0, feat1=feat1,
                grad1=grad1, loc=loc, homo=homo, bias0=bias0, basis0=basis0, bias1=bias1, basis1=basis1,
                mask=np.ones((H, W)), x=x, w=np.array([10.0, 9.0, 8.0]))


def photo_rows(orc, s, x):
    R10 = x["R1"].T @ x["R0"]; t10 = x["R1"].T @ (x["t0"] - x["t1"])
    return orc.photo_jac_error(R10, t10, x["R0"], x["t0"], x["R1"], x["t1"], s["bias0"], s["basis0"],
                               x["code0"], s["mask"], s["loc"], s["homo"], s["feat0"], s["feat1"], s["grad1"],
                               s["lo"], x["s0"], s["cams"], 1e-4, s["w"], prec="f64", want_rows=True)


def geo_rows(orc, s, x, loss=1e12):
    R10 = x["R1"].T @ x["R0"]; t10 = x["R1"].T @ (x["t0"] - x["t1"])
    H, W, CS = s["H"], s["W"], s["CS"]
    un = (s["bias1"] + s["basis1"] @ x["code1"]).reshape(1, H, W)
    g = orc.spatial_grad(un, prec="f64")[:, 0]
    # ramps: replicate-pad central differences are wrong on the border only; samples are interior
    return orc.geo_jac_error(R10, t10, x["R0"], x["t0"], x["R1"], x["t1"], s["bias0"], s["basis0"], x["code0"],
                             x["s1"] * un[0], x["s1"] * g, s["basis1"].reshape(H, W, CS), s["mask"],
                             s["loc"], s["homo"], x["s0"], x["s1"], s["cams"][0], 1e-4, loss, 1.0,
                             prec="f64", want_rows=True)


def perturb(x, name, d, CS):
    y = dict(x)
    if name == "pose0":
        y["R0"], y["t0"] = retract(x["R0"], x["t0"], d)
    elif name == "pose1":
        y["R1"], y["t1"] = retract(x["R1"], x["t1"], d)
    elif name in ("code0", "code1"):
        y[name] = x[name] + d
    else:
        y[name] = x[name] + d[0]
    return y


def test_photo_two_pose_code_scale_jacobian_fd(orc, scene):
    s, x, CS = scene, scene["x"], scene["CS"]
    base = photo_rows(orc, s, x)
    assert base["num_inliers"] == s["N"]
    J = base["J"].reshape(-1, 13 + CS); r0 = base["r"].reshape(-1)
    cols = {"pose0": slice(0, 6), "pose1": slice(6, 12), "code0": slice(12, 12 + CS), "s0": slice(12 + CS, 13 + CS)}
    rng = np.random.default_rng(0)
    for name, sl in cols.items():
        n = sl.stop - sl.start
        d = 1e-6 * rng.standard_normal(n)
        r1 = photo_rows(orc, s, perturb(x, name, d, CS))["r"].reshape(-1)
        lin = -J[:, sl] @ d                      # r = f0 - f1, J = +df1/dx
        err = np.linalg.norm((r1 - r0) - lin) / np.linalg.norm(lin)
        assert err < 1e-4, (name, err)


def test_geo_all_columns_jacobian_fd(orc, scene):
    s, x, CS = scene, scene["x"], scene["CS"]
    loss = 1e12
    base = geo_rows(orc, s, x, loss)
    assert base["num_inliers"] == s["N"]
    sc = np.sqrt(loss)                           # sqrt_w = 1/sqrt(rho^2 + c) ~= 1/sqrt(c)
    J = base["J"] * sc; r0 = base["r"] * sc
    cols = {"pose0": slice(0, 6), "pose1": slice(6, 12), "code0": slice(12, 12 + CS),
            "code1": slice(12 + CS, 12 + 2 * CS), "s0": slice(12 + 2 * CS, 13 + 2 * CS),
            "s1": slice(13 + 2 * CS, 14 + 2 * CS)}
    rng = np.random.default_rng(1)
    for name, sl in cols.items():
        n = sl.stop - sl.start
        d = 1e-6 * rng.standard_normal(n)
        r1 = geo_rows(orc, s, perturb(x, name, d, CS), loss)["r"] * sc
        lin = -J[:, sl] @ d                      # row = -d rho/dx, residual = +rho
        err = np.linalg.norm((r1 - r0) - lin) / np.linalg.norm(lin)
        assert err < 1e-4, (name, err)


def test_photo_error_only_equals_jac_path(orc, scene):
    s, x = scene, scene["x"]
    base = photo_rows(orc, s, x)
    R10 = x["R1"].T @ x["R0"]; t10 = x["R1"].T @ (x["t0"] - x["t1"])
    e, n = orc.photo_error(R10, t10, s["bias0"], s["basis0"], x["code0"], s["mask"], s["loc"], s["homo"],
                           s["feat0"], s["feat1"], s["lo"], x["s0"], s["cams"], 1e-4, s["w"], prec="f64")
    assert n == base["num_inliers"]
    assert e == pytest.approx(base["error"], rel=1e-9)


def test_error_model_matches_reference_relationship(orc, scene):
    """err(x+d) - err(x) ~= d^T AtA d - 2 Atb^T d  (photometric_factor.cpp:124-143)."""
    s, x, CS = scene, scene["x"], scene["CS"]
    base = photo_rows(orc, s, x)
    rng = np.random.default_rng(2)
    d = 1e-4 * rng.standard_normal(13 + CS)
    y = perturb(perturb(perturb(perturb(x, "pose0", d[0:6], CS), "pose1", d[6:12], CS),
                        "code0", d[12:12 + CS], CS), "s0", d[12 + CS:], CS)
    e1 = photo_rows(orc, s, y)["error"]
    pred = d @ base["AtA"] @ d - 2 * base["Atb"] @ d
    assert (e1 - base["error"]) == pytest.approx(pred, rel=2e-3)


def test_zero_overlap_fallback(orc, scene):
    """no inliers -> error = 10*sum(w), AtA = Atb = 0 (photometric...:1156-1161; geometric...:944)."""
    s, x, CS = scene, dict(scene["x"]), scene["CS"]
    x["t1"] = x["t1"] + x["R1"] @ np.array([0, 0, 50.0])     # every point ends up behind camera 1
    o = photo_rows(orc, s, x)
    assert o["num_inliers"] == 0 and o["error"] == pytest.approx(10 * s["w"].sum())
    assert not o["AtA"].any() and not o["Atb"].any()
    g = geo_rows(orc, s, x, 0.03)
    assert g["num_inliers"] == 0 and g["error"] == pytest.approx(10.0) and not g["AtA"].any()


def test_reprojection_factor_jacobian_and_reduction(orc):
    """f3 reprojection factor (cuda/reprojection_factor_kernels.cpp:27-213, :468-531): the oracle's unweighted rows
    (J / sqrt fair weight) against central differences of the projection  pi(T1^-1 T0 d x~)  written independently in
    numpy, for both world-frame poses (left retraction), the code and the scale; the fair-loss weights / error against
    their closed forms; AtA/Atb against the weighted rows; the no-inlier fallback; the tracker variant's 6-dof rows
    against the mapper rows of an identity-reference configuration."""
    rng = np.random.default_rng(12)
    H, W, CS, N = 48, 64, 8, 37
    cam = synth.Camera(58.0, 56.0, 31.3, 23.9, W, H)
    bias0 = 1.0 + 0.1 * rng.random(H * W); basis0 = 0.05 * rng.standard_normal((H * W, CS))
    code0 = 0.3 * rng.standard_normal(CS); s0 = 1.17
    ys = rng.integers(4, H - 4, N); xs = rng.integers(4, W - 4, N)
    loc = (ys * W + xs).astype(np.int32)
    homo = np.stack([(xs - cam.cx) / cam.fx, (ys - cam.cy) / cam.fy, np.ones(N)], 1)
    R0, t0 = rot([0.05, -0.08, 0.03]), np.array([0.02, -0.01, 0.03])
    R1, t1 = rot([-0.04, 0.06, 0.09]), np.array([-0.05, 0.02, -0.04])
    eps, c, wgt = 1e-4, 2.5, 0.7

    def project(R0_, t0_, R1_, t1_, code_, s_):
        d = s_ * (bias0[loc] + basis0[loc] @ code_)
        Xw = (R0_ @ (d[:, None] * homo).T).T + t0_
        X = (R1_.T @ (Xw - t1_).T).T
        return np.stack([X[:, 0] / X[:, 2] * cam.fx + cam.cx, X[:, 1] / X[:, 2] * cam.fy + cam.cy], 1), X

    p0, X = project(R0, t0, R1, t1, code0, s0)
    matched = p0 + rng.normal(0, 1.5, p0.shape)
    R10, t10 = R1.T @ R0, R1.T @ (t0 - t1)
    o = orc.reproj_jac_error(R10, t10, R0, t0, R1, t1, bias0, basis0, code0, loc, homo, matched, s0, cam, eps, c, wgt,
                             prec="f64", want_rows=True)
    D = 13 + CS
    # closed forms of the fair loss (:79-91)
    diff = matched - p0
    nrm = np.abs(diff) / np.sqrt(c)
    assert np.allclose(o["sw"], np.sqrt(1.0 / (c * (1 + nrm))), rtol=1e-12)
    assert np.isclose(o["error"], wgt * np.sum(2 * (nrm.sum(1) - np.log1p(nrm).sum(1))) / N, rtol=1e-12)
    assert o["num_inliers"] == N
    Ju = o["J"] / o["sw"][:, :, None]                       # unweighted d(proj)/d(theta)
    # central differences of the projection
    h = 1e-6
    def fd(f):
        return (f(+h) - f(-h)) / (2 * h)
    for j in range(6):
        e = np.zeros(6); e[j] = 1.0
        g0 = fd(lambda s: project(*retract(R0, t0, s * e), R1, t1, code0, s0)[0])
        g1 = fd(lambda s: project(R0, t0, *retract(R1, t1, s * e), code0, s0)[0])
        assert np.allclose(Ju[:, :, j], g0, rtol=1e-6, atol=1e-7), j
        assert np.allclose(Ju[:, :, 6 + j], g1, rtol=1e-6, atol=1e-7), j
    for i in range(CS):
        e = np.zeros(CS); e[i] = 1.0
        assert np.allclose(Ju[:, :, 12 + i], fd(lambda s: project(R0, t0, R1, t1, code0 + s * e, s0)[0]), rtol=1e-6, atol=1e-8)
    assert np.allclose(Ju[:, :, 12 + CS], fd(lambda s: project(R0, t0, R1, t1, code0, s0 + s)[0]), rtol=1e-6, atol=1e-8)
    # reduction (:507-520)
    Jw = o["J"].reshape(-1, D); rw = o["r"].reshape(-1)
    assert np.allclose(o["AtA"], (wgt / N) * Jw.T @ Jw, rtol=1e-12, atol=1e-14)
    assert np.allclose(o["Atb"], (wgt / N) * Jw.T @ rw, rtol=1e-12, atol=1e-14)
    assert np.allclose(rw.reshape(-1, 2), o["sw"] * diff, rtol=1e-12)
    e_only, n_only = orc.reproj_error(R10, t10, bias0, basis0, code0, loc, homo, matched, s0, cam, eps, c, wgt, prec="f64")
    assert np.isclose(e_only, o["error"], rtol=1e-13) and n_only == N
    # behind the camera: no inliers -> 10*weight, zeros (:522-527)
    ob = orc.reproj_jac_error(R10, t10 - np.array([0, 0, 50.0]), R0, t0, R1, t1, bias0, basis0, code0, loc, homo, matched,
                              s0, cam, eps, c, wgt, prec="f64")
    assert ob["num_inliers"] == 0 and ob["error"] == pytest.approx(10 * wgt) and not ob["AtA"].any() and not ob["Atb"].any()
    # tracker variant (:288-366): with T0 = identity the relative-pose rows are MINUS the pose-1 block's ... no: for a
    # left perturbation of the relative pose T10 the rows equal d(proj)/d(T10); check against central differences
    d0 = s0 * (bias0[loc] + basis0[loc] @ code0)
    ot = orc.tracker_reproj_jac_error(R10, t10, d0, homo, matched, cam, eps, c, wgt, prec="f64")
    def proj_rel(R_, t_):
        Xr = (R_ @ (d0[:, None] * homo).T).T + t_
        return np.stack([Xr[:, 0] / Xr[:, 2] * cam.fx + cam.cx, Xr[:, 1] / Xr[:, 2] * cam.fy + cam.cy], 1)
    Jt = np.zeros((N, 2, 6))
    for j in range(6):
        e = np.zeros(6); e[j] = 1.0
        Jt[:, :, j] = fd(lambda s: proj_rel(*retract(R10, t10, s * e)))
    Jtw = (o["sw"][:, :, None] * Jt).reshape(-1, 6)
    assert np.allclose(ot["AtA"], (wgt / N) * Jtw.T @ Jtw, rtol=1e-6, atol=1e-10)
    assert np.allclose(ot["Atb"], (wgt / N) * Jtw.T @ rw, rtol=1e-6, atol=1e-10)
    et, nt = orc.tracker_reproj_error(R10, t10, d0, homo, matched, cam, eps, c, wgt, prec="f64")
    assert np.isclose(et, o["error"], rtol=1e-12) and nt == N


def test_match_geometry_factor_family(orc):
    """f3 match-geometry factors (cuda/match_geometry_factor_kernels.cpp): for the mapper factor with every loss
    ("fair", "L2", "huber", "unbiased"), the loop factor and the two tracker variants, the oracle's unweighted rows
    against central differences of  X0_in_1 - X1_matched  written independently in numpy (both poses by left
    retraction, both codes, both scales -- the "unbiased" variant rescales both depths by s/(s0+s1), which couples
    the two scale columns), the loss weights/errors against closed forms, AtA/Atb against the weighted rows."""
    rng = np.random.default_rng(21)
    HW, CS, N = 600, 8, 29
    bias0 = 1.0 + 0.2 * rng.random(HW); bias1 = 1.1 + 0.2 * rng.random(HW)
    basis0 = 0.05 * rng.standard_normal((HW, CS)); basis1 = 0.05 * rng.standard_normal((HW, CS))
    code0 = 0.3 * rng.standard_normal(CS); code1 = 0.3 * rng.standard_normal(CS)
    s0, s1 = 1.2, 0.9
    loc0 = rng.integers(0, HW, N).astype(np.int32); loc1 = rng.integers(0, HW, N).astype(np.int32)
    homo0 = np.concatenate([rng.uniform(-0.5, 0.5, (N, 2)), np.ones((N, 1))], 1)
    homo1 = np.concatenate([rng.uniform(-0.5, 0.5, (N, 2)), np.ones((N, 1))], 1)
    R0, t0 = rot([0.05, -0.08, 0.03]), np.array([0.02, -0.01, 0.03])
    R1, t1 = rot([-0.04, 0.06, 0.09]), np.array([-0.05, 0.02, -0.04])
    R10, t10 = R1.T @ R0, R1.T @ (t0 - t1)
    c, wgt, h = 0.04, 0.6, 1e-6
    fd = lambda f: (f(+h) - f(-h)) / (2 * h)

    def g(R0_, t0_, R1_, t1_, c0, c1, a0, a1, unbiased):      # X0_in_1 - X1_matched, [N,3]
        d0 = (bias0[loc0] + basis0[loc0] @ c0); d1 = (bias1[loc1] + basis1[loc1] @ c1)
        if unbiased:
            d0, d1 = d0 * a0 / (a0 + a1), d1 * a1 / (a0 + a1)
        else:
            d0, d1 = d0 * a0, d1 * a1
        Xw = (R0_ @ (d0[:, None] * homo0).T).T + t0_
        return (R1_.T @ (Xw - t1_).T).T - d1[:, None] * homo1

    for loss in ("fair", "L2", "huber", "unbiased"):
        ub = loss == "unbiased"
        o = orc.match_geom_jac_error(0, loss, R10, t10, R0, t0, R1, t1, bias0, bias1, basis0, basis1, code0, code1,
                                     homo0=homo0, homo1=homo1, loc0=loc0, loc1=loc1, scale0=s0, scale1=s1, loss_param=c,
                                     weight=wgt, prec="f64", want_rows=True)
        D = 14 + 2 * CS
        diff = -g(R0, t0, R1, t1, code0, code1, s0, s1, ub)
        if loss == "L2":
            sw_ref = np.ones_like(diff); e_ref = (diff ** 2).sum()
        elif loss == "huber":
            sq = diff ** 2
            sw_ref = np.minimum(1.0, np.sqrt(c / sq)); e_ref = np.where(sq <= c, sq, 2 * np.sqrt(c * sq) - c).sum()
        else:
            nrm = np.abs(diff) / np.sqrt(c)
            sw_ref = np.sqrt(1 / (c * (1 + nrm))); e_ref = 2 * (nrm - np.log1p(nrm)).sum()
        assert np.allclose(o["sw"], sw_ref, rtol=1e-12), loss
        assert np.isclose(o["error"], wgt * e_ref / N, rtol=1e-12), loss
        if loss == "huber":
            assert (sw_ref < 1).any() and (sw_ref == 1).any()     # both branches exercised
        Ju = o["J"] / o["sw"][:, :, None]
        for j in range(6):
            e = np.zeros(6); e[j] = 1.0
            assert np.allclose(Ju[:, :, j], fd(lambda s: g(*retract(R0, t0, s * e), R1, t1, code0, code1, s0, s1, ub)), rtol=1e-6, atol=1e-8)
            assert np.allclose(Ju[:, :, 6 + j], fd(lambda s: g(R0, t0, *retract(R1, t1, s * e), code0, code1, s0, s1, ub)), rtol=1e-6, atol=1e-8)
        for i in range(CS):
            e = np.zeros(CS); e[i] = 1.0
            assert np.allclose(Ju[:, :, 12 + i], fd(lambda s: g(R0, t0, R1, t1, code0 + s * e, code1, s0, s1, ub)), rtol=1e-6, atol=1e-9)
            assert np.allclose(Ju[:, :, 12 + CS + i], fd(lambda s: g(R0, t0, R1, t1, code0, code1 + s * e, s0, s1, ub)), rtol=1e-6, atol=1e-9)
        assert np.allclose(Ju[:, :, 12 + 2 * CS], fd(lambda s: g(R0, t0, R1, t1, code0, code1, s0 + s, s1, ub)), rtol=1e-6, atol=1e-9), loss
        assert np.allclose(Ju[:, :, 13 + 2 * CS], fd(lambda s: g(R0, t0, R1, t1, code0, code1, s0, s1 + s, ub)), rtol=1e-6, atol=1e-9), loss
        Jw = o["J"].reshape(-1, D); rw = o["r"].reshape(-1)
        assert np.allclose(rw.reshape(-1, 3), o["sw"] * diff, rtol=1e-12)
        assert np.allclose(o["AtA"], (wgt / N) * Jw.T @ Jw, rtol=1e-12, atol=1e-14)
        assert np.allclose(o["Atb"], (wgt / N) * Jw.T @ rw, rtol=1e-12, atol=1e-14)
        eo = orc.match_geom_error(0, loss, R10, t10, bias0, bias1, basis0, basis1, code0, code1, homo0=homo0, homo1=homo1,
                                  loc0=loc0, loc1=loc1, scale0=s0, scale1=s1, loss_param=c, weight=wgt, prec="f64")
        assert np.isclose(eo, o["error"], rtol=1e-13)

    # loop factor: unscaled depths handed over, columns [pose0 pose1 scale0 scale1]
    u0 = bias0[loc0] + basis0[loc0] @ code0; u1 = bias1[loc1] + basis1[loc1] @ code1
    ol = orc.match_geom_jac_error(1, "fair", R10, t10, R0, t0, R1, t1, dpts0=u0, dpts1=u1, homo0=homo0, homo1=homo1,
                                  scale0=s0, scale1=s1, loss_param=c, weight=wgt, prec="f64", want_rows=True)
    om = orc.match_geom_jac_error(0, "fair", R10, t10, R0, t0, R1, t1, bias0, bias1, basis0, basis1, code0, code1,
                                  homo0=homo0, homo1=homo1, loc0=loc0, loc1=loc1, scale0=s0, scale1=s1, loss_param=c,
                                  weight=wgt, prec="f64", want_rows=True)
    assert np.allclose(ol["J"][:, :, :12], om["J"][:, :, :12], rtol=1e-12)
    assert np.allclose(ol["J"][:, :, 12:14], om["J"][:, :, 12 + 2 * CS:], rtol=1e-12)
    assert np.isclose(ol["error"], om["error"], rtol=1e-13)
    assert np.isclose(orc.match_geom_error(1, "fair", R10, t10, dpts0=u0, dpts1=u1, homo0=homo0, homo1=homo1, scale0=s0,
                                           scale1=s1, loss_param=c, weight=wgt, prec="f64"), ol["error"], rtol=1e-13)
    # tracker variants: relative pose rows E(X0_in_1) (left perturbation of T10), optional scale column
    d0 = s0 * u0; d1 = s1 * u1
    def g_rel(R_, t_, a0):
        return (R_ @ ((d0 * a0 / s0)[:, None] * homo0).T).T + t_ - d1[:, None] * homo1
    for mode in (2, 3):
        ot = orc.match_geom_jac_error(mode, "fair", R10, t10, dpts0=d0, dpts1=d1, homo0=homo0, homo1=homo1, scale0=s0,
                                      loss_param=c, weight=wgt, prec="f64", want_rows=True)
        Ju = ot["J"] / ot["sw"][:, :, None]
        for j in range(6):
            e = np.zeros(6); e[j] = 1.0
            assert np.allclose(Ju[:, :, j], fd(lambda s: g_rel(*retract(R10, t10, s * e), s0)), rtol=1e-6, atol=1e-8)
        if mode == 3:
            assert np.allclose(Ju[:, :, 6], fd(lambda s: g_rel(R10, t10, s0 + s)), rtol=1e-6, atol=1e-9)
        assert np.isclose(ot["error"], om["error"], rtol=1e-12)
