"""The drop-in boundary EXECUTED.  integration/compile_check/_bin/adapter_run is built in the build
container (sage_slam_amd/build.py:build_adapter_run -- it needs the reference's headers) from

    integration/sage_adapter.cpp              the replacement TU for cuda/{photometric,geometric}_factor_kernels.cpp      (7 df:: entry points)
    integration/sage_adapter_keypoints.cpp    the replacement TU for cuda/{reprojection,match_geometry}_factor_kernels.cpp (11 entry points)
    integration/compile_check/adapter_run.cpp a driver that includes the reference's OWN headers, builds at::Tensor arguments
                                              with PyTorch-ROCm's libtorch the way core/gtsam/*_factor.cpp and
                                              core/system/camera_tracker.cpp do, and calls all eighteen functions

and travels to the GPU box with the tree.  Here: a synthetic edge / keypoint set is written to a file, the binary runs on the
MI355X, and everything it got back through the reference's function signatures -- AtA, Atb, error -- is compared with the CPU
oracle (same bars as the C-ABI parity tests: rel-L2 2e-5, error rel 1e-5), together with the boundary's semantics: fresh device
output tensors of the factor's shape whatever the caller passed in, `error` as a host float, the zero-overlap fallback values,
the robust_loss_type string dispatch (an unknown string = no kernel in the reference: zeros), int64 locations converted."""
import os
import struct
import subprocess

import numpy as np
import pytest

from sage_slam_amd import synth
from tests.helpers import oracle_geo, oracle_photo, presample_source, rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "integration", "compile_check", "_bin", "adapter_run")
TOL_H = 2e-5
CS, FS = 32, 16            # DF_CODE_SIZE / DF_FEAT_SIZE the binary was compiled for (the reference's defaults)


def _write(path, arrays):
    with open(path, "wb") as f:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            dt = {np.dtype(np.float32): 0, np.dtype(np.int64): 1, np.dtype(np.int32): 2}[a.dtype]
            nb = name.encode()
            f.write(struct.pack("<I", len(nb))); f.write(nb); f.write(struct.pack("<BI", dt, a.ndim))
            f.write(struct.pack(f"<{a.ndim}q", *a.shape)); f.write(a.tobytes())


def _read(path):
    raw = open(path, "rb").read()
    o, out = 0, {}
    while o < len(raw):
        nl = struct.unpack_from("<I", raw, o)[0]; o += 4
        name = raw[o:o + nl].decode(); o += nl
        dt, nd = struct.unpack_from("<BI", raw, o); o += 5
        dims = struct.unpack_from(f"<{nd}q", raw, o); o += 8 * nd
        ty = (np.float32, np.int64, np.int32)[dt]
        n = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(raw, ty, n, o).reshape(dims).copy(); o += n * np.dtype(ty).itemsize
    return out


def _sc(v):
    return np.array([v], np.float32)


def _run(tmp_path, arrays):
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    _write(inp, arrays)
    r = subprocess.run([BIN, inp, outp], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    return _read(outp)


needs_bin = pytest.mark.skipif(not os.path.exists(BIN), reason="adapter_run is built in the build container (needs the "
                                                               "reference's headers) and travels with the tree")


@needs_bin
def test_dense_factor_adapter_executes_behind_the_reference_headers(tmp_path, orc):
    """the 7 entry points of photometric_factor_kernels.h:9-70 / geometric_factor_kernels.h:18-48 through sage_adapter.cpp"""
    import torch
    assert torch.cuda.is_available()
    w = synth.make_window(K=2, H=64, W=80, FS=FS, CS=CS, L=4, n_samples=3072, seed=61)      # BASELINE config 1 sizes
    a, b = w.keyframes[0], w.keyframes[1]
    R10, t10 = synth.relative_pose(a.R, a.t, b.R, b.t)
    D1, g1 = synth.depth_and_grad(b, w.H, w.W)
    feat0s = presample_source(orc, w, a)
    dpts0 = (np.float32(a.scale) * (a.bias + a.basis @ a.code))[a.loc1d].astype(np.float32)
    cam = w.cams[0]
    f32 = lambda x: np.ascontiguousarray(x, np.float32)
    arrays = dict(
        cam=f32([cam.fx, cam.fy, cam.cx, cam.cy, cam.w, cam.h]), levels=_sc(w.L), eps=_sc(w.eps),
        R10=f32(R10).reshape(3, 3), t10=f32(t10), R0=f32(a.R).reshape(3, 3), t0=f32(a.t), R1=f32(b.R).reshape(3, 3), t1=f32(b.t),
        bias_0=f32(a.bias), basis_0=f32(a.basis), code_0=f32(a.code), valid_mask_1=f32(w.mask).reshape(1, 1, w.H, w.W),
        loc1d_0=np.ascontiguousarray(a.loc1d, np.int64), homo_0=f32(a.homo), feat_pyr_0=f32(a.feat_pyr), feat_pyr_1=f32(b.feat_pyr),
        grad_pyr_1=f32(b.grad_pyr), level_offsets=np.ascontiguousarray(w.level_offsets, np.int64), photo_weights=f32(w.photo_weights),
        scale_0=_sc(a.scale), scale_1=_sc(b.scale), dpt_map_1=f32(D1).reshape(w.H, w.W), dpt_map_grad_1=f32(g1).reshape(2, w.H, w.W),
        basis_1=f32(b.basis).reshape(w.H, w.W, w.CS), geo_loss_param=_sc(w.geo_loss_param), geo_weight=_sc(w.geo_weight),
        sampled_dpts_0=dpts0, sampled_features_0=f32(feat0s))
    out = _run(tmp_path, arrays)
    t_far = np.array([0, 0, -100.0], np.float32)
    for sfx, tt in (("", t10), ("_far", t_far)):
        # mapper photometric pair
        o = orc.photo_jac_error(R10, tt, a.R, a.t, b.R, b.t, a.bias, a.basis, a.code, w.mask, a.loc1d, a.homo, a.feat_pyr,
                                b.feat_pyr, b.grad_pyr, w.level_offsets, a.scale, w.cams, w.eps, w.photo_weights)
        assert out["photo_fresh" + sfx][0] == 1.0
        assert out["photo_AtA" + sfx].shape == (13 + CS, 13 + CS) and out["photo_Atb" + sfx].shape == (13 + CS, 1)
        assert float(out["photo_err" + sfx][0]) == pytest.approx(o["error"], rel=1e-5)
        assert float(out["photo_err_only" + sfx][0]) == pytest.approx(o["error"], rel=1e-5)
        if o["num_inliers"] > 0:
            assert rel(out["photo_AtA" + sfx], o["AtA"]) < TOL_H and rel(out["photo_Atb" + sfx].reshape(-1), o["Atb"]) < TOL_H
        else:   # zero overlap (photometric_factor_kernels.cpp:1139-1141): 10 * sum(w), a zero system
            assert float(out["photo_err" + sfx][0]) == pytest.approx(10.0 * float(w.photo_weights.sum()), rel=1e-6)
            assert not out["photo_AtA" + sfx].any() and not out["photo_Atb" + sfx].any()
        # geometric pair
        og = orc.geo_jac_error(R10, tt, a.R, a.t, b.R, b.t, a.bias, a.basis, a.code, D1, g1, b.basis.reshape(w.H, w.W, w.CS),
                               w.mask, a.loc1d, a.homo, a.scale, b.scale, cam, w.eps, w.geo_loss_param, w.geo_weight)
        assert out["geo_fresh" + sfx][0] == 1.0
        assert float(out["geo_err" + sfx][0]) == pytest.approx(og["error"], rel=2e-5)
        assert float(out["geo_err_only" + sfx][0]) == pytest.approx(og["error"], rel=2e-5)
        if og["num_inliers"] > 0:
            assert rel(out["geo_AtA" + sfx], og["AtA"]) < TOL_H and rel(out["geo_Atb" + sfx].reshape(-1), og["Atb"]) < TOL_H
        else:
            assert float(out["geo_err" + sfx][0]) == pytest.approx(10.0 * w.geo_weight, rel=1e-6) and not out["geo_AtA" + sfx].any()
        # tracker trio
        for dof, key in ((6, "trk6"), (7, "trk7")):
            ot = orc.tracker_photo_jac_error(dof, R10, tt, w.mask, dpts0, a.homo, feat0s, b.feat_pyr, b.grad_pyr,
                                             w.level_offsets, w.cams, w.eps, w.photo_weights, scale0=a.scale)
            assert out[key + "_fresh" + sfx][0] == 1.0
            assert float(out[key + "_err" + sfx][0]) == pytest.approx(ot["error"], rel=1e-5)
            if ot["num_inliers"] > 0:
                assert rel(out[key + "_AtA" + sfx], ot["AtA"]) < TOL_H and rel(out[key + "_Atb" + sfx].reshape(-1), ot["Atb"]) < TOL_H
            else:
                assert not out[key + "_AtA" + sfx].any()
        oe, _ = orc.tracker_photo_error(R10, tt, w.mask, dpts0, a.homo, feat0s, b.feat_pyr, w.level_offsets, w.cams, w.eps,
                                        w.photo_weights)
        assert float(out["trk_err_only" + sfx][0]) == pytest.approx(oe, rel=1e-5)
    assert oracle_photo(orc, w, 0, 1)["num_inliers"] > 1000       # (the nominal case is a real overlap, the far one none)
    assert orc.photo_jac_error(R10, t_far, a.R, a.t, b.R, b.t, a.bias, a.basis, a.code, w.mask, a.loc1d, a.homo, a.feat_pyr,
                               b.feat_pyr, b.grad_pyr, w.level_offsets, a.scale, w.cams, w.eps, w.photo_weights)["num_inliers"] == 0


@needs_bin
def test_keypoint_factor_adapter_executes_behind_the_reference_headers(tmp_path, orc):
    """the 4 + 7 entry points of reprojection_factor_kernels.h:10-38 / match_geometry_factor_kernels.h:9-66 through
    sage_adapter_keypoints.cpp, incl. the robust_loss_type string dispatch (match_geometry_factor_kernels.cpp:1704-1807)"""
    from sage_slam_amd import capi
    rng = np.random.default_rng(77)
    H, W, N = 64, 80, 300
    cam = capi.SageCamera(72.0, 70.5, 39.5, 31.5, float(W), float(H))
    HW = H * W
    bias0 = (1.0 + 0.2 * rng.random(HW)).astype(np.float32); bias1 = (1.1 + 0.2 * rng.random(HW)).astype(np.float32)
    basis0 = (0.05 * rng.standard_normal((HW, CS))).astype(np.float32); basis1 = (0.05 * rng.standard_normal((HW, CS))).astype(np.float32)
    code0 = (0.3 * rng.standard_normal(CS)).astype(np.float32); code1 = (0.3 * rng.standard_normal(CS)).astype(np.float32)
    s0, s1 = 1.2, 0.9
    ys = rng.integers(0, H, N); xs = rng.integers(0, W, N)
    loc0 = (ys * W + xs).astype(np.int32); loc1 = rng.integers(0, HW, N).astype(np.int32)
    homo0 = np.stack([(xs - cam.cx) / cam.fx, (ys - cam.cy) / cam.fy, np.ones(N)], 1).astype(np.float32)
    homo1 = np.concatenate([rng.uniform(-0.5, 0.5, (N, 2)), np.ones((N, 1))], 1).astype(np.float32)
    R0 = synth.so3_exp(np.array([0.03, -0.05, 0.02])).astype(np.float32); t0 = np.array([0.02, -0.01, 0.03], np.float32)
    R1 = synth.so3_exp(np.array([-0.02, 0.04, 0.06])).astype(np.float32); t1 = np.array([-0.04, 0.02, -0.03], np.float32)
    R10 = (R1.T @ R0).astype(np.float32); t10 = (R1.T @ (t0 - t1)).astype(np.float32)
    u0 = (bias0[loc0] + basis0[loc0] @ code0).astype(np.float32); u1 = (bias1[loc1] + basis1[loc1] @ code1).astype(np.float32)
    d0 = (s0 * u0).astype(np.float32); d1 = (s1 * u1).astype(np.float32)
    X = (R10 @ (d0[:, None] * homo0).T).T + t10
    matched = (np.stack([X[:, 0] / X[:, 2] * cam.fx + cam.cx, X[:, 1] / X[:, 2] * cam.fy + cam.cy], 1)
               + rng.normal(0, 2.0, (N, 2))).astype(np.float32)
    eps, c, wgt = 1e-4, 0.05, 0.8
    arrays = dict(cam=np.array([cam.fx, cam.fy, cam.cx, cam.cy, cam.w, cam.h], np.float32), levels=_sc(1), eps=_sc(eps),
                  kp_R10=R10, kp_t10=t10, kp_R0=R0, kp_t0=t0, kp_R1=R1, kp_t1=t1, kp_bias_0=bias0, kp_bias_1=bias1,
                  kp_basis_0=basis0, kp_basis_1=basis1, kp_code_0=code0, kp_code_1=code1, kp_homo_0=homo0, kp_homo_1=homo1,
                  kp_loc_0=loc0, kp_loc_1=loc1, kp_matched_2d=matched, kp_dpts_0=d0, kp_dpts_1=d1, kp_unscaled_0=u0,
                  kp_unscaled_1=u1, kp_scale_0=_sc(s0), kp_scale_1=_sc(s1), kp_loss_param=_sc(c), kp_weight=_sc(wgt))
    out = _run(tmp_path, arrays)

    def same(key, o, D):
        assert out[key + "_fresh"][0] == 1.0, key
        assert out[key + "_AtA"].shape == (D, D) and out[key + "_Atb"].shape == (D, 1), key
        assert float(out[key + "_err"][0]) == pytest.approx(o["error"], rel=1e-5), key
        assert rel(out[key + "_AtA"], o["AtA"]) < TOL_H and rel(out[key + "_Atb"].reshape(-1), o["Atb"]) < TOL_H, key

    o = orc.reproj_jac_error(R10, t10, R0, t0, R1, t1, bias0, basis0, code0, loc0, homo0, matched, s0, cam, eps, c, wgt)
    assert o["num_inliers"] > 0
    same("reproj", o, 13 + CS)
    eo, _ = orc.reproj_error(R10, t10, bias0, basis0, code0, loc0, homo0, matched, s0, cam, eps, c, wgt)
    assert float(out["reproj_err_only"][0]) == pytest.approx(eo, rel=1e-5)
    assert float(out["reproj_err_only_i64"][0]) == float(out["reproj_err_only"][0])          # int64 locations: converted
    same("trk_reproj", orc.tracker_reproj_jac_error(R10, t10, d0, homo0, matched, cam, eps, c, wgt), 6)
    et, _ = orc.tracker_reproj_error(R10, t10, d0, homo0, matched, cam, eps, c, wgt)
    assert float(out["trk_reproj_err_only"][0]) == pytest.approx(et, rel=1e-5)
    for loss in ("fair", "L2", "huber", "unbiased"):
        om = orc.match_geom_jac_error(0, loss, R10, t10, R0, t0, R1, t1, bias0, bias1, basis0, basis1, code0, code1, homo0=homo0,
                                      homo1=homo1, loc0=loc0, loc1=loc1, scale0=s0, scale1=s1, loss_param=c, weight=wgt)
        same("mg_" + loss, om, 14 + 2 * CS)
        em = orc.match_geom_error(0, loss, R10, t10, bias0, bias1, basis0, basis1, code0, code1, homo0=homo0, homo1=homo1,
                                  loc0=loc0, loc1=loc1, scale0=s0, scale1=s1, loss_param=c, weight=wgt)
        assert float(out[f"mg_{loss}_err_only"][0]) == pytest.approx(em, rel=1e-5), loss
    # a string the reference has no branch for: no kernel runs there -- zero buffers, error 0 (match_geometry_factor_kernels.cpp:1811)
    assert out["mg_no_such_loss_fresh"][0] == 1.0 and not out["mg_no_such_loss_AtA"].any() and not out["mg_no_such_loss_Atb"].any()
    assert float(out["mg_no_such_loss_err"][0]) == 0.0 and float(out["mg_no_such_loss_err_only"][0]) == 0.0
    ol = orc.match_geom_jac_error(1, "fair", R10, t10, R0, t0, R1, t1, dpts0=u0, dpts1=u1, homo0=homo0, homo1=homo1, scale0=s0,
                                  scale1=s1, loss_param=c, weight=wgt)
    same("loop", ol, 14)
    assert float(out["loop_err_only"][0]) == pytest.approx(ol["error"], rel=1e-5)
    for mode, key, D in ((2, "trk_mg6", 6), (3, "trk_mg7", 7)):
        same(key, orc.match_geom_jac_error(mode, "fair", R10, t10, dpts0=d0, dpts1=d1, homo0=homo0, homo1=homo1, scale0=s0,
                                           loss_param=c, weight=wgt), D)
    assert float(out["trk_mg_err_only"][0]) == pytest.approx(
        orc.match_geom_error(2, "fair", R10, t10, dpts0=d0, dpts1=d1, homo0=homo0, homo1=homo1, loss_param=c, weight=wgt), rel=1e-5)
