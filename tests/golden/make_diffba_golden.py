#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own Python implementation of the
photometric / geometric BA terms (``representation/models/diff_ba.py``:
``photo_term`` :953-1061, ``geometry_term`` :1164-1287) and of the keypoint terms
(``match_geometry_term`` :891-951, the projection Jacobians :322-386 the reprojection term uses).

Runs ONLY in the build container (needs /root/reference); the outputs
(``tests/golden/diffba_*.npz``: seeded inputs + the reference's outputs) are
committed and travel, this script's imports of the reference do not.

``diff_ba.py`` imports a few packages that are absent here and unused by the two
functions called (cv2, torchgeometry, umap, the repo's own ``utils``/``models``
packages which pull h5py/tensorboardX/...); they are replaced by empty modules for the
duration of the import.  ``DiffBundleAdjustment.__init__`` calls ``.cuda()`` on a
constant (:37-38); ``torch.Tensor.cuda`` is made the identity so it constructs on CPU.
No arithmetic of the reference is touched.

Usage:  python tests/golden/make_diffba_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/representation"


def import_reference():
    for name in ["cv2", "torchgeometry", "umap", "utils", "models", "utils.logger"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["utils"].logger = types.SimpleNamespace(debug=lambda *a, **k: None, error=lambda *a, **k: None)
    torch.Tensor.cuda = lambda self, *a, **k: self
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_diff_ba", os.path.join(REF, "models", "diff_ba.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.DiffBundleAdjustment


def smooth(rng, C, H, W, cycles=1.5, waves=3):
    yy, xx = np.meshgrid(np.arange(H) / H, np.arange(W) / W, indexing="ij")
    out = np.zeros((C, H, W))
    for c in range(C):
        for _ in range(waves):
            kx, ky = rng.uniform(-cycles, cycles, 2)
            out[c] += rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * (kx * xx + ky * yy) + rng.uniform(0, 6.28))
    return out


def central_grad(img):
    p = np.pad(img, ((0, 0), (1, 1), (1, 1)), mode="edge")
    gx = 0.5 * (p[:, 1:-1, 2:] - p[:, 1:-1, :-2])
    gy = 0.5 * (p[:, 2:, 1:-1] - p[:, :-2, 1:-1])
    return gx, gy


def rot(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def make_case(seed, H, W, N, FS, CS, with_invalid):
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = 0.9 * W, 0.85 * W, W / 2.0 - 0.3, H / 2.0 + 0.2
    feat1 = smooth(rng, FS, H, W)
    gx, gy = central_grad(feat1)
    mask = np.ones((H, W))
    if with_invalid:
        mask[:, : W // 4] = 0            # a masked band so some samples land on mask == 0
    # sampled source pixels (integer locations), depths from bias + basis*code
    loc = rng.choice(H * W, N, replace=False)
    lx, ly = loc % W, loc // W
    homo = np.stack([(lx - cx) / fx, (ly - cy) / fy, np.ones(N)], 0)      # 3 x N
    bias = 1.0 + 0.2 * rng.standard_normal(N)
    basis = 0.05 * rng.standard_normal((N, CS))
    code = 0.1 * rng.standard_normal(CS)
    scale = 1.3
    if with_invalid:
        bias[:3] = -0.5                  # negative depth -> behind the camera
    src_feats = rng.uniform(-1, 1, (FS, N))
    R = rot(np.array([0.02, -0.03, 0.015]))
    t = np.array([0.05, -0.02, 0.03])
    # geometric inputs
    dmap = 1.2 + 0.1 * smooth(rng, 1, H, W)
    dgx, dgy = central_grad(dmap)
    return dict(H=H, W=W, N=N, FS=FS, CS=CS, intr=np.array([fx, fy, cx, cy]),
                feat1=feat1, gx=gx, gy=gy, mask=mask, loc=loc, homo=homo, bias=bias, basis=basis,
                code=code, scale=scale, src_feats=src_feats, R=R, t=t,
                dmap=dmap[0], dgx=dgx[0], dgy=dgy[0])


def run_case(ba, c):
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    H, W, FS = c["H"], c["W"], c["FS"]
    A, diff, valid = ba.photo_term(
        tgt_feature_map=f32(c["feat1"]).reshape(1, FS, H, W),
        tgt_feature_map_spatial_grad=f32(np.concatenate([c["gx"], c["gy"]], 0)).reshape(1, 2 * FS, H, W),
        tgt_valid_mask=f32(c["mask"]).reshape(1, 1, H, W),
        sampled_src_features=f32(c["src_feats"]),
        sampled_depth_bias=f32(c["bias"]),
        sampled_depth_jac_code_hierarchy=f32(c["basis"]),
        sampled_homo_2d_locations=f32(c["homo"]),
        camera_intrinsics=f32(c["intr"]).reshape(1, 4),
        guess_rotation=f32(c["R"]), guess_translation=f32(c["t"]),
        guess_code_hierarchy=f32(c["code"]), guess_scale=torch.tensor(c["scale"], dtype=torch.float32))
    mean_sq = float(np.mean(c["dmap"] ** 2))
    Ag, dg, eg, vg = ba.geometry_term(
        tgt_valid_mask=f32(c["mask"]).reshape(1, 1, H, W),
        tgt_depth_map=f32(c["dmap"]).reshape(1, 1, H, W),
        mean_squared_tgt_depth_value=torch.tensor(mean_sq, dtype=torch.float32),
        tgt_depth_map_spatial_grad=f32(np.stack([c["dgx"], c["dgy"]], 0)).reshape(1, 2, H, W),
        sampled_depth_bias=f32(c["bias"]),
        sampled_depth_jac_code_hierarchy=f32(c["basis"]),
        sampled_homo_2d_locations=f32(c["homo"]),
        camera_intrinsics=f32(c["intr"]).reshape(1, 4),
        guess_rotation=f32(c["R"]), guess_translation=f32(c["t"]),
        guess_code_hierarchy=f32(c["code"]), guess_scale=torch.tensor(c["scale"], dtype=torch.float32))
    # error-only evaluation (compute_geometry_error :1995-2063; weight = geometry_term_weight of the constructor)
    d0 = c["scale"] * (c["bias"] + c["basis"] @ c["code"])
    ge_only = ba.compute_geometry_error(
        tgt_valid_mask=f32(c["mask"]).reshape(1, 1, H, W), tgt_depth_map=f32(c["dmap"]).reshape(1, 1, H, W),
        mean_squared_tgt_depth_value=torch.tensor(mean_sq, dtype=torch.float32), sampled_depths=f32(d0),
        sampled_homo_2d_locations=f32(c["homo"]), camera_intrinsics=f32(c["intr"]).reshape(1, 4),
        guess_rotation=f32(c["R"]), guess_translation=f32(c["t"]))
    out = {k: np.asarray(v, dtype=np.float32) if isinstance(v, np.ndarray) and v.dtype.kind == "f" else v
           for k, v in c.items()}
    out.update(geo_error_only=np.float32(float(ge_only)), geo_term_weight=np.float32(GEO_WEIGHT))
    out.update(photo_A=A.detach().numpy(), photo_diff=diff.detach().numpy(), photo_valid=valid.detach().numpy(),
               geo_A=Ag.detach().numpy(), geo_diff=dg.detach().numpy(), geo_err=eg.detach().numpy(),
               geo_valid=vg.detach().numpy(), geo_mean_sq=np.float32(mean_sq),
               geo_cauchy_factor=np.float32(GEO_CAUCHY), depth_eps=np.float32(DEPTH_EPS))
    return out


def make_keypoint_case(seed, N=40, CS=16):
    """sparse keypoints with matches (match-geometry / reprojection factors): relative pose + code + scale"""
    rng = np.random.default_rng(seed)
    W, H = 80, 64
    fx, fy, cx, cy = 70.0, 68.0, 40.3, 31.8
    px = rng.uniform(5, W - 5, N); py = rng.uniform(5, H - 6, N)
    homo = np.stack([(px - cx) / fx, (py - cy) / fy, np.ones(N)], 0)         # 3 x N
    bias = 1.0 + 0.2 * rng.standard_normal(N)
    basis = 0.05 * rng.standard_normal((N, CS))
    code = 0.1 * rng.standard_normal(CS)
    scale = 1.3
    R = rot(np.array([0.02, -0.03, 0.015])); t = np.array([0.05, -0.02, 0.03])
    X = scale * (bias + basis @ code) * (R @ homo) + t[:, None]
    mh = X / X[2] + 0.01 * rng.standard_normal((3, N)); mh[2] = 1              # matched homogeneous coordinates
    md = X[2] * (1 + 0.05 * rng.standard_normal(N))                             # matched depths
    return dict(N=N, CS=CS, W=W, H=H, intr=np.array([fx, fy, cx, cy]), homo=homo, bias=bias, basis=basis, code=code,
                scale=scale, R=R, t=t, match_homo=mh, match_depths=md, mean_sq=float(np.mean(md ** 2)))


def run_keypoint_case(ba, c):
    """match_geometry_term (diff_ba.py:891-951) and the projection Jacobians the reprojection term is built from
    (:322-386; reproj_term itself reads attributes the constructor never sets)."""
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    sc = torch.tensor(c["scale"], dtype=torch.float32)
    A, diff, err = ba.match_geometry_term(f32(c["bias"]), f32(c["basis"]), f32(c["homo"]), f32(c["match_homo"]),
                                          f32(c["match_depths"]), torch.tensor(c["mean_sq"], dtype=torch.float32),
                                          f32(c["R"]), f32(c["t"]), f32(c["code"]), sc)
    d0 = c["scale"] * (c["bias"] + c["basis"] @ c["code"])
    X = d0 * (c["R"] @ c["homo"]) + c["t"][:, None]
    fx, fy = (torch.tensor(float(v), dtype=torch.float32) for v in c["intr"][:2])
    Jp = ba.jacobian_projected_2d_location_wrt_camera_pose(f32(X), fx, fy, mode="wh")
    Jd = ba.jacobian_projected_2d_location_wrt_src_depth(f32(c["R"] @ c["homo"]), f32(X), fx, fy, mode="wh")
    out = {k: (np.asarray(v, dtype=np.float32) if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    mg_only = ba.compute_match_geom_error(f32(d0), f32(c["homo"]), f32(c["match_homo"]), f32(c["match_depths"]),
                                          torch.tensor(c["mean_sq"], dtype=torch.float32), f32(c["R"]), f32(c["t"]))
    out.update(mg_error_only=np.float32(float(mg_only)), mg_term_weight=np.float32(MG_WEIGHT))
    out.update(mg_A=A.detach().numpy(), mg_diff=diff.detach().numpy(), mg_err=err.detach().numpy(),
               mg_param_factor=np.float32(MG_FACTOR), proj_J_pose=Jp.detach().numpy(), proj_J_depth=Jd.detach().numpy())
    return out


def import_reference_se3_exp():
    """representation/utils/processing.py:596-633 (se3_exp + so3_hat), the function DiffBundleAdjustment.update_variables
    (:830-842) retracts with.  The module imports torchgeometry at the top (absent, unused here): stubbed."""
    if "torchgeometry" not in sys.modules:
        sys.modules["torchgeometry"] = types.ModuleType("torchgeometry")
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_processing", os.path.join(REF, "utils", "processing.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.se3_exp


def run_retract_cases(DBA, se3_exp):
    """se3_exp on a spread of twists and update_variables (:830-842, left-multiplicative pose update, additive scale /
    code) -- what a13 / a8's UpdateVariables restate."""
    rng = np.random.default_rng(31)
    xis = [rng.standard_normal(6) * s for s in (1e-4, 1e-2, 0.1, 0.5, 1.0, 2.5) for _ in range(2)]
    xis = np.array(xis, dtype=np.float32)                                        # [omega(3), v(3)]
    exps = np.stack([se3_exp(torch.from_numpy(x)).numpy() for x in xis])          # [n, 3, 4]
    sys.modules["utils"].se3_exp = se3_exp                                        # diff_ba calls utils.se3_exp
    CS = 8
    R0 = rot(np.array([0.3, -0.2, 0.4])).astype(np.float32); t0 = np.array([0.2, -0.1, 0.5], np.float32)
    sols = (rng.standard_normal((4, 7 + CS)) * 0.05).astype(np.float32)           # [rot3, trans3, scale, code]
    code0 = (0.1 * rng.standard_normal(CS)).astype(np.float32)
    upd = []
    for sol in sols:
        R1, t1, s1, c1 = DBA.update_variables(torch.from_numpy(sol).reshape(-1, 1), torch.from_numpy(R0),
                                              torch.from_numpy(t0), torch.tensor([1.3]), torch.from_numpy(code0))
        upd.append(np.concatenate([R1.numpy().reshape(-1), t1.numpy().reshape(-1), s1.numpy().reshape(-1),
                                   c1.numpy().reshape(-1)]))
    return dict(xi=xis, exp=exps, R0=R0, t0=t0, scale0=np.float32(1.3), code0=code0, sol=sols, updated=np.array(upd))


def run_pyramid_case(DBA, ba):
    """generate_gaussian_pyramid (diff_ba.py:44-71): masked 3x3 binomial blur, stride 2, renormalised by the blurred mask
    -- the Python twin of the keyframe pyramid of mapping_utils.h / mapper.cpp:1384-1426 (f1 producer)."""
    rng = np.random.default_rng(41)
    FS, H, W, L = 6, 32, 40, 3
    feat = smooth(rng, FS, H, W).astype(np.float32)
    mask = np.ones((H, W), np.float32)
    mask[:, :6] = 0; mask[:3] = 0; mask[20:25, 30:] = 0
    pyr, mpyr = DBA.generate_gaussian_pyramid(torch.from_numpy(feat).reshape(1, FS, H, W),
                                              torch.from_numpy(mask).reshape(1, 1, H, W), True, L, ba.gauss_kernel)
    out = dict(feat=feat, mask=mask, L=L)
    for l, (f, m) in enumerate(zip(reversed(pyr), reversed(mpyr))):
        out[f"level{l}"] = f.numpy()[0]
        out[f"mask{l}"] = m.numpy()[0, 0]
    return out


def run_multilevel_photo_error(DBA, ba):
    """compute_photo_error (diff_ba.py:1853-1939) over a 3-level pyramid: per level, project with that level's
    intrinsics, grid_sample(align_corners=False), weight * sum / num_samples, summed over the levels.  The level
    intrinsics handed over are the ones the C++ kernels' coordinate rule implies (u_l = (u_0 + .5) * fx_l / fx_0 - .5,
    photometric_factor_kernels.cpp:142-160): fx_l = fx_0 * W_l / W_0, cx_l = (cx_0 + .5) * W_l / W_0 - .5.  All-valid
    masks and positive depths, so the per-level inlier counts of the Python code equal the level-0 count the C++ uses."""
    rng = np.random.default_rng(51)
    FS, H, W, L, N = 8, 32, 40, 3, 60
    fx0, fy0, cx0, cy0 = 0.9 * W, 0.85 * W, W / 2.0 - 0.3, H / 2.0 + 0.2
    feat = smooth(rng, FS, H, W).astype(np.float32)
    ones = np.ones((H, W), np.float32)
    pyr, mpyr = DBA.generate_gaussian_pyramid(torch.from_numpy(feat).reshape(1, FS, H, W),
                                              torch.from_numpy(ones).reshape(1, 1, H, W), True, L, ba.gauss_kernel)
    pyr = list(reversed(pyr)); mpyr = list(reversed(mpyr))                        # fine -> coarse
    px = rng.uniform(6, W - 7, N); py = rng.uniform(6, H - 7, N)
    homo = np.stack([(px - cx0) / fx0, (py - cy0) / fy0, np.ones(N)], 0).astype(np.float32)
    depths = (1.0 + 0.2 * rng.standard_normal(N)).astype(np.float32)
    R = rot(np.array([0.02, -0.03, 0.015])).astype(np.float32); t = np.array([0.05, -0.02, 0.03], np.float32)
    src = [rng.uniform(-1, 1, (FS, N)).astype(np.float32) for _ in range(L)]
    intr, scales = [], []
    for l in range(L):
        hl, wl = pyr[l].shape[2:]
        rx, ry = wl / W, hl / H
        intr.append(torch.tensor([[fx0 * rx, fy0 * ry, (cx0 + 0.5) * rx - 0.5, (cy0 + 0.5) * ry - 0.5]], dtype=torch.float32))
        scales.append(torch.tensor(0.5 ** l, dtype=torch.float32))
    err = ba.compute_photo_error(torch.from_numpy(homo), torch.from_numpy(depths), [torch.from_numpy(x) for x in src],
                                 intr, scales, pyr, mpyr, torch.from_numpy(R), torch.from_numpy(t))
    weights = np.array([abs(PHOTO_WEIGHT * 10) * (0.5 ** l) ** PHOTO_POW for l in range(L)], np.float32)
    out = dict(feat=feat, L=L, N=N, FS=FS, H=H, W=W, intr0=np.array([fx0, fy0, cx0, cy0], np.float32), homo=homo,
               depths=depths, R=R, t=t, src=np.stack(src), weights=weights, error=np.float32(float(err)),
               depth_eps=np.float32(DEPTH_EPS))
    for l in range(L):
        out[f"level{l}"] = pyr[l].numpy()[0]
    return out


GEO_CAUCHY = 0.03
DEPTH_EPS = 1.0e-4
MG_FACTOR = 0.1
PHOTO_WEIGHT = 1.0
PHOTO_POW = 1.0
MG_WEIGHT = 0.1
GEO_WEIGHT = 0.1


def main():
    DBA = import_reference()
    # ctor args (diff_ba.py:16-18): match_geom_param_factor, match_geom_term_weight, code_term_weight,
    # geometry_cauchy_param_factor, geometry_term_weight, scale_term_weight, photo_pow_factor,
    # photo_weight, num_photo_level, depth_eps, num_display_matches
    ba = DBA(MG_FACTOR, MG_WEIGHT, 1.0e-3, GEO_CAUCHY, GEO_WEIGHT, 1.0, PHOTO_POW, PHOTO_WEIGHT, 1, DEPTH_EPS, 0)
    with torch.no_grad():
        out = run_multilevel_photo_error(DBA, ba)
        np.savez_compressed(os.path.join(HERE, "diffba_photo_levels.npz"), **out)
        print("diffba_photo_levels error", out["error"], "weights", out["weights"])
        out = run_pyramid_case(DBA, ba)
        np.savez_compressed(os.path.join(HERE, "diffba_pyramid.npz"), **out)
        print("diffba_pyramid", [out[f"level{l}"].shape for l in range(out["L"])])
        out = run_retract_cases(DBA, import_reference_se3_exp())
        np.savez_compressed(os.path.join(HERE, "diffba_retract.npz"), **out)
        print("diffba_retract", out["exp"].shape, out["updated"].shape)
        for name, kw in {"diffba_keypoints": dict(seed=21, N=40, CS=16),
                         "diffba_keypoints32": dict(seed=22, N=25, CS=32)}.items():
            out = run_keypoint_case(ba, make_keypoint_case(**kw))
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
            print(name, "mg_A", out["mg_A"].shape, "proj_J_pose", out["proj_J_pose"].shape)
        for name, kw in {
            "diffba_allvalid": dict(seed=11, H=16, W=20, N=24, FS=16, CS=32, with_invalid=False),
            "diffba_invalid": dict(seed=12, H=16, W=20, N=24, FS=16, CS=16, with_invalid=True),
        }.items():
            c = make_case(**kw)
            out = run_case(ba, c)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
            print(name, "photo_A", out["photo_A"].shape, "valid", out["photo_valid"].sum(),
                  "geo_A", out["geo_A"].shape, "geo valid", out["geo_valid"].sum())


if __name__ == "__main__":
    main()
