#!/usr/bin/env python3
"""Second batch of golden vectors from the REFERENCE's Python implementation (representation/models/diff_ba.py),
for the columns the first batch (make_diffba_golden.py) could not reach ("parity holes"):

  diffba_autograd_geo_*.npz    torch.autograd through the reference's compute_geometry_error (:1995-2063) with
                               D1 = s1*(bias1 + basis1.code1) and d0 = s0*(bias0 + basis0.code0) formed in torch, and both
                               WORLD poses perturbed from the left (T <- exp(delta) T, gtsam_traits.h:45-70):
                               dE/d{delta0, delta1, code0, code1, s0, s1}.  Since E = (w/n) sum log(1 + (m rho)^2 / c) and the
                               C++ rows are sqrt(w)(-d rho/dx) with residual sqrt(w) rho, dE/dx = -2 Atb
                               (geometric_factor_kernels.cpp:684-716, :931-947) -- pins code1 / scale1 / both pose blocks
                               with general T1.
  diffba_autograd_photo_*.npz  the same through compute_photo_error (:1853-1939) over a 3-level pyramid:
                               dE/d{delta0, delta1, code0, s0} = -2 Atb (photometric_factor_kernels.cpp:241-363, :1139-1154)
                               -- pins the per-level weights INSIDE the Jacobian reduction, the level-0-inlier
                               normalisation with a partially masked frame, and the world-frame pose blocks (T1 != I).
  torch_cycle_match.npz        the literal tensor expression of match_geometry_factor.cpp:62-97 (== camera_tracker.cpp:
                               608-633) evaluated with the same ATen ops from Python, incl. exact ties (f4).

The two error functions are decorated with @torch.no_grad(); their undecorated bodies (``__wrapped__``) are called with
grad enabled -- no line of the reference is changed.  The feature / depth maps are PLANAR per channel (a x + b y + c):
the C++ Jacobians use bilinearly sampled central-difference gradient maps where autograd differentiates the bilinear
interpolant itself; on planar maps (away from the border texels) the two are the same function, so the comparison isolates
what is being pinned (coordinate chain, depth/code/scale chain, weights, normalisation).  The validity masks are column
bands with edges at multiples of 2^(L-1) pixels: then the reference's per-level nearest mask lookup and the C++ full-resolution lookup select
the same samples (u_l = (p + .5)/2^l - .5 rounds below W_l/4 exactly when p rounds below W/4).

Runs ONLY in the build container (needs /root/reference).  Usage:  python tests/golden/make_autograd_golden.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_diffba_golden import DEPTH_EPS, GEO_CAUCHY, GEO_WEIGHT, MG_FACTOR, MG_WEIGHT, PHOTO_POW, PHOTO_WEIGHT, import_reference, rot  # noqa: E402

T64 = torch.float64


def hat(w):
    z = torch.zeros((), dtype=w.dtype)
    return torch.stack([torch.stack([z, -w[2], w[1]]), torch.stack([w[2], z, -w[0]]), torch.stack([-w[1], w[0], z])])


def perturbed_relative_pose(R0, t0, R1, t1, d0, d1):
    """T_k <- exp(delta_k) T_k to first order in delta = [v, omega] (exact gradient at delta = 0), then
    R10 = R1^T R0, t10 = R1^T (t0 - t1) (core/gtsam/photometric_factor.cpp:280-281)."""
    I = torch.eye(3, dtype=R0.dtype)
    R0p = (I + hat(d0[3:])) @ R0; t0p = (I + hat(d0[3:])) @ t0 + d0[:3]
    R1p = (I + hat(d1[3:])) @ R1; t1p = (I + hat(d1[3:])) @ t1 + d1[:3]
    return R1p.T @ R0p, R1p.T @ (t0p - t1p)


def planar(rng, C, H, W, amp):
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    a = rng.uniform(-amp, amp, C) / W; b = rng.uniform(-amp, amp, C) / H; c = rng.uniform(-0.3, 0.3, C)
    return a[:, None, None] * xx + b[:, None, None] * yy + c[:, None, None]


def poses(rng):
    R1 = rot(np.array([0.3, -0.2, 0.25])); t1 = np.array([0.4, -0.3, 0.2])          # a general T1
    R10 = rot(np.array([0.012, -0.018, 0.01])); t10 = np.array([0.03, -0.015, 0.02])
    R0 = R1 @ R10; t0 = R1 @ t10 + t1                                             # T0 = T1 T10
    return R0, t0, R1, t1


def samples(rng, H, W, N, lo_x, hi_x, lo_y, hi_y, fx, fy, cx, cy):
    gx, gy = np.meshgrid(np.arange(lo_x, hi_x), np.arange(lo_y, hi_y))
    pick = rng.choice(gx.size, N, replace=False)                                  # distinct integer source pixels
    xs, ys = gx.reshape(-1)[pick], gy.reshape(-1)[pick]
    loc = ys * W + xs
    homo = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones(N)], 0)
    return loc, homo, xs, ys


def geo_case(DBA, ba, seed, band):
    rng = np.random.default_rng(seed)
    H, W, N, CS = 32, 40, 90, 8
    fx, fy, cx, cy = 0.9 * W, 0.85 * W, W / 2.0 - 0.3, H / 2.0 + 0.2
    R0, t0, R1, t1 = poses(rng)
    loc, homo, _, _ = samples(rng, H, W, N, 4, W - 5, 4, H - 5, fx, fy, cx, cy)
    bias0 = 1.2 + 0.05 * rng.standard_normal(N); basis0 = 0.05 * rng.standard_normal((N, CS))
    code0 = 0.1 * rng.standard_normal(CS); s0 = 1.3
    bias1 = 1.7 + planar(rng, 1, H, W, 0.3)[0]                                    # unscaled: D1 = s1 * (bias1 + basis1.code1)
    basis1 = np.moveaxis(planar(rng, CS, H, W, 0.4), 0, -1)                       # [H,W,CS]
    code1 = 0.2 * rng.standard_normal(CS); s1 = 0.9
    mask = np.ones((H, W))
    if band:
        mask[:, band[0]:band[1]] = 0
    tt = lambda a: torch.tensor(np.asarray(a, np.float64), dtype=T64)
    d0 = torch.zeros(6, dtype=T64, requires_grad=True); d1 = torch.zeros(6, dtype=T64, requires_grad=True)
    c0 = tt(code0).requires_grad_(); c1 = tt(code1).requires_grad_()
    ts0 = tt(s0).requires_grad_(); ts1 = tt(s1).requires_grad_()
    R10, t10 = perturbed_relative_pose(tt(R0), tt(t0), tt(R1), tt(t1), d0, d1)
    D1 = ts1 * (tt(bias1) + tt(basis1) @ c1)
    depths = ts0 * (tt(bias0) + tt(basis0) @ c0)
    mean_sq = torch.tensor(float(np.mean((s1 * (bias1 + basis1 @ code1)) ** 2)), dtype=T64)
    ba64 = ba
    E = DBA.compute_geometry_error.__wrapped__(
        ba64, tgt_valid_mask=tt(mask).reshape(1, 1, H, W), tgt_depth_map=D1.reshape(1, 1, H, W),
        mean_squared_tgt_depth_value=mean_sq, sampled_depths=depths, sampled_homo_2d_locations=tt(homo),
        camera_intrinsics=tt([fx, fy, cx, cy]).reshape(1, 4), guess_rotation=R10, guess_translation=t10)
    g = torch.autograd.grad(E, [d0, d1, c0, c1, ts0, ts1])
    return dict(H=H, W=W, N=N, CS=CS, intr=np.array([fx, fy, cx, cy]), R0=R0, t0=t0, R1=R1, t1=t1, loc=loc, homo=homo,
                bias0=bias0, basis0=basis0, code0=code0, s0=s0, bias1=bias1, basis1=basis1, code1=code1, s1=s1, mask=mask,
                mean_sq=float(mean_sq), cauchy_factor=GEO_CAUCHY, weight=GEO_WEIGHT, depth_eps=DEPTH_EPS,
                E=float(E.detach()), g_pose0=g[0].numpy(), g_pose1=g[1].numpy(), g_code0=g[2].numpy(), g_code1=g[3].numpy(),
                g_s0=float(g[4]), g_s1=float(g[5]))


def photo_case(DBA, ba, seed, band):
    rng = np.random.default_rng(seed)
    H, W, N, CS, FS, L = 48, 64, 70, 8, 6, 3
    fx0, fy0, cx0, cy0 = 0.9 * W, 0.85 * W, W / 2.0 - 0.3, H / 2.0 + 0.2
    R0, t0, R1, t1 = poses(rng)
    # keep source pixels and their projections >= 3 coarsest-level texels (4 px each) away from the border: the outermost
    # texel ring of every pyramid level is not planar (zero padding + mask renormalisation), the ring next to it has a
    # wrong central difference
    loc, homo, xs, ys = samples(rng, H, W, N, 14, W - 15, 14, H - 15, fx0, fy0, cx0, cy0)
    bias0 = 1.2 + 0.05 * rng.standard_normal(N); basis0 = 0.05 * rng.standard_normal((N, CS))
    code0 = 0.1 * rng.standard_normal(CS); s0 = 1.3
    feat0 = planar(rng, FS, H, W, 1.5); feat1 = planar(rng, FS, H, W, 1.5)
    ones = torch.ones(1, 1, H, W, dtype=T64)
    tt = lambda a: torch.tensor(np.asarray(a, np.float64), dtype=T64)
    gk = ba.gauss_kernel.to(T64)
    pyr1, _ = DBA.generate_gaussian_pyramid(tt(feat1).reshape(1, FS, H, W), ones, True, L, gk)
    pyr0, _ = DBA.generate_gaussian_pyramid(tt(feat0).reshape(1, FS, H, W), ones, True, L, gk)
    pyr1 = list(reversed(pyr1)); pyr0 = list(reversed(pyr0))                      # fine -> coarse
    mask = np.ones((H, W))
    if band:
        mask[:, band[0]:band[1]] = 0
    mpyr, intr, scales, src = [], [], [], []
    for l in range(L):
        hl, wl = pyr1[l].shape[2:]
        rx, ry = wl / W, hl / H
        intr.append(tt([[fx0 * rx, fy0 * ry, (cx0 + 0.5) * rx - 0.5, (cy0 + 0.5) * ry - 0.5]]))
        scales.append(torch.tensor(0.5 ** l, dtype=T64))
        ml = np.ones((hl, wl))
        if band:
            ml[:, band[0] >> l: band[1] >> l] = 0
        mpyr.append(tt(ml).reshape(1, 1, hl, wl))
        # source features: grid_sample at the integer source pixel, align_corners=False (camera_tracker.cpp:1092-1123)
        grid = torch.stack([(tt(xs) + 0.5) * (2.0 / W) - 1.0, (tt(ys) + 0.5) * (2.0 / H) - 1.0], 1).reshape(1, 1, N, 2)
        src.append(F.grid_sample(pyr0[l], grid, mode="bilinear", padding_mode="zeros", align_corners=False).reshape(FS, N))
    d0 = torch.zeros(6, dtype=T64, requires_grad=True); d1 = torch.zeros(6, dtype=T64, requires_grad=True)
    c0 = tt(code0).requires_grad_(); ts0 = tt(s0).requires_grad_()
    R10, t10 = perturbed_relative_pose(tt(R0), tt(t0), tt(R1), tt(t1), d0, d1)
    depths = ts0 * (tt(bias0) + tt(basis0) @ c0)
    E = DBA.compute_photo_error.__wrapped__(ba, tt(homo), depths, src, intr, scales, pyr1, mpyr, R10, t10)
    g = torch.autograd.grad(E, [d0, d1, c0, ts0])
    weights = np.array([abs(PHOTO_WEIGHT * 10) * (0.5 ** l) ** PHOTO_POW for l in range(L)])
    out = dict(H=H, W=W, N=N, CS=CS, FS=FS, L=L, intr0=np.array([fx0, fy0, cx0, cy0]), R0=R0, t0=t0, R1=R1, t1=t1,
               loc=loc, homo=homo, bias0=bias0, basis0=basis0, code0=code0, s0=s0, mask=mask, weights=weights,
               depth_eps=DEPTH_EPS, E=float(E.detach()), g_pose0=g[0].numpy(), g_pose1=g[1].numpy(), g_code0=g[2].numpy(),
               g_s0=float(g[3]))
    for l in range(L):
        out[f"feat0_level{l}"] = pyr0[l].numpy()[0]
        out[f"feat1_level{l}"] = pyr1[l].numpy()[0]
    return out


def cycle_match_case(seed, C, H, W, K, ties):
    """match_geometry_factor.cpp:62-97, op by op (torch C++ API == torch Python API over the same ATen kernels)."""
    rng = np.random.default_rng(seed)
    d0 = rng.standard_normal((1, C, H, W)).astype(np.float32)
    sh = (2, -1)
    d1 = (np.roll(d0, sh, (2, 3)) + 0.15 * rng.standard_normal((1, C, H, W))).astype(np.float32)
    if ties:
        # exact ties: a few descriptors of frame 1 (and of frame 0) are duplicated at a LATER flat index
        for (y, x), (y2, x2) in (((3, 4), (9, 7)), ((5, 1), (5, 2)), ((0, 0), (H - 1, W - 1))):
            d1[0, :, y2, x2] = d1[0, :, y, x]
            d0[0, :, y2, x2] = d0[0, :, y, x]
    loc = np.sort(rng.choice(H * W, K, replace=False)).astype(np.int64)
    if ties:
        loc[:3] = [(3 - 2) % H * W + (4 + 1) % W, 5 * W + 1, 0]                   # keypoints whose best match is duplicated
        loc = np.unique(loc); K = loc.size
    thresh = 2.0
    feature_desc_0, feature_desc_1 = torch.from_numpy(d0), torch.from_numpy(d1)
    keypoint_locations_1d = torch.from_numpy(loc)
    width, height, channel, num_keypoints = W, H, C, K
    keypoint_locations_2d_x = torch.fmod(keypoint_locations_1d, float(width))
    keypoint_locations_2d_y = torch.floor(keypoint_locations_1d / float(width))
    keypoint_features_0 = feature_desc_0.reshape(channel, height * width)[:, keypoint_locations_1d]
    feature_response_1 = -torch.sum(torch.square(keypoint_features_0.reshape(channel, num_keypoints, 1) -
                                                 feature_desc_1.reshape(channel, 1, height * width)), 0, False)
    raw_matched_locations_1d_1 = torch.max(feature_response_1, 1, False)[1]
    raw_matched_features_1 = feature_desc_1.reshape(channel, height * width)[:, raw_matched_locations_1d_1]
    feature_response_0 = -torch.sum(torch.square(raw_matched_features_1.reshape(channel, num_keypoints, 1) -
                                                 feature_desc_0.reshape(channel, 1, height * width)), 0, False)
    cyc_matched_locations_1d_0 = torch.max(feature_response_0, 1, False)[1]
    cyc_matched_locations_2d_x = torch.fmod(cyc_matched_locations_1d_0, float(width))
    cyc_matched_locations_2d_y = torch.floor(cyc_matched_locations_1d_0 / float(width))
    cyc_distances_sq = torch.square(keypoint_locations_2d_x - cyc_matched_locations_2d_x) + \
        torch.square(keypoint_locations_2d_y - cyc_matched_locations_2d_y)
    inlier = (cyc_distances_sq <= (thresh * thresh))
    return dict(desc0=d0[0], desc1=d1[0], kp=loc, thresh=np.float32(thresh), raw=raw_matched_locations_1d_1.numpy(),
                cyc=cyc_matched_locations_1d_0.numpy(), inlier=inlier.numpy().astype(np.int32),
                inlier_idx=torch.nonzero(inlier).reshape(-1).numpy())


def main():
    DBA = import_reference()
    ba = DBA(MG_FACTOR, MG_WEIGHT, 1.0e-3, GEO_CAUCHY, GEO_WEIGHT, 1.0, PHOTO_POW, PHOTO_WEIGHT, 1, DEPTH_EPS, 0)
    for name, band, pband in (("allvalid", None, None), ("band", (0, 8), (24, 32))):
        out = geo_case(DBA, ba, 61, band)
        np.savez_compressed(os.path.join(HERE, f"diffba_autograd_geo_{name}.npz"), **out)
        print("geo", name, "E", out["E"], "|g_pose0|", np.linalg.norm(out["g_pose0"]), "|g_code1|", np.linalg.norm(out["g_code1"]),
              "g_s1", out["g_s1"])
        out = photo_case(DBA, ba, 71, pband)
        np.savez_compressed(os.path.join(HERE, f"diffba_autograd_photo_{name}.npz"), **out)
        print("photo", name, "E", out["E"], "|g_pose0|", np.linalg.norm(out["g_pose0"]), "|g_code0|", np.linalg.norm(out["g_code0"]))
    with torch.no_grad():
        cases = {f"c{i}": cycle_match_case(**kw) for i, kw in enumerate((
            dict(seed=81, C=16, H=24, W=32, K=60, ties=False), dict(seed=82, C=32, H=16, W=20, K=25, ties=True),
            dict(seed=83, C=16, H=32, W=40, K=120, ties=False)))}
    flat = {f"{k}_{kk}": v for k, c in cases.items() for kk, v in c.items()}
    np.savez_compressed(os.path.join(HERE, "torch_cycle_match.npz"), **flat)
    for k, c in cases.items():
        print("cycle", k, "K", c["kp"].size, "inliers", int(c["inlier"].sum()))


if __name__ == "__main__":
    main()
