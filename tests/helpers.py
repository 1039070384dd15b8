"""Shared helpers: run the same edge through the CPU oracle (checker) and the HIP engine (product)."""
import numpy as np

from sage_slam_amd import synth


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def oracle_photo(orc, w, k0, k1, jac=True, prec="f32"):
    a, b = w.keyframes[k0], w.keyframes[k1]
    R10, t10 = synth.relative_pose(a.R, a.t, b.R, b.t)
    if jac:
        return orc.photo_jac_error(R10, t10, a.R, a.t, b.R, b.t, a.bias, a.basis, a.code, w.mask, a.loc1d, a.homo,
                                   a.feat_pyr, b.feat_pyr, b.grad_pyr, w.level_offsets, a.scale, w.cams, w.eps,
                                   w.photo_weights, prec=prec)
    e, n = orc.photo_error(R10, t10, a.bias, a.basis, a.code, w.mask, a.loc1d, a.homo, a.feat_pyr, b.feat_pyr,
                           w.level_offsets, a.scale, w.cams, w.eps, w.photo_weights, prec=prec)
    return dict(error=e, num_inliers=n)


def oracle_geo(orc, w, k0, k1, jac=True, prec="f32"):
    a, b = w.keyframes[k0], w.keyframes[k1]
    R10, t10 = synth.relative_pose(a.R, a.t, b.R, b.t)
    D1, g1 = synth.depth_and_grad(b, w.H, w.W)
    if jac:
        return orc.geo_jac_error(R10, t10, a.R, a.t, b.R, b.t, a.bias, a.basis, a.code, D1, g1,
                                 b.basis.reshape(w.H, w.W, w.CS), w.mask, a.loc1d, a.homo, a.scale, b.scale,
                                 w.cams[0], w.eps, w.geo_loss_param, w.geo_weight, prec=prec)
    e, n = orc.geo_error(R10, t10, a.bias, a.basis, a.code, D1, w.mask, a.loc1d, a.homo, a.scale, w.cams[0],
                         w.eps, w.geo_loss_param, w.geo_weight, prec=prec)
    return dict(error=e, num_inliers=n)


def presample_source(orc_or_none, w, kf):
    """[L,N,FS] source features at the sampled pixels (tracker pre-sampling, camera_tracker.cpp:1092-1123);
    numpy bilinear with zero padding == grid_sample(align_corners=False)."""
    L, FS = w.L, w.FS
    N = kf.homo.shape[0]
    out = np.zeros((L, N, FS), np.float32)
    lx = (kf.loc1d % w.W).astype(np.float32); ly = (kf.loc1d // w.W).astype(np.float32)
    for l, cam in enumerate(w.cams):
        Wl, Hl = int(cam.w), int(cam.h)
        u = (lx + np.float32(0.5)) * np.float32(Wl / w.W) - np.float32(0.5)
        v = (ly + np.float32(0.5)) * np.float32(Hl / w.H) - np.float32(0.5)
        xf = np.floor(u).astype(int); yf = np.floor(v).astype(int)
        img = kf.feat_pyr[:, w.level_offsets[l]:w.level_offsets[l] + Wl * Hl].reshape(FS, Hl, Wl)
        acc = np.zeros((FS, N), np.float32)
        for dx, dy in ((0, 0), (1, 1), (0, 1), (1, 0)):
            x = xf + dx; y = yf + dy
            wx = (1 - np.abs(u - x)).astype(np.float32); wy = (1 - np.abs(v - y)).astype(np.float32)
            ok = (x >= 0) & (x < Wl) & (y >= 0) & (y < Hl)
            xs = np.clip(x, 0, Wl - 1); ys = np.clip(y, 0, Hl - 1)
            acc += np.where(ok, wx * wy, 0).astype(np.float32) * img[:, ys, xs]
        out[l] = acc.T
    return out


def damped_delta(H, g, damp):
    """(H + damp*diag(H)) d = g in fp64 (reference LM damping, camera_tracker.cpp:1182)."""
    Hd = H + damp * np.diag(np.diag(H))
    return np.linalg.solve(Hd, g)
