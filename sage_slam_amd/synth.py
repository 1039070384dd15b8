"""Synthetic keyframe windows for the dense-BA hot path (tests + bench data).

The reference ships no TorchScript nets and no data, so
every test and benchmark runs on synthetic keyframes built here, in the layouts
the reference's ``Frame``/``Keyframe`` structs hold (SURVEY.md s8 a10,
``core/mapping/frame.h:17-125``):

* ``feat_pyr  [FS, P]``      feature pyramid, levels concatenated fine->coarse
* ``grad_pyr  [2, FS, P]``   central-difference gradients (x then y)
* ``bias      [H*W]``, ``basis [H*W, CS]``, ``code [CS]``, ``scale``
* ``loc1d     [N]`` int64, ``homo [N, 3]`` sampled source pixels
* shared ``mask [H, W]`` (destination validity), ``level_offsets [L]``, camera pyramid
* ``pose_wk`` = world-from-keyframe ``(R, t)``

The scene is a textured plane seen from cameras on a smooth arc: feature maps are
geometrically consistent across keyframes (``feat_k(pixel) = F(world point)``) and
the true depth lies in the span of the linear depth code, so LM on the window
actually converges (the survey's "realistic gradients matter" note, s7).

This module is host-side *data generation* (numpy); it is not on the hot path.
The pyramid/gradient producer mirrors ``Mapper::GenerateGaussianPyramidWithGrad``
(``core/mapping/mapper.cpp:1384-1426``) and ``ComputeSpatialGrad``
(``core/mapping/mapping_utils.h:236-252``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- cameras
@dataclass
class Camera:
    """``PinholeCamera<float>`` (``common/pinhole_camera.h:44-131``)."""
    fx: float
    fy: float
    cx: float
    cy: float
    w: float
    h: float

    def as_array(self) -> np.ndarray:
        return np.array([self.fx, self.fy, self.cx, self.cy, self.w, self.h], dtype=F32)


def camera_pyramid(cam: Camera, levels: int) -> List[Camera]:
    """``CameraPyramid`` ctor (``common/camera_pyramid.h:18-32``): level i is level i-1
    resized to ``(size_t)(w/2), (size_t)(h/2)`` with fx,u0 / fy,v0 scaled by the ratio
    (``pinhole_camera_impl.h:120-132``), all in fp32."""
    out = [Camera(*[F32(v) for v in (cam.fx, cam.fy, cam.cx, cam.cy, cam.w, cam.h)])]
    for _ in range(1, levels):
        p = out[-1]
        nw = int(F32(p.w) / F32(2))
        nh = int(F32(p.h) / F32(2))
        xr = F32(nw) / F32(p.w)
        yr = F32(nh) / F32(p.h)
        out.append(Camera(F32(p.fx * xr), F32(p.fy * yr), F32(p.cx * xr), F32(p.cy * yr), F32(nw), F32(nh)))
    return out


def level_offsets_of(cams: List[Camera]) -> Tuple[np.ndarray, int]:
    sizes = [int(c.w) * int(c.h) for c in cams]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    return offs, int(sum(sizes))


# --------------------------------------------------------------------------- producers
def spatial_grad(img: np.ndarray) -> np.ndarray:
    """img [C,H,W] -> [2,C,H,W]; 0.5*(next-prev) with replicate padding, x then y."""
    p = np.pad(img, ((0, 0), (1, 1), (1, 1)), mode="edge")
    gx = F32(0.5) * (p[:, 1:-1, 2:] - p[:, 1:-1, :-2])
    gy = F32(0.5) * (p[:, 2:, 1:-1] - p[:, :-2, 1:-1])
    return np.stack([gx, gy], 0).astype(F32)


_G = (np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], dtype=F32) / F32(16.0)).astype(F32)


def _conv_s2(img: np.ndarray) -> np.ndarray:
    """3x3 Gaussian, stride 2, zero pad 1 on [C,H,W] -> [C, H // 2, W // 2] (an odd last row / column is dropped: the size the
    camera pyramid gives the level, ``common/camera_pyramid.h:26-27``)."""
    C, H, W = img.shape
    p = np.pad(img, ((0, 0), (1, 1), (1, 1)))
    h2, w2 = H // 2, W // 2
    out = np.zeros((C, h2, w2), dtype=F32)
    for dy in range(3):
        for dx in range(3):
            out += _G[dy, dx] * p[:, dy:dy + 2 * h2:2, dx:dx + 2 * w2:2]
    return out


def gaussian_pyramid_with_grad(feat: np.ndarray, mask: np.ndarray, levels: int, allow_odd: bool = False):
    """feat [FS,H,W], mask [H,W] -> (pyr [FS,P], grad [2,FS,P]).  Masked stride-2
    Gaussian: level k+1 = conv(level_k * mask_k) / (conv(mask_k) + 1e-8); the mask
    pyramid is nearest-neighbour decimation (source index 2i)."""
    FS, H, W = feat.shape
    # (allow_odd: test inputs with NON-DYADIC camera pyramids, w = 62 -> 31 -> 15 -- levels cropped to the camera pyramid's
    #  floor sizes; the reference's own producer disagrees with its camera pyramid there, the factor kernels do not care)
    assert allow_odd or (H % (1 << (levels - 1)) == 0 and W % (1 << (levels - 1)) == 0), \
        "level sizes must stay even (the reference's conv/camera/mask pyramids only agree then)"
    cur = feat.astype(F32)
    curm = mask.astype(F32)
    pyr, grad = [], []
    for l in range(levels):
        g = spatial_grad(cur)
        pyr.append(cur.reshape(FS, -1))
        grad.append(g.reshape(2, FS, -1))
        if l == levels - 1:
            break
        raw = _conv_s2(cur * curm[None])
        rm = _conv_s2(curm[None])
        cur = (raw / (rm + F32(1.0e-8))).astype(F32)
        curm = curm[::2, ::2][:cur.shape[1], :cur.shape[2]].copy()
    return np.ascontiguousarray(np.concatenate(pyr, 1)), np.ascontiguousarray(np.concatenate(grad, 2))


# --------------------------------------------------------------------------- SE3 helpers (fp64 on host)
def so3_exp(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


# --------------------------------------------------------------------------- data model
@dataclass
class Keyframe:
    feat_pyr: np.ndarray
    grad_pyr: np.ndarray
    bias: np.ndarray
    basis: np.ndarray
    code: np.ndarray
    scale: float
    loc1d: np.ndarray
    homo: np.ndarray
    R: np.ndarray
    t: np.ndarray
    # ground truth (for convergence tests)
    code_true: np.ndarray = None
    scale_true: float = 1.0
    R_true: np.ndarray = None
    t_true: np.ndarray = None


@dataclass
class Window:
    H: int
    W: int
    L: int
    FS: int
    CS: int
    cams: List[Camera]
    level_offsets: np.ndarray
    P: int
    mask: np.ndarray                      # [H,W] video mask (destination validity)
    keyframes: List[Keyframe]
    links: List[Tuple[int, int]] = field(default_factory=list)   # undirected (older, newer)
    photo_weights: np.ndarray = None      # [L]
    geo_weight: float = 0.1
    geo_loss_param: float = 0.03
    eps: float = 1.0e-4

    def directed_edges(self) -> List[Tuple[int, int]]:
        """Each link contributes both directions (``core/mapping/mapper.cpp:346-374``)."""
        out = []
        for a, b in self.links:
            out.append((a, b))
            out.append((b, a))
        return out


def _texture(rng, FS, n_waves, kmin, kmax):
    ang = rng.uniform(0, 2 * np.pi, (FS, n_waves))
    kk = rng.uniform(kmin, kmax, (FS, n_waves))
    k = np.stack([kk * np.cos(ang), kk * np.sin(ang)], -1)          # [FS,J,2] cycles / world unit
    phase = rng.uniform(0, 2 * np.pi, (FS, n_waves))
    amp = rng.uniform(0.5, 1.0, (FS, n_waves))
    amp /= np.sqrt((amp ** 2).sum(1, keepdims=True))
    return k, phase, amp


def _smooth_fields(rng, n, H, W, sigma, max_cycles=2.0, n_waves=4):
    """n smooth random fields over the image, std ~= sigma."""
    yy, xx = np.meshgrid(np.arange(H) / H, np.arange(W) / W, indexing="ij")
    out = np.zeros((n, H, W))
    for i in range(n):
        f = np.zeros((H, W))
        for _ in range(n_waves):
            kx, ky = rng.uniform(-max_cycles, max_cycles, 2)
            f += rng.uniform(0.5, 1.0) * np.sin(2 * np.pi * (kx * xx + ky * yy) + rng.uniform(0, 2 * np.pi))
        out[i] = f / (f.std() + 1e-12) * sigma
    return out


def make_window(K: int, H: int, W: int, FS: int = 16, CS: int = 32, L: int = 4,
                n_samples: int = 0, back_links: int = 3, seed: int = 0,
                baseline: float = 0.025, pose_noise: float = 1.0, code_noise: float = 0.03,
                border: int = 2, erode: int = 6, loop_radius: float = 0.0, allow_odd: bool = False) -> Window:
    """Build a K-keyframe window (SURVEY.md s8d "synthetic inputs").

    ``n_samples == 0`` -> dense sampling (all pixels of the eroded mask, row-major like
    ``torch.nonzero``); otherwise a seeded shuffle keeps the first ``n_samples``
    (``core/mapping/mapper.cpp:1222-1237``).  ``loop_radius > 0``: the camera walks once around a circle of that
    radius in the x-y plane (keyframe K-1 ends next to keyframe 0: loop closures have overlap) instead of along the
    smooth arc."""
    rng = np.random.default_rng(seed)
    cam0 = Camera(0.9 * W, 0.9 * W, W / 2.0, H / 2.0, W, H)
    cams = camera_pyramid(cam0, L)
    offs, P = level_offsets_of(cams)
    fx, fy, cx, cy = float(cams[0].fx), float(cams[0].fy), float(cams[0].cx), float(cams[0].cy)

    # masks: destination validity = video mask, sampling from the eroded one
    mask = np.zeros((H, W), dtype=F32)
    mask[border:H - border, border:W - border] = 1
    smask = np.zeros((H, W), dtype=bool)
    smask[border + erode:H - border - erode, border + erode:W - border - erode] = True
    valid = np.nonzero(smask.reshape(-1))[0].astype(np.int64)

    # scene: tilted plane n.X = h, texture on in-plane coordinates
    n = np.array([0.08, -0.05, 1.0])
    n /= np.linalg.norm(n)
    h = 1.0
    e1 = np.cross(n, [0, 1, 0]); e1 /= np.linalg.norm(e1)
    e2 = np.cross(n, e1)
    px = h / fx                                   # world units per pixel at depth ~h
    kvec, phase, amp = _texture(rng, FS, 6, 1.0 / (48 * px), 1.0 / (10 * px))

    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    rays = np.stack([(xx - cx) / fx, (yy - cy) / fy, np.ones_like(xx)], -1)      # [H,W,3]

    kfs: List[Keyframe] = []
    for k in range(K):
        # smooth arc, small rotations that keep the plane in view
        s = k * baseline
        t_true = np.array([s, 0.15 * baseline * np.sin(0.7 * k), 0.05 * baseline * np.cos(0.3 * k)])
        if loop_radius > 0:
            th = 2 * np.pi * k / K
            t_true = np.array([loop_radius * np.sin(th), loop_radius * (1 - np.cos(th)), 0.05 * baseline * np.cos(0.3 * k)])
        R_true = so3_exp(np.array([0.01 * np.sin(0.5 * k), -0.02 * np.sin(0.2 * k), 0.015 * np.sin(0.3 * k)]))
        dirs = rays @ R_true.T
        depth = (h - n @ t_true) / (dirs @ n)                                  # z-depth along the ray
        Xw = dirs * depth[..., None] + t_true
        uv = np.stack([Xw @ e1, Xw @ e2], -1)                                   # [H,W,2]
        arg = 2 * np.pi * np.einsum("hwd,cjd->cjhw", uv, kvec) + phase[:, :, None, None]
        feat = (amp[:, :, None, None] * np.sin(arg)).sum(1).astype(F32)         # [FS,H,W] in [-1,1]
        pyr, grad = gaussian_pyramid_with_grad(feat, mask, L, allow_odd)

        krng = np.random.default_rng(seed * 100003 + k)                         # seed = kf id
        basis = _smooth_fields(krng, CS, H, W, 0.05).reshape(CS, -1).T          # [HW,CS]
        code_true = krng.normal(0, 0.1, CS)
        scale_true = float(np.median(depth))
        bias = depth.reshape(-1) / scale_true - basis @ code_true

        if n_samples and n_samples < valid.size:
            perm = krng.permutation(valid.size)[:n_samples]
            loc = valid[perm]
        else:
            loc = valid
        import os as _os
        if _os.environ.get("SAGE_SYNTH_SHUFFLE") == "1":                        # dev experiment: same samples, shuffled order
            loc = loc[krng.permutation(loc.size)]
        _tile = int(_os.environ.get("SAGE_SYNTH_TILE_ORDER", "0"))              # dev experiment: 2-D tile order of the samples
        if _tile:
            ty, tx = (loc // W) // _tile, (loc % W) // _tile
            loc = loc[np.lexsort((loc % W, loc // W, tx, ty))]
        lx = (loc % W).astype(np.float64)
        ly = (loc // W).astype(np.float64)
        homo = np.stack([(lx - cx) / fx, (ly - cy) / fy, np.ones_like(lx)], -1)

        # perturbed initial estimate
        dw = krng.normal(0, 0.003 * pose_noise, 3)
        dv = krng.normal(0, 0.004 * pose_noise, 3)
        if k == 0:
            dw[:] = 0
            dv[:] = 0
        R0 = so3_exp(dw) @ R_true
        t0 = t_true + dv
        code0 = code_true + krng.normal(0, code_noise, CS)
        scale0 = scale_true * (1 + krng.normal(0, 0.01 * pose_noise))

        kfs.append(Keyframe(
            feat_pyr=pyr, grad_pyr=grad,
            bias=bias.astype(F32), basis=np.ascontiguousarray(basis, dtype=F32),
            code=code0.astype(F32), scale=float(F32(scale0)),
            loc1d=loc.astype(np.int64), homo=homo.astype(F32),
            R=R0.astype(F32), t=t0.astype(F32),
            code_true=code_true.astype(F32), scale_true=scale_true,
            R_true=R_true.astype(F32), t_true=t_true.astype(F32)))

    links = [(j, i) for i in range(K) for j in range(max(0, i - back_links), i)]
    bias2 = float(np.mean([np.mean((kf.scale * kf.bias) ** 2) for kf in kfs]))
    return Window(H=H, W=W, L=L, FS=FS, CS=CS, cams=cams, level_offsets=offs, P=P, mask=mask,
                  keyframes=kfs, links=links,
                  photo_weights=np.array([10, 9, 8, 7, 6, 5][:L], dtype=F32),
                  geo_weight=0.1, geo_loss_param=float(0.03 * bias2), eps=1.0e-4)


def relative_pose(R0, t0, R1, t1):
    """``R10 = R1^T R0``, ``t10 = R1^T (t0 - t1)`` in fp32
    (``core/gtsam/photometric_factor.cpp:280-281``)."""
    R0 = R0.astype(F32); R1 = R1.astype(F32)
    return (R1.T @ R0).astype(F32), (R1.T @ (t0.astype(F32) - t1.astype(F32))).astype(F32)


def depth_and_grad(kf: Keyframe, H: int, W: int):
    """Caller-side precompute of the geometric factor (``geometric_factor.cpp:317-347``):
    ``D1 = s1*(bias1 + basis1*code1)`` and ``gradD1 = s1*centraldiff(bias1 + basis1*code1)``."""
    unscaled = (kf.bias + kf.basis @ kf.code).astype(F32).reshape(1, H, W)
    g = spatial_grad(unscaled)[:, 0]
    return (F32(kf.scale) * unscaled[0]).astype(F32), (F32(kf.scale) * g).astype(F32)
