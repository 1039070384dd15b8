"""ctypes front-end of the CPU oracle (``oracle/sage_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg; never by anything under ``sage_slam_amd/``.
See ``oracle/sage_oracle.h`` for the parity-pinning status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, seconds).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("sage_oracle.c", "sage_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def set_threads(n: int) -> None:
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def _dt(prec):
    return (np.float32, C.c_float, "_f32") if prec == "f32" else (np.float64, C.c_double, "_f64")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _cams(cams, dt):
    """list of objects with fx,fy,cx,cy,w,h (or [L,6] array) -> [L,6] array of REAL."""
    if isinstance(cams, np.ndarray):
        return _arr(cams.reshape(-1, 6), dt)
    return _arr([[c.fx, c.fy, c.cx, c.cy, c.w, c.h] for c in cams], dt)


def camera_pyramid(base6, levels, prec="f32"):
    dt, ct, sfx = _dt(prec)
    base = _arr(base6, dt)
    out = np.zeros((levels, 6), dtype=dt)
    getattr(lib(), "orc_camera_pyramid" + sfx)(_p(base), C.c_int(levels), _p(out))
    return out


def photo_jac_error(R10, t10, R0, t0, R1, t1, bias0, basis0, code0, mask1, loc1d, homo,
                    feat0, feat1, grad1, level_offsets, scale0, cams, eps, weights,
                    prec="f32", want_rows=False):
    dt, ct, sfx = _dt(prec)
    cams = _cams(cams, dt)
    L = cams.shape[0]
    homo = _arr(homo, dt)
    N = homo.shape[0]
    feat0 = _arr(feat0, dt)
    FS, P = feat0.shape
    basis0 = _arr(basis0, dt)
    CS = basis0.shape[1]
    D = 13 + CS
    AtA = np.zeros((D, D), dtype=dt)
    Atb = np.zeros((D,), dtype=dt)
    err = ct(0)
    nin = ct(0)
    J = np.zeros((L, N, FS, D), dtype=dt) if want_rows else None
    r = np.zeros((L, N, FS), dtype=dt) if want_rows else None
    e = np.zeros((L, N), dtype=dt) if want_rows else None
    v = np.zeros((L, N), dtype=dt) if want_rows else None
    args = [_arr(x, dt) for x in (R10, t10, R0, t0, R1, t1, bias0)]
    code0 = _arr(code0, dt)
    mask1 = _arr(mask1, dt)
    loc1d = _arr(loc1d, np.int64)
    feat1 = _arr(feat1, dt)
    grad1 = _arr(grad1, dt)
    lo = _arr(level_offsets, np.int32)
    w = _arr(weights, dt)
    getattr(lib(), "orc_photo_jac_error" + sfx)(
        _p(AtA), _p(Atb), C.byref(err), C.byref(nin),
        *[_p(a) for a in args], _p(basis0), _p(code0), _p(mask1), _p(loc1d), _p(homo),
        _p(feat0), _p(feat1), _p(grad1), _p(lo), ct(scale0), _p(cams),
        C.c_int(L), C.c_int(N), C.c_int(FS), C.c_int(CS), C.c_int(P), ct(eps), _p(w),
        _p(J), _p(r), _p(e), _p(v))
    out = dict(AtA=AtA, Atb=Atb, error=float(err.value), num_inliers=float(nin.value))
    if want_rows:
        out.update(J=J, r=r, err_rows=e, valid=v)
    return out


def photo_error(R10, t10, bias0, basis0, code0, mask1, loc1d, homo, feat0, feat1,
                level_offsets, scale0, cams, eps, weights, prec="f32"):
    dt, ct, sfx = _dt(prec)
    cams = _cams(cams, dt)
    L = cams.shape[0]
    homo = _arr(homo, dt)
    N = homo.shape[0]
    feat0 = _arr(feat0, dt)
    FS, P = feat0.shape
    basis0 = _arr(basis0, dt)
    CS = basis0.shape[1]
    nin = ct(0)
    f = getattr(lib(), "orc_photo_error" + sfx)
    f.restype = ct
    a = [_arr(x, dt) for x in (R10, t10, bias0)]
    code0 = _arr(code0, dt); mask1 = _arr(mask1, dt); loc1d = _arr(loc1d, np.int64)
    feat1 = _arr(feat1, dt); lo = _arr(level_offsets, np.int32); w = _arr(weights, dt)
    e = f(*[_p(x) for x in a], _p(basis0), _p(code0), _p(mask1), _p(loc1d), _p(homo),
          _p(feat0), _p(feat1), _p(lo), ct(scale0), _p(cams),
          C.c_int(L), C.c_int(N), C.c_int(FS), C.c_int(CS), C.c_int(P), ct(eps), _p(w), C.byref(nin))
    return float(e), float(nin.value)


def tracker_photo_jac_error(dof, R, t, mask1, dpts0, homo, feat0s, feat1, grad1,
                            level_offsets, cams, eps, weights, scale0=1.0, prec="f32", want_rows=False):
    dt, ct, sfx = _dt(prec)
    cams = _cams(cams, dt)
    L = cams.shape[0]
    homo = _arr(homo, dt)
    N = homo.shape[0]
    feat1 = _arr(feat1, dt)
    FS, P = feat1.shape
    AtA = np.zeros((dof, dof), dtype=dt)
    Atb = np.zeros((dof,), dtype=dt)
    err = ct(0); nin = ct(0)
    J = np.zeros((L, N, FS, dof), dtype=dt) if want_rows else None
    r = np.zeros((L, N, FS), dtype=dt) if want_rows else None
    R = _arr(R, dt); t = _arr(t, dt); mask1 = _arr(mask1, dt); dpts0 = _arr(dpts0, dt)
    feat0s = _arr(feat0s, dt); grad1 = _arr(grad1, dt)
    lo = _arr(level_offsets, np.int32); w = _arr(weights, dt)
    getattr(lib(), "orc_tracker_photo_jac_error" + sfx)(
        _p(AtA), _p(Atb), C.byref(err), C.byref(nin), C.c_int(dof),
        _p(R), _p(t), _p(mask1), _p(dpts0), _p(homo), _p(feat0s), _p(feat1), _p(grad1),
        _p(lo), _p(cams), C.c_int(L), C.c_int(N), C.c_int(FS), C.c_int(P),
        ct(scale0), ct(eps), _p(w), _p(J), _p(r))
    out = dict(AtA=AtA, Atb=Atb, error=float(err.value), num_inliers=float(nin.value))
    if want_rows:
        out.update(J=J, r=r)
    return out


def tracker_photo_error(R, t, mask1, dpts0, homo, feat0s, feat1, level_offsets, cams, eps, weights,
                        prec="f32"):
    dt, ct, sfx = _dt(prec)
    cams = _cams(cams, dt)
    L = cams.shape[0]
    homo = _arr(homo, dt)
    N = homo.shape[0]
    feat1 = _arr(feat1, dt)
    FS, P = feat1.shape
    nin = ct(0)
    f = getattr(lib(), "orc_tracker_photo_error" + sfx)
    f.restype = ct
    R = _arr(R, dt); t = _arr(t, dt); mask1 = _arr(mask1, dt); dpts0 = _arr(dpts0, dt)
    feat0s = _arr(feat0s, dt); lo = _arr(level_offsets, np.int32); w = _arr(weights, dt)
    e = f(_p(R), _p(t), _p(mask1), _p(dpts0), _p(homo), _p(feat0s), _p(feat1), _p(lo), _p(cams),
          C.c_int(L), C.c_int(N), C.c_int(FS), C.c_int(P), ct(eps), _p(w), C.byref(nin))
    return float(e), float(nin.value)


def geo_jac_error(R10, t10, R0, t0, R1, t1, bias0, basis0, code0, dpt1, dpt_grad1, basis1, mask1,
                  loc1d, homo, scale0, scale1, cam, eps, loss_param, weight, prec="f32", want_rows=False):
    dt, ct, sfx = _dt(prec)
    cam = _cams([cam] if not isinstance(cam, np.ndarray) else cam, dt)
    homo = _arr(homo, dt)
    N = homo.shape[0]
    basis0 = _arr(basis0, dt)
    CS = basis0.shape[1]
    D = 14 + 2 * CS
    AtA = np.zeros((D, D), dtype=dt)
    Atb = np.zeros((D,), dtype=dt)
    err = ct(0); nin = ct(0)
    J = np.zeros((N, D), dtype=dt) if want_rows else None
    r = np.zeros((N,), dtype=dt) if want_rows else None
    a = [_arr(x, dt) for x in (R10, t10, R0, t0, R1, t1, bias0)]
    code0 = _arr(code0, dt); dpt1 = _arr(dpt1, dt); dpt_grad1 = _arr(dpt_grad1, dt)
    basis1 = _arr(basis1, dt); mask1 = _arr(mask1, dt); loc = _arr(loc1d, np.int32)
    getattr(lib(), "orc_geo_jac_error" + sfx)(
        _p(AtA), _p(Atb), C.byref(err), C.byref(nin), *[_p(x) for x in a],
        _p(basis0), _p(code0), _p(dpt1), _p(dpt_grad1), _p(basis1), _p(mask1), _p(loc), _p(homo),
        ct(scale0), ct(scale1), _p(cam), C.c_int(N), C.c_int(CS), ct(eps), ct(loss_param), ct(weight),
        _p(J), _p(r))
    out = dict(AtA=AtA, Atb=Atb, error=float(err.value), num_inliers=float(nin.value))
    if want_rows:
        out.update(J=J, r=r)
    return out


def geo_error(R10, t10, bias0, basis0, code0, dpt1, mask1, loc1d, homo, scale0, cam, eps,
              loss_param, weight, prec="f32"):
    dt, ct, sfx = _dt(prec)
    cam = _cams([cam] if not isinstance(cam, np.ndarray) else cam, dt)
    homo = _arr(homo, dt)
    N = homo.shape[0]
    basis0 = _arr(basis0, dt)
    CS = basis0.shape[1]
    nin = ct(0)
    f = getattr(lib(), "orc_geo_error" + sfx)
    f.restype = ct
    a = [_arr(x, dt) for x in (R10, t10, bias0)]
    code0 = _arr(code0, dt); dpt1 = _arr(dpt1, dt); mask1 = _arr(mask1, dt); loc = _arr(loc1d, np.int32)
    e = f(*[_p(x) for x in a], _p(basis0), _p(code0), _p(dpt1), _p(mask1), _p(loc), _p(homo),
          ct(scale0), _p(cam), C.c_int(N), C.c_int(CS), ct(eps), ct(loss_param), ct(weight), C.byref(nin))
    return float(e), float(nin.value)


def update_depth(bias, basis, code, scale, prec="f32"):
    dt, ct, sfx = _dt(prec)
    bias = _arr(bias, dt); basis = _arr(basis, dt); code = _arr(code, dt)
    out = np.zeros_like(bias)
    getattr(lib(), "orc_update_depth" + sfx)(_p(out), _p(bias), _p(basis), _p(code), ct(scale),
                                             C.c_int(bias.size), C.c_int(basis.shape[1]))
    return out


def spatial_grad(img, prec="f32"):
    dt, ct, sfx = _dt(prec)
    img = _arr(img, dt)
    Cc, H, W = img.shape
    out = np.zeros((2, Cc, H, W), dtype=dt)
    getattr(lib(), "orc_spatial_grad" + sfx)(_p(out), _p(img), C.c_int(Cc), C.c_int(H), C.c_int(W))
    return out


def gaussian_pyramid_with_grad(feat, mask, L, level_offsets, P, prec="f32"):
    dt, ct, sfx = _dt(prec)
    feat = _arr(feat, dt); mask = _arr(mask, dt)
    FS, H, W = feat.shape
    pyr = np.zeros((FS, P), dtype=dt)
    grad = np.zeros((2, FS, P), dtype=dt)
    lo = _arr(level_offsets, np.int32)
    getattr(lib(), "orc_gaussian_pyramid_with_grad" + sfx)(
        _p(pyr), _p(grad), _p(feat), _p(mask), C.c_int(FS), C.c_int(H), C.c_int(W), C.c_int(L), _p(lo), C.c_int(P))
    return pyr, grad


def se3_exp(omega, v, prec="f32"):
    dt, ct, sfx = _dt(prec)
    omega = _arr(omega, dt); v = _arr(v, dt)
    R = np.zeros((3, 3), dtype=dt); t = np.zeros((3,), dtype=dt)
    getattr(lib(), "orc_se3_exp" + sfx)(_p(omega), _p(v), _p(R), _p(t))
    return R, t


def valid_locations(mask, cam, prec="f32"):
    """mapping_utils.h:254-287 -> (loc1d int64 [n], homo [n,3])."""
    dt, ct, sfx = _dt(prec)
    mask = _arr(mask, dt)
    H, W = mask.shape
    loc = np.zeros(H * W, np.int64); homo = np.zeros((H * W, 3), dt)
    c = _cams([cam], dt)          # OrcCamera = 6 REALs {fx, fy, cx, cy, w, h}
    fn = getattr(lib(), "orc_valid_locations" + sfx)
    fn.restype = C.c_int
    n = fn(loc.ctypes.data_as(C.c_void_p), _p(homo), _p(mask), _p(c))
    return loc[:n].copy(), homo[:n].copy()


def shuffle_indices(n, seed):
    """std::iota + std::mt19937(seed) + std::shuffle as libstdc++ 11 does it (mapper.cpp:1326-1333)."""
    idx = np.zeros(max(n, 1), np.int64)
    lib().orc_shuffle_indices_f32(idx.ctypes.data_as(C.c_void_p), C.c_longlong(n), C.c_longlong(seed))
    return idx[:n].copy()


def reproj_jac_error(R10, t10, R0, t0, R1, t1, bias0, basis0, code0, loc1d, homo, matched, scale0, cam, eps,
                     loss_param, weight, prec="f32", want_rows=False):
    """cuda/reprojection_factor_kernels.cpp:27-213 + :468-531 (mapper factor, D = 13+CS)."""
    dt, ct, sfx = _dt(prec)
    basis0 = _arr(basis0, dt); CS = basis0.shape[-1]; D = 13 + CS
    homo = _arr(homo, dt); N = homo.shape[0]
    AtA = np.zeros((D, D), dt); Atb = np.zeros(D, dt); err = np.zeros(1, dt); nin = np.zeros(1, dt)
    J = np.zeros((max(N, 1), 2, D), dt) if want_rows else None
    r = np.zeros((max(N, 1), 2), dt) if want_rows else None
    sw = np.zeros((max(N, 1), 2), dt) if want_rows else None
    getattr(lib(), "orc_reproj_jac_error" + sfx)(
        _p(AtA), _p(Atb), _p(err), _p(nin), _p(_arr(R10, dt)), _p(_arr(t10, dt)), _p(_arr(R0, dt)), _p(_arr(t0, dt)),
        _p(_arr(R1, dt)), _p(_arr(t1, dt)), _p(_arr(bias0, dt)), _p(basis0), _p(_arr(code0, dt)),
        _p(_arr(loc1d, np.int32)), _p(homo), _p(_arr(matched, dt)), ct(scale0), _p(_cams([cam], dt)), C.c_int(N),
        C.c_int(CS), ct(eps), ct(loss_param), ct(weight), _p(J), _p(r), _p(sw))
    out = dict(AtA=AtA, Atb=Atb, error=float(err[0]), num_inliers=float(nin[0]))
    if want_rows:
        out.update(J=J[:N], r=r[:N], sw=sw[:N])
    return out


def reproj_error(R10, t10, bias0, basis0, code0, loc1d, homo, matched, scale0, cam, eps, loss_param, weight, prec="f32"):
    dt, ct, sfx = _dt(prec)
    basis0 = _arr(basis0, dt); CS = basis0.shape[-1]
    homo = _arr(homo, dt); N = homo.shape[0]
    nin = np.zeros(1, dt)
    fn = getattr(lib(), "orc_reproj_error" + sfx); fn.restype = ct
    e = fn(_p(_arr(R10, dt)), _p(_arr(t10, dt)), _p(_arr(bias0, dt)), _p(basis0), _p(_arr(code0, dt)),
           _p(_arr(loc1d, np.int32)), _p(homo), _p(_arr(matched, dt)), ct(scale0), _p(_cams([cam], dt)), C.c_int(N),
           C.c_int(CS), ct(eps), ct(loss_param), ct(weight), _p(nin))
    return float(e), float(nin[0])


def tracker_reproj_jac_error(R, t, dpts0, homo, matched, cam, eps, loss_param, weight, prec="f32"):
    """cuda/reprojection_factor_kernels.cpp:288-366 + :533-593 (tracker, D = 6)."""
    dt, ct, sfx = _dt(prec)
    homo = _arr(homo, dt); N = homo.shape[0]
    AtA = np.zeros((6, 6), dt); Atb = np.zeros(6, dt); err = np.zeros(1, dt); nin = np.zeros(1, dt)
    getattr(lib(), "orc_tracker_reproj_jac_error" + sfx)(
        _p(AtA), _p(Atb), _p(err), _p(nin), _p(_arr(R, dt)), _p(_arr(t, dt)), _p(_arr(dpts0, dt)), _p(homo),
        _p(_arr(matched, dt)), _p(_cams([cam], dt)), C.c_int(N), ct(eps), ct(loss_param), ct(weight))
    return dict(AtA=AtA, Atb=Atb, error=float(err[0]), num_inliers=float(nin[0]))


def tracker_reproj_error(R, t, dpts0, homo, matched, cam, eps, loss_param, weight, prec="f32"):
    dt, ct, sfx = _dt(prec)
    homo = _arr(homo, dt); N = homo.shape[0]
    nin = np.zeros(1, dt)
    fn = getattr(lib(), "orc_tracker_reproj_error" + sfx); fn.restype = ct
    e = fn(_p(_arr(R, dt)), _p(_arr(t, dt)), _p(_arr(dpts0, dt)), _p(homo), _p(_arr(matched, dt)), _p(_cams([cam], dt)),
           C.c_int(N), ct(eps), ct(loss_param), ct(weight), _p(nin))
    return float(e), float(nin[0])


MG_LOSS = {"fair": 0, "L2": 1, "huber": 2, "unbiased": 3}


def match_geom_jac_error(mode, loss, R10, t10, R0=None, t0=None, R1=None, t1=None, bias0=None, bias1=None, basis0=None,
                         basis1=None, code0=None, code1=None, dpts0=None, dpts1=None, homo0=None, homo1=None, loc0=None,
                         loc1=None, scale0=1.0, scale1=1.0, loss_param=1.0, weight=1.0, CS=0, prec="f32",
                         want_rows=False):
    """cuda/match_geometry_factor_kernels.cpp, all variants (see sage_oracle.h).  mode 0 mapper / 1 loop / 2 tracker /
    3 tracker with scale; loss "fair" | "L2" | "huber" | "unbiased"."""
    dt, ct, sfx = _dt(prec)
    homo0 = _arr(homo0, dt); N = homo0.shape[0]
    if mode == 0:
        CS = np.asarray(basis0).shape[-1]
    D = {0: 14 + 2 * CS, 1: 14, 2: 6, 3: 7}[mode]
    a = lambda x: None if x is None else _arr(x, dt)
    ai = lambda x: None if x is None else _arr(x, np.int32)
    AtA = np.zeros((D, D), dt); Atb = np.zeros(D, dt); err = np.zeros(1, dt)
    J = np.zeros((N, 3, D), dt) if want_rows else None
    r = np.zeros((N, 3), dt) if want_rows else None
    sw = np.zeros((N, 3), dt) if want_rows else None
    keep = [a(R10), a(t10), a(R0), a(t0), a(R1), a(t1), a(bias0), a(bias1), a(basis0), a(basis1), a(code0), a(code1),
            a(dpts0), a(dpts1), homo0, a(homo1), ai(loc0), ai(loc1)]
    getattr(lib(), "orc_match_geom_jac_error" + sfx)(
        _p(AtA), _p(Atb), _p(err), C.c_int(mode), C.c_int(MG_LOSS[loss]), *[_p(k) for k in keep], ct(scale0), ct(scale1),
        C.c_int(N), C.c_int(CS), ct(loss_param), ct(weight), _p(J), _p(r), _p(sw))
    out = dict(AtA=AtA, Atb=Atb, error=float(err[0]))
    if want_rows:
        out.update(J=J, r=r, sw=sw)
    return out


def match_geom_error(mode, loss, R10, t10, bias0=None, bias1=None, basis0=None, basis1=None, code0=None, code1=None,
                     dpts0=None, dpts1=None, homo0=None, homo1=None, loc0=None, loc1=None, scale0=1.0, scale1=1.0,
                     loss_param=1.0, weight=1.0, CS=0, prec="f32"):
    dt, ct, sfx = _dt(prec)
    homo0 = _arr(homo0, dt); N = homo0.shape[0]
    if mode == 0:
        CS = np.asarray(basis0).shape[-1]
    a = lambda x: None if x is None else _arr(x, dt)
    ai = lambda x: None if x is None else _arr(x, np.int32)
    keep = [a(R10), a(t10), a(bias0), a(bias1), a(basis0), a(basis1), a(code0), a(code1), a(dpts0), a(dpts1), homo0,
            a(homo1), ai(loc0), ai(loc1)]
    fn = getattr(lib(), "orc_match_geom_error" + sfx); fn.restype = ct
    return float(fn(C.c_int(mode), C.c_int(MG_LOSS[loss]), *[_p(k) for k in keep], ct(scale0), ct(scale1), C.c_int(N),
                    C.c_int(CS), ct(loss_param), ct(weight)))


def cycle_match(desc0, desc1, kp_loc0, H, W, cyc_thresh, prec="f32"):
    """match_geometry_factor.cpp:62-97 / camera_tracker.cpp:608-633 -> (raw_matched1, cyc_matched0, inlier flags)."""
    dt, ct, sfx = _dt(prec)
    desc0 = _arr(desc0, dt).reshape(-1, H * W); desc1 = _arr(desc1, dt).reshape(-1, H * W)
    kp = np.ascontiguousarray(kp_loc0, np.int64); K = kp.shape[0]
    m1 = np.zeros(max(K, 1), np.int64); c0 = np.zeros(max(K, 1), np.int64); fl = np.zeros(max(K, 1), np.int32)
    fn = getattr(lib(), "orc_cycle_match" + sfx); fn.restype = C.c_int
    fn(m1.ctypes.data_as(C.c_void_p), c0.ctypes.data_as(C.c_void_p), fl.ctypes.data_as(C.c_void_p), _p(desc0), _p(desc1),
       kp.ctypes.data_as(C.c_void_p), C.c_int(K), C.c_int(desc0.shape[0]), C.c_int(H), C.c_int(W), ct(cyc_thresh))
    return m1[:K], c0[:K], fl[:K]
