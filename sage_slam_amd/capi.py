"""ctypes binding of the C ABI (``include/sage_ba.h``) + thin torch plumbing.

PyTorch is used only for device memory (``torch.Tensor.data_ptr()``), streams and
``torch.distributed``; every compute entry point goes through ``libsage_ba.so``.
There is no CPU fallback: if the library is missing or no HIP device is present the
calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsage_ba.so")
SAGE_MAX_LEVELS = 8


class SageError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        msg = lib().sage_error_string(code).decode() if _lib is not None else "?"
        super().__init__(f"{where}: {msg} (code {code})")


class SageCamera(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "w", "h")]


class SagePyramid(C.Structure):
    _fields_ = [("levels", C.c_int32), ("P", C.c_int32), ("level_offsets", C.c_int32 * SAGE_MAX_LEVELS),
                ("cam", SageCamera * SAGE_MAX_LEVELS)]


class SageLmConfig(C.Structure):
    _fields_ = [("max_num_iters", C.c_int), ("min_grad_thresh", C.c_float), ("min_param_inc_thresh", C.c_float),
                ("init_damp", C.c_float), ("min_damp", C.c_float), ("max_damp", C.c_float),
                ("damp_dec_factor", C.c_float), ("damp_inc_factor", C.c_float),
                ("jac_update_err_inc_threshold", C.c_float), ("max_inner_evals", C.c_int),
                ("no_overlap_error", C.c_float), ("linearize_at_candidate", C.c_int)]


class SageLmTraceEntry(C.Structure):
    _fields_ = [("damp", C.c_float), ("error", C.c_float), ("candidate_error", C.c_float),
                ("accepted", C.c_int), ("relinearized", C.c_int)]


class SageTrackProblem(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("use_photo", C.c_int32), ("mask1_dev", C.c_void_p), ("dpts0_dev", C.c_void_p),
                ("homo_dev", C.c_void_p), ("feat0s_dev", C.c_void_p), ("feat1_dev", C.c_void_p),
                ("grad1_dev", C.c_void_p), ("weights_dev", C.c_void_p), ("pyr", SagePyramid), ("eps", C.c_float),
                ("N", C.c_int32), ("FS", C.c_int32), ("use_keypoints", C.c_int32), ("NK", C.c_int32),
                ("kp_dpts0_dev", C.c_void_p), ("kp_homo0_dev", C.c_void_p), ("kp_matched_2d_dev", C.c_void_p),
                ("kp_matched_dpts1_dev", C.c_void_p), ("kp_matched_homo1_dev", C.c_void_p),
                ("kp_loss_param", C.c_float), ("kp_weight", C.c_float)]


class SageKeyframeView(C.Structure):
    _fields_ = [("feat_pyr", C.c_void_p), ("grad_pyr", C.c_void_p), ("bias", C.c_void_p), ("basis", C.c_void_p),
                ("loc1d", C.c_void_p), ("homo", C.c_void_p), ("N", C.c_int32)]


class SageWindowConfig(C.Structure):
    _fields_ = [("pyr", SagePyramid), ("FS", C.c_int32), ("CS", C.c_int32), ("mask_dev", C.c_void_p),
                ("photo_weights", C.c_float * SAGE_MAX_LEVELS), ("geo_weight", C.c_float),
                ("geo_loss_param", C.c_float), ("eps", C.c_float), ("code_prior_weight", C.c_float),
                ("scale_prior_weight", C.c_float), ("pose_prior_weight", C.c_float),
                ("use_photo", C.c_int32), ("use_geo", C.c_int32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)


class SageLmState(C.Structure):
    _fields_ = [("damp", C.c_double), ("error", C.c_double), ("candidate_error", C.c_double),
                ("accepted", C.c_int), ("iters", C.c_int)]


TRACK_LIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float),
                           C.POINTER(C.c_float), C.POINTER(C.c_float))
TRACK_ERR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float))

_lib = None

# every symbol include/sage_ba.h declares (checked by tests/test_capi_symbols.py against the header)
SYMBOLS = [
    "sage_camera_pyramid", "sage_version", "sage_error_string", "sage_workspace_create", "sage_workspace_destroy",
    "sage_photometric_jac_error_calculate", "sage_photometric_error_calculate",
    "sage_tracker_photo_jac_error_calculate", "sage_tracker_photo_error_calculate",
    "sage_geometric_jac_error_calculate", "sage_geometric_error_calculate", "sage_depth_and_grad",
    "sage_gaussian_pyramid_with_grad", "sage_se3_exp", "sage_pose_retract", "sage_nearest_psd", "sage_nearest_psd_reference", "sage_factor_block_count", "sage_factor_hessian_blocks",
    "sage_damped_solve_qr_f32", "sage_block_solve", "sage_solve_lookahead_count", "sage_lm_config_default", "sage_track_lm", "sage_track_frame",
    "sage_window_create", "sage_window_destroy", "sage_window_add_keyframe", "sage_window_add_link", "sage_window_set_link_geo_loss",
    "sage_window_set_shard", "sage_window_finalize", "sage_window_num_keyframes", "sage_window_num_links",
    "sage_window_block_size", "sage_window_packed_count", "sage_window_packed_dev",
    "sage_window_residuals_per_linearize", "sage_window_bytes_per_linearize", "sage_window_linearize",
    "sage_window_error", "sage_window_tune_runs", "sage_window_set_runs", "sage_window_error_dev", "sage_window_solve", "sage_window_total_error",
    "sage_window_accept", "sage_window_reset", "sage_window_get_keyframe", "sage_window_set_keyframe", "sage_window_get_delta",
    "sage_window_get_edge", "sage_window_prepass", "sage_window_factor", "sage_window_factor_error", "sage_window_prepare_factors", "sage_factor_psd", "sage_factor_cut_blocks", "sage_window_set_profiling", "sage_window_get_kernel_time", "sage_window_get_phase_time", "sage_window_lm_step", "sage_window_lm_run", "sage_window_lm_run_timed", "sage_window_sync_variables", "sage_window_set_allreduce", "sage_shard_plan_create", "sage_shard_plan_create_domains", "sage_block_solve_domains", "sage_shard_plan_destroy", "sage_shard_sep_count", "sage_shard_num_separators", "sage_shard_num_interior", "sage_shard_keyframe_owner", "sage_shard_keyframe_is_local", "sage_shard_eliminate", "sage_shard_solve", "sage_rccl_unique_id", "sage_rccl_comm_create", "sage_rccl_comm_destroy", "sage_rccl_comm_info", "sage_window_use_rccl", "sage_window_emulate_peers", "sage_sort_locations", "sage_bind_thread_to_device", "sage_solver_helper_cpus", "sage_solver_placement_moves", "sage_placement_monitor", "sage_shutdown", "sage_host_threads_running",
    "sage_valid_locations", "sage_shuffle_indices", "sage_sample_locations",
    "sage_reprojection_jac_error_calculate", "sage_reprojection_error_calculate",
    "sage_tracker_reproj_jac_error_calculate", "sage_tracker_reproj_error_calculate",
    "sage_match_geometry_jac_error_calculate", "sage_match_geometry_error_calculate",
    "sage_loop_mg_jac_error_calculate", "sage_loop_mg_error_calculate",
    "sage_tracker_match_geom_jac_error_calculate", "sage_tracker_match_geom_error_calculate", "sage_cycle_match",
]


def lib():
    """Load ``libsage_ba.so`` (built in-tree by ``sage_slam_amd.build``); fails loudly when absent."""
    global _lib
    if _lib is None:
        LIB_PATH = os.environ.get("SAGE_BA_LIB", globals()["LIB_PATH"])   # dev: A/B a variant build
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -m sage_slam_amd.build` "
                              "(the HIP engine has no fallback path)")
        # torch first: it bundles its own libamdhip64 (same SONAME); loading it before our library makes both
        # share ONE HIP runtime, so torch device pointers are valid in our launches.  (A C++ host without torch
        # simply binds /opt/rocm's runtime.)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.sage_version.restype = C.c_char_p
        L.sage_error_string.restype = C.c_char_p
        L.sage_window_packed_count.restype = C.c_size_t
        L.sage_shutdown.restype = None
        L.sage_window_packed_dev.restype = C.c_void_p
        L.sage_window_error_dev.restype = C.c_void_p
        L.sage_window_residuals_per_linearize.restype = C.c_double
        L.sage_window_bytes_per_linearize.restype = C.c_double
        for name in ("sage_window_packed_count", "sage_window_packed_dev", "sage_window_error_dev",
                     "sage_window_residuals_per_linearize", "sage_window_bytes_per_linearize",
                     "sage_window_num_keyframes", "sage_window_num_links", "sage_window_block_size"):
            getattr(L, name).argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _chk(code: int, where: str):
    if code != 0:
        raise SageError(code, where)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def dptr(t) -> C.c_void_p:
    """device pointer of a torch tensor (must be contiguous)."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "device buffers handed to the C ABI must be contiguous"
    return C.c_void_p(t.data_ptr())


# --------------------------------------------------------------------------- host helpers
def make_pyramid(cam, levels: int) -> SagePyramid:
    """``CameraPyramid`` (``common/camera_pyramid.h:18-32``) through the C ABI."""
    base = SageCamera(*[float(v) for v in (cam.fx, cam.fy, cam.cx, cam.cy, cam.w, cam.h)])
    out = SagePyramid()
    _chk(lib().sage_camera_pyramid(C.byref(base), levels, C.byref(out)), "sage_camera_pyramid")
    return out


def se3_exp(omega, v):
    omega = _f32(omega); v = _f32(v)
    R = np.zeros(9, np.float32); t = np.zeros(3, np.float32)
    lib().sage_se3_exp(_fp(omega), _fp(v), _fp(R), _fp(t))
    return R.reshape(3, 3), t


def pose_retract(pose12, delta6):
    pose12 = _f32(pose12).reshape(12); delta6 = _f32(delta6)
    out = np.zeros(12, np.float32)
    lib().sage_pose_retract(_fp(pose12), _fp(delta6), _fp(out))
    return out


def nearest_psd(M):
    M = np.ascontiguousarray(M, dtype=np.float64)
    out = np.zeros_like(M)
    _chk(lib().sage_nearest_psd(M.ctypes.data_as(C.POINTER(C.c_double)), M.shape[0],
                                out.ctypes.data_as(C.POINTER(C.c_double))), "sage_nearest_psd")
    return out


def nearest_psd_reference(M):
    M = np.ascontiguousarray(M, np.float64)
    out = np.zeros_like(M)
    _chk(lib().sage_nearest_psd_reference(M.ctypes.data_as(C.POINTER(C.c_double)), M.shape[0],
                                          out.ctypes.data_as(C.POINTER(C.c_double))), "sage_nearest_psd_reference")
    return out


def factor_hessian_blocks(type_: int, CS: int, AtA, Atb, psd_mode: int = 1):
    """sage_factor_hessian_blocks -> (dict {(i, j): G_ij}, [g_i], dims): the HessianFactor blocks of a per-edge system."""
    L = lib()
    A = _f32(AtA); b = _f32(Atb)
    n = L.sage_factor_block_count(type_, CS)
    if n < 0:
        raise SageError(n, "sage_factor_block_count")
    G = np.zeros(n, np.float64); g = np.zeros(b.size, np.float64)
    dims = (C.c_int32 * 6)(); nk = C.c_int32()
    _chk(L.sage_factor_hessian_blocks(type_, CS, _fp(A), _fp(b), psd_mode, G.ctypes.data_as(C.POINTER(C.c_double)),
                                      g.ctypes.data_as(C.POINTER(C.c_double)), dims, C.byref(nk)),
         "sage_factor_hessian_blocks")
    dims = [int(dims[i]) for i in range(nk.value)]
    blocks, o = {}, 0
    for i in range(nk.value):
        for j in range(i, nk.value):
            blocks[(i, j)] = G[o:o + dims[i] * dims[j]].reshape(dims[i], dims[j]); o += dims[i] * dims[j]
    offs = np.concatenate([[0], np.cumsum(dims)])
    return blocks, [g[offs[i]:offs[i + 1]] for i in range(nk.value)], dims


def damped_solve_qr_f32(A, b, damp):
    A = _f32(A); b = _f32(b)
    x = np.zeros(b.shape[0], np.float32)
    _chk(lib().sage_damped_solve_qr_f32(_fp(A), _fp(b), b.shape[0], C.c_float(damp), _fp(x)), "sage_damped_solve_qr_f32")
    return x


def block_solve(packed, K, links, B, damp, diag_add=None, g_add=None):
    packed = np.ascontiguousarray(packed, dtype=np.float64)
    lk = np.ascontiguousarray(np.asarray(links, dtype=np.int32).reshape(-1))
    delta = np.zeros(K * B, np.float64)
    dp = lambda a: None if a is None else np.ascontiguousarray(a, np.float64).ctypes.data_as(C.POINTER(C.c_double))
    da = None if diag_add is None else np.ascontiguousarray(diag_add, np.float64)
    ga = None if g_add is None else np.ascontiguousarray(g_add, np.float64)
    _chk(lib().sage_block_solve(packed.ctypes.data_as(C.POINTER(C.c_double)), K, len(lk) // 2, lk.ctypes.data_as(C.POINTER(C.c_int32)), B,
                                C.c_double(damp), dp(da), dp(ga), delta.ctypes.data_as(C.POINTER(C.c_double))),
         "sage_block_solve")
    return delta


def solve_lookahead_count() -> int:
    f = lib().sage_solve_lookahead_count
    f.restype = C.c_longlong
    return int(f())


def block_solve_domains(packed, K, links, B, damp, ndomains, diag_add=None, g_add=None):
    """sage_block_solve by domain decomposition on `ndomains` host threads (long windows, loop closures)."""
    p = np.ascontiguousarray(packed, np.float64)
    lk = np.ascontiguousarray(np.asarray(links, np.int32).reshape(-1))
    out = np.zeros(K * B, np.float64)
    dp = lambda a: None if a is None else np.ascontiguousarray(a, np.float64).ctypes.data_as(C.POINTER(C.c_double))
    da = None if diag_add is None else np.ascontiguousarray(diag_add, np.float64)
    ga = None if g_add is None else np.ascontiguousarray(g_add, np.float64)
    _chk(lib().sage_block_solve_domains(p.ctypes.data_as(C.POINTER(C.c_double)), K, len(links),
                                        lk.ctypes.data_as(C.POINTER(C.c_int32)), B, C.c_double(damp), dp(da), dp(ga),
                                        ndomains, out.ctypes.data_as(C.POINTER(C.c_double))), "sage_block_solve_domains")
    return out


def lm_config_default() -> SageLmConfig:
    cfg = SageLmConfig()
    lib().sage_lm_config_default(C.byref(cfg))
    return cfg


def pack_pose(R, t) -> np.ndarray:
    return np.concatenate([_f32(R).reshape(9), _f32(t).reshape(3)])


def track_lm(cfg: SageLmConfig, dof: int, lin_fn, err_fn, pose12, scale: float, trace_cap: int = 256):
    """Run the tracker LM policy (``camera_tracker.cpp:1156-1279``) with Python evaluation callbacks.
    ``lin_fn(pose12, scale) -> (AtA, Atb, error)``; ``err_fn(pose12, scale) -> error``."""
    def _lin(ctx, p, s, AtA, Atb, err):
        A, b, e = lin_fn(np.ctypeslib.as_array(p, (12,)).copy(), float(s))
        A = _f32(A).reshape(-1); b = _f32(b).reshape(-1)
        for i in range(dof * dof):
            AtA[i] = A[i]
        for i in range(dof):
            Atb[i] = b[i]
        err[0] = e
        return 0

    def _err(ctx, p, s, err):
        err[0] = err_fn(np.ctypeslib.as_array(p, (12,)).copy(), float(s))
        return 0

    pose = _f32(pose12).reshape(12).copy()
    sc = C.c_float(scale)
    fe = C.c_float(0); it = C.c_int(0); tl = C.c_int(0)
    trace = (SageLmTraceEntry * trace_cap)()
    cb1, cb2 = TRACK_LIN_FN(_lin), TRACK_ERR_FN(_err)
    _chk(lib().sage_track_lm(C.byref(cfg), dof, cb1, cb2, None, _fp(pose), C.byref(sc), C.byref(fe), C.byref(it),
                             trace, trace_cap, C.byref(tl)), "sage_track_lm")
    tr = [dict(damp=trace[i].damp, error=trace[i].error, candidate_error=trace[i].candidate_error,
               accepted=trace[i].accepted, relinearized=trace[i].relinearized) for i in range(tl.value)]
    return pose, sc.value, fe.value, it.value, tr


# --------------------------------------------------------------------------- device side
def track_frame(cfg: SageLmConfig, dof: int, prob: "SageTrackProblem", pose12, scale: float, trace_cap: int = 256):
    """sage_track_frame: the tracker LM wired to the HIP kernels.  Returns (rc, pose12, scale, final_error, iters, trace)."""
    L = lib()
    p = _f32(pose12).copy()
    sc = C.c_float(scale); fe = C.c_float(); it = C.c_int(); tl = C.c_int()
    tr = (SageLmTraceEntry * trace_cap)()
    rc = L.sage_track_frame(C.byref(cfg), dof, C.byref(prob), _fp(p), C.byref(sc), C.byref(fe), C.byref(it), tr,
                            trace_cap, C.byref(tl))
    trace = [dict(damp=tr[i].damp, error=tr[i].error, candidate_error=tr[i].candidate_error,
                  accepted=tr[i].accepted, relinearized=tr[i].relinearized) for i in range(tl.value)]
    return rc, p, sc.value, fe.value, it.value, trace


class Workspace:
    """``SageWorkspace``: stream + scratch of one host thread."""

    def __init__(self, stream=None):
        self.h = C.c_void_p()
        sp = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(0)
        _chk(lib().sage_workspace_create(sp, C.byref(self.h)), "sage_workspace_create")

    def close(self):
        if self.h:
            lib().sage_workspace_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _dev(x, dtype=None):
    import torch
    if isinstance(x, torch.Tensor):
        return x.contiguous()
    a = np.ascontiguousarray(x, dtype=dtype)
    return torch.from_numpy(a).cuda()


class DeviceKeyframe:
    """device-resident copy of a synthetic/real keyframe in the reference layouts (a10)."""

    def __init__(self, kf, H, W):
        import torch
        self.feat_pyr = _dev(kf.feat_pyr, np.float32)
        self.grad_pyr = _dev(kf.grad_pyr, np.float32)
        self.bias = _dev(kf.bias, np.float32)
        self.basis = _dev(kf.basis, np.float32)
        self.loc1d = _dev(kf.loc1d, np.int64)
        self.loc1d_i32 = self.loc1d.to(torch.int32)
        self.homo = _dev(kf.homo, np.float32)
        self.N = int(self.homo.shape[0])
        self.H, self.W = H, W

    def view(self) -> SageKeyframeView:
        return SageKeyframeView(self.feat_pyr.data_ptr(), self.grad_pyr.data_ptr(), self.bias.data_ptr(),
                                self.basis.data_ptr(), self.loc1d.data_ptr(), self.homo.data_ptr(), self.N)


def photometric_jac_error(ws: Workspace, R10, t10, R0, t0, R1, t1, bias0, basis0, code0, mask1, loc1d, homo,
                          feat0, feat1, grad1, scale0, pyr: SagePyramid, eps, weights, FS, CS):
    """``df::photometric_jac_error_calculate<CS,FS>``; poses/code are small host arrays uploaded here (the
    reference does the same H2D copies in ``PhotometricFactor::ComputeJacobianAndError``)."""
    import torch
    D = 13 + CS
    small = torch.from_numpy(np.concatenate([_f32(x).reshape(-1) for x in (R10, t10, R0, t0, R1, t1, code0)])).cuda()
    o = [0, 9, 12, 21, 24, 33, 36]
    AtA = torch.empty((D, D), dtype=torch.float32, device="cuda")
    Atb = torch.empty((D,), dtype=torch.float32, device="cuda")
    err = C.c_float(); nin = C.c_float()
    w = _f32(weights)
    N = int(homo.shape[0])
    base = small.data_ptr()
    p = lambda i: C.c_void_p(base + 4 * o[i])
    _chk(lib().sage_photometric_jac_error_calculate(
        ws.h, dptr(AtA), dptr(Atb), C.byref(err), C.byref(nin), p(0), p(1), p(2), p(3), p(4), p(5),
        dptr(bias0), dptr(basis0), p(6), dptr(mask1), dptr(loc1d), dptr(homo), dptr(feat0), dptr(feat1),
        dptr(grad1), C.c_float(scale0), C.byref(pyr), C.c_float(eps), _fp(w), N, FS, CS),
        "sage_photometric_jac_error_calculate")
    return dict(AtA=AtA.cpu().numpy(), Atb=Atb.cpu().numpy(), error=err.value, num_inliers=nin.value)


def photometric_error(ws: Workspace, R10, t10, bias0, basis0, code0, mask1, loc1d, homo, feat0, feat1, scale0,
                      pyr: SagePyramid, eps, weights, FS, CS):
    import torch
    small = torch.from_numpy(np.concatenate([_f32(x).reshape(-1) for x in (R10, t10, code0)])).cuda()
    base = small.data_ptr()
    err = C.c_float(); nin = C.c_float()
    w = _f32(weights)
    N = int(homo.shape[0])
    _chk(lib().sage_photometric_error_calculate(
        ws.h, C.byref(err), C.byref(nin), C.c_void_p(base), C.c_void_p(base + 36), dptr(bias0), dptr(basis0),
        C.c_void_p(base + 48), dptr(mask1), dptr(loc1d), dptr(homo), dptr(feat0), dptr(feat1), C.c_float(scale0),
        C.byref(pyr), C.c_float(eps), _fp(w), N, FS, CS), "sage_photometric_error_calculate")
    return err.value, nin.value


def tracker_photo_jac_error(ws: Workspace, dof, R, t, mask1, dpts0, homo, feat0s, feat1, grad1, pyr, scale0, eps,
                            weights_dev, FS):
    import torch
    small = torch.from_numpy(np.concatenate([_f32(R).reshape(-1), _f32(t).reshape(-1)])).cuda()
    base = small.data_ptr()
    AtA = torch.empty((dof, dof), dtype=torch.float32, device="cuda")
    Atb = torch.empty((dof,), dtype=torch.float32, device="cuda")
    err = C.c_float(); nin = C.c_float()
    N = int(homo.shape[0])
    _chk(lib().sage_tracker_photo_jac_error_calculate(
        ws.h, dof, dptr(AtA), dptr(Atb), C.byref(err), C.byref(nin), C.c_void_p(base), C.c_void_p(base + 36),
        dptr(mask1), dptr(dpts0), dptr(homo), dptr(feat0s), dptr(feat1), dptr(grad1), C.byref(pyr),
        C.c_float(scale0), C.c_float(eps), dptr(weights_dev), N, FS), "sage_tracker_photo_jac_error_calculate")
    return dict(AtA=AtA.cpu().numpy(), Atb=Atb.cpu().numpy(), error=err.value, num_inliers=nin.value)


def tracker_photo_error(ws: Workspace, R, t, mask1, dpts0, homo, feat0s, feat1, pyr, eps, weights_dev, FS):
    import torch
    small = torch.from_numpy(np.concatenate([_f32(R).reshape(-1), _f32(t).reshape(-1)])).cuda()
    base = small.data_ptr()
    err = C.c_float(); nin = C.c_float()
    N = int(homo.shape[0])
    _chk(lib().sage_tracker_photo_error_calculate(
        ws.h, C.byref(err), C.byref(nin), C.c_void_p(base), C.c_void_p(base + 36), dptr(mask1), dptr(dpts0),
        dptr(homo), dptr(feat0s), dptr(feat1), C.byref(pyr), C.c_float(eps), dptr(weights_dev), N, FS),
        "sage_tracker_photo_error_calculate")
    return err.value, nin.value


def geometric_jac_error(ws: Workspace, R10, t10, R0, t0, R1, t1, bias0, basis0, code0, dpt1, dgrad1, basis1, mask1,
                        loc1d_i32, homo, scale0, scale1, cam: SageCamera, eps, loss_param, weight, CS):
    import torch
    D = 14 + 2 * CS
    small = torch.from_numpy(np.concatenate([_f32(x).reshape(-1) for x in (R10, t10, R0, t0, R1, t1, code0)])).cuda()
    o = [0, 9, 12, 21, 24, 33, 36]
    base = small.data_ptr()
    p = lambda i: C.c_void_p(base + 4 * o[i])
    AtA = torch.empty((D, D), dtype=torch.float32, device="cuda")
    Atb = torch.empty((D,), dtype=torch.float32, device="cuda")
    err = C.c_float(); nin = C.c_float()
    N = int(homo.shape[0])
    _chk(lib().sage_geometric_jac_error_calculate(
        ws.h, dptr(AtA), dptr(Atb), C.byref(err), C.byref(nin), p(0), p(1), p(2), p(3), p(4), p(5),
        dptr(bias0), dptr(basis0), p(6), dptr(dpt1), dptr(dgrad1), dptr(basis1), dptr(mask1), dptr(loc1d_i32),
        dptr(homo), C.c_float(scale0), C.c_float(scale1), C.byref(cam), C.c_float(eps), C.c_float(loss_param),
        C.c_float(weight), N, CS), "sage_geometric_jac_error_calculate")
    return dict(AtA=AtA.cpu().numpy(), Atb=Atb.cpu().numpy(), error=err.value, num_inliers=nin.value)


def geometric_error(ws: Workspace, R10, t10, bias0, basis0, code0, dpt1, mask1, loc1d_i32, homo, scale0,
                    cam: SageCamera, eps, loss_param, weight, CS):
    import torch
    small = torch.from_numpy(np.concatenate([_f32(x).reshape(-1) for x in (R10, t10, code0)])).cuda()
    base = small.data_ptr()
    err = C.c_float(); nin = C.c_float()
    N = int(homo.shape[0])
    _chk(lib().sage_geometric_error_calculate(
        ws.h, C.byref(err), C.byref(nin), C.c_void_p(base), C.c_void_p(base + 36), dptr(bias0), dptr(basis0),
        C.c_void_p(base + 48), dptr(dpt1), dptr(mask1), dptr(loc1d_i32), dptr(homo), C.c_float(scale0),
        C.byref(cam), C.c_float(eps), C.c_float(loss_param), C.c_float(weight), N, CS),
        "sage_geometric_error_calculate")
    return err.value, nin.value


def depth_and_grad(ws: Workspace, bias, basis, code, scale, H, W, CS):
    import torch
    dpt = torch.empty((H, W), dtype=torch.float32, device="cuda")
    grad = torch.empty((2, H, W), dtype=torch.float32, device="cuda")
    code_d = _dev(code, np.float32)
    _chk(lib().sage_depth_and_grad(ws.h, dptr(dpt), dptr(grad), dptr(bias), dptr(basis), dptr(code_d),
                                   C.c_float(scale), H, W, CS), "sage_depth_and_grad")
    # the producer is ASYNCHRONOUS on the workspace's stream: its inputs must outlive the launch.  `code_d` rides on the
    # result (torch would otherwise hand its block to the next allocation of any thread while the kernel is queued --
    # found by tests/test_gpu_threads.py)
    dpt._sage_keepalive = code_d
    grad._sage_keepalive = code_d    # (a caller may keep only one of the two)
    return dpt, grad


def gaussian_pyramid_with_grad(ws: Workspace, feat, mask, pyr: SagePyramid, FS):
    import torch
    out = torch.empty((FS, pyr.P), dtype=torch.float32, device="cuda")
    grad = torch.empty((2, FS, pyr.P), dtype=torch.float32, device="cuda")
    _chk(lib().sage_gaussian_pyramid_with_grad(ws.h, dptr(out), dptr(grad), dptr(feat), dptr(mask), C.byref(pyr), FS),
         "sage_gaussian_pyramid_with_grad")
    return out, grad


class Window:
    """``SageWindow``: batched K-keyframe BA window on one GPU (one shard of the edge set)."""

    def __init__(self, win, rank: int = 0, world: int = 1, stream=None, use_photo=True, use_geo=True,
                 code_prior_weight=1.0e-3, scale_prior_weight=1.0e4, pose_prior_weight=1.0e4):
        import torch
        self.win = win
        self.pyr = make_pyramid(win.cams[0], win.L)
        assert self.pyr.P == win.P
        self.mask = _dev(win.mask, np.float32)
        self.kfs = [DeviceKeyframe(kf, win.H, win.W) for kf in win.keyframes]
        cfg = SageWindowConfig()
        cfg.pyr = self.pyr
        cfg.FS, cfg.CS = win.FS, win.CS
        cfg.mask_dev = self.mask.data_ptr()
        for l in range(win.L):
            cfg.photo_weights[l] = float(win.photo_weights[l])
        cfg.geo_weight, cfg.geo_loss_param, cfg.eps = win.geo_weight, win.geo_loss_param, win.eps
        cfg.code_prior_weight, cfg.scale_prior_weight, cfg.pose_prior_weight = (
            code_prior_weight, scale_prior_weight, pose_prior_weight)
        cfg.use_photo, cfg.use_geo = int(use_photo), int(use_geo)
        self.cfg = cfg
        self.h = C.c_void_p()
        sp = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(0)
        L = lib()
        _chk(L.sage_window_create(C.byref(cfg), sp, C.byref(self.h)), "sage_window_create")
        for kf, dk in zip(win.keyframes, self.kfs):
            v = dk.view()
            pose = pack_pose(kf.R, kf.t)
            code = _f32(kf.code)
            r = L.sage_window_add_keyframe(self.h, C.byref(v), _fp(pose), _fp(code), C.c_float(kf.scale))
            if r < 0:
                raise SageError(r, "sage_window_add_keyframe")
        for a, b in win.links:
            r = L.sage_window_add_link(self.h, a, b)
            if r < 0:
                raise SageError(r, "sage_window_add_link")
            gl = getattr(win, "link_geo_loss", None)     # optional per-link Cauchy parameters (mapper.cpp:367-373)
            if gl is not None and gl[r] > 0:
                _chk(L.sage_window_set_link_geo_loss(self.h, r, C.c_float(gl[r])), "sage_window_set_link_geo_loss")
        _chk(L.sage_window_set_shard(self.h, rank, world), "sage_window_set_shard")
        _chk(L.sage_window_finalize(self.h), "sage_window_finalize")
        self.K = L.sage_window_num_keyframes(self.h)
        self.B = L.sage_window_block_size(self.h)
        self.nlinks = L.sage_window_num_links(self.h)
        self.packed_count = L.sage_window_packed_count(self.h)
        self.residuals_per_linearize = L.sage_window_residuals_per_linearize(self.h)
        self.bytes_per_linearize = L.sage_window_bytes_per_linearize(self.h)

    # raw-pointer views for torch.distributed (plumbing only)
    def packed_tensor(self):
        return _tensor_from_ptr(lib().sage_window_packed_dev(self.h), self.packed_count)

    def error_tensor(self):
        return _tensor_from_ptr(lib().sage_window_error_dev(self.h), 4)

    def linearize(self):
        _chk(lib().sage_window_linearize(self.h), "sage_window_linearize")

    def error(self, which=1):
        _chk(lib().sage_window_error(self.h, which), "sage_window_error")

    def tune_runs(self):
        """``sage_window_tune_runs``: measure the photometric kernels' run lengths on this window, keep the fastest ->
        dict(tpb, tpb_rule, ms_rule, ms_best)."""
        tpb, rule = C.c_int(0), C.c_int(0)
        ms_rule, ms_best = C.c_float(0), C.c_float(0)
        _chk(lib().sage_window_tune_runs(self.h, C.byref(tpb), C.byref(rule), C.byref(ms_rule), C.byref(ms_best)),
             "sage_window_tune_runs")
        return dict(tpb=tpb.value, tpb_rule=rule.value, ms_rule=ms_rule.value, ms_best=ms_best.value)

    def set_runs(self, tpb):
        _chk(lib().sage_window_set_runs(self.h, int(tpb)), "sage_window_set_runs")

    def solve(self, damp, want_norm=True):
        """Damped step + candidate variables.  With ``want_norm=False`` the device solver is only enqueued (no
        synchronisation); a non-positive pivot is then reported by the next ``total_error``/``accept``."""
        if not want_norm:
            _chk(lib().sage_window_solve(self.h, C.c_double(damp), None), "sage_window_solve")
            return None
        sn = C.c_double()
        _chk(lib().sage_window_solve(self.h, C.c_double(damp), C.byref(sn)), "sage_window_solve")
        return sn.value

    def total_error(self, from_linearize: bool):
        e = C.c_double()
        _chk(lib().sage_window_total_error(self.h, int(from_linearize), C.byref(e)), "sage_window_total_error")
        return e.value

    def accept(self):
        _chk(lib().sage_window_accept(self.h), "sage_window_accept")

    def reset(self):
        _chk(lib().sage_window_reset(self.h), "sage_window_reset")

    def set_allreduce(self, dist_module, group=None):
        """Install ``torch.distributed.all_reduce`` as the window's all-reduce hook (``sage_window_set_allreduce``):
        ``lm_step`` then drives a sharded window too, entering Python only for the two collectives."""
        tensors = {}
        for t in (self.packed_tensor(), self.error_tensor()):
            tensors[(t.data_ptr(), t.numel())] = t

        def hook(ptr, n, _user):
            try:
                t = tensors.get((ptr, n))
                if t is None:
                    t = tensors[(ptr, n)] = _tensor_from_ptr(ptr, n)
                dist_module.all_reduce(t, group=group)
                return 0
            except Exception:                      # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1

        self._allreduce_cb = ALLREDUCE_FN(hook)    # keep the trampoline alive as long as the window
        _chk(lib().sage_window_set_allreduce(self.h, self._allreduce_cb, None), "sage_window_set_allreduce")

    def sync_variables(self):
        _chk(lib().sage_window_sync_variables(self.h), "sage_window_sync_variables")

    def use_rccl(self, comm):
        """native RCCL all-reduce on the window's stream (sage_window_use_rccl); comm from rccl_comm_create()."""
        _chk(lib().sage_window_use_rccl(self.h, C.c_void_p(comm)), "sage_window_use_rccl")

    def emulate_peers(self, rest):
        """``sage_window_emulate_peers``: rest = [n_iterates, packed_count] float64 cuda tensor (kept alive here), the
        packed systems of the ranks that are not there at the LM iterates since reset; None switches it off."""
        if rest is None:
            self._emu_rest = None
            _chk(lib().sage_window_emulate_peers(self.h, None, 0), "sage_window_emulate_peers")
            return
        assert rest.is_cuda and rest.dim() == 2 and rest.shape[1] == self.packed_count and rest.is_contiguous()
        self._emu_rest = rest
        _chk(lib().sage_window_emulate_peers(self.h, C.c_void_p(rest.data_ptr()), int(rest.shape[0])),
             "sage_window_emulate_peers")

    def lm_step(self, state: SageLmState, cfg: SageLmConfig):
        _chk(lib().sage_window_lm_step(self.h, C.byref(state), C.byref(cfg)), "sage_window_lm_step")
        return state

    def lm_run(self, state: SageLmState, cfg: SageLmConfig, n: int):
        """n LM iterations without returning to Python in between -> [n, 4] trace {error, candidate, accepted, damp}."""
        tr = np.zeros((n, 4), np.float64)
        done = C.c_int()
        _chk(lib().sage_window_lm_run(self.h, C.byref(state), C.byref(cfg), n, tr.ctypes.data_as(C.POINTER(C.c_double)),
                                      C.byref(done)), "sage_window_lm_run")
        return tr[:done.value]

    def lm_run_timed(self, state: SageLmState, cfg: SageLmConfig, n: int):
        """lm_run + the host wall time of every iteration in seconds -> ([n, 4] trace, [n] seconds)."""
        tr = np.zeros((n, 4), np.float64); sec = np.zeros(n, np.float64)
        done = C.c_int()
        _chk(lib().sage_window_lm_run_timed(self.h, C.byref(state), C.byref(cfg), n, tr.ctypes.data_as(C.POINTER(C.c_double)),
                                            C.byref(done), sec.ctypes.data_as(C.POINTER(C.c_double))), "sage_window_lm_run_timed")
        return tr[:done.value], sec[:done.value]

    def delta(self):
        d = np.zeros(self.K * self.B, np.float64)
        _chk(lib().sage_window_get_delta(self.h, d.ctypes.data_as(C.POINTER(C.c_double))), "sage_window_get_delta")
        return d

    def get_keyframe(self, k):
        pose = np.zeros(12, np.float32); code = np.zeros(self.win.CS, np.float32); s = C.c_float()
        _chk(lib().sage_window_get_keyframe(self.h, k, _fp(pose), _fp(code), C.byref(s)), "sage_window_get_keyframe")
        return pose, code, s.value

    def set_keyframe(self, k, pose12, code, scale):
        pose12 = _f32(pose12); code = _f32(code)
        _chk(lib().sage_window_set_keyframe(self.h, k, _fp(pose12), _fp(code), C.c_float(scale)),
             "sage_window_set_keyframe")

    def get_edge(self, type_, e):
        D = 13 + self.win.CS if type_ == 0 else 14 + 2 * self.win.CS
        A = np.zeros((D, D), np.float32); b = np.zeros(D, np.float32)
        err = C.c_float(); nin = C.c_float()
        _chk(lib().sage_window_get_edge(self.h, type_, e, _fp(A), _fp(b), C.byref(err), C.byref(nin)),
             "sage_window_get_edge")
        return dict(AtA=A, Atb=b, error=err.value, num_inliers=nin.value)

    def prepass(self, poses12, codes, scales, jacobians: bool = True) -> bool:
        """f2: evaluate the whole window once at these values (or hit the cache). Returns True when kernels ran."""
        P = _f32(np.asarray(poses12).reshape(-1)); Cd = _f32(np.asarray(codes).reshape(-1)); S = _f32(np.asarray(scales).reshape(-1))
        K = self.K
        assert P.size == K * 12 and Cd.size == K * self.win.CS and S.size == K
        rec = C.c_int32()
        _chk(lib().sage_window_prepass(self.h, _fp(P), _fp(Cd), _fp(S), int(jacobians), C.byref(rec)), "sage_window_prepass")
        return bool(rec.value)

    def factor(self, type_, e, psd_mode: int = 1):
        """f2: the HessianFactor of directed edge e from the prepass cache -> (blocks, gs, f, dims)."""
        L = lib(); CS = self.win.CS
        n = L.sage_factor_block_count(type_, CS)
        D = 13 + CS if type_ == 0 else 14 + 2 * CS
        G = np.zeros(n, np.float64); g = np.zeros(D, np.float64); f = C.c_double()
        dims = (C.c_int32 * 6)(); nk = C.c_int32()
        _chk(L.sage_window_factor(self.h, type_, e, psd_mode, G.ctypes.data_as(C.POINTER(C.c_double)),
                                  g.ctypes.data_as(C.POINTER(C.c_double)), C.byref(f), dims, C.byref(nk)), "sage_window_factor")
        dims = [int(dims[i]) for i in range(nk.value)]
        blocks, o = {}, 0
        for i in range(nk.value):
            for j in range(i, nk.value):
                blocks[(i, j)] = G[o:o + dims[i] * dims[j]].reshape(dims[i], dims[j]); o += dims[i] * dims[j]
        offs = np.concatenate([[0], np.cumsum(dims)])
        return blocks, [g[offs[i]:offs[i + 1]] for i in range(nk.value)], f.value, dims

    def prepare_factors(self, psd_mode: int = 1, n_threads: int = 0):
        """f2: NearestPsd of every cached factor on host threads; factor(.., psd_mode) then only cuts blocks."""
        _chk(lib().sage_window_prepare_factors(self.h, psd_mode, n_threads), "sage_window_prepare_factors")

    def factor_error(self, type_, e) -> float:
        v = C.c_double()
        _chk(lib().sage_window_factor_error(self.h, type_, e, C.byref(v)), "sage_window_factor_error")
        return v.value

    def set_profiling(self, on):
        """False / 0 off, True / 1 every hot kernel + phase marks, 2 the photometric linearize only."""
        _chk(lib().sage_window_set_profiling(self.h, int(on)), "sage_window_set_profiling")

    def kernel_time(self, which: int):
        """(total_ms, launches) of hot kernel `which` since the last call (0 photo lin, 1 geo lin, 2 photo err, 3 geo err)."""
        ms = C.c_double(); n = C.c_int()
        _chk(lib().sage_window_get_kernel_time(self.h, which, C.byref(ms), C.byref(n)), "sage_window_get_kernel_time")
        return ms.value, n.value

    def phase_time(self):
        """({linearize, allreduce, solve, error_pass} total ms, iterations) of the LM iterations since the last call."""
        ms = (C.c_double * 4)(); n = C.c_int()
        _chk(lib().sage_window_get_phase_time(self.h, ms, C.byref(n)), "sage_window_get_phase_time")
        return dict(zip(("linearize", "allreduce", "solve", "error_pass"), (float(v) for v in ms))), n.value

    def packed_host(self):
        return self.packed_tensor().cpu().numpy()

    def close(self):
        if self.h:
            lib().sage_window_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _tensor_from_ptr(ptr: int, n: int):
    """wrap engine-owned device memory as a torch tensor (zero-copy) so torch.distributed can all-reduce it."""
    import torch

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = dict(shape=(int(n),), typestr="<f8", data=(int(ptr), False), version=2)
    return torch.as_tensor(h, device="cuda")


# --------------------------------------------------------------------------- host-side window algebra
# (pure numpy; used by the multi-process CPU tests of the sharded reduction and by parity tests)
class ShardPlan:
    """sage_shard_* : domain-decomposed solve of a link-sharded window (host, double)."""

    def __init__(self, K, links, B, rank, world):
        L = lib()
        L.sage_shard_sep_count.restype = C.c_size_t
        L.sage_shard_sep_count.argtypes = [C.c_void_p]
        lk = np.ascontiguousarray(np.asarray(links, np.int32).reshape(-1))
        self.h = C.c_void_p()
        _chk(L.sage_shard_plan_create(K, len(links), lk.ctypes.data_as(C.POINTER(C.c_int32)), B, rank, world,
                                      C.byref(self.h)), "sage_shard_plan_create")
        self.K, self.B = K, B
        self.sep_count = int(L.sage_shard_sep_count(self.h))
        self.n_sep = L.sage_shard_num_separators(self.h)
        self.n_interior = L.sage_shard_num_interior(self.h)

    def owner(self, kf):
        return lib().sage_shard_keyframe_owner(self.h, kf)

    def is_local(self, kf):
        return bool(lib().sage_shard_keyframe_is_local(self.h, kf))

    def eliminate(self, packed_local, damp, diag_add=None, g_add=None):
        p = np.ascontiguousarray(packed_local, np.float64)
        out = np.zeros(self.sep_count, np.float64)
        dp = lambda a: None if a is None else np.ascontiguousarray(a, np.float64).ctypes.data_as(C.POINTER(C.c_double))
        self._keep = (diag_add, g_add)
        _chk(lib().sage_shard_eliminate(self.h, p.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(damp), dp(diag_add),
                                        dp(g_add), out.ctypes.data_as(C.POINTER(C.c_double))), "sage_shard_eliminate")
        return out

    def solve(self, sep_reduced):
        s = np.ascontiguousarray(sep_reduced, np.float64)
        delta = np.zeros(self.K * self.B, np.float64)
        _chk(lib().sage_shard_solve(self.h, s.ctypes.data_as(C.POINTER(C.c_double)),
                                    delta.ctypes.data_as(C.POINTER(C.c_double))), "sage_shard_solve")
        return delta

    def close(self):
        if self.h:
            lib().sage_shard_plan_destroy(self.h)
            self.h = C.c_void_p()


def rccl_unique_id() -> bytes:
    buf = (C.c_ubyte * 128)()
    _chk(lib().sage_rccl_unique_id(buf), "sage_rccl_unique_id")
    return bytes(buf)


def rccl_comm_create(uid: bytes, rank: int, world: int) -> int:
    """ncclCommInitRank on the current device -> opaque communicator handle (int)."""
    buf = (C.c_ubyte * 128).from_buffer_copy(uid)
    comm = C.c_void_p()
    _chk(lib().sage_rccl_comm_create(buf, rank, world, C.byref(comm)), "sage_rccl_comm_create")
    return comm.value


def rccl_comm_destroy(comm: int):
    lib().sage_rccl_comm_destroy(C.c_void_p(comm))


def rccl_comm_info(comm: int):
    """(ranks, rank) as the communicator itself reports them (``ncclCommCount`` / ``ncclCommUserRank``)."""
    n, r = C.c_int(), C.c_int()
    _chk(lib().sage_rccl_comm_info(C.c_void_p(comm), C.byref(n), C.byref(r)), "sage_rccl_comm_info")
    return n.value, r.value


def shard_links(nlinks: int, rank: int, world: int) -> List[int]:
    """link ownership rule of windows that use the domain-decomposed solve (``SAGE_SHARD_SCHUR`` / K >= 256) and of
    ``sage_shard_plan_create``: rank r owns the contiguous range [r*n/world, (r+1)*n/world) of the link list."""
    return list(range(nlinks * rank // world, nlinks * (rank + 1) // world))


def shard_edges(nlinks: int, rank: int, world: int) -> List[int]:
    """ownership rule of ``sage_window_set_shard`` (r05): rank r owns the contiguous range [r*2n/world, (r+1)*2n/world) of
    the DIRECTED edges 2*link + direction (both factor types of a direction together); the two directions of a link may
    belong to two ranks."""
    return list(range(2 * nlinks * rank // world, 2 * nlinks * (rank + 1) // world))


def edge_col(type_: int, role: int, bi: int, CS: int) -> int:
    """B-index (pose 6, code CS, scale) -> column of the per-edge system (mirror of the assemble kernel)."""
    if bi < 6:
        return role * 6 + bi
    if type_ == 0:
        if role == 1:
            return -1
        return 12 + (bi - 6) if bi < 6 + CS else 12 + CS
    if bi < 6 + CS:
        return 12 + role * CS + (bi - 6)
    return 12 + 2 * CS + role


def assemble_packed(K: int, links: Sequence, CS: int, edge_results: dict) -> np.ndarray:
    """Sum per-edge normal equations into the packed block layout.
    ``edge_results[(type, link, dir)] = dict(AtA, Atb, error, num_inliers)`` for the edges present."""
    B = 7 + CS
    BB = B * B
    diag = np.zeros((K, B, B), np.float64)
    lnk = np.zeros((len(links), B, B), np.float64)
    g = np.zeros((K, B), np.float64)
    tail = np.zeros(4, np.float64)
    cols = {(t, r): np.array([edge_col(t, r, bi, CS) for bi in range(B)]) for t in (0, 1) for r in (0, 1)}
    for (t, l, d), res in edge_results.items():
        a, b = links[l]
        k0, k1 = (a, b) if d == 0 else (b, a)
        A = np.asarray(res["AtA"], np.float64); v = np.asarray(res["Atb"], np.float64)
        for kf, role in ((k0, 0), (k1, 1)):
            c = cols[(t, role)]
            m = c >= 0
            diag[kf][np.ix_(m, m)] += A[np.ix_(c[m], c[m])]
            g[kf][m] += v[c[m]]
        # link block rows = older keyframe a, cols = newer keyframe b
        ra, rb = (0, 1) if d == 0 else (1, 0)
        ca, cb = cols[(t, ra)], cols[(t, rb)]
        ma, mb = ca >= 0, cb >= 0
        lnk[l][np.ix_(ma, mb)] += A[np.ix_(ca[ma], cb[mb])]
        tail[t] += res["error"]
        tail[2 + t] += res["num_inliers"]
    return np.concatenate([diag.reshape(-1), lnk.reshape(-1), g.reshape(-1), tail])


def unpack_dense(packed: np.ndarray, K: int, links: Sequence, CS: int):
    """packed block layout -> dense (K*B x K*B) H, g, tail."""
    B = 7 + CS
    BB = B * B
    n = K * B
    H = np.zeros((n, n), np.float64)
    diag = packed[:K * BB].reshape(K, B, B)
    lnk = packed[K * BB:(K + len(links)) * BB].reshape(len(links), B, B)
    g = packed[(K + len(links)) * BB:(K + len(links)) * BB + n].astype(np.float64)
    for k in range(K):
        H[k * B:(k + 1) * B, k * B:(k + 1) * B] = 0.5 * (diag[k] + diag[k].T)
    for l, (a, b) in enumerate(links):
        H[a * B:(a + 1) * B, b * B:(b + 1) * B] += lnk[l]
        H[b * B:(b + 1) * B, a * B:(a + 1) * B] += lnk[l].T
    return H, g, packed[-4:]


def shuffle_indices(n: int, seed: int) -> np.ndarray:
    """Host helper: the keyframe sampling permutation (mapper.cpp:1326-1333)."""
    idx = np.zeros(max(n, 1), np.int64)
    _chk(lib().sage_shuffle_indices(C.c_int64(seed), C.c_int64(n), idx.ctypes.data_as(C.POINTER(C.c_int64))),
         "sage_shuffle_indices")
    return idx[:n].copy()


def valid_locations(ws: "Workspace", mask, cam):
    """mask: [H,W] float cuda tensor; cam: SageCamera -> (loc1d int64 [n], homo [n,3]) cuda tensors."""
    import torch
    H, W = mask.shape
    loc = torch.zeros(H * W, dtype=torch.int64, device="cuda")
    homo = torch.zeros(H * W, 3, dtype=torch.float32, device="cuda")
    n = C.c_int()
    _chk(lib().sage_valid_locations(ws.h, C.c_void_p(mask.data_ptr()), C.byref(cam), C.c_void_p(loc.data_ptr()),
                                    C.c_void_p(homo.data_ptr()), C.byref(n)), "sage_valid_locations")
    return loc[:n.value].clone(), homo[:n.value].clone()


def sample_locations(ws: "Workspace", vloc, vhomo, seed: int, num_samples: int):
    import torch
    nv = int(vloc.shape[0])
    loc = torch.zeros(max(min(num_samples, nv), 1), dtype=torch.int64, device="cuda")
    homo = torch.zeros(max(min(num_samples, nv), 1), 3, dtype=torch.float32, device="cuda")
    n = C.c_int()
    _chk(lib().sage_sample_locations(ws.h, C.c_void_p(vloc.data_ptr()), C.c_void_p(vhomo.data_ptr()), nv,
                                     C.c_int64(seed), num_samples, C.c_void_p(loc.data_ptr()),
                                     C.c_void_p(homo.data_ptr()), C.byref(n)), "sage_sample_locations")
    return loc[:n.value].clone(), homo[:n.value].clone()


def _dptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def bind_thread_to_device(device: int = 0) -> int:
    """``sage_bind_thread_to_device``: keep the calling thread on the GPU's NUMA node (returns the number of CPUs)."""
    return int(lib().sage_bind_thread_to_device(int(device)))


def solver_helper_cpus():
    """``sage_solver_helper_cpus``: CPUs the hybrid solve's helper threads are pinned to (-1: not placed yet)."""
    buf = (C.c_int * 3)()
    n = int(lib().sage_solver_helper_cpus(buf, 3))
    return [int(buf[i]) for i in range(n)]


def solver_placement_moves() -> int:
    """``sage_solver_placement_moves``: helpers the placement monitor has moved off crowded cores so far."""
    return int(lib().sage_solver_placement_moves())


def placement_monitor(enable: bool) -> None:
    """``sage_placement_monitor``: the background re-pinning of crowded helper threads is opt-in (r06)."""
    lib().sage_placement_monitor(1 if enable else 0)


def shutdown() -> None:
    """``sage_shutdown``: stop and join every host thread the library started (they restart on demand)."""
    lib().sage_shutdown()


def host_threads_running() -> int:
    """``sage_host_threads_running``: helper / pool / monitor threads of the library that are alive."""
    return int(lib().sage_host_threads_running())


def sort_locations(ws, loc1d, homo, H, W):
    """``sage_sort_locations``: raster-ordered copies of a keyframe's sampled locations -> (loc1d, homo, sorted?)."""
    import torch
    n = int(loc1d.shape[0])
    lo = torch.empty_like(loc1d)
    ho = torch.empty_like(homo)
    flag = C.c_int()
    _chk(lib().sage_sort_locations(ws.h, _dptr(loc1d), _dptr(homo), n, H, W, _dptr(lo), _dptr(ho), C.byref(flag)),
         "sage_sort_locations")
    return lo, ho, bool(flag.value)


def reprojection_jac_error(ws, R10, t10, R0, t0, R1, t1, bias0, basis0, code0, loc1d_i32, homo, matched, scale0, cam,
                           eps, loss_param, weight, CS):
    """All array arguments are cuda tensors (poses 9/3 floats).  Returns dict(AtA [D,D], Atb [D] tensors, error, num_inliers)."""
    import torch
    N = int(homo.shape[0]); D = 13 + CS
    AtA = torch.zeros(D, D, device="cuda"); Atb = torch.zeros(D, device="cuda")
    e = C.c_float(); n = C.c_float()
    _chk(lib().sage_reprojection_jac_error_calculate(
        ws.h, _dptr(AtA), _dptr(Atb), C.byref(e), C.byref(n), _dptr(R10), _dptr(t10), _dptr(R0), _dptr(t0), _dptr(R1),
        _dptr(t1), _dptr(bias0), _dptr(basis0), _dptr(code0), _dptr(loc1d_i32), _dptr(homo), _dptr(matched),
        C.c_float(scale0), C.byref(cam), C.c_float(eps), C.c_float(loss_param), C.c_float(weight), N, CS),
        "sage_reprojection_jac_error_calculate")
    return dict(AtA=AtA, Atb=Atb, error=e.value, num_inliers=n.value)


def reprojection_error(ws, R10, t10, bias0, basis0, code0, loc1d_i32, homo, matched, scale0, cam, eps, loss_param,
                       weight, CS):
    N = int(homo.shape[0])
    e = C.c_float(); n = C.c_float()
    _chk(lib().sage_reprojection_error_calculate(
        ws.h, C.byref(e), C.byref(n), _dptr(R10), _dptr(t10), _dptr(bias0), _dptr(basis0), _dptr(code0),
        _dptr(loc1d_i32), _dptr(homo), _dptr(matched), C.c_float(scale0), C.byref(cam), C.c_float(eps),
        C.c_float(loss_param), C.c_float(weight), N, CS), "sage_reprojection_error_calculate")
    return e.value, n.value


def tracker_reproj_jac_error(ws, R, t, dpts0, homo, matched, cam, eps, loss_param, weight):
    import torch
    N = int(homo.shape[0])
    AtA = torch.zeros(6, 6, device="cuda"); Atb = torch.zeros(6, device="cuda")
    e = C.c_float(); n = C.c_float()
    _chk(lib().sage_tracker_reproj_jac_error_calculate(
        ws.h, _dptr(AtA), _dptr(Atb), C.byref(e), C.byref(n), _dptr(R), _dptr(t), _dptr(dpts0), _dptr(homo),
        _dptr(matched), C.byref(cam), C.c_float(eps), C.c_float(loss_param), C.c_float(weight), N),
        "sage_tracker_reproj_jac_error_calculate")
    return dict(AtA=AtA, Atb=Atb, error=e.value, num_inliers=n.value)


def tracker_reproj_error(ws, R, t, dpts0, homo, matched, cam, eps, loss_param, weight):
    N = int(homo.shape[0])
    e = C.c_float(); n = C.c_float()
    _chk(lib().sage_tracker_reproj_error_calculate(
        ws.h, C.byref(e), C.byref(n), _dptr(R), _dptr(t), _dptr(dpts0), _dptr(homo), _dptr(matched), C.byref(cam),
        C.c_float(eps), C.c_float(loss_param), C.c_float(weight), N), "sage_tracker_reproj_error_calculate")
    return e.value, n.value


MG_LOSS = {"fair": 0, "L2": 1, "huber": 2, "unbiased": 3}


def match_geometry(ws, loss, jac, R10, t10, R0, t0, R1, t1, bias0, bias1, basis0, basis1, code0, code1, homo0, homo1,
                   loc0, loc1, scale0, scale1, loss_param, weight, CS):
    """Mapper match-geometry factor; cuda tensors in, dict(AtA, Atb, error) or the error out."""
    import torch
    N = int(homo0.shape[0]); D = 14 + 2 * CS
    e = C.c_float()
    if jac:
        AtA = torch.zeros(D, D, device="cuda"); Atb = torch.zeros(D, device="cuda")
        _chk(lib().sage_match_geometry_jac_error_calculate(
            ws.h, _dptr(AtA), _dptr(Atb), C.byref(e), _dptr(R10), _dptr(t10), _dptr(R0), _dptr(t0), _dptr(R1), _dptr(t1),
            _dptr(bias0), _dptr(bias1), _dptr(basis0), _dptr(basis1), _dptr(code0), _dptr(code1), _dptr(homo0),
            _dptr(homo1), _dptr(loc0), _dptr(loc1), C.c_float(scale0), C.c_float(scale1), C.c_float(loss_param),
            C.c_float(weight), MG_LOSS[loss], N, CS), "sage_match_geometry_jac_error_calculate")
        return dict(AtA=AtA, Atb=Atb, error=e.value)
    _chk(lib().sage_match_geometry_error_calculate(
        ws.h, C.byref(e), _dptr(R10), _dptr(t10), _dptr(bias0), _dptr(bias1), _dptr(basis0), _dptr(basis1), _dptr(code0),
        _dptr(code1), _dptr(homo0), _dptr(homo1), _dptr(loc0), _dptr(loc1), C.c_float(scale0), C.c_float(scale1),
        C.c_float(loss_param), C.c_float(weight), MG_LOSS[loss], N, CS), "sage_match_geometry_error_calculate")
    return e.value


def loop_mg(ws, jac, R10, t10, R0, t0, R1, t1, udpts0, udpts1, homo0, homo1, scale0, scale1, loss_param, weight):
    import torch
    N = int(homo0.shape[0])
    e = C.c_float()
    if jac:
        AtA = torch.zeros(14, 14, device="cuda"); Atb = torch.zeros(14, device="cuda")
        _chk(lib().sage_loop_mg_jac_error_calculate(
            ws.h, _dptr(AtA), _dptr(Atb), C.byref(e), _dptr(R10), _dptr(t10), _dptr(R0), _dptr(t0), _dptr(R1), _dptr(t1),
            _dptr(udpts0), _dptr(udpts1), _dptr(homo0), _dptr(homo1), C.c_float(scale0), C.c_float(scale1),
            C.c_float(loss_param), C.c_float(weight), N), "sage_loop_mg_jac_error_calculate")
        return dict(AtA=AtA, Atb=Atb, error=e.value)
    _chk(lib().sage_loop_mg_error_calculate(
        ws.h, C.byref(e), _dptr(R10), _dptr(t10), _dptr(udpts0), _dptr(udpts1), _dptr(homo0), _dptr(homo1),
        C.c_float(scale0), C.c_float(scale1), C.c_float(loss_param), C.c_float(weight), N), "sage_loop_mg_error_calculate")
    return e.value


def tracker_match_geom(ws, jac, with_scale, R, t, dpts0, dpts1, homo0, homo1, scale0, loss_param, weight):
    import torch
    N = int(homo0.shape[0]); D = 7 if with_scale else 6
    e = C.c_float()
    if jac:
        AtA = torch.zeros(D, D, device="cuda"); Atb = torch.zeros(D, device="cuda")
        _chk(lib().sage_tracker_match_geom_jac_error_calculate(
            ws.h, _dptr(AtA), _dptr(Atb), C.byref(e), _dptr(R), _dptr(t), _dptr(dpts0), _dptr(dpts1), _dptr(homo0),
            _dptr(homo1), C.c_float(scale0), C.c_float(loss_param), C.c_float(weight), int(with_scale), N),
            "sage_tracker_match_geom_jac_error_calculate")
        return dict(AtA=AtA, Atb=Atb, error=e.value)
    _chk(lib().sage_tracker_match_geom_error_calculate(
        ws.h, C.byref(e), _dptr(R), _dptr(t), _dptr(dpts0), _dptr(dpts1), _dptr(homo0), _dptr(homo1),
        C.c_float(loss_param), C.c_float(weight), N), "sage_tracker_match_geom_error_calculate")
    return e.value


def cycle_match(ws, desc0, desc1, kp_loc0, H, W, cyc_thresh):
    """desc0/desc1: [C,H,W] cuda float tensors; kp_loc0: int64 cuda tensor [K] -> (raw_matched1, cyc_matched0, flags, n)."""
    import torch
    K = int(kp_loc0.shape[0]); Cc = int(desc0.shape[0])
    m1 = torch.zeros(max(K, 1), dtype=torch.int64, device="cuda"); c0 = torch.zeros_like(m1)
    fl = torch.zeros(max(K, 1), dtype=torch.int32, device="cuda")
    n = C.c_int()
    _chk(lib().sage_cycle_match(ws.h, _dptr(desc0), _dptr(desc1), _dptr(kp_loc0), K, Cc, H, W, C.c_float(cyc_thresh),
                                _dptr(m1), _dptr(c0), _dptr(fl), C.byref(n)), "sage_cycle_match")
    return m1[:K], c0[:K], fl[:K], n.value


# ----------------------------------------------------------------------------------------------------------------------
# keyframe ingest: the outputs of the (TorchScript) feature / depth networks -> the tensor bundle of a keyframe, on the
# device end to end (what Mapper::BuildKeyframe does with libtorch ops, core/mapping/mapper.cpp:1318-1426, 1242-1252)
# ----------------------------------------------------------------------------------------------------------------------
class IngestedKeyframe:
    """device tensors of one keyframe in the reference layouts (a10) + the host-side variables; duck-types a
    ``synth.Keyframe`` for :class:`Window` / :class:`DeviceKeyframe` (no host copy is made of the big tensors)."""

    def __init__(self, feat_pyr, grad_pyr, bias, basis, loc1d, homo, R, t, code, scale, avg_squared_dpt_bias):
        self.feat_pyr, self.grad_pyr, self.bias, self.basis = feat_pyr, grad_pyr, bias, basis
        self.loc1d, self.homo = loc1d, homo
        self.R, self.t, self.code, self.scale = R, t, code, scale
        self.avg_squared_dpt_bias = avg_squared_dpt_bias


def keyframe_from_net_outputs(ws: "Workspace", feat_map, dpt_bias, dpt_jac_code, mask, pyr: SagePyramid, seed: int,
                              num_samples: int, R=None, t=None):
    """feat_map [1,FS,H,W] or [FS,H,W] (feature net), dpt_bias [1,1,H,W] or [H,W], dpt_jac_code [H*W,CS] (depth net),
    mask [H,W] float 0/1 -- CUDA tensors that stay where they are.  Masked Gaussian pyramid + central-difference
    gradients (mapper.cpp:1384-1426), valid-pixel enumeration (mapping_utils.h:254-287), seeded std::shuffle subsample
    with seed = timestamp (mapper.cpp:1326-1340), zero initial code, unit scale (mapper.cpp:1242, :1318)."""
    import torch
    feat = feat_map.reshape(feat_map.shape[-3], feat_map.shape[-2], feat_map.shape[-1]).contiguous().float()
    FS, H, W = feat.shape
    bias = dpt_bias.reshape(-1).contiguous().float()
    basis = dpt_jac_code.reshape(H * W, -1).contiguous().float()
    CS = int(basis.shape[1])
    assert bias.numel() == H * W and mask.shape == (H, W)
    feat_pyr, grad_pyr = gaussian_pyramid_with_grad(ws, feat, mask, pyr, FS)
    vloc, vhomo = valid_locations(ws, mask, pyr.cam[0])
    loc, homo = sample_locations(ws, vloc, vhomo, seed, num_samples)
    # avg_squared_dpt_bias (mapper.cpp:1248-1250): the per-link Cauchy parameter of the geometric factor derives from it
    avg = float((torch.sum(torch.square(bias * mask.reshape(-1))) / torch.sum(mask)).item())
    R = np.eye(3, dtype=np.float32) if R is None else np.asarray(R, np.float32)
    t = np.zeros(3, np.float32) if t is None else np.asarray(t, np.float32)
    return IngestedKeyframe(feat_pyr, grad_pyr, bias, basis, loc, homo, R, t, np.zeros(CS, np.float32), 1.0, avg)
