"""f2 glue EXECUTED: integration/sage_gtsam_prepass.h -- the gtsam-side type conversion around
sage_window_prepass / sage_window_factor (Sophus::SE3f -> [R|t], Eigen::Map of the upper-triangular blocks into
gtsam::HessianFactor(keys, Gs, gs, f); core/gtsam/photometric_factor.cpp:151-218, geometric_factor.cpp:120-218) -- is
compiled in the build container into integration/compile_check/_bin/prepass_run (real Eigen + Sophus, RECORDING
stand-ins for gtsam::Values / gtsam::HessianFactor that assemble the augmented information matrix block by block like
gtsam's constructor) and run here on a K = 4 window written to a file.  What arrives on the gtsam side is compared with
(i) the CPU oracle's AtA / Atb / error of the same directed edge at the SAME values in the reference's column layout,
(ii) sage_window_factor through ctypes.  A transposed Eigen::Map (the rectangular pose x code blocks), a wrong block
order, a wrong dims[] or a mis-packed rotation fails it."""
import os
import struct
import subprocess

import numpy as np
import pytest

from sage_slam_amd import synth
from tests.helpers import oracle_geo, oracle_photo, rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "integration", "compile_check", "_bin", "prepass_run")


def _quat(R):
    """unit quaternion (w, x, y, z) of a rotation matrix (double)"""
    R = np.asarray(R, np.float64)
    w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    x = (R[2, 1] - R[1, 2]) / (4 * w); y = (R[0, 2] - R[2, 0]) / (4 * w); z = (R[1, 0] - R[0, 1]) / (4 * w)
    q = np.array([w, x, y, z]); return (q / np.linalg.norm(q)).astype(np.float32)


def _write_window(path, w, values):
    f32 = lambda a: np.ascontiguousarray(a, np.float32).tobytes()
    cam = w.cams[0]
    with open(path, "wb") as f:
        f.write(struct.pack("<7i", len(w.keyframes), w.H, w.W, w.FS, w.CS, w.L, len(w.links)))
        f.write(f32([cam.fx, cam.fy, cam.cx, cam.cy, cam.w, cam.h]))
        f.write(f32(w.photo_weights[:w.L]))
        f.write(f32([w.geo_weight, w.geo_loss_param, w.eps, 1.0e-3, 1.0e4, 1.0e4]))
        f.write(f32(w.mask))
        for k in w.keyframes:
            f.write(struct.pack("<i", k.homo.shape[0]))
            for a in (k.feat_pyr, k.grad_pyr, k.bias, k.basis):
                f.write(f32(a))
            f.write(np.ascontiguousarray(k.loc1d, np.int64).tobytes())
            f.write(f32(k.homo))
            f.write(f32(np.concatenate([np.asarray(k.R).reshape(-1), np.asarray(k.t).reshape(-1)])))
            f.write(f32(k.code)); f.write(f32([k.scale]))
        f.write(np.ascontiguousarray(w.links, np.int32).tobytes())
        for q, t, code, scale in values:
            f.write(f32(q)); f.write(f32(t)); f.write(f32(code)); f.write(f32([scale]))


def _read_out(path, K, nlinks, CS):
    raw = open(path, "rb").read()
    o = 0
    poses = np.frombuffer(raw, np.float32, K * 12, o).reshape(K, 12).copy(); o += K * 48
    recomputed = struct.unpack_from("<i", raw, o)[0]; o += 4
    out = {}
    for t in (0, 1):
        for e in range(2 * nlinks):
            nk = struct.unpack_from("<i", raw, o)[0]; o += 4
            keys = np.frombuffer(raw, np.uint64, nk, o).copy(); o += 8 * nk
            dims = np.frombuffer(raw, np.int32, nk, o).copy(); o += 4 * nk
            D = int(dims.sum())
            info = np.frombuffer(raw, np.float64, (D + 1) ** 2, o).reshape(D + 1, D + 1).copy(); o += 8 * (D + 1) ** 2
            err = struct.unpack_from("<d", raw, o)[0]; o += 8
            out[(t, e)] = dict(keys=keys, dims=dims, info=info, error=err)
    assert o == len(raw)
    return poses, recomputed, out


@pytest.mark.skipif(not os.path.exists(BIN), reason="prepass_run is built in the build container (needs the reference's "
                                                    "vendored Eigen/Sophus) and travels with the tree")
@pytest.mark.parametrize("CS", [32, 16])
def test_gtsam_glue_header_executes_and_delivers_the_reference_blocks(tmp_path, orc, CS):
    import copy
    import torch
    assert torch.cuda.is_available()
    from sage_slam_amd import capi
    w = synth.make_window(K=4, H=64, W=80, FS=16, CS=CS, L=4, seed=7, back_links=2)
    # the Values to linearise at differ from the variables the keyframes were added with: the header has to carry them
    rng = np.random.default_rng(3)
    values = []
    for k in w.keyframes:
        dR = synth.so3_exp(0.003 * rng.standard_normal(3))
        values.append((_quat(dR @ np.asarray(k.R, np.float64)), np.asarray(k.t, np.float32) + np.float32(0.003) * rng.standard_normal(3).astype(np.float32),
                       np.asarray(k.code, np.float32) + np.float32(0.01) * rng.standard_normal(CS).astype(np.float32),
                       np.float32(k.scale * (1 + 0.01 * rng.standard_normal()))))
    wb, ob = str(tmp_path / "window.bin"), str(tmp_path / "out.bin")
    _write_window(wb, w, values)
    res = {}
    for psd in (0, 1):
        r = subprocess.run([BIN, wb, ob, str(psd)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
        res[psd] = _read_out(ob, len(w.keyframes), len(w.links), CS)
    poses, recomputed, out0 = res[0]
    assert recomputed == 1                                     # first Prepare ran the kernels, the second was a cache hit
    # the window at the values the Sophus::SE3f objects actually hold (formed by the driver, not by the header)
    w2 = copy.deepcopy(w)
    for k, kf in enumerate(w2.keyframes):
        R = poses[k, :9].reshape(3, 3)
        assert np.abs(R.astype(np.float64) @ R.T.astype(np.float64) - np.eye(3)).max() < 1e-6
        kf.R = R.copy(); kf.t = poses[k, 9:].copy()
        kf.code = values[k][2].copy(); kf.scale = float(values[k][3])
        assert np.abs(poses[k, 9:] - values[k][1]).max() == 0
    win = capi.Window(w)                                       # (added with the ORIGINAL variables, like the driver's)
    codes = np.stack([v[2] for v in values]); scales = np.array([v[3] for v in values], np.float32)
    assert win.prepass(poses, codes, scales, jacobians=True)
    for l, (a, b) in enumerate(w.links):
        for d, (k0, k1) in enumerate(((a, b), (b, a))):
            e = 2 * l + d
            for t, ofn in ((0, oracle_photo), (1, oracle_geo)):
                got = out0[(t, e)]
                exp_keys = [1000 + k0, 1000 + k1, 2000 + k0, 3000 + k0] if t == 0 else \
                           [1000 + k0, 1000 + k1, 2000 + k0, 2000 + k1, 3000 + k0, 3000 + k1]
                exp_dims = [6, 6, CS, 1] if t == 0 else [6, 6, CS, CS, 1, 1]
                assert list(got["keys"]) == exp_keys and list(got["dims"]) == exp_dims
                D = sum(exp_dims)
                info = got["info"]
                assert np.array_equal(info, info.T)
                ref = ofn(orc, w2, k0, k1)                     # reference column layout == key order
                assert rel(info[:D, :D], ref["AtA"].astype(np.float64)) < 2e-5, (t, e)
                assert rel(info[:D, D], ref["Atb"].astype(np.float64)) < 2e-4, (t, e)
                assert info[D, D] == pytest.approx(ref["error"], rel=1e-5)
                assert got["error"] == pytest.approx(ref["error"], rel=1e-5)
                # bit for bit what sage_window_factor hands over through ctypes, both PSD modes
                for psd in (0, 1):
                    blocks, gs, f, dims = win.factor(t, e, psd_mode=psd)
                    inf = res[psd][2][(t, e)]["info"]
                    offs = np.concatenate([[0], np.cumsum(dims)])
                    for (i, j), blk in blocks.items():
                        assert np.array_equal(inf[offs[i]:offs[i + 1], offs[j]:offs[j + 1]], blk), (t, e, psd, i, j)
                    assert np.array_equal(inf[:D, D], np.concatenate(gs)) and inf[D, D] == f
    win.close()
