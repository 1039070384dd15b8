"""Host side of the BASELINE config-1 tracker scene (2 keyframes, 64x80x16 feature maps, 32-dim code, N = 3072 seeded
samples, NK matched keypoints) and the reference's term composition over the CPU oracle's kernels
(CameraTracker::ComputeJacobianAndError / ComputeError, camera_tracker.cpp:220-374).  No GPU, no engine library: shared by
tests/test_gpu_tracker.py (which adds the device buffers), tests/golden/make_lm_trace_golden.py and tests/test_lm_trace_golden.py."""
import numpy as np

from sage_slam_amd import synth
from tests.helpers import presample_source


def pack_pose(R, t):
    return np.concatenate([np.asarray(R, np.float32).reshape(-1), np.asarray(t, np.float32).reshape(-1)]).astype(np.float32)


class HostScene:
    """frame 0 (tracked) -> frame 1 (reference keyframe) of a synthetic window, plus NK matched keypoints"""

    def __init__(self, orc, seed=31, NK=160, kp_noise_px=0.4, kp_noise_depth=0.002):
        self.orc = orc
        w = synth.make_window(K=2, H=64, W=80, FS=16, CS=32, L=4, n_samples=3072, seed=seed, pose_noise=0.0)
        self.w = w
        a, b = w.keyframes[0], w.keyframes[1]
        self.a, self.b = a, b
        self.feat0s = presample_source(orc, w, a)
        self.unscaled = (a.bias + a.basis @ a.code_true)[a.loc1d].astype(np.float32)       # dpt_map_0 / dpt_scale_0
        self.s_true = np.float32(a.scale_true)
        self.R10, self.t10 = synth.relative_pose(a.R_true, a.t_true, b.R_true, b.t_true)
        rng = np.random.default_rng(seed + 1)
        cam = w.cams[0]
        xs = rng.integers(10, w.W - 10, NK); ys = rng.integers(10, w.H - 10, NK)
        loc = ys * w.W + xs
        self.kp_homo0 = np.stack([(xs - cam.cx) / cam.fx, (ys - cam.cy) / cam.fy, np.ones(NK)], 1).astype(np.float32)
        self.kp_unscaled = (a.bias + a.basis @ a.code_true)[loc].astype(np.float32)
        X1 = (self.R10.astype(np.float64) @ (float(self.s_true) * self.kp_unscaled[:, None] * self.kp_homo0).T).T + self.t10
        self.kp_matched_2d = (np.stack([X1[:, 0] / X1[:, 2] * cam.fx + cam.cx, X1[:, 1] / X1[:, 2] * cam.fy + cam.cy], 1)
                              + rng.normal(0, kp_noise_px, (NK, 2))).astype(np.float32)
        self.kp_dpts1 = (X1[:, 2] + rng.normal(0, kp_noise_depth, NK)).astype(np.float32)
        self.kp_homo1 = np.stack([X1[:, 0] / X1[:, 2] + rng.normal(0, kp_noise_px / cam.fx, NK),
                                  X1[:, 1] / X1[:, 2] + rng.normal(0, kp_noise_px / cam.fy, NK), np.ones(NK)], 1).astype(np.float32)
        self.NK = NK
        self.reproj_loss_param = 1e-4 * w.W * w.W           # reproj_loss_param_factor * width^2 (camera_tracker.cpp:1076)
        self.mg_loss_param = 0.1 * float(np.mean(a.bias ** 2))
        self.reproj_weight, self.mg_weight = 0.05, 3.0

    def oracle_callbacks(self, dof, use_photo, use_kp):
        """ComputeJacobianAndError / ComputeError restated over the oracle kernels: fp32 sums term by term"""
        orc, w, a, b = self.orc, self.w, self.a, self.b
        cam = w.cams[0]
        F = np.float32

        def depths(s):
            if dof == 7:
                return F(s) * self.unscaled, F(s) * self.kp_unscaled
            return self.s_true * self.unscaled, self.s_true * self.kp_unscaled

        def lin(p, s):
            R, t = p[:9].reshape(3, 3), p[9:]
            dp, kdp = depths(s)
            A = np.zeros((dof, dof), F); g = np.zeros(dof, F); e = F(0)
            if use_photo:
                o = orc.tracker_photo_jac_error(dof, R, t, w.mask, dp, a.homo, self.feat0s, b.feat_pyr, b.grad_pyr,
                                                w.level_offsets, w.cams, w.eps, w.photo_weights, scale0=s)
                A = A + o["AtA"].astype(F); g = g + o["Atb"].astype(F); e = F(e + F(o["error"]))
            if use_kp and dof == 6:
                o = orc.tracker_reproj_jac_error(R, t, kdp, self.kp_homo0, self.kp_matched_2d, cam, w.eps,
                                                 self.reproj_loss_param, self.reproj_weight)
                A = A + o["AtA"].astype(F); g = g + o["Atb"].astype(F); e = F(e + F(o["error"]))
            if use_kp and dof == 7:
                o = orc.match_geom_jac_error(3, "fair", R, t, dpts0=kdp, dpts1=self.kp_dpts1, homo0=self.kp_homo0,
                                             homo1=self.kp_homo1, scale0=s, loss_param=self.mg_loss_param,
                                             weight=self.mg_weight)
                A = A + o["AtA"].astype(F); g = g + o["Atb"].astype(F); e = F(e + F(o["error"]))
            return A, g, float(e)

        def err(p, s):
            R, t = p[:9].reshape(3, 3), p[9:]
            dp, kdp = depths(s)
            e = F(0)
            if use_photo:
                e = F(e + F(orc.tracker_photo_error(R, t, w.mask, dp, a.homo, self.feat0s, b.feat_pyr, w.level_offsets,
                                                    w.cams, w.eps, w.photo_weights)[0]))
            if use_kp and dof == 6:
                e = F(e + F(orc.tracker_reproj_error(R, t, kdp, self.kp_homo0, self.kp_matched_2d, cam, w.eps,
                                                     self.reproj_loss_param, self.reproj_weight)[0]))
            if use_kp and dof == 7:
                e = F(e + F(orc.match_geom_error(2, "fair", R, t, dpts0=kdp, dpts1=self.kp_dpts1, homo0=self.kp_homo0,
                                                 homo1=self.kp_homo1, loss_param=self.mg_loss_param,
                                                 weight=self.mg_weight)))
            return float(e)

        return lin, err

    def start_pose(self, rot=(0.004, -0.003, 0.002), trans=(0.004, -0.003, 0.002)):
        return pack_pose(synth.so3_exp(np.array(rot, np.float64)) @ self.R10,
                                   self.t10 + np.array(trans, np.float32))
