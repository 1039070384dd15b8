"""Tracker front-end timing (dev tool): one sage_track_frame LM at the reference's tracker size (64x80, N=3072, L=4)
and at 128x160 dense; prints ms per call and per LM iteration.  usage: python scripts/track_probe.py [shuffle]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sage_slam_amd import capi, synth
from tests.helpers import presample_source
shuffle = len(sys.argv) > 1
for (H, W, ns) in ((64, 80, 3072), (128, 160, 0)):
    w = synth.make_window(K=2, H=H, W=W, FS=16, CS=32, L=4, n_samples=ns, seed=31, pose_noise=0.0)
    a, b = w.keyframes[0], w.keyframes[1]
    if not shuffle and ns:
        o = np.argsort(a.loc1d); a.loc1d = a.loc1d[o]; a.homo = np.ascontiguousarray(a.homo[o])
    ws = capi.Workspace()
    pyr = capi.make_pyramid(w.cams[0], w.L)
    mask = torch.from_numpy(w.mask).cuda()
    kfs = [capi.DeviceKeyframe(k, w.H, w.W) for k in w.keyframes]
    feat0s = presample_source(None, w, a)
    dpts0 = (np.float32(a.scale_true) * (a.bias + a.basis @ a.code_true))[a.loc1d].astype(np.float32)
    R10, t10 = synth.relative_pose(a.R_true, a.t_true, b.R_true, b.t_true)
    pose0 = capi.pack_pose(synth.so3_exp(np.array([0.004, -0.003, 0.002])) @ R10, t10 + np.array([0.004, -0.003, 0.002], np.float32))
    cfg = capi.lm_config_default()
    prob = capi.SageTrackProblem()
    f0 = torch.from_numpy(feat0s).cuda(); dp = torch.from_numpy(dpts0).cuda(); wd = torch.from_numpy(w.photo_weights).cuda()
    prob.ws = ws.h; prob.mask1_dev = mask.data_ptr(); prob.dpts0_dev = dp.data_ptr()
    prob.homo_dev = kfs[0].homo.data_ptr(); prob.feat0s_dev = f0.data_ptr(); prob.feat1_dev = kfs[1].feat_pyr.data_ptr()
    prob.grad1_dev = kfs[1].grad_pyr.data_ptr(); prob.weights_dev = wd.data_ptr(); prob.pyr = pyr
    prob.eps = w.eps; prob.N = a.homo.shape[0]; prob.FS = w.FS
    for rep in range(4):
        ph = pose0.copy(); sc = C.c_float(1.0); fe = C.c_float(); it = C.c_int()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = capi.lib().sage_track_frame(C.byref(cfg), 6, C.byref(prob), ph.ctypes.data_as(C.POINTER(C.c_float)),
                                         C.byref(sc), C.byref(fe), C.byref(it))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print(f"{H}x{W} N={prob.N} shuffled={shuffle}: rc {rc}  {it.value} LM iterations  {dt:.3f} ms  = {dt / max(1, it.value) * 1e3:.0f} us/iteration  final error {fe.value:.5f}")
