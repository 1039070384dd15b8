"""dev tool (GPU): cost of one per-Values batched prepass (f2) on the headline window: kernels + D2H of the per-edge results,
then NearestPsd of all factors on host threads and the block cutting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sage_slam_amd import capi, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = synth.make_window(K=K, H=128, W=160, FS=16, CS=32, L=4, seed=0)
win = capi.Window(w)
poses = np.stack([np.concatenate([np.asarray(k.R, np.float32).reshape(-1), np.asarray(k.t, np.float32).reshape(-1)]) for k in w.keyframes])
codes = np.stack([np.asarray(k.code, np.float32).reshape(-1) for k in w.keyframes]); scales = np.array([k.scale for k in w.keyframes], np.float32)
win.prepass(poses, codes, scales, True)
for rep in range(3):
    p2 = poses.copy(); p2[1, 9:] += np.float32(1e-4 * (rep + 1))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert win.prepass(p2, codes, scales, True)
    t1 = time.perf_counter()
    win.prepare_factors(2, 0)
    t2 = time.perf_counter()
    win.prepare_factors(1, 0)
    t3 = time.perf_counter()
    print(f"K={K}: prepass (batched linearize + D2H of {4 * len(w.links)} factors) {1e3 * (t1 - t0):.2f} ms | NearestPsd as written, all factors, host threads {1e3 * (t2 - t1):.1f} ms | Higham {1e3 * (t3 - t2):.1f} ms", flush=True)
win.close()
