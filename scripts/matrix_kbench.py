"""dev tool (GPU): kernel times over the template matrix (CS, FS) x resolution, K = 16 dense windows."""
import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sage_slam_amd import capi, synth
for (H, W) in ((128, 160), (64, 80)):
    for CS in (16, 32):
        for FS in (16, 32):
            K = 16 if H == 128 else 48
            w = synth.make_window(K=K, H=H, W=W, FS=FS, CS=CS, L=4, seed=0)
            win = capi.Window(w)
            win.linearize(); win.error(1); torch.cuda.synchronize()
            win.set_profiling(True)
            for _ in range(6):
                win.linearize(); win.error(1)
            res = {}
            for i, nm in enumerate(["photo_lin", "geo_lin", "photo_err"]):
                ms, c = win.kernel_time(i); res[nm] = round(ms / max(1, c), 4)
            npx = 2 * len(w.links) * w.keyframes[0].homo.shape[0]
            print(f"{H}x{W} K={K} CS={CS} FS={FS}: {json.dumps(res)}  Gpx/s photo {npx / res['photo_lin'] / 1e6:.2f} geo {npx / res['geo_lin'] / 1e6:.2f} err {npx / res['photo_err'] / 1e6:.2f}", flush=True)
            win.close()
