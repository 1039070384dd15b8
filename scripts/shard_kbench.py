"""dev tool (GPU): kernel times of ONE rank's shard of the headline window (world = 8) and of the K = 16 window, for the
sub-tiles-per-workgroup heuristic (SAGE_PHOTO_TPB / SAGE_GEO_TPB override it)."""
import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sage_slam_amd import capi, synth
for (K, world) in ((64, 8), (64, 4), (64, 2), (16, 1)):
    w = synth.make_window(K=K, H=128, W=160, FS=16, CS=32, L=4, seed=0)
    win = capi.Window(w, rank=0, world=world)
    win.linearize(); win.error(1); torch.cuda.synchronize()
    win.set_profiling(True)
    for _ in range(8):
        win.linearize(); win.error(1)
    res = {}
    for i, nm in enumerate(["photo_lin", "geo_lin", "photo_err"]):
        ms, c = win.kernel_time(i); res[nm] = round(ms / max(1, c), 4)
    print(f"K={K} world={world}:", json.dumps(res), flush=True)
    win.close()
