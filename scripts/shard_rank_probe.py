"""dev tool (GPU): linearize kernel times of single ranks' shards (separate kernels, sage_window_linearize) for a config.
usage: python scripts/shard_rank_probe.py config world rank [rank ...]     (env SAGE_PHOTO_TPB / SAGE_GEO_TPB apply)"""
import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sage_slam_amd import capi, synth
cfgn, world = int(sys.argv[1]), int(sys.argv[2])
K, H, W, FS, CS = {3: (64, 128, 160, 16, 32), 2: (16, 128, 160, 16, 32), 4: (16, 256, 320, 32, 32)}[cfgn]
w = synth.make_window(K=K, H=H, W=W, FS=FS, CS=CS, L=4, seed=0)
for r in [int(a) for a in sys.argv[3:]]:
    win = capi.Window(w, rank=r, world=world)
    for _ in range(3):
        win.linearize()
    torch.cuda.synchronize()
    win.set_profiling(True)
    for _ in range(10):
        win.linearize()
    res = {}
    for i, nm in enumerate(["photo_lin", "geo_lin"]):
        ms, c = win.kernel_time(i); res[nm] = round(ms / max(1, c), 4)
    edges = capi.shard_edges(len(w.links), r, world)
    print(f"config {cfgn} rank {r}/{world}: {len(edges)} directed edges", json.dumps(res), flush=True)
    win.close()
