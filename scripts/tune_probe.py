"""dev tool (GPU): what sage_window_tune_runs picks on a few window shapes, and the kernel times before / after."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sage_slam_amd import capi, synth
shapes = [(64, 128, 160, 16), (16, 128, 160, 16), (16, 256, 320, 32), (16, 192, 256, 32), (32, 192, 256, 16), (16, 256, 320, 16), (16, 128, 160, 32), (8, 384, 480, 32)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for K, H, W, FS in shapes:
    w = synth.make_window(K=K, H=H, W=W, FS=FS, CS=32, L=4, seed=0)
    win = capi.Window(w)
    def times():
        win.set_profiling(True)
        win.linearize(); win.error(1); [win.kernel_time(i) for i in range(4)]
        for _ in range(3):
            win.linearize(); win.error(1)
        t = [win.kernel_time(i) for i in range(4)]
        win.set_profiling(False)
        return round(t[0][0] / t[0][1], 4), round(t[2][0] / t[2][1], 4)
    before = times()
    r = win.tune_runs()
    after = times()
    print(json.dumps(dict(K=K, H=H, W=W, FS=FS, before=before, after=after, **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})))
    del win
