"""Kernel-only micro-bench on the headline window: N linearize + N error passes, HIP-event kernel times (dev tool;
also the workload for the rocprofv3 --pmc passes)."""
import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sage_slam_amd import capi, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W, FS, CS = (int(a) for a in sys.argv[3:7]) if len(sys.argv) > 6 else (128, 160, 16, 32)   # config 4: 16 3 256 320 32 32
w = synth.make_window(K=K, H=H, W=W, FS=FS, CS=CS, L=4, seed=0)
win = capi.Window(w)
win.linearize(); win.error(1); torch.cuda.synchronize()
win.set_profiling(True)
for _ in range(n):
    win.linearize(); win.error(1)
# the LM iteration's own kernels (r05: sage_window_lm_step linearizes with the MERGED pair -- photo_kernel<.., 2> and
# geo_kernel<.., true, true>; win.linearize() above runs the separate ones): a few classic iterations from the initial
# estimate, so that a counter pass sees both sets under their own kernel names
sep = [win.kernel_time(i) for i in range(4)]
cfg = capi.lm_config_default(); cfg.max_inner_evals = 1; cfg.linearize_at_candidate = -1
for _ in range(n):
    win.reset()
    win.lm_step(capi.SageLmState(), cfg)
mer = [win.kernel_time(i) for i in range(4)]
names = ["photo_lin", "geo_lin", "photo_err", "geo_err"]
res = {}
for i, nm in enumerate(names):
    ms, c = sep[i]
    res[nm] = round(ms / max(1, c), 4)
res["photo_lin_merged"] = round(mer[0][0] / max(1, mer[0][1]), 4)
res["geo_lin_merged"] = round(mer[1][0] / max(1, mer[1][1]), 4)
N = w.keyframes[0].homo.shape[0]; E = 2 * len(w.links)
rho = w.P / (w.H * w.W)
bp = 4 * (4 * w.FS * rho + w.CS + 6) * N * E; bg = 4 * (2 * w.CS + 9) * N * E
res["photo_GBs"] = round(bp / res["photo_lin"] / 1e6, 1); res["geo_GBs"] = round(bg / res["geo_lin"] / 1e6, 1)
res["photo_frac"] = round(res["photo_GBs"] / 8000, 4)
print(json.dumps(res))
