"""dev tool (GPU): one rank of an N-rank job on this one device (bench.shard_emulation) without the rest of bench.py --
the workload for a rocprofv3 kernel trace of the emulated rank's LM iteration (scripts/timeline.py <csv> solve_scatter).
usage: python scripts/shard_emu_probe.py [config] [R/N] [steps]     (config 3 = headline K = 64, 2, 4)"""
import json, os, sys
sys.stdout.flush(); _fd = os.dup(1); os.dup2(2, 1)     # RCCL prints its banner to fd 1
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sage_slam_amd import capi, synth
cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 3
r, n = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0/8").split("/"))
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
K, H, W, FS, CS = {3: (64, 128, 160, 16, 32), 2: (16, 128, 160, 16, 32), 4: (16, 256, 320, 32, 32)}[cfgn]
capi.bind_thread_to_device(0)
wh = synth.make_window(K=K, H=H, W=W, FS=FS, CS=CS, L=4, seed=0)
win = capi.Window(wh)
# the one-GPU classic step, measured the plain way
cfg = capi.lm_config_default(); cfg.max_inner_evals = 1; cfg.linearize_at_candidate = -1
st = capi.SageLmState()
import time
def run(nst):
    i = 0
    while i < nst:
        win.reset(); st.iters = 0; st.damp = float(cfg.init_damp)
        m = min(3, nst - i); win.lm_run(st, cfg, m); i += m
run(30); torch.cuda.synchronize(); t0 = time.perf_counter(); run(30); torch.cuda.synchronize()
ms1 = 1e3 * (time.perf_counter() - t0) / 30
out = bench.shard_emulation(capi, torch, win, wh, r, n, steps, 3, ms1, compare_one_gpu=os.environ.get("EMU_COMPARE") == "1")
os.write(_fd, (json.dumps(out) + "\n").encode())
