"""sage_window_lm_step timing on the headline window (dev tool; SAGE_DEBUG_TIMING=1 prints the solve's phases)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sage_slam_amd import capi, synth
w = synth.make_window(K=64, H=128, W=160, FS=16, CS=32, L=4, seed=0)
win = capi.Window(w)
cfg = capi.lm_config_default(); cfg.max_inner_evals = 1
st = capi.SageLmState()
for it in range(8):
    if it % 4 == 0:
        win.reset(); st.iters = 0; st.damp = float(cfg.init_damp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    win.lm_step(st, cfg)
    torch.cuda.synchronize()
    print(f"lm_step {it}: {(time.perf_counter() - t0) * 1e3:.3f} ms accepted {st.accepted}", file=sys.stderr)
