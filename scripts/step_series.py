"""Per-step wall time of the headline LM loop (dev tool): does the step time settle after the warm-up?  Steps are run
three at a time (restart every 3, like bench.py) through sage_window_lm_run."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sage_slam_amd import capi, synth
w = synth.make_window(K=64, H=128, W=160, FS=16, CS=32, L=4, seed=0)
win = capi.Window(w)
cfg = capi.lm_config_default(); cfg.max_inner_evals = 1
st = capi.SageLmState()
out = []
for it in range(40):
    win.reset(); st.iters = 0; st.damp = float(cfg.init_damp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    list(win.lm_run(st, cfg, 3))
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) * 1e3 / 3)
print(" ".join(f"{v:.3f}" for v in out))
