import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sage_slam_amd import capi, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w = synth.make_window(K=K, H=128, W=160, FS=16, CS=32, L=4, seed=0)
for maxd in (1e-2, 1e2):
    win = capi.Window(w)
    cfg = capi.lm_config_default(); cfg.max_damp = maxd; cfg.max_inner_evals = 1
    st = capi.SageLmState()
    for i in range(14):
        t0 = time.perf_counter(); win.lm_step(st, cfg); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"maxd {maxd} it {i}: err {st.error:.4f} cand {st.candidate_error:.4f} acc {st.accepted} damp {st.damp:.1e} {dt*1e3:.2f} ms")
    # distance to truth
    dp = [np.linalg.norm(win.get_keyframe(k)[0][9:] - w.keyframes[k].t_true) for k in range(K)]
    print("mean |t - t_true|", np.mean(dp), "initial", np.mean([np.linalg.norm(w.keyframes[k].t - w.keyframes[k].t_true) for k in range(K)]))
    win.close()
