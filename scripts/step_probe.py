"""Where does one LM step go?  Python-level timers with device syncs around each phase (dev tool)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sage_slam_amd import capi, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = synth.make_window(K=K, H=128, W=160, FS=16, CS=32, L=4, seed=0)
win = capi.Window(w)
def T(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for it in range(6):
    t_lin = T(win.linearize)
    t_e0 = T(lambda: win.total_error(True))
    t_solve = T(lambda: win.solve(1e-4, want_norm=False))
    t_err = T(lambda: win.error(1))
    t_e1 = T(lambda: win.total_error(False))
    t_acc = T(win.accept)
    print(f"it {it}: linearize {t_lin:.2f} total_error {t_e0:.2f} solve {t_solve:.2f} error {t_err:.2f} total_error {t_e1:.2f} accept {t_acc:.2f} ms")
win.set_profiling(True)
for _ in range(5):
    win.linearize(); win.error(1)
print("kernel ms (photo lin, geo lin, photo err, geo err):", [tuple(round(x, 3) for x in (lambda a: (a[0] / max(1, a[1]), a[1]))(win.kernel_time(i))) for i in range(4)])
