"""dev tool (GPU box): one process = one K = 64 window; prints where the LM thread ran, how busy the box was BEFORE the window
existed (per-CPU /proc/stat deltas over 0.3 s: other tenants' load), and the step / solve times -- run it N times to see
what a slow-solve process has in common (DESIGN s7: ~1 process in 8 measures +60..200 us per step for its whole life)."""
import ctypes, json, os, sys, time
_libc = ctypes.CDLL(None)
getcpu = lambda: int(_libc.sched_getcpu())
sys.stdout.flush(); _fd = os.dup(1); os.dup2(2, 1)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cpu_times():
    out = {}
    for ln in open("/proc/stat"):
        if ln.startswith("cpu") and ln[3].isdigit():
            f = ln.split()
            v = [int(x) for x in f[1:9]]
            out[int(f[0][3:])] = (sum(v), v[3] + v[4])     # total, idle + iowait
    return out


def busy(a, b):
    return {c: 1.0 - (b[c][1] - a[c][1]) / max(1, b[c][0] - a[c][0]) for c in a if c in b}


def l3_of(cpu):
    try:
        return open(f"/sys/devices/system/cpu/cpu{cpu}/cache/index3/shared_cpu_list").read().strip()
    except OSError:
        return "?"


a = cpu_times(); time.sleep(0.3); b = cpu_times()
pre = busy(a, b)
import torch
from sage_slam_amd import capi, synth
ncpu = capi.bind_thread_to_device(0)
allowed = sorted(os.sched_getaffinity(0))
wh = synth.make_window(K=64, H=128, W=160, FS=16, CS=32, L=4, seed=0)
win = capi.Window(wh)
cfg = capi.lm_config_default(); cfg.max_inner_evals = 1; cfg.linearize_at_candidate = -1
st = capi.SageLmState()


def run(nst):
    i = 0
    while i < nst:
        win.reset(); st.iters = 0; st.damp = float(cfg.init_damp)
        m = min(3, nst - i); win.lm_run(st, cfg, m); i += m


run(30); torch.cuda.synchronize()
cpu0 = getcpu()
a = cpu_times(); t0 = time.perf_counter(); run(60); torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 60; b = cpu_times()
cpu1 = getcpu()
dur = busy(a, b)
busy_pre = sorted(c for c in allowed if pre.get(c, 0) > 0.2)
hot = sorted(c for c in dur if dur[c] > 0.5)
hog_trace = None
if len(sys.argv) > 1 and sys.argv[1] == "hog":
    # crowd the helpers' cores with busy loops and watch the placement monitor move them (DESIGN s7)
    import subprocess
    before = [c for c in capi.solver_helper_cpus() if c >= 0]
    hogs = [subprocess.Popen([sys.executable, "-c", "import os\nos.sched_setaffinity(0, {%d})\nwhile True: pass" % c]) for c in set(before)]
    hog_trace = []
    try:
        for _ in range(8):
            t0 = time.perf_counter(); run(120); torch.cuda.synchronize()
            hog_trace.append((round(1e3 * (time.perf_counter() - t0) / 120, 4), capi.solver_helper_cpus(), capi.solver_placement_moves()))
    finally:
        for h in hogs:
            h.kill()
out = dict(ms_per_step=round(ms, 4), helper_cpus=capi.solver_helper_cpus(), monitor_moves=capi.solver_placement_moves(), hog_trace=hog_trace, cpu_start=cpu0, cpu_end=cpu1, l3=l3_of(cpu1), allowed=f"{allowed[0]}..{allowed[-1]} ({len(allowed)})",
           busy_before_in_mask=busy_pre, n_busy_before_all=sum(1 for v in pre.values() if v > 0.2),
           hot_during=hot, loadavg=open("/proc/loadavg").read().split()[:3])
os.write(_fd, (json.dumps(out) + "\n").encode())
