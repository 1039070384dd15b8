"""Timeline of one steady-state LM step from a rocprofv3 --kernel-trace CSV (dev tool).
usage: python scripts/timeline.py <kernel_trace.csv> [marker_kernel_substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "depth_batch"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# steps start at every other depth_batch launch (linearize pass, then error pass)
starts = [i for i, e in enumerate(ev) if marker in e[2]]
if len(starts) < 8:
    sys.exit("not enough marker kernels")
if marker == "depth_batch":
    i0, i1 = starts[-6], starts[-4]      # one full step well inside steady state
else:
    i0, i1 = starts[-2], starts[-1]      # the last full span between two marker launches
t0 = ev[i0][0]
prev_end = t0
for s, e, n in ev[i0:i1]:
    name = n.split("(")[0][-60:]
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {name}")
    prev_end = e
print(f"step span {(ev[i1][0] - t0) / 1e3:.1f} us")
