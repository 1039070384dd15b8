#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel over the passes in a directory (csv output)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sage_slam_amd.build import kernel_source_sha16
print("# kernel_source_sha16:", kernel_source_sha16(), "(photo_kernels.hip geo_kernels.hip sage_device.h sage_internal.h)")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            short = k.split("(")[0].replace("void sage::", "").replace("sage::", "")
            if "kernel" not in short or "at::" in k:
                continue
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
# r06: the kernels' durations IN the pass that collected GRBM_GUI_ACTIVE (kernel trace of the same rocprofv3 run): the engine clock
# of the counter pass is GRBM_GUI_ACTIVE / 8 XCDs / that duration (bench.py: measured_clock_hz)
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        if "GRBM_GUI_ACTIVE" not in fh.read():
            continue
    kt = f.replace("counter_collection.csv", "kernel_trace.csv")
    if not os.path.exists(kt):
        continue
    with open(kt) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            short = k.split("(")[0].replace("void sage::", "").replace("sage::", "")
            if "kernel" not in short or "at::" in k:
                continue
            dur[short].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
for k in sorted(acc):
    print(k)
    if dur.get(k):
        v = dur[k]
        print(f"    {'DURATION_US_IN_GRBM_PASS':42s} avg/dispatch {sum(v)/len(v):.6g}   (n={len(v)})")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:42s} avg/dispatch {sum(v)/len(v):.6g}   (n={len(v)})")
