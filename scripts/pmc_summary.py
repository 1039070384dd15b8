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
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:42s} avg/dispatch {sum(v)/len(v):.6g}   (n={len(v)})")
