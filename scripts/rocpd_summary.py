#!/usr/bin/env python3
"""Dump the per-kernel statistics of a rocprofv3 (rocpd sqlite) result into a CSV under profiles/.
usage: python scripts/rocpd_summary.py gpurun_out/prof_r01/r01_results.db profiles/r01_kernel_stats.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, tot, avg, pct in rows:
        if len(name) > 120:
            name = name[:117] + "..."
        w.writerow([name, calls, f"{tot:.3f}", f"{avg:.3f}", f"{pct:.2f}"])
print(f"wrote {len(rows)} kernels to {sys.argv[2]}")
