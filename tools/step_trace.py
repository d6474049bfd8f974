#!/usr/bin/env python3
"""Print the kernels of the last training step from a rocprofv3 --kernel-trace CSV (ordered by start time)."""
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
f = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True))
rows = list(csv.DictReader(open(f[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_adam")]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
tot = 0.0
for r in rows[a + 1:b + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if d > thr:
        print("%8.1f us  @%8.1f  %s grid=%s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e3, r["Kernel_Name"][:70], r.get("Grid_Size_X", "")))
print("sum of kernel time %.1f us, step span %.1f us, %d launches" % (tot, (int(rows[b]["End_Timestamp"]) - t0) / 1e3, b - a))
