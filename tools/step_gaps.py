#!/usr/bin/env python3
"""Idle gaps (no kernel running on any stream) inside the last training step of a rocprofv3 --kernel-trace CSV."""
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
f = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True))
rows = list(csv.DictReader(open(f[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_adam")]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
cur_end, last, idle = t0, "k_adam", 0.0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_end:
        g = (s - cur_end) / 1e3
        idle += g
        if g >= thr:
            print("%6.1f us idle @%8.1f  after %-40s before %s" % (g, (cur_end - t0) / 1e3, last[:40], r["Kernel_Name"][:40]))
    if e > cur_end:
        cur_end, last = e, r["Kernel_Name"]
print("idle %.1f us of %.1f us" % (idle, (int(rows[b]["End_Timestamp"]) - t0) / 1e3))
