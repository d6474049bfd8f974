#!/usr/bin/env python3
"""Every counter of one or more rocprofv3 --pmc passes, per kernel, averaged per launch (developer tool).
usage: pmc_all.py [--filter substr] dir [dir ...]"""
import collections
import csv
import glob
import sys

args = sys.argv[1:]
flt = None
if args and args[0] == "--filter":
    flt = args[1]
    args = args[2:]
data = collections.defaultdict(lambda: collections.defaultdict(list))
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "")
            data[n + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(data):
    if flt and flt not in k:
        continue
    c = {n: sum(v) / len(v) for n, v in data[k].items()}
    print(k)
    wc = c.get("SQ_WAVE_CYCLES")
    for n in sorted(c):
        extra = ""
        if wc and n.startswith(("SQ_WAIT", "SQ_ACTIVE_INST", "SQ_INST_CYCLES")):
            extra = "  (%.3f of wave cycles)" % (c[n] / wc)
        if n.startswith("SQ_INSTS") and c.get("SQ_WAVES"):
            extra = "  (%.1f per wave)" % (c[n] / c["SQ_WAVES"])
        print("    %-28s %16.0f%s" % (n, c[n], extra))
