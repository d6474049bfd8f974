#!/usr/bin/env python3
"""Per-kernel SQ counter summary from rocprofv3 --pmc passes (one directory per pass, counters averaged per launch).
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); LDS busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES;
the wave-cycle fractions (waiting / issuing) are relative to SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import sys


def short_name(full):
    """k_name<template args> without return type, namespace or the parameter list ("void (anonymous namespace)::k_x<1, 2>(Args)")."""
    n = full.replace("(anonymous namespace)::", "").replace("void ", "")
    depth, out = 0, []
    for ch in n:
        if ch == "<": depth += 1
        if ch == "(" and depth == 0: break
        if ch == ">": depth -= 1
        out.append(ch)
    return "".join(out).strip()


data = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = short_name(r["Kernel_Name"])
            data[n + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
keep = ("nt16", "tn16", "tn_ring", "chain", "render", "pw_stack", "conv0", "count_kl", "k_dec", "k_conv")
print("%-58s %9s %9s %9s %9s %9s %9s %9s" % ("kernel", "mfma_util", "lds_busy", "bank_conf", "wait_any", "wait_inst", "issue", "valu"))
for k in sorted(data):
    if not any(t in k for t in keep):
        continue
    c = {n: sum(v) / len(v) for n, v in data[k].items()}
    cu = max(1.0, c.get("SQ_BUSY_CU_CYCLES", 0.0))
    wc = max(1.0, c.get("SQ_WAVE_CYCLES", 0.0))
    ia = max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0.0))
    print("%-58s %9.3f %9.3f %9.3f %9.3f %9.3f %9.3f %9.3f" % (k[:58], c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * cu), c.get("SQ_LDS_IDX_ACTIVE", 0) / cu,
          c.get("SQ_LDS_BANK_CONFLICT", 0) / ia, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
          c.get("SQ_ACTIVE_INST_VALU", 0) / wc))
