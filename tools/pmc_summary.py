#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as MI355X_MICROARCH.md prescribes:
FETCH_SIZE on gfx950 tallies 128-B requests at 64 B -> doubled; WRITE_SIZE as reported.  Both counters are in KB."""
import csv
import collections
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



def load(path):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        name = short_name(r["Kernel_Name"])
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return acc



if __name__ == "__main__":
    f = load(sys.argv[1])
    w = load(sys.argv[2])
    print("%-46s %8s %12s %12s %12s" % ("kernel", "calls", "read MB", "write MB", "total MB"))
    rows = []
    for k in sorted(set(f) | set(w)):
        calls = max(f[k][1], w[k][1])
        rd = 2.0 * f[k][0] / max(1, f[k][1]) / 1024.0        # KB -> MB, x2 (gfx950 correction)
        wr = w[k][0] / max(1, w[k][1]) / 1024.0
        rows.append((rd + wr, k, calls, rd, wr))
    for tot, k, calls, rd, wr in sorted(rows, reverse=True)[:24]:
        print("%-46s %8d %12.1f %12.1f %12.1f" % (k[:46], calls, rd, wr, tot))

    if len(sys.argv) > 3:      # optional: the four kernels bench.py prices, as JSON (matched by substring: template arguments / mangling vary)
        import json
        names = {"chain_bwd": "k_chain_bwd", "chain_fwd": "k_chain_fwd", "render_fwd": "k_render_fwd", "render_bwd": "k_render_bwd"}
        out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 3 --warmup 1`; "
                         "FETCH_SIZE doubled (gfx950), per launch"}
        # hashes of the kernel sources this was collected on: bench.py refuses to quote the traffic once they change
        import hashlib, os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out["src_sha16"] = {}
        for src in ("spair_pytorch_amd/csrc/chain.hip", "spair_pytorch_amd/csrc/render2.hip", "spair_pytorch_amd/csrc/render3.hip"):
            out["src_sha16"][src] = hashlib.sha256(open(os.path.join(root, src), "rb").read()).hexdigest()[:16]
        for key, kn in names.items():
            for tot, k, calls, rd, wr in rows:
                if kn in k:
                    out[key] = {"read_MB": round(rd, 1), "write_MB": round(wr, 1), "kernel": k[:60]}
        json.dump(out, open(sys.argv[3], "w"), indent=1)
