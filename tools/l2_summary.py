#!/usr/bin/env python3
"""Per-kernel L1 -> L2 read traffic and its latency from one rocprofv3 --pmc pass of
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum (no trace domain in the same pass).
TCP_TCC_READ_REQ counts the read requests the CUs' vector L1s send to L2, TCP_TCC_READ_REQ_LATENCY the cycles they were outstanding (summed
over requests): latency / requests = the average L2 round trip a CU sees.  Requests per launch x 64 B and x 128 B bracket the L2 -> L1 bytes
(gfx950's L1 asks for 64-B or 128-B lines); the fused chain kernels' modelled weight stream (bench.py `l2_stream`) is printed beside them."""
import collections
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_summary import short_name  # noqa: E402


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(path)):
        k = short_name(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
    return acc, calls


if __name__ == "__main__":
    acc, calls = load(sys.argv[1])
    print("%-46s %6s %12s %10s %10s %10s %8s" % ("kernel", "calls", "rd req/launch", "MB @64B", "MB @128B", "lat cyc", "L2 hit"))
    rows = []
    for k, c in acc.items():
        n = max(1, calls[k].get("TCP_TCC_READ_REQ_sum", 0))
        req = c.get("TCP_TCC_READ_REQ_sum", 0.0) / n
        lat = c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / max(1.0, c.get("TCP_TCC_READ_REQ_sum", 0.0))
        hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
        rows.append((req, k, n, lat, hit / max(1.0, hit + miss)))
    for req, k, n, lat, hr in sorted(rows, reverse=True)[:20]:
        print("%-46s %6d %12.3e %10.1f %10.1f %10.0f %8.3f" % (k[:46], n, req, req * 64 / 1e6, req * 128 / 1e6, lat, hr))
