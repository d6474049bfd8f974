#!/usr/bin/env python3
"""Diagnostic: per-K-step timing of the patch-resident dgrad kernel (build: tools/build_variant.sh dg_stamp conv_s2_dgrad.hip -DDG_STAMP;
run with SPAIR_HIP_LIB=build/libspair_dg_stamp.so)."""
import ctypes, os, sys, runpy
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
runpy.run_path(os.path.join(root, "tools", "bench_conv.py"))
from spair_pytorch_amd import _lib as L
out = (ctypes.c_ulonglong * (2 * 8 * 256))()
L.check(L.lib().spair_dg_stamps(out), "stamps")
st = np.array(out[:], dtype=np.float64).reshape(2, 8, 256)
for wg in range(2):
    for w in (0, 3, 4, 7):
        t = st[wg, w, :128].reshape(32, 4)
        arr, ex, ep = t[:, 0], t[:, 1], t[:, 2]
        print("wg%d wave%d (%s): steps" % (wg, w, "compute" if w < 4 else "loader"), np.round(np.diff(arr)[:17]).astype(int).tolist(),
              "| barrier wait", int((ex - arr)[1:31].mean()), "| step body", int((arr[1:] - ex[:-1])[1:31].mean()))
        if w < 4:
            print("     last step of a class: body before the epilogue", [int(ep[k] - ex[k]) for k in (7, 15, 23)], " epilogue", [int(arr[k + 1] - ep[k]) for k in (7, 15, 23)])
