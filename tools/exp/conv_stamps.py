#!/usr/bin/env python3
"""[needs `git apply tools/exp/patches/conv_s2_switches.patch` first: the hooks are not in the product source]
Diagnostic: per-K-step timing of the patch-resident conv kernel's consumer and loader waves (build: tools/build_variant.sh cp_stamp conv_s2.hip
-DCP_STAMP; run with SPAIR_HIP_LIB=build/libspair_cp_stamp.so).  Stamps per step: arrival at the barrier | barrier exit | (loaders) issue done."""
import ctypes, os, sys, runpy
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
runpy.run_path(os.path.join(root, "tools", "bench_conv.py"))
from spair_pytorch_amd import _lib as L
out = (ctypes.c_ulonglong * (2 * 8 * 128))()
L.check(L.lib().spair_cp_stamps(out), "stamps")
st = np.array(out[:], dtype=np.float64).reshape(2, 8, 128)
for wg in range(2):
    for w in range(8):
        t = st[wg, w, :96].reshape(32, 3)
        arr, ex, iss = t[:, 0], t[:, 1], t[:, 2]
        step = np.diff(arr)[2:30]
        if w < 4:
            print("wg%d consumer %d: step %.0f cycles (min %.0f max %.0f) | barrier wait %.0f | compute %.0f" %
                  (wg, w, step.mean(), step.min(), step.max(), (ex - arr)[2:30].mean(), (arr[1:] - ex[:-1])[2:30].mean()))
        else:
            print("wg%d loader   %d: step %.0f cycles | barrier wait %.0f | issue %.0f | landing wait %.0f" %
                  (wg, w, step.mean(), (ex - arr)[2:30].mean(), (iss - ex)[2:30].mean(), (arr[1:] - iss[:-1])[2:30].mean()))
t = st[0, 0, :96].reshape(32, 3)
print("consumer 0 steps:", np.round(np.diff(t[:, 0])[:16]))
t = st[0, 4, :96].reshape(32, 3)
print("loader 4 issue per step:", np.round((t[:, 2] - t[:, 1])[:16]))
