#!/usr/bin/env python3
"""[needs `git apply tools/exp/patches/dec_fused_switches.patch` first: the hooks are not in the product source]
Diagnostic: where a stage-3 slot of the fused decoder forward spends its cycles (build: tools/build_variant.sh df_stamp dec_fused.hip -DDF_STAMP;
run with SPAIR_HIP_LIB=build/libspair_df_stamp.so).  Stamps per slot: top | after the counted wait | after the barrier."""
import ctypes, os, sys, runpy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_dec.py"))
from spair_pytorch_amd import _lib as L
out = (ctypes.c_ulonglong * (8 * 256))()
L.check(L.lib().spair_df_stamps(out), "stamps")
st = np.array(out[:], dtype=np.float64).reshape(2, 4, 256)
for wg in range(2):
    for w in range(4):
        t = st[wg, w, :150].reshape(50, 3)
        top, aw, ab = t[:, 0], t[:, 1], t[:, 2]
        slot = np.diff(top)[2:47]
        print("wg%d wave%d: slot %.0f cycles (min %.0f max %.0f) | lgkm+vmcnt wait %.0f | barrier %.0f | body %.0f" %
              (wg, w, slot.mean(), slot.min(), slot.max(), (aw - top)[2:47].mean(), (ab - aw)[2:47].mean(), (top[1:] - ab[:-1])[2:47].mean()))
t = st[0, 0, :150].reshape(50, 3)
print("per slot (even = first half of a pair, odd = second half + epilogue):", np.round(np.diff(t[:, 0])[2:14]))
