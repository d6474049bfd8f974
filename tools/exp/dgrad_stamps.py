#!/usr/bin/env python3
"""Diagnostic: per-K-step s_memtime stamps of the patch-resident dgrad kernel.  The stamps are NOT in the shipped kernel: apply
tools/exp/dgrad_stamps.patch (git apply), rebuild, run this on the GPU box, then revert.  Every stamp drains lgkmcnt (s_memtime returns
through it), so the phases inside a step are serialised by the stamps themselves; the step periods and the class-boundary steps are what
to read.  Measured (round 3, pipelined kernel, conv_1 shape): steady step 1,850 cycles, class-boundary step 5,300-6,300 (epilogue),
loader: issue 520-640, DMA wait 350, waiting for the computing waves 1,470-1,570."""
import ctypes, os, sys, runpy
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
runpy.run_path(os.path.join(root, "tools", "bench_conv.py"))
from spair_pytorch_amd import _lib as L
out = (ctypes.c_ulonglong * (2 * 8 * 256))()
L.check(L.lib().spair_dg_stamps(out), "stamps")
st = np.array(out[:], dtype=np.float64).reshape(2, 8, 256)
for wg in range(2):
    for w in (0, 3):
        t = st[wg, w, :128].reshape(32, 4)
        pre, post, end, lg = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
        print("wg%d wave%d compute: period" % (wg, w), int(np.diff(pre)[1:30].mean()), "| reads B + mma A", int((pre[1:] - end[:-1])[1:30].mean()),
              "| lgkm wait", int((lg - pre)[1:31].mean()), "| barrier wait", int((post - lg)[1:31].mean()), "| reads A + mma B", int((end - post)[1:31].mean()))
        print("     periods", np.round(np.diff(pre)[:20]).astype(int).tolist())
    for w in (4, 7):
        t = st[wg, w, :128].reshape(32, 4)
        b, i, wdone = t[:, 0], t[:, 1], t[:, 2]
        print("wg%d wave%d loader: period" % (wg, w), int(np.diff(b)[2:30].mean()), "| issue", int((i - b)[2:31].mean()), "| dma wait", int((wdone - i)[2:31].mean()),
              "| barrier wait", int((b[1:] - wdone[:-1])[2:30].mean()))
