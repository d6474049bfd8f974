#!/usr/bin/env python3
"""Pricing of "two samples per workgroup" for the chain kernels WITHOUT building it: the per-wavefront time of k_chain_fwd / k_chain_bwd
(sample 0's s_memtime stamps) against the number of cells on the wavefront (1..8 at G = 16).  T(n) = a + b n fitted per kernel and per stage:
the fixed part `a` (weight streaming, barriers, the stage chain) is what a second sample in the workgroup would share, `b n` is what it would
pay again.  T1 = sum_t (a + b n_t) is today's kernel, T2 = sum_t (a + 2 b n_t) the linear estimate of the two-sample form (a LOWER bound: it
assumes the 9th..16th cell of a wavefront costs what the 2nd..8th does)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spair_pytorch_amd import _lib as L, config as cfg, models
from spair_pytorch_amd.data import scattered_digits
I, B, G = 128, int(os.environ.get("STAMP_BATCH", "256")), 16
cfg.set_grid(I, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)
m = models.SPAIR([1, I, I], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
x = torch.from_numpy(scattered_digits(1, B, I, 11)[0]).cuda()
models.STEP_FLAGS = 2
acc_f, acc_b = [], []
for it in range(6):
    m.zero_grad()
    loss = m(x, 2000)[0]
    loss.backward()
    torch.cuda.synchronize()
    e = m._last["engine"]
    T = L.lib().spair_chain_stamp_wavefronts(ctypes.byref(e["dims"]))
    ns, gl, nb = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    L.check(L.lib().spair_chain_stamp_layout(ctypes.byref(ns), ctypes.byref(gl), ctypes.byref(nb)), "layout")
    out = torch.zeros(4096, dtype=torch.int64, device="cuda")
    L.check(L.lib().spair_chain_stamps(ctypes.byref(e["dims"]), L.ptr(e["workspace"]), L.ptr(out), 4096, L.stream()), "stamps")
    o = out.cpu().numpy().astype(np.float64)
    if it >= 2:
        acc_f.append(o[:T * ns.value].reshape(T, ns.value))
        acc_b.append(o[2048:2048 + T * nb.value].reshape(T, nb.value))
tick_us = 1.0 / 2.1e3
ncell = np.array([sum(1 for h in range(G) for w in range(G) if 2 * h + w == t) for t in range(3 * G - 2)])
for name, acc, rev in (("forward", acc_f, False), ("backward", acc_b, True)):
    st = np.mean(acc, axis=0)
    # wavefront-to-wavefront time (includes the loop overhead): stamp 0 of consecutive wavefronts
    per = np.abs(np.diff(st[:, 0])) * tick_us
    n = ncell[::-1][:-1] if rev else ncell[:-1]
    print("%s: wavefront time by cell count (us)" % name)
    for k in range(1, 9):
        sel = per[n == k]
        if len(sel):
            print("  n=%d  wavefronts=%2d  mean %.2f  min %.2f  max %.2f" % (k, len(sel), sel.mean(), sel.min(), sel.max()))
    A = np.stack([np.ones_like(n, dtype=np.float64), n.astype(np.float64)], 1)
    (a, b), *_ = np.linalg.lstsq(A, per, rcond=None)
    T1 = (a + b * n).sum()
    T2 = (a + 2 * b * n).sum()
    print("  fit T(n) = %.2f + %.3f n us;  sum over %d wavefronts: T1 = %.1f us (measured %.1f), T2 (two samples) >= %.1f us = %.2f x T1"
          % (a, b, len(n), T1, per.sum(), T2, T2 / T1))
    d = np.diff(st, axis=1) * tick_us
    nn = ncell[::-1] if rev else ncell
    print("  per stage: fixed a / per-cell b (us)")
    for s in range(d.shape[1]):
        A2 = np.stack([np.ones(len(nn)), nn.astype(np.float64)], 1)
        (a2, b2), *_ = np.linalg.lstsq(A2, d[:, s], rcond=None)
        print("    stage %2d  a %.2f  b %.3f  (n=1: %.2f  n=8: %.2f)" % (s, a2, b2, d[nn == 1, s].mean(), d[nn == 8, s].mean()))
