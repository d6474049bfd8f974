#!/usr/bin/env python3
"""Diagnostic: per-stage time of the fused forward chain kernel (sample 0) from s_memtime stamps."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spair_pytorch_amd import _lib as L, config as cfg, models
from spair_pytorch_amd.data import scattered_digits
cfg.set_grid(128, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)
m = models.SPAIR([1, 128, 128], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
x = torch.from_numpy(scattered_digits(1, 256, 128, 11)[0]).cuda()
models.STEP_FLAGS = 2
for _ in range(3):
    with torch.no_grad():
        m(x, 2000)
torch.cuda.synchronize()
e = m._last["engine"]
T = 3 * 16 - 2
NS = 21
out = torch.zeros(4096, dtype=torch.int64, device="cuda")
L.check(L.lib().spair_chain_stamps(ctypes.byref(e["dims"]), L.ptr(e["workspace"]), L.ptr(out), T * NS, L.stream()), "stamps")
st = out.cpu().numpy()[:T * NS].reshape(T, NS).astype(np.float64)
d = np.diff(st, axis=1)          # [T, NS-1] stage durations in s_memtime ticks (100 MHz => 10 ns)
names = ["rows", "S0 ctx", "BOX0", "BOX1", "BOXH", "box", "glimpse", "ENC0", "ENC1", "ENC2", "attr", "Z0", "Z1", "ZH", "depth", "OBJ0", "OBJ1", "OBJ2", "pres", "x"]
tick_ns = 10.0
print("per-wavefront mean stage time (us), over %d wavefronts; total %.1f us/step" % (T, d[:, :19].sum(1).mean() * tick_ns / 1e3))
for i in range(19):
    print("%-8s %7.2f" % (names[i], d[:, i].mean() * tick_ns / 1e3))
print("step-to-step (incl. loop overhead): %.2f us" % (np.diff(st[:, 0]).mean() * tick_ns / 1e3))
