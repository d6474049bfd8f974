#!/usr/bin/env python3
"""Diagnostic: per-stage time of the fused forward chain kernel (sample 0) from s_memtime stamps."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spair_pytorch_amd import _lib as L, config as cfg, models
from spair_pytorch_amd.data import scattered_digits
IMG_SIDE = int(os.environ.get("STAMP_IMAGE", "128"))           # STAMP_IMAGE=256 STAMP_BATCH=64: configs[3] (the stamping workgroup is sample 0's top band)
BATCH = int(os.environ.get("STAMP_BATCH", "256"))
cfg.set_grid(IMG_SIDE, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)
m = models.SPAIR([1, IMG_SIDE, IMG_SIDE], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
x = torch.from_numpy(scattered_digits(1, BATCH, IMG_SIDE, 11)[0]).cuda()
models.STEP_FLAGS = 2
for _ in range(2):
    m.zero_grad()
    loss = m(x, 2000)[0]
    loss.backward()
torch.cuda.synchronize()
e = m._last["engine"]
T = L.lib().spair_chain_stamp_wavefronts(ctypes.byref(e["dims"]))
_ns, _gl, _nb = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
L.check(L.lib().spair_chain_stamp_layout(ctypes.byref(_ns), ctypes.byref(_gl), ctypes.byref(_nb)), "stamp_layout")
NS = _ns.value
out = torch.zeros(4096, dtype=torch.int64, device="cuda")
L.check(L.lib().spair_chain_stamps(ctypes.byref(e["dims"]), L.ptr(e["workspace"]), L.ptr(out), T * NS, L.stream()), "stamps")
st = out.cpu().numpy()[:T * NS].reshape(T, NS).astype(np.float64)
d = np.diff(st, axis=1)          # [T, NS-1] stage durations in s_memtime ticks (100 MHz => 10 ns)
names = ["rows", "S0 ctx", "BOX0", "BOX1", "BOXH+box", "glimpse", "ENC0", "ENC1", "ENC2", "attr", "Z0", "Z1", "ZH+depth", "OBJ0", "OBJ1+obj2", "pres"]
tick_ns = 1.0 / 2.1   # s_memtime counts shader cycles (~2.1 GHz under load): report in us assuming that clock
print("per-wavefront mean stage time (us), over %d wavefronts; total %.1f us/step" % (T, d[:, :NS - 1].sum(1).mean() * tick_ns / 1e3))
for i in range(NS - 1):
    print("%-8s %7.2f" % (names[i], d[:, i].mean() * tick_ns / 1e3))
print("step-to-step (incl. loop overhead): %.2f us" % (np.diff(st[:, 0]).mean() * tick_ns / 1e3))

NB = _nb.value
out2 = torch.zeros(4096, dtype=torch.int64, device="cuda")
L.check(L.lib().spair_chain_stamps(ctypes.byref(e["dims"]), L.ptr(e["workspace"]), L.ptr(out2), 4096, L.stream()), "stamps")
sb = out2.cpu().numpy()[2048:2048 + T * NB].reshape(T, NB).astype(np.float64)
db = np.diff(sb, axis=1)
bn = ["rows", "grec", "pres+dHo2", "OBJ1", "OBJ0", "depth", "ZH", "Z1", "Z0", "attr", "ENC2", "ENC1", "ENC0+stn", "box", "BOXH", "BOX1", "BOX0", "dfeat/edge"]
print("backward: total %.1f us/step" % (db.sum(1).mean() * tick_ns / 1e3))
for i in range(db.shape[1]):
    print("%-10s %7.2f" % (bn[i] if i < len(bn) else "?", db[:, i].mean() * tick_ns / 1e3))
print("step-to-step: %.2f us" % (np.abs(np.diff(sb[:, 0])).mean() * tick_ns / 1e3))
