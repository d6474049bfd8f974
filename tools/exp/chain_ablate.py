"""Timing-only ablations of k_chain_fwd (results are WRONG by construction): run with SPAIR_HIP_LIB=build/libspair_abl_<x>.so"""
import os, sys, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spair_pytorch_amd import _lib as L, config as cfg, models
from spair_pytorch_amd.data import scattered_digits
B = int(os.environ.get("B", "256"))
cfg.set_grid(128, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)
m = models.SPAIR([1, 128, 128], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
x = torch.from_numpy(scattered_digits(1, B, 128, 11)[0]).cuda()
lib = L.lib()
for _ in range(3):
    with torch.no_grad():
        m(x, 2000)
lib.spair_prof_select(ctypes.c_ulonglong(0xFFFFFFFFFFFFFFFF)); lib.spair_prof_enable(1); lib.spair_prof_enable(0)
torch.cuda.synchronize()
n = 10
for i in range(n):
    lib.spair_prof_enable(2)
    m.zero_grad()
    loss = m(x, 2000)[0]
    if os.environ.get("BWD", "0") == "1":
        loss.backward()
torch.cuda.synchronize()
ms = (ctypes.c_float * 18)(); cnt = (ctypes.c_int * 18)()
lib.spair_prof_read(ms, cnt, 18)
print(os.environ.get("SPAIR_HIP_LIB", "default"), "cells_fwd %.4f ms" % (ms[2] / max(cnt[2], 1)), "cells_bwd %.4f ms" % (ms[9] / max(cnt[9], 1)))
