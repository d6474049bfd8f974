#!/usr/bin/env python3
"""The stem conv alone at BASELINE configs[1] (B=256, 128 -> 142 padded -> 70): bf16 (matrix cores) (the SPAIR_CONV0_VALU switch that forced the FMA kernel was removed in round 3)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spair_pytorch_amd import _lib as L
lib = L.lib()
B, I, pre, post = 256, 128, 7, 7
Hin = I + pre + post; Hout = (Hin - 4) // 2 + 1
x = torch.rand(B, I, I, device="cuda"); w = torch.randn(128, 16, device="cuda") * 0.3; b = torch.randn(128, device="cuda") * 0.1
o = torch.zeros(B, Hout, Hout, 128, device="cuda", dtype=torch.bfloat16)
f = lambda: L.check(lib.spair_stem_conv_fwd(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(o), B, I, pre, Hin, Hout, 128, 2, 1, L.stream()), "stem")
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
print("stem conv: %.1f us, output %.0f MB -> %.2f TB/s written" % (t * 1e3, o.numel() * 2 / 1e6, o.numel() * 2 / t / 1e9))
