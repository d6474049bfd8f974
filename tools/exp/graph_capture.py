#!/usr/bin/env python3
"""Capture one whole training step (forward + backward + fused Adam) into a HIP graph and replay it (GPU box; experiment)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spair_pytorch_amd import config as cfg
from spair_pytorch_amd.models import SPAIR
from spair_pytorch_amd.optim import FusedAdam
from spair_pytorch_amd import _lib as L

B, I, C = int(os.environ.get("GB", 256)), 128, int(os.environ.get("GC", 1))      # GC=3: the colour-image variant (per-wavefront launches)
cfg.set_grid(I, (2, 2, 2, 1, 1, 1))
cfg.INPUT_IMAGE_SHAPE[0] = C
torch.manual_seed(3)
dev = torch.device("cuda:0")
model = SPAIR([C, I, I], None, dev, compute_dtype="bf16").to(dev)
opt = FusedAdam(model, lr=1e-4)
x = (torch.rand(B, C, I, I, device=dev) > 0.9).float()
L.lib().spair_init()
def step():
    loss, recon, zw, zp = model(x, 2000)
    loss.backward()
    opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    lg = step()
torch.cuda.synchronize()
for _ in range(5): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("graph replay %.3f ms/step, eager %.3f ms/step, loss %.1f" % ((t1 - t0) * 20, (t2 - t1) * 20, float(lg)))
