"""Timing-only ablations of the renderer inside the real step (results wrong by construction): SPAIR_HIP_LIB=build/libspair_rf_<x>.so"""
import os, sys, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spair_pytorch_amd import _lib as L, config as cfg, models
from spair_pytorch_amd.data import scattered_digits
from spair_pytorch_amd.optim import FusedAdam
I = int(os.environ.get("I", "128")); B = int(os.environ.get("B", "256"))
cfg.set_grid(I, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)
m = models.SPAIR([1, I, I], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
opt = FusedAdam(m, lr=1e-4)
x = torch.from_numpy(scattered_digits(1234, B, I, 11)[0]).cuda()
torch.manual_seed(7)
lib = L.lib()
FWD_ONLY = os.environ.get("FWD_ONLY", "1") == "1"      # forward passes from the seed-3 initial weights: every build sees the same boxes
def step(i):
    if FWD_ONLY:
        with torch.no_grad():
            m(x, 2000 + i)
    else:
        opt.zero_grad(); loss = m(x, 2000 + i)[0]; loss.backward()      # no optimizer step: every build sees the seed-3 boxes
for i in range(10): step(i)
lib.spair_prof_select(ctypes.c_ulonglong(0xFFFFFFFFFFFFFFFF)); lib.spair_prof_enable(1); lib.spair_prof_enable(0)
torch.cuda.synchronize()
for i in range(10):
    lib.spair_prof_enable(2); step(10 + i)
torch.cuda.synchronize()
ms = (ctypes.c_float * 18)(); cnt = (ctypes.c_int * 18)()
lib.spair_prof_read(ms, cnt, 18)
print(os.environ.get("SPAIR_HIP_LIB", "default"), "render_fwd %.4f ms  render_bwd %.4f ms" % (ms[5] / max(cnt[5], 1), ms[7] / max(cnt[7], 1)))
