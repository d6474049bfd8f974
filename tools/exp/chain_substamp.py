import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from spair_pytorch_amd import _lib as L, config as cfg, models
from spair_pytorch_amd.data import scattered_digits
cfg.set_grid(128, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)
m = models.SPAIR([1, 128, 128], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
x = torch.from_numpy(scattered_digits(1, 256, 128, 11)[0]).cuda()
models.STEP_FLAGS = 2
for _ in range(2):
    m.zero_grad(); loss = m(x, 2000)[0]; loss.backward()
torch.cuda.synchronize()
e = m._last["engine"]; T = 46; NS = 22
out = torch.zeros(4096, dtype=torch.int64, device="cuda")
L.check(L.lib().spair_chain_stamps(ctypes.byref(e["dims"]), L.ptr(e["workspace"]), L.ptr(out), T * NS, L.stream()), "stamps")
st = out.cpu().numpy()[:T * NS].reshape(T, NS).astype(np.float64)
d = np.diff(st, axis=1)
names = ["rows", "S0 ctx", "BOX0", "BOX1:gemm", "BOX1:store", "BOX1:barrier", "BOXH", "box", "glimpse", "ENC0", "ENC1", "ENC2", "attr", "Z0", "Z1", "ZH", "depth", "OBJ0", "OBJ1", "OBJ2", "pres"]
for i in range(21):
    print("%-14s %7.0f cycles" % (names[i], d[:, i].mean()))
