"""Step time of the fp32 per-wavefront step with the MLP and with the convolutional object encoder / decoder (row f4)."""
import sys, time, torch
sys.path.insert(0, ".")
from spair_pytorch_amd import config as cfg
from spair_pytorch_amd.models import SPAIR
from spair_pytorch_amd.optim import FusedAdam
from spair_pytorch_amd.data import scattered_digits
I, B = 48, int(sys.argv[1]) if len(sys.argv) > 1 else 256
DT = sys.argv[2] if len(sys.argv) > 2 else "f32"
from spair_pytorch_amd import models as _m
if DT == "bf16": _m.STEP_FLAGS = 1          # the MLP pair on the same per-wavefront launches (the fused chain is the benchmarked bf16 path)
cfg.set_grid(I, (2, 2, 2, 1, 1, 1))
x = torch.from_numpy(scattered_digits(4, B, I, 4)[0]).cuda()
for kind in ("mlp", "conv"):
    torch.manual_seed(3)
    m = SPAIR([1, I, I], None, torch.device("cuda"), compute_dtype=DT, object_encoder=kind).to("cuda")
    opt = FusedAdam(m, lr=1e-4)
    def step(i):
        m.zero_grad(); l = m(x, i)[0]; l.backward(); opt.step(); return l
    for i in range(3): step(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(10): l = step(3 + i)
    torch.cuda.synchronize()
    print(DT + " per-wavefront step, object_encoder=%s: B %d  %dx%d  %.2f ms/step  loss %.1f" % (kind, B, I, I, (time.perf_counter() - t) * 100, l.item()))
