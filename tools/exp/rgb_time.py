"""Step time of the colour-image variant (C = 3, generic-channel renderer, per-wavefront launches) at the bench geometry.
usage (GPU box): python tools/exp/rgb_time.py [batch]"""
import sys, time, torch
sys.path.insert(0, ".")
from spair_pytorch_amd import config as cfg, models
from spair_pytorch_amd.optim import FusedAdam
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg.set_grid(128, (2, 2, 2, 1, 1, 1))
cfg.INPUT_IMAGE_SHAPE[0] = 3
for dt in ("bf16", "f32"):
    torch.manual_seed(3)
    m = models.SPAIR([3, 128, 128], None, torch.device("cuda"), compute_dtype=dt).to("cuda")
    opt = FusedAdam(m, lr=1e-4)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.rand(B, 3, 128, 128, device="cuda", generator=g) > 0.9).float() * torch.rand(B, 3, 128, 128, device="cuda", generator=g)
    def step():
        opt.zero_grad(); loss, *_ = m(x, 2000); loss.backward(); opt.step(); return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): l = step()
    torch.cuda.synchronize(); dt_ms = (time.perf_counter() - t0) / n * 1e3
    print("%s B=%d 128x128 RGB: %.2f ms/step, %.0f images/s, loss %.1f" % (dt, B, dt_ms, B / dt_ms * 1e3, float(l.detach())))
