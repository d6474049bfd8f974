"""Why does the bench's density axis land at 0.66-0.67 for a target of 0.70 (and exactly on 0.05 / 0.30)?  Replays the calibration for one point and
prints the mean presence of calibration-style forwards and of whole steps at the calibrated bias."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spair_pytorch_amd import config as cfg, models
from spair_pytorch_amd.data import scattered_digits
from spair_pytorch_amd.optim import FusedAdam
cfg.set_grid(128, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)
m = models.SPAIR([1, 128, 128], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
opt = FusedAdam(m, lr=1e-4)
x = torch.from_numpy(scattered_digits(1234, 256, 128, 11)[0]).cuda()
torch.manual_seed(7)
gs = 2000
def step():
    opt.zero_grad(); loss, recon, zw, zp = m(x, gs); loss.backward(); opt.step(); return float(zp.mean()), float(loss)
for i in range(40): step()
opt.lr = 0.0
sd = m.state_dict()
pres_b = sd["obj_network.out.bias"]
def mean_pres():
    with torch.enable_grad():
        return float(m(x, gs)[3].mean().item())
for target in (0.3, 0.7, 0.9):
    base = pres_b.clone()
    lo, hi = -12.0, 12.0
    with torch.no_grad():
        for _ in range(14):
            mid = 0.5 * (lo + hi)
            pres_b.copy_(base + mid)
            if mean_pres() < target: lo = mid
            else: hi = mid
        pres_b.copy_(base + 0.5 * (lo + hi))
    print("target %.2f: bias shift %.4f" % (target, 0.5 * (lo + hi)))
    print("   calibration-style forwards:", ["%.4f" % mean_pres() for _ in range(4)])
    print("   whole steps (lr 0)        :", ["%.4f (loss %.1f)" % step() for _ in range(6)])
    print("   forwards again            :", ["%.4f" % mean_pres() for _ in range(3)])
    print("   status", m.step_status(), "bias now", float(pres_b[0]), "params finite", bool(torch.isfinite(m.flat_parameters()).all()))
    with torch.no_grad(): pres_b.copy_(base)
