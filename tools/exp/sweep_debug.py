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
def step(gs):
    opt.zero_grad(); loss, recon, zw, zp = m(x, gs); loss.backward(); opt.step(); return loss, zw, zp
for i in range(70):
    loss, zw, zp = step(2000 + i)
    if i % 10 == 0: print(i, float(loss), float(zw[:, 2:4].mean()) * 128, float(zp.mean()), [float(zw[:, c].mean()) for c in range(4)])
snap = (m.flat_parameters().clone(), {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state_dict().items()})
for rep in range(3):
    m.flat_parameters().copy_(snap[0]); opt.load_state_dict(snap[1])
    for i in range(12):
        loss, zw, zp = step(2000 + i)
        if i in (0, 11): print("rep", rep, i, float(loss), float(zw[:, 2:4].mean()) * 128, float(zp.mean()))
