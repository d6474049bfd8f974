"""Dump the bf16 step's outputs / gradients of one golden case to an .npz (A/B between two builds via SPAIR_HIP_LIB)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import golden_inputs as gi
from helpers import load_case
from spair_pytorch_amd import config as cfg
from spair_pytorch_amd.models import SPAIR
name, tag = sys.argv[1], sys.argv[2]
z, case = load_case(name)
cfg.set_grid(case["I"], case["strides"])
m = SPAIR([1, case["I"], case["I"]], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
m.load_state_dict({k: torch.from_numpy(v) for k, v in gi.make_weights(case["wseed"], case["wscale"]).items()})
x = torch.from_numpy(z["x"]).cuda()
noise = {k: torch.from_numpy(z[k]).cuda() for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}
m.zero_grad()
loss, recon, z_where, z_pres = m(x, int(z["global_step"]), noise=noise)
out = dict(loss=loss.item(), recon=recon.cpu().numpy(), z_where=z_where.cpu().numpy(), z_pres=z_pres.cpu().numpy(),
           z_attr=m.export_map(0).cpu().numpy(), z_depth=m.export_map(1).cpu().numpy())
for i in range(2, 15):
    out["map%d" % i] = m.export_map(i).cpu().numpy()
loss.backward()
for k, p in m.named_parameters():
    if p.grad is not None:
        out["g_" + k] = p.grad.cpu().numpy()
np.savez(os.path.join(ROOT, "gpurun_out", "dump_%s_%s.npz" % (name, tag)), **out)
