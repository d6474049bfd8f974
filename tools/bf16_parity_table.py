#!/usr/bin/env python3
"""Per-tensor gradient agreement of the bf16 step with the reference's fixtures (tests/golden): |g|/|g_ref| and cosine for every parameter
of every golden case, loss / recon / z_where errors.  Written to stdout as a table (committed under profiles/ as the evidence behind the
thresholds of tests/test_engine_gpu.py::test_bf16_step_within_north_star_tolerance)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import golden_inputs as gi
from helpers import load_case


def run(name, dtype="bf16"):
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.models import SPAIR
    z, case = load_case(name)
    cfg.set_grid(case["I"], case["strides"])
    m = SPAIR([1, case["I"], case["I"]], None, torch.device("cuda"), compute_dtype=dtype).to("cuda")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gi.make_weights(case["wseed"], case["wscale"]).items()})
    x = torch.from_numpy(z["x"]).cuda()
    noise = {k: torch.from_numpy(z[k]).cuda() for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}
    m.zero_grad()
    loss, recon, z_where, z_pres = m(x, int(z["global_step"]), noise=noise)
    loss.backward()
    print("== %s (%s): loss rel err %.2e, max|recon err| %.3g, max|z_where err| %.3g, max|z_pres err| %.3g" % (
        name, dtype, abs(loss.item() - float(z["loss"])) / abs(float(z["loss"])), np.abs(recon.cpu().numpy() - z["recon_x"]).max(),
        np.abs(z_where.cpu().numpy() - z["z_where"]).max(), np.abs(z_pres.cpu().numpy() - z["z_pres"]).max()))
    worst_cos, worst_norm = 1.0, 0.0
    for k, p in m.named_parameters():
        if k.startswith("attn."):
            continue
        gn = float(p.grad.double().norm().item())
        ref_n = float(z["gradnorm_" + k])
        g = p.grad.detach().double().cpu().flatten().numpy()
        if "grad_" + k in z.files:
            ref, kind = z["grad_" + k].astype(np.float64).flatten(), "full"
        else:
            g, ref, kind = g[z["gradidx_" + k]], z["gradsample_" + k].astype(np.float64), "sample"
        cos = float(np.dot(g, ref) / (np.linalg.norm(g) * np.linalg.norm(ref) + 1e-30)) if np.linalg.norm(ref) > 1e-6 * max(1.0, ref_n) else float("nan")
        print("   %-42s |g|/|ref| %.4f  cos %.5f (%s)  |ref| %.3e" % (k, gn / max(ref_n, 1e-30), cos, kind, ref_n))
        if cos == cos:
            worst_cos = min(worst_cos, cos)
        worst_norm = max(worst_norm, abs(gn / max(ref_n, 1e-30) - 1.0))
    print("   worst cosine %.5f, worst norm deviation %.4f" % (worst_cos, worst_norm))


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(gi.CASES)):
        run(n)
