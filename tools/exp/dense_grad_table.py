#!/usr/bin/env python3
"""Per-parameter gradient error of the HIP step against the CPU oracle in the dense-presence regime (tests/test_countkl_gpu.py::
test_dense_presence_step_vs_oracle's inputs): max |g - g_ref| / max |g_ref| and the cosine, per tensor, both dtypes, for several presence biases."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_inputs as gi
from oracle import spair_oracle as orc
from spair_pytorch_amd import config as cfg
from spair_pytorch_amd.models import SPAIR
from spair_pytorch_amd.data import scattered_digits
I, B, strides, gs = 128, 3, (2, 2, 2, 1, 1, 1), 2000
G = gi.grid_side(I, strides)
cfg.set_grid(I, strides)
for bias in [float(v) for v in (sys.argv[1:] or ["0", "1.3", "7"])]:
    w = gi.make_weights(61, 1.0)
    w["obj_network.out.bias"] = np.full_like(w["obj_network.out.bias"], bias)
    x = scattered_digits(62, B, I, 11)[0]
    noise = gi.make_noise(63, B, G)
    p = {k: torch.from_numpy(v).double().clone().requires_grad_(not k.startswith("attn.")) for k, v in w.items()}
    ocfg = orc.OracleConfig(image_shape=(1, I, I), conv_strides=strides, inverse_mode="closed")
    ref = orc.forward(p, torch.from_numpy(x).double(), gs, {k: torch.from_numpy(v).double() for k, v in noise.items()}, ocfg, fast=True)
    ref["loss"].backward()
    print("== presence bias %.1f: oracle(f64) loss %.4f mean z_pres %.4f" % (bias, ref["loss"].item(), ref["z_pres"].mean().item()))
    for dtype in ("f32", "bf16"):
        m = SPAIR([1, I, I], None, torch.device("cuda"), compute_dtype=dtype).to("cuda")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        m.zero_grad()
        loss, recon, zw, zp = m(torch.from_numpy(x).cuda(), gs, noise={k: torch.from_numpy(v).cuda() for k, v in noise.items()})
        loss.backward()
        print("  %s: loss rel %.2e  recon max abs %.2e  z_where %.2e  z_pres %.2e" % (
            dtype, abs(loss.item() - ref["loss"].item()) / abs(ref["loss"].item()), (recon.cpu().double() - ref["recon_x"].detach()).abs().max().item(),
            (zw.cpu().double() - ref["z_where"].detach()).abs().max().item(), (zp.cpu().double() - ref["z_pres"].detach()).abs().max().item()))
        rows = []
        for k, pt in m.named_parameters():
            if k.startswith("attn."):
                continue
            g, r = pt.grad.double().cpu().flatten(), p[k].grad.flatten()
            rows.append(((g - r).abs().max().item() / (r.abs().max().item() + 1e-30), float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30)), k))
        for e, c, k in sorted(rows, reverse=True)[:10 if dtype == "f32" else 6]:
            print("     %.3e  cos %.6f  %s" % (e, c, k))
