#!/usr/bin/env python3
"""Where the count-prior kernel's p_z differs most from the float64 recursion (the oracle's compute_kl on a float64 z_pres), per pattern kind
and global_step -- and how far the fp32 oracle itself is from float64 at the same cell."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as gi
from oracle import spair_oracle as orc
import test_countkl_gpu as T
from spair_pytorch_amd.data import scattered_digits
for I, strides, B in T.DENSE_GEOMS:
    G = gi.grid_side(I, strides); HW = G * G
    m = T._model(I, strides, "f32")
    x = torch.from_numpy(scattered_digits(5, B, I, 9)[0]).cuda()
    base = {k: torch.from_numpy(v) for k, v in gi.make_noise(4, B, G).items()}
    ocfg = orc.OracleConfig(image_shape=(1, I, I), conv_strides=strides)
    for kind in ("one_off", "iid", "runs"):
        on = T.dense_patterns(kind, B, HW)
        noise = dict(base); noise["u_pres"] = torch.from_numpy(np.where(on, T.ON, T.OFF).astype(np.float32).reshape(B, 1, G, G))
        for step in (1, 2000, 6000, 12000):
            with torch.no_grad():
                z = m(x, step, noise={k: v.cuda() for k, v in noise.items()})[3].cpu()
            pz = m.export_map(14).cpu().double().flatten(1)
            o32, o64 = [], []
            orc.compute_kl({}, z, step, ocfg, p_z_out=o32); orc.compute_kl({}, z.double(), step, ocfg, p_z_out=o64)
            r32, r64 = o32[0].double().flatten(1), o64[0].flatten(1)
            e32 = (pz - r32).abs().max().item()          # against the fp32 oracle = the reference's arithmetic (what the test bounds)
            e = (pz - r64).abs()
            b, i = divmod(int(e.argmax()), HW)
            print("G=%2d %-7s step %5d: kernel-oracle32 max %.2e | kernel-f64 max %.2e at sample %3d cell %4d (p_z %.6f, %d on before, density %.2f); oracle32-f64 there %.2e, max %.2e; rel err max %.2e"
                  % (G, kind, step, e32, e.max().item(), b, i, r64[b, i].item(), int(on[b, :i].sum()), on[b].mean(), abs(r32[b, i] - r64[b, i]).item(),
                     (r32 - r64).abs().max().item(), (e / r64.clamp(min=1e-12)).max().item()), flush=True)
