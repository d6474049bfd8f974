"""Shared helpers for the parity tests (CPU side): load golden vectors, build oracle inputs."""
import os

import numpy as np
import torch

import golden_inputs as gi
from oracle import spair_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    case = gi.CASES[name]
    return z, case


def oracle_cfg(case, **kw):
    return orc.OracleConfig(image_shape=(1, case["I"], case["I"]), conv_strides=tuple(case["strides"]), **kw)


def case_weights(case, requires_grad=False):
    w = gi.make_weights(case["wseed"], case["wscale"])
    return {k: torch.from_numpy(v).clone().requires_grad_(requires_grad and not k.startswith("attn."))
            for k, v in w.items()}


def case_noise(z):
    return {k: torch.from_numpy(z[k]) for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}


KL_NAMES = ["cy_logit", "cx_logit", "height_logit", "width_logit", "attr", "depth_logit", "pres_dist"]
