"""Shared helpers for the parity tests (CPU side): load golden vectors, build oracle inputs."""
import os

import numpy as np
import torch

import golden_inputs as gi
from oracle import spair_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return z, gi.all_cases()[name]


def oracle_cfg(case, **kw):
    return orc.OracleConfig(image_shape=(case.get("in_chan", 1), case["I"], case["I"]), conv_strides=tuple(case["strides"]), n_lookback=case.get("lookback", 1), **kw)


def case_weights(case, requires_grad=False):
    w = gi.make_weights(case["wseed"], case["wscale"], in_chan=case.get("in_chan", 1), lookback=case.get("lookback", 1))
    return {k: torch.from_numpy(v).clone().requires_grad_(requires_grad and not k.startswith("attn."))
            for k, v in w.items()}


def case_noise(z):
    return {k: torch.from_numpy(z[k]) for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}


KL_NAMES = ["cy_logit", "cx_logit", "height_logit", "width_logit", "attr", "depth_logit", "pres_dist"]


def assert_adam_updates_close(pa, pb, lr, tight=2e-6, outlier_frac=1e-5):
    """Parameters after Adam steps of two runs whose gradients agree to rounding (fp32 atomics in the bias / edge sums): Adam's update
    lr * m / (sqrt(v) + eps) is sign-like, so a gradient that is itself rounding noise (|g| <~ eps) may move its parameter by up to
    +-lr differently in the two runs.  All but a few elements in a million must agree to `tight`, none may differ by more than 2.2 lr."""
    d = np.abs(np.asarray(pa, np.float64) - np.asarray(pb, np.float64))
    assert d.max() <= 2.2 * lr, d.max()
    assert (d > tight).mean() <= outlier_frac, ((d > tight).sum(), d.size)
