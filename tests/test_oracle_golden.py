"""Pin the CPU oracle (oracle/spair_oracle.py) against vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from helpers import KL_NAMES, case_noise, case_weights, load_case, oracle_cfg
from oracle import spair_oracle as orc

# c4 (32x32 grid, 1024 sequential cells) is the slow one: ~1 min
# (the N_LOOKBACK = 2 / 3 fixtures pin the generalised context gather, the rgb_* ones the C = 3 channel plumbing)
CASES = list(gi.CASES) + list(gi.LOOKBACK_CASES) + list(gi.RGB_CASES)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("name", CASES)
def test_forward_and_grads_match_reference(name):
    z, case = load_case(name)
    cfg = oracle_cfg(case)
    p = case_weights(case, requires_grad=True)
    out = orc.forward(p, torch.from_numpy(z["x"]), int(z["global_step"]), case_noise(z), cfg)
    assert float(out["wheel"]) == float(z["training_wheel"])
    # fp32, same op order up to reductions: tight tolerances
    assert abs(out["loss"].item() - float(z["loss"])) <= 2e-6 * abs(float(z["loss"]))
    assert abs(out["terms"]["recon"].item() - float(z["recon_loss"])) <= 2e-6 * float(z["recon_loss"])
    for n in KL_NAMES:
        ref = float(z["kl_" + n])
        assert abs(out["terms"]["kl_" + n].item() - ref) <= 1e-5 * abs(ref) + 1e-5, n
    for k in ("recon_x", "z_where", "z_pres", "z_depth", "z_attr"):
        assert rel(out[k].detach().numpy(), z[k]) < 2e-5, k
    for n, (mu, sg) in out["dist"].items():
        assert rel(mu.detach().numpy(), z["mean_" + n]) < 2e-5, n
        assert rel(sg.detach().numpy(), z["sigma_" + n]) < 2e-5, n
    out["loss"].backward()
    for k, t in p.items():
        if k.startswith("attn."):
            assert ("gradnone_" + k) in z  # dead Self_Attn: never receives a gradient
            continue
        g = t.grad.numpy()
        gn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        ref_n = float(z["gradnorm_" + k])
        assert abs(gn - ref_n) <= 2e-4 * ref_n + 1e-7, (k, gn, ref_n)
        if ("grad_" + k) in z:
            assert np.abs(g - z["grad_" + k]).max() <= 2e-4 * np.abs(z["grad_" + k]).max() + 1e-7, k
        else:
            smp = g.reshape(-1)[z["gradidx_" + k]]
            assert np.abs(smp - z["gradsample_" + k]).max() <= 2e-4 * ref_n / np.sqrt(g.size) * 30 + 1e-7, k


def test_closed_form_inverse_close_to_lu():
    """The HIP renderer uses the closed-form inverse affine; the reference a batched 3x3
    inverse (modules.py:258-261).  They agree to ~1e-5 on the image."""
    z, case = load_case("c1_b8_step1001")
    p = case_weights(case)
    x = torch.from_numpy(z["x"])
    outs = []
    for mode in ("lu", "closed"):
        cfg = oracle_cfg(case, inverse_mode=mode)
        with torch.no_grad():
            outs.append(orc.forward(p, x, int(z["global_step"]), case_noise(z), cfg))
    assert (outs[0]["recon_x"] - outs[1]["recon_x"]).abs().max() < 1e-4
    assert abs(outs[0]["loss"].item() - outs[1]["loss"].item()) < 1e-5 * outs[0]["loss"].item()


def test_units_stn_and_decay(golden_dir):
    import os
    u = np.load(os.path.join(golden_dir, "units.npz"))
    img, zw = torch.from_numpy(u["stn_img"]), torch.from_numpy(u["stn_zw"]).requires_grad_(True)
    for fast in (False, True):
        zw.grad = None
        g = orc.stn(img, zw, (28, 28), inverse=False, fast=fast)
        assert np.abs(g.detach().numpy() - u["stn_glimpse"]).max() < 1e-5
        (g * torch.from_numpy(u["stn_gw"])).sum().backward()
        assert np.abs(zw.grad.numpy() - u["stn_dzw"]).max() <= 2e-4 * np.abs(u["stn_dzw"]).max()
    spr = torch.from_numpy(u["inv_sprite"]).requires_grad_(True)
    zw2 = torch.from_numpy(u["stn_zw"]).requires_grad_(True)
    inv = orc.stn(spr, zw2, (64, 64), inverse=True, inverse_mode="lu")
    assert np.abs(inv.detach().numpy() - u["inv_out"]).max() < 1e-5
    (inv * torch.from_numpy(u["inv_gout"])).sum().backward()
    assert np.abs(spr.grad.numpy() - u["inv_dsprite"]).max() <= 1e-4 * np.abs(u["inv_dsprite"]).max()
    assert np.abs(zw2.grad.numpy() - u["inv_dzw"]).max() <= 5e-4 * np.abs(u["inv_dzw"]).max()
    c = orc.OracleConfig()
    for s, wv, cv in zip(u["decay_steps"], u["decay_wheel"], u["decay_count_log"]):
        assert float(orc.exponential_decay(int(s), **c.wheel)) == float(wv)
        assert float(orc.exponential_decay(int(s), **c.count_prior)) == float(cv)


def test_backbone_geometry_known_answers():
    """Receptive-field printout recorded in the reference's notebook (test_notebook.ipynb
    cell 10): rf 31, cell 12, 11 cells, pre 9, post 14 for the default topology at 128."""
    pre, post, G, cell, rf = orc.backbone_geometry(128, (4, 4, 4, 1, 1, 1), (3, 2, 2, 1, 1, 1))
    assert (rf, cell, G, pre, post) == (31, 12, 11, 9, 14)
    assert rf + (G - 1) * cell == 151
    pre, post, G, cell, rf = orc.backbone_geometry(128, (4, 4, 4, 1, 1, 1), (2, 2, 2, 1, 1, 1))
    assert (pre, post, G, cell) == (7, 7, 16, 8)


def test_freeze_gradient_known_answer():
    """Notebook cell 3: g*y.detach() + (1-g)*y scales the gradient by (1-g)."""
    y = torch.tensor([2.0], requires_grad=True)
    out = orc.freeze(0.6, y * 100)
    (out * 0.001).sum().backward()
    assert abs(y.grad.item() - 0.04) < 1e-7


# ---- evaluation metrics: oracle vs the reference's own spair/metric.py ------------------------
@pytest.mark.parametrize("name", ["m_b4_g6", "m_b3_g16", "m_b2_g11"])
def test_metric_oracle_matches_reference(name, golden_dir):
    import os
    from oracle import metric_oracle as mo
    z = np.load(os.path.join(golden_dir, "metrics.npz"))
    B, G, I, K = (int(v) for v in z[name + "/dims"])
    zw, zp = torch.from_numpy(z[name + "/z_where"]), torch.from_numpy(z[name + "/z_pres"])
    bb, cnt = torch.from_numpy(z[name + "/bbox"]), torch.from_numpy(z[name + "/count"])
    keep = [t.clone() for t in (zw, zp, bb, cnt)]
    m = mo.mAP(zw, zp, bb, cnt, I)
    acc = mo.object_count_accuracy(zp, cnt)
    a, b = mo.corners(zw, bb, I)
    iou = mo.batch_jaccard(a, b)
    for t, k in zip((zw, zp, bb, cnt), keep):
        assert torch.equal(t, k)                      # unlike the reference, inputs are not mutated
    assert abs(float(m) - float(z[name + "/mAP"])) <= 1e-6 * max(1.0, abs(float(z[name + "/mAP"])))
    assert abs(float(acc) - float(z[name + "/count_accuracy"])) <= 1e-5 * max(1.0, abs(float(z[name + "/count_accuracy"])))
    assert np.allclose(iou.numpy(), z[name + "/iou"], rtol=1e-6, atol=1e-7, equal_nan=True)
