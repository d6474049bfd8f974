#!/usr/bin/env python3
"""Generate golden vectors by importing the reference (/root/reference) in THIS container.

Runs only in the build container; nothing under tests/, bench.py or smoke() imports the
reference at run time -- they read the .npz files this script wrote.  The reference is
never modified: a stub ``tensorboardX`` is injected (the reference imports it only for a
type annotation, models.py:11), config constants are mutated in-process before
``spair.models`` is imported, and the 7 per-cell random draws are
*injected* (numpy-generated noise returned in the reference's draw order) by wrapping
``torch.distributions.normal._standard_normal`` and ``torch.rand``.

Usage:  python tests/golden/make_golden.py            # all cases (one subprocess each)
        python tests/golden/make_golden.py --case c1_b16_step1
        python tests/golden/make_golden.py --units    # stn / decay / init-hash fixtures
"""
import argparse
import contextlib
import hashlib
import io
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_inputs as gi  # noqa: E402

REF = "/root/reference"
SMALL = 4096  # tensors up to this many elements get their full gradient stored


def _import_reference(I, strides, B, G, lookback=1, in_chan=1):
    import matplotlib
    matplotlib.use("Agg")
    tb = types.ModuleType("tensorboardX")

    class SummaryWriter:  # no-op writer
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, n):
            return lambda *a, **k: None

    tb.SummaryWriter = SummaryWriter
    sys.modules["tensorboardX"] = tb
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from spair import config as cfg
    cfg.INPUT_IMAGE_SHAPE[:] = [in_chan, I, I]
    cfg.BATCH_SIZE = B
    cfg.N_LOOKBACK = lookback
    for layer, s in zip(cfg.DEFAULT_BACKBONE_TOPOLOGY, strides):
        layer["stride"] = s
    from spair import models, modules, debug_tools
    debug_tools.GRID_SIZE = G
    return cfg, models, modules, SummaryWriter


def run_case(name):
    import torch
    case = gi.all_cases()[name]
    in_chan = case.get("in_chan", 1)
    I, strides, B, step = case["I"], case["strides"], case["B"], case["step"]
    G = gi.grid_side(I, strides)
    cfg, models, modules, SummaryWriter = _import_reference(I, strides, B, G, case.get("lookback", 1), in_chan)
    torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.SPAIR(cfg.INPUT_IMAGE_SHAPE, SummaryWriter(), torch.device("cpu"))
    assert tuple(m.feature_space_dim) == (100, G, G), m.feature_space_dim
    w = gi.make_weights(case["wseed"], case["wscale"], in_chan=in_chan, lookback=case.get("lookback", 1))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)

    x = gi.make_image(100 + case["wseed"], B, I, case["max_objects"], in_chan=in_chan)
    noise = gi.make_noise(200 + case["wseed"], B, G)

    # ---- inject the noise in the reference's draw order (models.py:333-336,84,95,402-403)
    import torch.distributions.normal as tdn
    state = dict(cell=0, k=0)

    def cell_hw():
        return divmod(state["cell"], G)

    def fake_standard_normal(shape, dtype, device):
        h, w_ = cell_hw()
        k = state["k"]
        if k < 4:
            assert tuple(shape) == (B, 1), shape
            out = noise["eps_box"][:, k:k + 1, h, w_]
        elif k == 4:
            assert tuple(shape) == (B, gi.N_ATTR), shape
            out = noise["eps_attr"][:, :, h, w_]
        elif k == 5:
            assert tuple(shape) == (B, 1), shape
            out = noise["eps_depth"][:, :, h, w_]
        else:
            raise AssertionError("unexpected normal draw")
        state["k"] += 1
        return torch.from_numpy(np.ascontiguousarray(out))

    real_rand = torch.rand

    def fake_rand(*shape, **kw):
        shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        assert state["k"] == 6 and shp == (B, 1), (state, shp)
        h, w_ = cell_hw()
        state["k"] = 0
        state["cell"] += 1
        return torch.from_numpy(np.ascontiguousarray(noise["u_pres"][:, :, h, w_]))

    captured = {}
    real_render, real_build_loss = m._render, m._build_loss

    def render_spy(z_attr, z_where, z_depth, z_pres, xin):
        captured["z_attr"], captured["z_depth"] = z_attr, z_depth
        return real_render(z_attr, z_where, z_depth, z_pres, xin)

    def loss_spy(xin, recon, kl):
        captured["kl"] = kl
        captured["recon_loss"] = torch.nn.functional.binary_cross_entropy(recon, xin, reduction="sum")
        return real_build_loss(xin, recon, kl)

    m._render, m._build_loss = render_spy, loss_spy
    tdn._standard_normal = fake_standard_normal
    torch.rand = fake_rand
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            xt = torch.from_numpy(x)
            loss, recon, z_where, z_pres = m(xt, step)
            loss.backward(retain_graph=True)
    finally:
        torch.rand = real_rand
    assert state["cell"] == G * G

    out = dict(x=x, global_step=np.int64(step), **noise)
    out["loss"] = np.float32(loss.item())
    out["recon_loss"] = np.float32(captured["recon_loss"].item())
    for nm, z_kl in captured["kl"].items():
        out["kl_" + nm] = np.float32(torch.mean(torch.sum(z_kl, dim=[1, 2, 3])).item())
    out["training_wheel"] = np.float32(float(m.training_wheel))
    out["recon_x"] = recon.detach().numpy()
    out["z_where"] = z_where.detach().numpy()
    out["z_pres"] = z_pres.detach().numpy()
    out["z_depth"] = captured["z_depth"].detach().numpy()
    out["z_attr"] = captured["z_attr"].detach().numpy()
    for dn, dp in m.dist_param.items():
        out[f"mean_{dn}"] = dp["mean"].detach().numpy()
        out[f"sigma_{dn}"] = dp["sigma"].detach().numpy()
    # gradients: per-tensor L2 norm for all, full tensor when small, fixed sample otherwise
    srng = np.random.default_rng(999)
    for k, p in m.named_parameters():
        if p.grad is None:
            out[f"gradnone_{k}"] = np.int8(1)
            continue
        g = p.grad.detach().numpy()
        out[f"gradnorm_{k}"] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        if g.size <= SMALL:
            out[f"grad_{k}"] = g
        else:
            idx = srng.choice(g.size, 256, replace=False)
            out[f"gradidx_{k}"] = idx.astype(np.int64)
            out[f"gradsample_{k}"] = g.reshape(-1)[idx]
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={out['loss']:.4f} recon={out['recon_loss']:.4f} wheel={out['training_wheel']}"
          f" -> {os.path.getsize(path) / 1e6:.2f} MB")


def run_units():
    """Unit fixtures: stn forward/inverse, exponential_decay, init replay hashes."""
    import torch
    I = 64
    cfg, models, modules, SummaryWriter = _import_reference(I, (2, 2, 2, 1, 1, 1), 4, 8)
    rng = np.random.default_rng(5)
    out = {}
    # --- stn forward (border) and inverse (zeros), modules.py:216-273
    n = 6
    img = rng.uniform(0, 1, (n, 1, I, I)).astype(np.float32)
    zw = np.stack([rng.uniform(-0.1, 1.1, n), rng.uniform(-0.1, 1.1, n),
                   rng.uniform(0.05, 0.9, n), rng.uniform(0.05, 0.9, n)], -1).astype(np.float32)
    it = torch.from_numpy(img)
    zt = torch.from_numpy(zw).requires_grad_(True)
    g = modules.stn(it, zt, [28, 28], torch.device("cpu"))
    gw = rng.standard_normal(tuple(g.shape)).astype(np.float32)
    (g * torch.from_numpy(gw)).sum().backward()
    out.update(stn_img=img, stn_zw=zw, stn_glimpse=g.detach().numpy(), stn_gw=gw,
               stn_dzw=zt.grad.numpy().copy())
    spr = rng.uniform(0, 1, (n, 3, 28, 28)).astype(np.float32)
    st = torch.from_numpy(spr).requires_grad_(True)
    zt2 = torch.from_numpy(zw).requires_grad_(True)
    inv = modules.stn(st, zt2, [I, I], torch.device("cpu"), inverse=True)
    gi_ = rng.standard_normal(tuple(inv.shape)).astype(np.float32)
    (inv * torch.from_numpy(gi_)).sum().backward()
    out.update(inv_sprite=spr, inv_out=inv.detach().numpy(), inv_gout=gi_,
               inv_dsprite=st.grad.numpy().copy(), inv_dzw=zt2.grad.numpy().copy())
    # --- exponential_decay, modules.py:191-213 with config.py:58-69
    steps = np.array([0, 1, 999, 1000, 1001, 2000, 4000, 6000, 7000, 7001, 8000, 10000], np.int64)
    dev = torch.device("cpu")
    out["decay_steps"] = steps
    out["decay_wheel"] = np.array([float(modules.exponential_decay(int(s), dev, **cfg.LATENT_VAR_TRAINING_WHEEL_PARAM)) for s in steps], np.float32)
    out["decay_count_log"] = np.array([float(modules.exponential_decay(int(s), dev, **cfg.OBJ_PRES_COUNT_LOG_PRIOR)) for s in steps], np.float32)
    np.savez_compressed(os.path.join(HERE, "units.npz"), **out)
    print("units.npz written")


def run_init_hashes():
    """sha256 of every tensor of the reference's own init under torch.manual_seed(3)
    (train.py:39), reference defaults (128x128, strides 3,2,2)."""
    import torch
    cfg, models, modules, SummaryWriter = _import_reference(128, (3, 2, 2, 1, 1, 1), 32, 11)
    torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.SPAIR(cfg.INPUT_IMAGE_SHAPE, SummaryWriter(), torch.device("cpu"))
    out = {}
    for k, v in m.state_dict().items():
        out["sha_" + k] = np.frombuffer(hashlib.sha256(v.numpy().tobytes()).digest(), np.uint8)
        out["head_" + k] = v.reshape(-1)[:4].numpy().copy()
    out["feature_space_dim"] = np.array(tuple(m.feature_space_dim), np.int64)
    out["pixels_per_cell"] = np.array(m.pixels_per_cell, np.int64)
    np.savez_compressed(os.path.join(HERE, "init_seed3.npz"), **out)
    print("init_seed3.npz written")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case")
    ap.add_argument("--units", action="store_true")
    ap.add_argument("--init", action="store_true")
    a = ap.parse_args()
    if a.case:
        run_case(a.case)
    elif a.units:
        run_units()
    elif a.init:
        run_init_hashes()
    else:
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        for nm in gi.all_cases():
            subprocess.check_call([sys.executable, __file__, "--case", nm], env=env)
        subprocess.check_call([sys.executable, __file__, "--units"], env=env)
        subprocess.check_call([sys.executable, __file__, "--init"], env=env)
