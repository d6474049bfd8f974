"""End-to-end parity of the HIP training step (spair_forward / spair_backward through the Python
surface) with the golden vectors produced by the reference itself (tests/golden/*.npz).

fp32 MFMA mode must agree to fp32 round-off accumulated over the sequential chain; bf16 mode to
the north-star tolerance (ELBO within 1e-3 relative)."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from helpers import KL_NAMES, load_case

pytestmark = pytest.mark.gpu


def build_model(case, dtype):
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.models import SPAIR
    cfg.set_grid(case["I"], case["strides"])
    m = SPAIR([1, case["I"], case["I"]], None, torch.device("cuda"), compute_dtype=dtype).to("cuda")
    w = gi.make_weights(case["wseed"], case["wscale"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return m


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("name", list(gi.CASES))
def test_fp32_step_matches_reference(name):
    z, case = load_case(name)
    m = build_model(case, "f32")
    x = torch.from_numpy(z["x"]).cuda()
    noise = {k: torch.from_numpy(z[k]).cuda() for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}
    m.zero_grad()
    loss, recon, z_where, z_pres = m(x, int(z["global_step"]), noise=noise)
    t = m.loss_terms().cpu().numpy()
    assert abs(t[0] - float(z["loss"])) <= 2e-5 * abs(float(z["loss"]))
    assert abs(t[1] - float(z["recon_loss"])) <= 2e-5 * float(z["recon_loss"])
    for i, n in enumerate(KL_NAMES):
        ref = float(z["kl_" + n])
        assert abs(t[2 + i] - ref) <= 1e-4 * abs(ref) + 1e-4, (n, t[2 + i], ref)
    assert rel(z_where.cpu().numpy(), z["z_where"]) < 1e-4
    assert rel(z_pres.cpu().numpy(), z["z_pres"]) < 1e-4
    assert rel(recon.cpu().numpy(), z["recon_x"]) < 2e-4
    assert rel(m.export_map(0).cpu().numpy(), z["z_attr"]) < 1e-4
    assert rel(m.export_map(1).cpu().numpy(), z["z_depth"]) < 1e-4
    for i, n in enumerate(KL_NAMES[:6]):
        assert rel(m.dist_param[n]["mean"].cpu().numpy(), z["mean_" + n]) < 1e-4, n
        assert rel(m.dist_param[n]["sigma"].cpu().numpy(), z["sigma_" + n]) < 1e-4, n
    loss.backward(retain_graph=True)
    bad = []
    for k, p in m.named_parameters():
        if k.startswith("attn."):
            assert p.grad is None
            continue
        g = p.grad.cpu().numpy()
        gn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        ref_n = float(z["gradnorm_" + k])
        if abs(gn - ref_n) > 2e-3 * ref_n + 1e-6:
            bad.append((k, gn, ref_n))
            continue
        if ("grad_" + k) in z:
            if np.abs(g - z["grad_" + k]).max() > 2e-3 * np.abs(z["grad_" + k]).max() + 1e-6:
                bad.append((k, "elements"))
        else:
            smp = g.reshape(-1)[z["gradidx_" + k]]
            if np.abs(smp - z["gradsample_" + k]).max() > 2e-3 * np.abs(z["gradsample_" + k]).max() + 1e-6:
                bad.append((k, "samples"))
    assert not bad, bad


# Per-case bounds of the bf16 step against the reference's fixtures.  Evidence: profiles/r04_bf16_parity_table.txt (tools/bf16_parity_table.py,
# every tensor of every case; the step has no atomics, so the table repeats bit for bit).
# loss: BASELINE.json asks for 1e-3 relative; 3e-5 .. 1.6e-4 is observed.  recon: absolute, on a [0, 1] image.
# Gradients: |g| / |g_ref| - 1 and the cosine, per tensor.
# Round 4: the box network's forward runs as split-bf16 products (hi + lo operands, chain.hip) -- its outputs place the glimpse and the sprite on
# the pixel grid.  z_where now agrees to 1e-6 .. 3e-5 (2e-4 before), the reconstruction to 3e-4 .. 6e-3 (2e-3 .. 5e-2), and the gradient direction of
# every tensor of five fixtures is >= 0.995 (0.918 on the 11 x 11 grid before); the sharp-count-prior fixture with doubled weights, whose
# encoder also needs the precision (tools/exp/f32nets_table.py: 0.995 with box AND encoder on fp32 operands), is at 0.957 (0.937).
# The bounds leave room for the re-roll every later bf16 rounding gets from a last-bit change upstream (two builds that differ in one
# sigmoid's last bit land 0.9963 / 0.9946 on c1_b16_step1's encoder.dense0).
BF16_BOUNDS = {
    #                          loss    recon   z_where  norm    cos        observed (r04 table): recon / z_where / norm / cos
    "c1_b16_step1":            (2.0e-4, 0.002,  1.0e-4,  0.03,   0.990),    # 4.4e-4 / 9.6e-6 / -     / 0.9950  (training wheel: encoder + decoder only)
    "c1_b8_step1001":          (2.0e-4, 0.002,  1.0e-4,  0.02,   0.990),    # 4.5e-4 / 6.8e-6 / 0.004 / 0.9979
    "c1_b8_step7001":          (2.0e-4, 0.020,  3.0e-3,  0.08,   0.940),    # 6.1e-3 / 8.6e-4 / 0.049 / 0.9569  (sharp count prior, weights x2: the hardest fixture)
    "c2_b2_step1001":          (2.0e-4, 0.002,  5.0e-5,  0.02,   0.990),    # 3.6e-4 / 3.2e-6 / 0.008 / 0.9978  (the bench geometry: 128x128, 16x16 grid)
    "ref_default_b2_step1001": (2.5e-4, 0.012,  2.0e-4,  0.04,   0.985),    # 3.8e-3 / 2.6e-5 / 0.018 / 0.9957  (the reference's default 11 x 11 grid)
    "c4_b1_step1001":          (2.0e-4, 0.002,  5.0e-5,  0.02,   0.990),    # 3.1e-4 / 1.2e-6 / 0.009 / 0.9985  (256x256, 32x32 grid)
}


@pytest.mark.parametrize("name", list(gi.CASES))
def test_bf16_step_within_north_star_tolerance(name):
    """BASELINE.json: ELBO within 1e-3 relative of the CPU reference on the same batch and noise (observed 3e-5 .. 1.6e-4, bound 2e-4; 2.5e-4 on the 11 x 11 fixture: profiles/r05_bf16_parity_table.txt)."""
    tol_loss, tol_recon, tol_zw, tol_norm, min_cos = BF16_BOUNDS[name]
    z, case = load_case(name)
    m = build_model(case, "bf16")
    x = torch.from_numpy(z["x"]).cuda()
    noise = {k: torch.from_numpy(z[k]).cuda() for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}
    m.zero_grad()
    loss, recon, z_where, z_pres = m(x, int(z["global_step"]), noise=noise)
    assert abs(loss.item() - float(z["loss"])) <= tol_loss * abs(float(z["loss"]))
    assert np.abs(recon.cpu().numpy() - z["recon_x"]).max() < tol_recon
    assert np.abs(z_where.cpu().numpy() - z["z_where"]).max() < tol_zw
    loss.backward()
    bad = []
    for k, p in m.named_parameters():
        if k.startswith("attn."):
            continue
        gn = float(p.grad.double().norm().item())
        ref_n = float(z["gradnorm_" + k])
        if abs(gn - ref_n) > tol_norm * ref_n + 1e-5:
            bad.append((k, "norm", gn / max(ref_n, 1e-30)))
        # direction: full tensor when the fixture holds it, the fixed sample of elements otherwise
        g = p.grad.detach().double().cpu().flatten().numpy()
        if "grad_" + k in z.files:
            ref = z["grad_" + k].astype(np.float64).flatten()
        else:
            g, ref = g[z["gradidx_" + k]], z["gradsample_" + k].astype(np.float64)
        if np.linalg.norm(ref) > 1e-6 * max(1.0, ref_n):
            cos = float(np.dot(g, ref) / (np.linalg.norm(g) * np.linalg.norm(ref) + 1e-30))
            if cos < min_cos:
                bad.append((k, "cos", cos))
    assert not bad, bad


def _bench_model(dtype, I, seed=3):
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.models import SPAIR
    cfg.set_grid(I, (2, 2, 2, 1, 1, 1))
    torch.manual_seed(seed)                    # train.py:39
    return SPAIR([1, I, I], None, torch.device("cuda"), compute_dtype=dtype).to("cuda")


def test_bf16_vs_f32_mode_full_bench_batch():
    """BASELINE configs[1] at its full size (B=256, 128x128, 16x16 grid), which the CPU reference cannot run (85 GB): the bf16 step against
    the fp32-MFMA mode of the same engine (itself pinned to the reference on the small fixtures) on the same weights, scenes and noise."""
    from spair_pytorch_amd.data import scattered_digits
    x = torch.from_numpy(scattered_digits(1234, 256, 128, 11)[0]).cuda()
    out = {}
    for dtype in ("f32", "bf16"):
        m = _bench_model(dtype, 128)
        torch.manual_seed(7)                   # the noise seed comes from torch's CPU generator: same draws in both modes
        m.zero_grad()
        loss, recon, z_where, z_pres = m(x, 2000)
        loss.backward()
        out[dtype] = dict(loss=loss.item(), terms=m.loss_terms().cpu().numpy().copy(), recon=recon.cpu().numpy(), z_where=z_where.cpu().numpy(),
                          z_pres=z_pres.cpu().numpy(), grad=m.flat_gradients().double().cpu().numpy().copy(),
                          slices={k: v for k, v in m._slices.items()})
        del m
        torch.cuda.empty_cache()
    a, b = out["f32"], out["bf16"]
    assert np.isfinite(b["grad"]).all()
    assert abs(b["loss"] - a["loss"]) <= 3e-4 * abs(a["loss"]), (a["loss"], b["loss"])          # observed 1.0e-4
    assert np.abs(b["terms"][1:9] - a["terms"][1:9]).max() <= 3e-4 * abs(a["loss"])
    assert np.abs(b["z_where"] - a["z_where"]).max() < 5e-3 and np.abs(b["z_pres"] - a["z_pres"]).max() < 5e-3
    assert np.abs(b["recon"] - a["recon"]).mean() < 1e-3
    # 65,536 rows per gradient sum: every tensor's direction and size agree closely.  Observed (tools/exp/fullbatch_bf16_f32.py): cos >= 0.9997
    # and |g| within 0.4 % for 43 of the 49 tensors; the stem weight (cos 0.9897), the box net's first layer (cos 0.9887) and the box net's
    # body / latent head (|g| 2.4 - 3.3 % low, cos >= 0.994) are the outliers, the same with the fused and the per-wavefront chain.
    bad = []
    for k, (off, cnt, shp) in a["slices"].items():
        if k.startswith("attn."):
            continue
        ga, gb = a["grad"][off:off + cnt], b["grad"][off:off + cnt]
        na, nb_ = np.linalg.norm(ga), np.linalg.norm(gb)
        cos = float(np.dot(ga, gb) / (na * nb_ + 1e-30))
        if cos < 0.985 or abs(nb_ / na - 1.0) > 0.04:
            bad.append((k, cos, nb_ / na))
    assert not bad, bad


def test_bf16_trajectory_tracks_f32_mode():
    """60 optimizer steps from the same initial weights, scenes and noise stream: the bf16 ELBO follows the fp32-mode ELBO step by step while
    the two runs are still the same trajectory (20 steps: <= 1e-3 relative, 1.2e-4 observed; afterwards the two
    trajectories have separated by more than rounding: the fp32 MODE still sums its split-K partials with fp32 atomics, so even two fp32 runs drift), both fall, and they end within 5 % of each other."""
    from spair_pytorch_amd.data import scattered_digits
    from spair_pytorch_amd.optim import FusedAdam
    x = torch.from_numpy(scattered_digits(5, 16, 48, 3)[0]).cuda()
    traj = {}
    for dtype in ("f32", "bf16"):
        m = _bench_model(dtype, 48)
        opt = FusedAdam(m, lr=1e-4)
        torch.manual_seed(11)
        losses = []
        for s in range(60):
            opt.zero_grad()
            loss = m(x, 1000 + s)[0]
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        traj[dtype] = torch.stack(losses).cpu().numpy().astype(np.float64)
    a, b = traj["f32"], traj["bf16"]
    assert np.isfinite(b).all()
    assert (np.abs(b[:20] - a[:20]) / np.abs(a[:20])).max() <= 1e-3
    assert b[-1] < 0.8 * b[0] and a[-1] < 0.8 * a[0]
    assert abs(b[-1] - a[-1]) <= 0.05 * abs(a[-1])


def test_adam_step_matches_torch():
    import ctypes
    from spair_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(0)
    n = 100003
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-4)
    p, m, v = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    grd = gr.cuda()
    for step in (1, 2, 3):
        ref.grad = gr.clone()
        opt.step()
        L.check(L.lib().spair_adam(L.ptr(p), L.ptr(grd), L.ptr(m), L.ptr(v), ctypes.c_int64(n), ctypes.c_float(1e-4), ctypes.c_float(0.9),
                                   ctypes.c_float(0.999), ctypes.c_float(1e-8), step, L.stream()), "adam")
    assert (p.cpu() - ref.detach()).abs().max() < 1e-6


def test_noise_statistics():
    import ctypes
    from spair_pytorch_amd import _lib as L
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.models import make_dims
    cfg.set_grid(128, (2, 2, 2, 1, 1, 1))
    topo = [dict(filters=128, kernel_size=k, stride=s) for k, s in zip((4, 4, 4, 1, 1, 1), (2, 2, 2, 1, 1, 1))]
    d = make_dims(16, [1, 128, 128], topo, "f32")
    B, G, A = 16, d.G, d.A
    eb, ea = torch.empty(B, 4, G, G, device="cuda"), torch.empty(B, A, G, G, device="cuda")
    ed, up = torch.empty(B, 1, G, G, device="cuda"), torch.empty(B, 1, G, G, device="cuda")
    L.check(L.lib().spair_noise_fill(ctypes.byref(d), ctypes.c_uint64(1234), L.ptr(eb), L.ptr(ea), L.ptr(ed), L.ptr(up), L.stream()), "noise")
    assert abs(ea.mean().item()) < 0.01 and abs(ea.std().item() - 1) < 0.01
    assert 0 < up.min().item() and up.max().item() < 1 and abs(up.mean().item() - 0.5) < 0.03
    ea2 = torch.empty_like(ea)
    L.check(L.lib().spair_noise_fill(ctypes.byref(d), ctypes.c_uint64(1234), L.ptr(eb), L.ptr(ea2), L.ptr(ed), L.ptr(up), L.stream()), "noise")
    assert torch.equal(ea, ea2)   # counter-based: same seed, same stream


def test_internal_noise_follows_torch_seed():
    """Without injected noise the 7 per-cell draws come from SpairStep.draw_noise (filled inside spair_forward on its helper stream)
    with a seed taken from torch's CPU generator: same torch seed -> same step to the bit, another seed -> another sample; and the
    maps the backward reads are the ones the forward drew (the gradient of a repeated step is identical)."""
    z, case = load_case("c1_b8_step1001")
    m = build_model(case, "bf16")
    x = torch.from_numpy(z["x"]).cuda()

    def run(seed):
        torch.manual_seed(seed)
        m.zero_grad()
        loss = m(x, 1001)[0]
        loss.backward()
        return loss.item(), m.flat_gradients().clone()

    la, ga = run(5)
    lb, gb = run(5)
    lc, gc = run(6)
    assert la == lb
    assert torch.equal(ga, gb)      # the bf16 step has no atomics: the same seed repeats to the bit
    assert lc != la
    eps = m._last["engine"]["noise"]["eps_attr"]
    assert abs(eps.mean().item()) < 0.05 and abs(eps.std().item() - 1.0) < 0.05


def test_batch_split_equals_full_batch():
    """Size-independent property at the bench geometry (128x128, 16x16 grid): samples are independent, so a batch of 32 run as two
    'ranks' of 16 (world_size = 2: each back-propagates BCE_sum_local + KL_sum_local / (B_local * 2)) must give the full batch's loss
    and gradient when the two are SUMMED -- the data-parallel contract of ddp.py, checked here without any collective."""
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.models import SPAIR
    from spair_pytorch_amd.data import scattered_digits
    I, strides, B = 128, (2, 2, 2, 1, 1, 1), 32
    cfg.set_grid(I, strides)
    G = gi.grid_side(I, strides)
    x = torch.from_numpy(scattered_digits(11, B, I, 11)[0]).cuda()
    noise = {k: torch.from_numpy(v).cuda() for k, v in gi.make_noise(12, B, G).items()}
    w = {k: torch.from_numpy(v) for k, v in gi.make_weights(13, 1.0).items()}

    def run(xs, ns, world):
        m = SPAIR([1, I, I], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
        m.load_state_dict(w)
        m.world_size = world
        m.zero_grad()
        loss, recon, z_where, z_pres = m(xs, 3000, noise=ns)
        loss.backward()
        return loss.item(), m.flat_gradients().double().clone(), recon

    lf, gf, rf = run(x, noise, 1)
    h = B // 2
    la, ga, ra = run(x[:h], {k: v[:h] for k, v in noise.items()}, 2)
    lb, gb, rb = run(x[h:], {k: v[h:] for k, v in noise.items()}, 2)
    assert abs((la + lb) - lf) <= 1e-5 * abs(lf)
    assert ((ga + gb) - gf).norm().item() <= 2e-3 * gf.norm().item()      # bf16 GEMM partial sums are grouped differently
    assert (torch.cat([ra, rb]) - rf).abs().max().item() == 0.0


def test_full_size_step_is_finite_and_repeatable():
    """BASELINE configs[1] at full size (B = 256, 128x128): every launch configuration the bench uses (grouped weight gradients, split-K
    scratch, fused stem partials, XCD-ordered grids) -- the step must be finite, repeat to the bit in the loss, and its first
    Adam update must lower the loss on the same batch and noise."""
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.models import SPAIR
    from spair_pytorch_amd.optim import FusedAdam
    from spair_pytorch_amd.data import scattered_digits
    I, strides, B = 128, (2, 2, 2, 1, 1, 1), 256
    cfg.set_grid(I, strides)
    G = gi.grid_side(I, strides)
    x = torch.from_numpy(scattered_digits(21, B, I, 11)[0]).cuda()
    noise = {k: torch.from_numpy(v).cuda() for k, v in gi.make_noise(22, B, G).items()}
    torch.manual_seed(3)
    m = SPAIR([1, I, I], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
    opt = FusedAdam(m, lr=1e-4)

    def step(update):
        opt.zero_grad()
        loss, recon, z_where, z_pres = m(x, 2500, noise=noise)
        loss.backward()
        g = m.flat_gradients()
        assert torch.isfinite(loss).item() and torch.isfinite(g).all().item() and torch.isfinite(recon).all().item()
        assert z_where.shape == (B, 4, G, G) and z_pres.shape == (B, 1, G, G)
        if update:
            opt.step()
        return loss.item(), g.double().norm().item()

    l0, n0 = step(False)
    l1, n1 = step(True)
    assert l0 == l1 and abs(n0 - n1) <= 1e-5 * n0
    l2, _ = step(False)
    assert l2 < l1
