"""GPU parity of the STN gather and the fused renderer, each through its C-ABI entry point, against
the reference's own vectors (units.npz) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import spair_oracle as orc

pytestmark = pytest.mark.gpu


def _L():
    from spair_pytorch_amd import _lib as L
    return L


def test_stn_glimpse_fwd_bwd_vs_reference(golden_dir):
    L = _L()
    u = np.load(os.path.join(golden_dir, "units.npz"))
    img, zw = torch.from_numpy(u["stn_img"]).cuda(), torch.from_numpy(u["stn_zw"]).cuda()
    n, C, I, P = img.shape[0], 1, img.shape[-1], 28
    out = torch.zeros(n, C * P * P, device="cuda")
    L.check(L.lib().spair_stn_glimpse_fwd(L.ptr(img), L.ptr(zw), n, L.ptr(out), C * P * P, n, C, I, P, 0, L.stream()), "stn fwd")
    assert np.abs(out.cpu().numpy().reshape(n, C, P, P) - u["stn_glimpse"]).max() < 1e-5
    gw = torch.from_numpy(u["stn_gw"]).reshape(n, -1).contiguous().cuda()
    dzw = torch.zeros(n, 4, device="cuda")
    L.check(L.lib().spair_stn_glimpse_bwd(L.ptr(img), L.ptr(zw), n, L.ptr(gw), C * P * P, L.ptr(dzw), n, C, I, P, 0, L.stream()), "stn bwd")
    assert np.abs(dzw.cpu().numpy() - u["stn_dzw"]).max() <= 3e-4 * np.abs(u["stn_dzw"]).max()


def _render_oracle(S, nbox, pres, depth, x, B, HW, I, P):
    """Composite + BCE with the oracle's inverse STN (closed-form inverse), rows r = k*B + b."""
    N = B * HW
    grey, a0 = S[..., 0], S[..., 1]
    alpha = a0 * pres.view(N, 1, 1)
    imp = torch.clamp(alpha * depth.view(N, 1, 1), min=0.01)
    objs = torch.stack([grey, alpha, imp], 1)                       # [N,3,P,P]
    t = orc.stn(objs, nbox, (I, I), inverse=True, inverse_mode="closed")  # [N,3,I,I]
    t = t.view(HW, B, 3, I, I).permute(1, 0, 2, 3, 4)               # [B,HW,3,I,I]
    colour, al, im = t[:, :, 0:1], t[:, :, 1:2], t[:, :, 2:3] + 1e-9
    im = im / im.sum(1, keepdim=True)
    rec = torch.clamp((al * colour * im).sum(1), 0, 1)
    return rec, torch.nn.functional.binary_cross_entropy(rec, x, reduction="sum")


@pytest.mark.parametrize("B,G,I,smin,srange", [(2, 3, 48, 0.08, 0.5), (8, 4, 64, 0.08, 0.5),
                                                (1, 2, 128, 0.7, 0.5),     # magnified sprites: tiled + row-chunked backward
                                                (2, 4, 96, 0.02, 0.1)])    # minified sprites (objects smaller than 28 px)
def test_render_fwd_bwd_vs_oracle(B, G, I, smin, srange):
    L = _L()
    P, HW = 28, G * G
    N = B * HW
    g = torch.Generator().manual_seed(B + G + I)
    logits = torch.randn(N, P, P, 2, generator=g)
    logits[..., 1] += 1.0
    S = torch.sigmoid(logits).requires_grad_(True)
    nbox = torch.stack([torch.rand(N, generator=g) * 1.2 - 0.1, torch.rand(N, generator=g) * 1.2 - 0.1,
                        torch.rand(N, generator=g) * srange + smin, torch.rand(N, generator=g) * srange + smin], 1).requires_grad_(True)
    pres = torch.rand(N, generator=g).requires_grad_(True)
    depth = (torch.rand(N, generator=g) * 4).requires_grad_(True)
    x = (torch.rand(B, 1, I, I, generator=g) > 0.7).float() * torch.rand(B, 1, I, I, generator=g)
    rec_o, bce_o = _render_oracle(S, nbox, pres, depth, x, B, HW, I, P)
    bce_o.backward()

    Sd = S.detach().reshape(N, -1).contiguous().cuda()
    nb, pr, dp, xd = nbox.detach().cuda(), pres.detach().cuda(), depth.detach().cuda(), x.cuda()
    recon = torch.zeros(B, 1, I, I, device="cuda")
    aux = torch.zeros(B, I, I, 4, device="cuda")
    nblk = B * ((I + 15) // 16) ** 2
    part = torch.zeros(nblk, device="cuda")
    ld = P * P * 2
    L.check(L.lib().spair_render_fwd(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(xd), L.ptr(recon), L.ptr(aux), L.ptr(part),
                                     B, HW, 1, I, P, 0, L.stream()), "render fwd")
    assert (recon.cpu() - rec_o.detach()).abs().max() < 2e-5
    assert abs(part.sum().item() - bce_o.item()) <= 2e-5 * bce_o.item()
    gl = torch.ones((), device="cuda")
    dlog = torch.zeros(N, ld, device="cuda")
    dnb, dpr, ddp = torch.zeros(N, 4, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    L.check(L.lib().spair_render_bwd(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(aux), L.ptr(gl), L.ptr(dlog), L.ptr(dnb),
                                     L.ptr(dpr), L.ptr(ddp), B, HW, 1, I, P, 0, ctypes_f(2.0), ctypes_f(0.1), L.stream()), "render bwd")
    s = S.detach()
    scale = torch.tensor([2.0, 0.1]).view(1, 1, 1, 2)
    ref_dlog = (S.grad * s * (1 - s) * scale).reshape(N, -1)

    def close(a, b, tol):
        d = (a.cpu() - b).abs()
        i = int(d.argmax())
        ok = d.max().item() <= tol * b.abs().max().item() + 1e-7
        if not ok:
            print("worst @%d: got %g want %g (max |want| %g)" % (i, a.cpu().flatten()[i], b.flatten()[i], b.abs().max()))
            if a.dim() == 2 and a.shape[1] == 4:
                print("nbox row", nbox.detach()[i // 4], "pres", pres.detach()[i // 4], "depth", depth.detach()[i // 4])
        return ok

    assert close(dlog, ref_dlog, 2e-4)
    assert close(dpr, pres.grad, 2e-4)
    assert close(ddp, depth.grad, 2e-4)
    assert close(dnb, nbox.grad, 1e-3)


def ctypes_f(v):
    import ctypes
    return ctypes.c_float(v)


@pytest.mark.parametrize("B,G,I,C,smin,srange", [(2, 3, 48, 3, 0.08, 0.5), (3, 4, 64, 3, 0.08, 0.5), (1, 2, 96, 2, 0.7, 0.5),
                                                  (2, 4, 80, 3, 0.02, 0.1)])
def test_render_rgb_fwd_bwd_vs_oracle(B, G, I, C, smin, srange):
    """The generic-channel renderer (render_c.hip; cfg.INPUT_IMAGE_SHAPE[0] = C > 1) against the oracle's own composite
    (models.py:452-547 with C colour channels) and its autograd: forward 2e-5, every gradient 2e-4 (d z_where 1e-3) of the largest
    element -- the bounds of the greyscale fp32 kernels -- and bit-identical from run to run."""
    L = _L()
    P, HW = 28, G * G
    N = B * HW
    g = torch.Generator().manual_seed(B + G + I + C)
    logits = torch.randn(N, P, P, C + 1, generator=g)
    logits[..., C] += 1.0
    S = torch.sigmoid(logits).requires_grad_(True)
    nbox = torch.stack([torch.rand(N, generator=g) * 1.2 - 0.1, torch.rand(N, generator=g) * 1.2 - 0.1,
                        torch.rand(N, generator=g) * srange + smin, torch.rand(N, generator=g) * srange + smin], 1).requires_grad_(True)
    pres = torch.rand(N, generator=g).requires_grad_(True)
    depth = (torch.rand(N, generator=g) * 4).requires_grad_(True)
    x = (torch.rand(B, C, I, I, generator=g) > 0.7).float() * torch.rand(B, C, I, I, generator=g)
    # the oracle's composite (oracle.render's formulas on given sprites), rows r = k*B + b
    alpha = S[..., C] * pres.view(N, 1, 1)
    imp = torch.clamp(alpha * depth.view(N, 1, 1), min=0.01)
    objs = torch.cat([S[..., :C].permute(0, 3, 1, 2), alpha[:, None], imp[:, None]], 1)          # [N,C+2,P,P]
    t = orc.stn(objs, nbox, (I, I), inverse=True, inverse_mode="closed").view(HW, B, C + 2, I, I).permute(1, 0, 2, 3, 4)
    colour, al, im = t[:, :, :C], t[:, :, C:C + 1], t[:, :, C + 1:C + 2] + 1e-9
    im = im / im.sum(1, keepdim=True)
    rec_o = torch.clamp((al * colour * im).sum(1), 0, 1)
    bce_o = torch.nn.functional.binary_cross_entropy(rec_o, x, reduction="sum")
    bce_o.backward()

    Sd = S.detach().reshape(N, -1).contiguous().cuda()
    nb, pr, dp, xd = nbox.detach().cuda(), pres.detach().cuda(), depth.detach().cuda(), x.cuda()
    ld = P * P * (C + 1)
    nblk = B * ((I + 15) // 16) ** 2
    gl = torch.ones((), device="cuda")
    outs = []
    for rep in range(2):
        recon = torch.zeros(B, C, I, I, device="cuda")
        aux = torch.zeros(B, C, I, I, 2, device="cuda")
        part = torch.zeros(nblk, device="cuda")
        L.check(L.lib().spair_render_fwd_rgb(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(xd), L.ptr(recon), L.ptr(aux), L.ptr(part),
                                             B, HW, C, I, P, 0, L.stream()), "render fwd rgb")
        dlog = torch.zeros(N, ld, device="cuda")
        dnb, dpr, ddp = torch.zeros(N, 4, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
        L.check(L.lib().spair_render_bwd_rgb(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(aux), L.ptr(gl), L.ptr(dlog), L.ptr(dnb),
                                             L.ptr(dpr), L.ptr(ddp), B, HW, C, I, P, 0, ctypes_f(2.0), ctypes_f(0.1), L.stream()), "render bwd rgb")
        outs.append([t_.cpu() for t_ in (recon, part, dlog, dnb, dpr, ddp)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)                                   # run-to-run bit-identical (one wave per object, no global atomics)
    recon, part, dlog, dnb, dpr, ddp = outs[0]
    assert (recon - rec_o.detach()).abs().max() < 2e-5
    assert abs(part.sum().item() - bce_o.item()) <= 2e-5 * bce_o.item()
    s = S.detach()
    scale = torch.tensor([2.0] * C + [0.1]).view(1, 1, 1, C + 1)
    ref_dlog = (S.grad * s * (1 - s) * scale).reshape(N, -1)

    def close(a, b, tol):
        return (a - b).abs().max().item() <= tol * b.abs().max().item() + 1e-7

    assert close(dlog, ref_dlog, 2e-4)
    assert close(dpr, pres.grad, 2e-4)
    assert close(ddp, depth.grad, 2e-4)
    assert close(dnb, nbox.grad, 1e-3)


@pytest.mark.parametrize("B,G,I,smin,srange", [(8, 4, 64, 0.08, 0.5), (4, 8, 128, 0.12, 0.12),      # the bench geometry's object sizes
                                                (1, 2, 128, 0.7, 0.5),                                # magnified: several pixel chunks per object
                                                (2, 4, 96, 0.02, 0.1)])                               # minified
def test_render16_fwd_bwd_vs_oracle(B, G, I, smin, srange):
    """The renderer kernels of the bf16 step (k_render_fwd3 on fp16 sprites, k_render_bwd2: one wave per object, sampling transpose on the
    matrix cores) through spair_render_fwd16 / _bwd16 against the oracle evaluated on the SAME fp16-rounded sprites.  Forward: fp32 math,
    2e-5.  Backward: the adjoints and hat weights enter the MFMAs as bf16 and the d-logits leave as bf16 -> 1e-2 relative to the largest
    element, cosine >= 0.9999; d z_where is fp32 throughout (1e-3), d pres / d depth come from the bf16-product texel sums (1e-2)."""
    L = _L()
    P, HW = 28, G * G
    N = B * HW
    g = torch.Generator().manual_seed(B + G + I + 1)
    logits = torch.randn(N, P, P, 2, generator=g)
    logits[..., 1] += 1.0
    S = torch.sigmoid(logits).half().float().requires_grad_(True)          # exactly fp16-representable sprites
    nbox = torch.stack([torch.rand(N, generator=g) * 1.2 - 0.1, torch.rand(N, generator=g) * 1.2 - 0.1,
                        torch.rand(N, generator=g) * srange + smin, torch.rand(N, generator=g) * srange + smin], 1).requires_grad_(True)
    pres = torch.rand(N, generator=g).requires_grad_(True)
    depth = (torch.rand(N, generator=g) * 4).requires_grad_(True)
    x = (torch.rand(B, 1, I, I, generator=g) > 0.7).float() * torch.rand(B, 1, I, I, generator=g)
    rec_o, bce_o = _render_oracle(S, nbox, pres, depth, x, B, HW, I, P)
    bce_o.backward()

    Sd = S.detach().reshape(N, -1).half().contiguous().cuda()
    nb, pr, dp, xd = nbox.detach().cuda(), pres.detach().cuda(), depth.detach().cuda(), x.cuda()
    recon = torch.zeros(B, 1, I, I, device="cuda")
    aux = torch.zeros(B, I, I, 2, device="cuda")
    part = torch.zeros(B * ((I + 15) // 16) ** 2, device="cuda")
    ld = P * P * 2
    L.check(L.lib().spair_render_fwd16(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(xd), L.ptr(recon), L.ptr(aux), L.ptr(part),
                                       B, HW, 1, I, P, 0, L.stream()), "render fwd16")
    assert (recon.cpu() - rec_o.detach()).abs().max() < 2e-5
    assert abs(part.sum().item() - bce_o.item()) <= 2e-5 * bce_o.item()
    gl = torch.ones((), device="cuda")
    dlog = torch.zeros(N, ld, device="cuda", dtype=torch.bfloat16)
    dnb, dpr, ddp = torch.zeros(N, 4, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    L.check(L.lib().spair_render_bwd16(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(aux), L.ptr(gl), L.ptr(dlog), L.ptr(dnb),
                                       L.ptr(dpr), L.ptr(ddp), B, HW, 1, I, P, 0, ctypes_f(2.0), ctypes_f(0.1), L.stream()), "render bwd16")
    s = S.detach()
    ref_dlog = (S.grad * s * (1 - s) * torch.tensor([2.0, 0.1]).view(1, 1, 1, 2)).reshape(N, -1)
    got = dlog.float().cpu()
    assert (got - ref_dlog).abs().max().item() <= 1e-2 * ref_dlog.abs().max().item() + 1e-7
    cos = float((got.double() * ref_dlog.double()).sum() / (got.double().norm() * ref_dlog.double().norm() + 1e-30))
    assert cos >= 0.9999, cos
    assert (dnb.cpu() - nbox.grad).abs().max().item() <= 1e-3 * nbox.grad.abs().max().item() + 1e-7
    assert (dpr.cpu() - pres.grad).abs().max().item() <= 1e-2 * pres.grad.abs().max().item() + 1e-7
    assert (ddp.cpu() - depth.grad).abs().max().item() <= 1e-2 * depth.grad.abs().max().item() + 1e-7


@pytest.mark.parametrize("B,G,I,smin,srange", [(8, 4, 64, 0.08, 0.5), (4, 8, 128, 0.12, 0.12),      # the bench geometry's object sizes
                                                (8, 16, 128, 0.1, 0.2),                               # B % 8 == 0: the XCD-aware tile order
                                                (1, 2, 128, 0.7, 0.5),                                # magnified: one 16-row tile per pair
                                                (2, 4, 96, 0.02, 0.1),                                # minified: both tiles
                                                (3, 5, 72, 0.05, 0.6)])                               # image side not a multiple of 16
def test_render16m_fwd_vs_oracle(B, G, I, smin, srange):
    """The matrix-core forward renderer of the bf16 step (render3.hip: spair_render_prep + spair_render_fwd16m) against the oracle on the
    SAME fp16 sprites, and against the tap kernel (spair_render_fwd16).  Its sampling is not fp32: source coordinates are rounded to odd
    multiples of 2^-11 texel (so that the hat weights are exact fp16 pairs summing to 1 and never 0: render3.h), the x-interpolated rows
    are rounded to fp16 once (2^-12 relative, nearest-even), the importance floor is an fp16 number (where many objects overlap that moves
    the importance WEIGHTS between objects) -- per pixel <= 1.5e-3 (observed <= 9e-4), rms <= 1.5e-4 (observed 6.5e-5), and WITHOUT bias:
    the mean signed error stays below 6e-6 (observed <= 3e-6, the 4-object case) and the BCE sum within 2e-5 (observed 2e-6).  The backward (k_render_bwd2) reads this kernel's aux record, so the
    record of the tap kernel and of this one must agree to the same bound."""
    L = _L()
    P, HW = 28, G * G
    N = B * HW
    g = torch.Generator().manual_seed(B + G + I + 2)
    logits = torch.randn(N, P, P, 2, generator=g)
    logits[..., 1] += 1.0
    S = torch.sigmoid(logits).half().float()
    nbox = torch.stack([torch.rand(N, generator=g) * 1.2 - 0.1, torch.rand(N, generator=g) * 1.2 - 0.1,
                        torch.rand(N, generator=g) * srange + smin, torch.rand(N, generator=g) * srange + smin], 1)
    pres = torch.rand(N, generator=g)
    depth = torch.rand(N, generator=g) * 4
    if N > 4:       # degenerate objects: off-screen, vanishing, absent -- culled or drawn with weight ~0, never NaN
        nbox[0] = torch.tensor([5.0, 5.0, 0.2, 0.2]); nbox[1] = torch.tensor([0.5, 0.5, 1e-12, 0.3]); pres[2] = 0.0
        nbox[3] = torch.tensor([-0.2, 0.98, 0.5, 0.5])
    x = (torch.rand(B, 1, I, I, generator=g) > 0.7).float() * torch.rand(B, 1, I, I, generator=g)
    rec_o, bce_o = _render_oracle(S, nbox, pres, depth, x, B, HW, I, P)
    Sd = S.reshape(N, -1).half().contiguous().cuda()
    nb, pr, dp, xd = nbox.cuda(), pres.cuda(), depth.cuda(), x.cuda()
    ld = P * P * 2
    nblk = B * ((I + 15) // 16) ** 2
    out = {}
    for name in ("taps", "mma"):
        recon = torch.zeros(B, 1, I, I, device="cuda")
        aux = torch.zeros(B, I, I, 2, device="cuda")
        part = torch.zeros(nblk, device="cuda")
        if name == "taps":
            L.check(L.lib().spair_render_fwd16(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(xd), L.ptr(recon), L.ptr(aux), L.ptr(part),
                                               B, HW, 1, I, P, 0, L.stream()), "render fwd16")
        else:
            recs = torch.zeros(N * 16, device="cuda", dtype=torch.int32)
            L.check(L.lib().spair_render_prep(L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(recs), B, HW, I, P, 0, L.stream()), "render prep")
            L.check(L.lib().spair_render_fwd16m(L.ptr(Sd), ld, L.ptr(recs), L.ptr(xd), L.ptr(recon), L.ptr(aux), L.ptr(part),
                                                B, HW, 1, I, P, 0, L.stream()), "render fwd16m")
        torch.cuda.synchronize()
        out[name] = (recon.cpu(), aux.cpu(), part.sum().item())
    rec_m, aux_m, bce_m = out["mma"]
    rec_t, aux_t, bce_t = out["taps"]
    assert torch.isfinite(rec_m).all() and torch.isfinite(aux_m).all()
    err = rec_m - rec_o
    rms = err.double().pow(2).mean().sqrt().item()
    print("mma vs oracle: max %.3g rms %.3g mean %.3g | taps vs oracle: max %.3g | bce rel %.3g" %
          (err.abs().max(), rms, err.mean(), (rec_t - rec_o).abs().max(), abs(bce_m - bce_o.item()) / bce_o.item()))
    assert err.abs().max() < 1.5e-3
    assert rms < 1.5e-4
    assert abs(err.mean().item()) < 6e-6
    assert abs(bce_m - bce_o.item()) <= 2e-5 * bce_o.item()
    assert (aux_m[..., 1] - aux_t[..., 1]).abs().max() < 1.5e-3         # pre (unclamped)
    d0 = (aux_m[..., 0] - aux_t[..., 0]).abs()                         # dBCE/dpre / D
    assert (d0 <= 1e-2 * aux_t[..., 0].abs() + 1e-3 * aux_t[..., 0].abs().max()).all()


@pytest.mark.parametrize("B,G,I,smin,srange", [(8, 4, 64, 0.08, 0.5), (4, 8, 128, 0.12, 0.12), (3, 5, 72, 0.05, 0.6)])
def test_render16_bwd_from_records_equals_plain(B, G, I, smin, srange):
    """spair_render_bwd16r (inverse-affine parameters and pixel footprints read from the records of spair_render_prep: what the training
    step runs) against spair_render_bwd16 (recomputed per object): the records hold the same expressions' results, so every output is
    bit-equal -- including objects that are off-screen (empty footprint: all gradients zero)."""
    L = _L()
    P, HW = 28, G * G
    N = B * HW
    g = torch.Generator().manual_seed(B + G + I + 3)
    Sd = torch.sigmoid(torch.randn(N, P * P * 2, generator=g)).half().contiguous().cuda()
    nbox = torch.stack([torch.rand(N, generator=g) * 1.2 - 0.1, torch.rand(N, generator=g) * 1.2 - 0.1,
                        torch.rand(N, generator=g) * srange + smin, torch.rand(N, generator=g) * srange + smin], 1)
    nbox[0] = torch.tensor([5.0, 5.0, 0.2, 0.2]); nbox[1] = torch.tensor([-0.2, 0.98, 0.5, 0.5])
    pres, depth = torch.rand(N, generator=g), torch.rand(N, generator=g) * 4
    x = (torch.rand(B, 1, I, I, generator=g) > 0.7).float() * torch.rand(B, 1, I, I, generator=g)
    nb, pr, dp, xd = nbox.cuda(), pres.cuda(), depth.cuda(), x.cuda()
    ld = P * P * 2
    recon = torch.zeros(B, 1, I, I, device="cuda")
    aux = torch.zeros(B, I, I, 2, device="cuda")
    part = torch.zeros(B * ((I + 15) // 16) ** 2, device="cuda")
    recs = torch.zeros(N * 16, device="cuda", dtype=torch.int32)
    L.check(L.lib().spair_render_prep(L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(recs), B, HW, I, P, 0, L.stream()), "render prep")
    L.check(L.lib().spair_render_fwd16m(L.ptr(Sd), ld, L.ptr(recs), L.ptr(xd), L.ptr(recon), L.ptr(aux), L.ptr(part), B, HW, 1, I, P, 0,
                                        L.stream()), "render fwd16m")
    gl = torch.ones((), device="cuda")
    out = []
    for with_rec in (False, True):
        dlog = torch.zeros(N, ld, device="cuda", dtype=torch.bfloat16)
        dnb, dpr, ddp = torch.zeros(N, 4, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
        if with_rec:
            L.check(L.lib().spair_render_bwd16r(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(recs), L.ptr(aux), L.ptr(gl), L.ptr(dlog),
                                                L.ptr(dnb), L.ptr(dpr), L.ptr(ddp), B, HW, 1, I, P, 0, ctypes_f(2.0), ctypes_f(0.1), L.stream()), "bwd16r")
        else:
            L.check(L.lib().spair_render_bwd16(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(aux), L.ptr(gl), L.ptr(dlog),
                                               L.ptr(dnb), L.ptr(dpr), L.ptr(ddp), B, HW, 1, I, P, 0, ctypes_f(2.0), ctypes_f(0.1), L.stream()), "bwd16")
        torch.cuda.synchronize()
        out.append((dlog.float().cpu(), dnb.cpu(), dpr.cpu(), ddp.cpu()))
    for a, b in zip(out[0], out[1]):
        assert torch.isfinite(b).all()
        assert torch.equal(a, b)
    assert out[1][0].abs().max() > 0 and (out[1][1][0] == 0).all()     # the off-screen object: no gradient


@pytest.mark.parametrize("B,I,pre,post", [(3, 128, 7, 7), (2, 48, 3, 5), (1, 32, 0, 2)])
def test_stem_conv_mfma_vs_torch(B, I, pre, post):
    """The bf16-mode stem (k_conv0_fwd_c1k4_mfma: split-bf16 operands on the matrix cores) against torch's fp32 conv on the CPU: before
    the bf16 store the two agree to ~2^-15, so after it every element is within ONE bf16 step (2^-8 relative) and all but a few per
    thousand are identical to the rounded fp32 result; the fp32-output path of the same entry point (the FMA kernel) to 2e-6."""
    L = _L()
    Hin = I + pre + post
    Hout = (Hin - 4) // 2 + 1
    g = torch.Generator().manual_seed(I + pre)
    x = torch.rand(B, 1, I, I, generator=g) * (torch.rand(B, 1, I, I, generator=g) > 0.5)
    w = torch.randn(128, 1, 4, 4, generator=g) * 0.3
    b = torch.randn(128, generator=g) * 0.1
    ref = torch.relu(torch.nn.functional.conv2d(torch.nn.functional.pad(x, (pre, post, pre, post)), w, b, stride=2)).permute(0, 2, 3, 1).contiguous()
    xd, wd, bd = x.cuda().contiguous(), w.reshape(128, 16).cuda().contiguous(), b.cuda()
    o16 = torch.zeros(B, Hout, Hout, 128, device="cuda", dtype=torch.bfloat16)
    o32 = torch.zeros(B, Hout, Hout, 128, device="cuda")
    L.check(L.lib().spair_stem_conv_fwd(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(o16), B, I, pre, Hin, Hout, 128, 2, 1, L.stream()), "stem bf16")
    L.check(L.lib().spair_stem_conv_fwd(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(o32), B, I, pre, Hin, Hout, 128, 2, 0, L.stream()), "stem fp32")
    assert (o32.cpu() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    got, want = o16.float().cpu(), ref.bfloat16().float()
    assert (got - ref).abs().max().item() <= 2.0 ** -8 * max(1.0, ref.abs().max().item())
    assert (got != want).float().mean().item() < 5e-3
    # the same kernel leaving the sign-bit mask of its output (conv_1's data-gradient gate): same output bits, byte (pixel, g) bit e = channel 8g+e > 0
    o16m = torch.zeros(B, Hout, Hout, 128, device="cuda", dtype=torch.bfloat16)
    mask = torch.full((B, Hout, Hout, 16), 0xAA, device="cuda", dtype=torch.uint8)
    L.check(L.lib().spair_stem_conv_fwd_mask(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(o16m), L.ptr(mask), B, I, pre, Hin, Hout, L.stream()), "stem mask")
    assert torch.equal(o16m, o16)
    bits = (o16.float() > 0).view(B, Hout, Hout, 16, 8).to(torch.int32)
    want_mask = (bits * (2 ** torch.arange(8, device="cuda", dtype=torch.int32))).sum(-1).to(torch.uint8)
    assert torch.equal(mask, want_mask)


@pytest.mark.parametrize("N", [72, 1000, 8192])
def test_decoder_fused_bwd_vs_torch(N):
    """dec_fused_bwd.hip (the bf16 step's decoder data-gradient chain d-logits -> dH2 -> dH1 -> d z_attr as ONE kernel, backward of
    models.py:474-492) through its C-ABI entry point against torch fp32 on the same bf16 operands, layer by layer on the kernel's own
    stored input of that layer.  N = 72 / 1000: partial 128-row blocks; K = 1568 is not a multiple of the 64-deep stage."""
    import ctypes
    L = _L()
    A, LDR, H1, H2, NO = 50, 56, 128, 256, 1568
    g = torch.Generator().manual_seed(N + 1)
    rb = lambda t: t.to(torch.bfloat16).float()
    dL = rb(torch.randn(N, NO, generator=g) * 0.05)
    W0, W1, W2 = torch.randn(H1, A, generator=g) * 0.2, torch.randn(H2, H1, generator=g) * 0.1, torch.randn(NO, H2, generator=g) * 0.08
    h1 = rb(torch.relu(torch.randn(N, H1, generator=g)))            # stored forward activations: the relu gates
    h2 = rb(torch.relu(torch.randn(N, H2, generator=g)))
    dv = dict(dL=dL.to(torch.bfloat16).cuda(), W2t=W2.t().contiguous().to(torch.bfloat16).cuda(), W1t=W1.t().contiguous().to(torch.bfloat16).cuda(),
              W0t=W0.t().contiguous().to(torch.bfloat16).cuda(), H2=h2.to(torch.bfloat16).cuda(), H1=h1.to(torch.bfloat16).cuda())
    dH2 = torch.full((N + 1, H2), -7.0, dtype=torch.bfloat16, device="cuda")
    dH1 = torch.full((N + 1, H1), -7.0, dtype=torch.bfloat16, device="cuda")
    dza = torch.full((N + 1, LDR), -7.0, device="cuda")
    L.check(L.lib().spair_decoder_bwd16(L.ptr(dv["dL"]), NO, L.ptr(dv["W2t"]), NO, L.ptr(dv["W1t"]), L.ptr(dv["W0t"]), L.ptr(dv["H2"]), L.ptr(dv["H1"]),
                                        L.ptr(dH2), L.ptr(dH1), L.ptr(dza), LDR, ctypes.c_longlong(N), A, NO, L.stream()), "decoder bwd16")
    torch.cuda.synchronize()
    assert (dH2[N] == -7.0).all().item() and (dH1[N] == -7.0).all().item() and (dza[N] == -7.0).all().item()      # nothing past the last row
    assert (dza[:N, A:] == -7.0).all().item()                                                                       # ... or past the A columns
    g2, g1, gz = dH2[:N].float().cpu(), dH1[:N].float().cpu(), dza[:N, :A].cpu()
    want2 = (dL @ rb(W2)) * (h2 > 0)
    want1 = (g2 @ rb(W1)) * (h1 > 0)
    wantz = g1 @ rb(W0)
    for got, want, name in ((g2, want2, "dH2"), (g1, want1, "dH1")):
        d = (got - want).abs()
        assert (d <= 0.0079 * want.abs() + 2e-6 * want.abs().max()).all().item(), (name, float(d.max()))      # one bf16 step of the fp32 sum
        assert ((got == 0) == (want == 0)).float().mean().item() > 0.999, name                                 # the gate
    assert (gz - wantz).abs().max().item() <= 2e-5 * wantz.abs().max().item() + 1e-6                         # fp32 output: summation order only


@pytest.mark.parametrize("N", [72, 1000, 8192])
def test_decoder_fused_fwd_vs_torch(N):
    """dec_fused.hip (the bf16 step's decoder forward, models.py:474-492, as ONE activation-stationary kernel) through its C-ABI entry
    point against torch fp32 on the same bf16-rounded weights: the stored hidden activations (what the weight gradients / relu gates
    read) to one bf16 step, the fp16 sprites to the rounding of a bf16-operand product.  N = 72 / 1000: partial 256-row blocks."""
    import ctypes
    L = _L()
    A, LDZ, H1, H2, NO, LDS_ = 50, 56, 128, 256, 1568, 1568
    g = torch.Generator().manual_seed(N)
    za = torch.randn(N, LDZ, generator=g)
    za[:, A:] = 3.0                                   # pad columns: finite garbage, must not matter
    W0, b0 = torch.randn(H1, A, generator=g) * 0.2, torch.randn(H1, generator=g) * 0.1
    W1, b1 = torch.randn(H2, H1, generator=g) * 0.1, torch.randn(H2, generator=g) * 0.1
    W2, b2 = torch.randn(NO, H2, generator=g) * 0.08, torch.randn(NO, generator=g) * 0.1
    obj_s, al_s, al_b = 2.0, 0.1, 5.0
    rb = lambda t: t.to(torch.bfloat16).float()
    za16 = za.to(torch.bfloat16)
    h1 = rb(torch.relu(za16.float()[:, :A] @ rb(W0).t() + b0))
    h2 = rb(torch.relu(h1 @ rb(W1).t() + b1))
    lg = (h2 @ rb(W2).t() + b2).view(N, NO // 2, 2)
    ref = torch.stack([torch.sigmoid(lg[..., 0] * obj_s), torch.sigmoid(lg[..., 1] * al_s + al_b)], -1).view(N, NO)
    lib = L.lib()
    lib.spair_decoder_fwd16_scratch_bytes.restype = ctypes.c_int64
    scratch = torch.zeros(int(lib.spair_decoder_fwd16_scratch_bytes(NO)), dtype=torch.uint8, device="cuda")
    H1d = torch.full((N, H1), -7.0, dtype=torch.bfloat16, device="cuda")
    H2d = torch.full((N, H2), -7.0, dtype=torch.bfloat16, device="cuda")
    S = torch.full((N + 1, LDS_), -7.0, dtype=torch.float16, device="cuda")       # one guard row behind the last
    dv = [t.cuda().contiguous() for t in (W0, b0, W1, b1, W2, b2)]
    zad = za16.cuda()
    L.check(lib.spair_decoder_fwd16(L.ptr(zad), LDZ, L.ptr(dv[0]), L.ptr(dv[1]), L.ptr(dv[2]), L.ptr(dv[3]), L.ptr(dv[4]), L.ptr(dv[5]),
                                    L.ptr(H1d), L.ptr(H2d), L.ptr(S), LDS_, ctypes.c_longlong(N), A, NO, ctypes.c_float(obj_s),
                                    ctypes.c_float(al_s), ctypes.c_float(al_b), L.ptr(scratch), L.stream()), "decoder fwd16")
    torch.cuda.synchronize()
    assert (S[N] == -7.0).all().item()                                            # nothing written past the last row
    # layer by layer, each against torch on the kernel's OWN stored input of that layer (a flipped bf16 rounding of h1 would otherwise
    # propagate into h2 as an absolute error unrelated to h2's magnitude): one bf16 step (2^-7) where the fp32 sums differ in the last bits
    H1c, H2c = H1d.float().cpu(), H2d.float().cpu()
    h2_own = rb(torch.relu(H1c @ rb(W1).t() + b1))
    # decoder.out's rows are packed pre-multiplied by -scale * log2(e) (so the accumulator is the exp2 argument of the sigmoid): the bf16
    # rounding point of those weights is AFTER the scaling, mirrored here; `ref` above keeps the reference's own order (scale the logit)
    L2E = 1.4426950408889634
    sc = torch.where(torch.arange(NO) % 2 == 1, torch.tensor(-al_s * L2E), torch.tensor(-obj_s * L2E)).float()
    bsc = torch.where(torch.arange(NO) % 2 == 1, -(b2 * al_s + al_b) * L2E, -(b2 * obj_s) * L2E)
    ref_own = 1.0 / (1.0 + torch.exp2(H2c @ rb(W2 * sc[:, None]).t() + bsc))
    for got, want, name in ((H1c, h1, "h1"), (H2c, h2_own, "h2")):
        d = (got - want).abs()
        assert (d <= 0.0079 * want.abs() + 1e-6).all().item(), (name, float(d.max()))
        assert (d > 0).float().mean().item() < 0.02, name                            # ... and only where a rounding boundary was crossed
    ds = (S[:N].float().cpu() - ref_own).abs()
    assert ds.max().item() < 1.5e-3, float(ds.max())                                 # fp16 store of a value in (0, 1) + fp32 summation order
    assert ds.mean().item() < 2e-4                                                   # fp16 rounding of values in (0.5, 1): 2^-12 on average
    # end to end against the all-torch chain (decoder.out's weights rounded BEFORE the scaling there): the same up to the propagated
    # rounding flips and the two independent bf16 roundings of decoder.out's 256-term rows
    assert (H2c - h2).abs().mean().item() < 2e-4 and (S[:N].float().cpu() - ref).abs().max().item() < 1e-2


@pytest.mark.parametrize("B,Hout", [(3, 34), (2, 16), (5, 7), (1, 40), (3, 40), (2, 66)])
def test_conv_s2k4_patch_fwd_vs_torch(B, Hout):
    """conv_s2.hip (conv_1 / conv_2 of the bf16 step, modules.py:59-64: Conv2d(128, 128, 4, stride 2) + ReLU on the pre-padded NHWC input) against
    torch's fp32 conv on the same bf16-rounded operands.  34 / 16: the two layers of BASELINE configs[1] (tiles cross image boundaries at 34);
    7: tiles spanning several images, partial last tile; 40 and 66 (conv_1 of BASELINE configs[3], 256 x 256 images): the patch is a linear
    window of the sub-lattice and, where a 256-row tile of the whole batch would not fit, the tiles restart at every image -- the kernel must
    ACCEPT these sizes (round 3's whole-row patches refused them and the step fell back to the implicit GEMM)."""
    L = _L()
    Hin, C = 2 * Hout + 2, 128
    g = torch.Generator().manual_seed(Hout * 10 + B)
    x = torch.randn(B, C, Hin, Hin, generator=g).to(torch.bfloat16)
    w = (torch.randn(C, C, 4, 4, generator=g) / 45.0).to(torch.bfloat16)
    bias = torch.randn(C, generator=g) * 0.1
    ref = torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), bias, stride=2))            # [B, 128, Hout, Hout]
    ref = ref.permute(0, 2, 3, 1).reshape(B * Hout * Hout, C)
    # tap-parity K order (gemm.h GemmNT::ktab): column ((cls * 2 + half) * 4 + tap) * 64 + c
    wf = torch.empty(C, 2048, dtype=torch.bfloat16)
    for py in range(2):
        for px in range(2):
            for half in range(2):
                for dy in range(2):
                    for dx in range(2):
                        blk = ((py * 2 + px) * 2 + half) * 4 + dy * 2 + dx
                        wf[:, blk * 64:(blk + 1) * 64] = w[:, half * 64:(half + 1) * 64, py + 2 * dy, px + 2 * dx]
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()                                                 # NHWC
    out = torch.full((B * Hout * Hout + 1, C), -3.0, dtype=torch.bfloat16, device="cuda")
    wfd, bd = wf.cuda(), bias.cuda()             # (kept alive: a temporary's block would be handed to the next allocation)
    rc = L.lib().spair_conv_s2k4_fwd16(L.ptr(xin), L.ptr(wfd), L.ptr(bd), L.ptr(out), B, Hin, Hout, L.stream())
    L.check(rc, "conv_s2k4")
    torch.cuda.synchronize()
    assert (out[-1] == -3.0).all().item()
    got = out[:-1].float().cpu()
    d = (got - ref).abs()
    assert (d <= 0.0079 * ref.abs() + 2e-3).all().item(), float(d.max())       # one bf16 step of the stored value + fp32 summation order
    assert d.mean().item() < 1e-3


@pytest.mark.parametrize("B,Ho", [(3, 34), (2, 16), (5, 7), (1, 14), (3, 40), (2, 66)])
def test_conv_s2k4_patch_dgrad_vs_torch(B, Ho):
    """conv_s2_dgrad.hip (data gradient of conv_1 / conv_2 in the bf16 step + the ReLU gate of the layer below) against torch autograd on the
    same bf16-rounded operands: d x = conv2d_backward_input(d out, W) * (x > 0), all four output-parity classes from one staged d-out patch."""
    L = _L()
    Hi, C = 2 * (Ho + 1), 128
    g = torch.Generator().manual_seed(Ho * 7 + B)
    x = torch.randn(B, C, Hi, Hi, generator=g).to(torch.bfloat16)                      # the stored activation of the layer below (its sign gates)
    w = (torch.randn(C, C, 4, 4, generator=g) / 45.0).to(torch.bfloat16)             # [co][ci][ky][kx]
    dout = torch.randn(B, C, Ho, Ho, generator=g).to(torch.bfloat16)
    ref = torch.nn.grad.conv2d_input((B, C, Hi, Hi), w.float(), dout.float(), stride=2) * (x.float() > 0)
    ref = ref.permute(0, 2, 3, 1).reshape(B * Hi * Hi, C)
    wd = []
    for py in range(2):
        for px in range(2):
            m = torch.empty(C, 4 * C, dtype=torch.bfloat16)                          # [ci][(ty * 2 + tx) * 128 + co]
            for ty in range(2):
                for tx in range(2):
                    m[:, (ty * 2 + tx) * C:(ty * 2 + tx + 1) * C] = w[:, :, py + 2 * ty, px + 2 * tx].t()
            wd.append(m.cuda())
    dd = dout.permute(0, 2, 3, 1).contiguous().cuda()
    gate = x.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.full((B * Hi * Hi + 1, C), -3.0, dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().spair_conv_s2k4_dgrad16(L.ptr(dd), L.ptr(wd[0]), L.ptr(wd[1]), L.ptr(wd[2]), L.ptr(wd[3]), L.ptr(gate), L.ptr(out), B, Ho,
                                            L.stream()), "conv_s2k4 dgrad")
    torch.cuda.synchronize()
    assert (out[-1] == -3.0).all().item()
    got = out[:-1].float().cpu()
    d = (got - ref).abs()
    assert (d <= 0.0079 * ref.abs() + 2e-3).all().item(), float(d.max())
    assert ((got == 0) == (ref == 0)).float().mean().item() > 0.999                  # the gate
    # the gate as sign bits (one byte per pixel and 8 channels: what the training step reads for conv_1): bit-identical result
    gbits = ((gate.float() > 0).view(B, Hi, Hi, 16, 8).to(torch.int32) * (2 ** torch.arange(8, device="cuda", dtype=torch.int32))).sum(-1).to(torch.uint8).contiguous()
    out2 = torch.full((B * Hi * Hi + 1, C), -3.0, dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().spair_conv_s2k4_dgrad16_bits(L.ptr(dd), L.ptr(wd[0]), L.ptr(wd[1]), L.ptr(wd[2]), L.ptr(wd[3]), L.ptr(gbits), L.ptr(out2), B, Ho,
                                                 L.stream()), "conv_s2k4 dgrad bits")
    assert torch.equal(out2, out)
