"""CPU oracle for the SPAIR training step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / reported baseline.  The product path
(``spair_pytorch_amd``) never imports it and has no CPU fallback.

It is an independent torch-CPU (fp32) restatement of the reference's algorithm, written
from the formulas with the reference file:line each function
follows.  Differences from the reference are confined to plumbing: parameters are a flat
``{state_dict key: tensor}`` mapping, the 7 per-cell random draws are *inputs*
(``noise``), and nothing is logged or printed.

Parity pin: ``tests/test_oracle_golden.py`` checks this file against every vector in
``tests/golden/*.npz``, which were produced by importing the reference itself
(``tests/golden/make_golden.py``).  Pinned at torch 2.10 semantics
(``align_corners=False``, BCE log clamp -100, closed-form ``kl_normal_normal``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# Hyper-parameters (reference: spair/config.py:3-81)
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    image_shape: Tuple[int, int, int] = (1, 128, 128)          # config.py:4
    conv_kernels: Sequence[int] = (4, 4, 4, 1, 1, 1)           # config.py:7-14
    conv_strides: Sequence[int] = (3, 2, 2, 1, 1, 1)
    n_backbone_features: int = 100                             # config.py:22
    n_passthrough: int = 100                                   # config.py:24
    n_attr: int = 50                                           # config.py:27
    n_lookback: int = 1                                        # config.py:31
    object_shape: Tuple[int, int] = (28, 28)                   # config.py:33
    anchorbox: Tuple[int, int] = (48, 48)                      # config.py:34
    max_yx: float = 1.5                                        # config.py:38-41
    min_yx: float = -0.5
    max_hw: float = 1.0
    min_hw: float = 0.0
    priors: Dict[str, Tuple[float, float]] = field(default_factory=lambda: {  # config.py:45-52
        "cy_logit": (0.0, 1.0), "cx_logit": (0.0, 1.0),
        "height_logit": (7.0, 0.5), "width_logit": (7.0, 0.5),
        "attr": (0.0, 1.0), "depth_logit": (0.0, 1.0)})
    vae_beta: float = 1.0                                      # config.py:55
    wheel: Dict = field(default_factory=lambda: dict(start=1.0, end=0.0, decay_rate=0.0,
                                                     decay_step=1000.0, staircase=True))
    count_prior: Dict = field(default_factory=lambda: dict(start=1000000.0, end=0.0125,
                                                           decay_rate=0.1, decay_step=1000.0,
                                                           log_space=True))
    obj_logit_scale: float = 2.0                               # config.py:74-76
    alpha_logit_scale: float = 0.1
    alpha_logit_bias: float = 5.0
    align_corners: bool = False                                # torch>=1.3 default
    inverse_mode: str = "lu"   # "lu": batched 3x3 inverse like modules.py:258-261; "closed": 1/xs form
    # Convolutional object encoder / decoder variant: (filters, kernel, stride) per layer of CONV_OBJECT_ENCODER_TOPOLOGY
    # (config.py:15-20), None = the live MLP pair.  PARITY UNPINNED: models.py:606-665 cannot run (Linear(123, ..), undefined
    # self.linear, list.reverse() is None); this restates what the sketch says it means -- Conv2d + ReLU per layer, flatten in
    # (C,H,W) order into `out`; decoder: `inp` Linear to the encoder's last map, the mirrored ConvTranspose2d stack with ReLU
    # between layers, whose C+1 output channels are the per-pixel sprite logits (colour.., alpha).
    object_conv: Optional[Sequence[Tuple[int, int, int]]] = None


def exponential_decay(global_step, start, end, decay_rate, decay_step, staircase=False,
                      log_space=False) -> torch.Tensor:
    """modules.py:191-213 -- fp32 tensor arithmetic, ``//`` for staircase."""
    gs = torch.tensor(global_step, dtype=torch.float32)
    t = gs // decay_step if staircase else gs / decay_step
    value = (start - end) * (decay_rate ** t) + end
    if log_space:
        value = (value + 1e-6).log()
    return value


def backbone_geometry(image_hw: int, kernels: Sequence[int], strides: Sequence[int]):
    """modules.py:68-105: receptive-field padding.  Returns (pad_pre, pad_post, G, cell_px, rf)."""
    j, r = 1, 1
    for k, s in zip(kernels, strides):
        r = r + (k - 1) * j
        j = j * s
    cell = j
    pre = int(math.floor(r / 2 - cell / 2))
    G = int(math.ceil(image_hw / cell))
    required = r + (G - 1) * cell
    post = required - image_hw - pre
    return pre, post, G, cell, r


def backbone_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
    """modules.py:107-111 with the Sequential built at modules.py:43-66."""
    pre, post, _, _, _ = backbone_geometry(cfg.image_shape[1], cfg.conv_kernels, cfg.conv_strides)
    h = F.pad(x, (pre, post, pre, post))
    for i, s in enumerate(cfg.conv_strides):
        h = F.relu(F.conv2d(h, p[f"backbone.net.conv_{i}.weight"], p[f"backbone.net.conv_{i}.bias"], stride=s))
    return F.conv2d(h, p["backbone.net.conv_out.weight"], p["backbone.net.conv_out.bias"])


def _mlp(p, prefix, x, n_hidden=2, multi=False):
    """modules.py:124-165,276-284."""
    body = prefix + (".body" if multi else "")
    for i in range(n_hidden):
        x = F.relu(F.linear(x, p[f"{body}.dense{i}.weight"], p[f"{body}.dense{i}.bias"]))
    if multi:
        return [F.linear(x, p[f"{prefix}.output_layers.{i}.weight"], p[f"{prefix}.output_layers.{i}.bias"])
                for i in range(2)]
    return F.linear(x, p[f"{prefix}.out.weight"], p[f"{prefix}.out.bias"])


def object_encoder(p, glimpse, cfg: "OracleConfig"):
    """models.py:152 (MLP) / models.py:606-631 (conv sketch).  glimpse [B,C,P,P] -> [B,2A]."""
    if cfg.object_conv is None:
        return _mlp(p, "object_encoder", glimpse.flatten(1))
    h = glimpse
    for i, (_, k, s) in enumerate(cfg.object_conv):
        h = F.relu(F.conv2d(h, p[f"object_encoder.conv.conv_{i}.weight"], p[f"object_encoder.conv.conv_{i}.bias"], stride=s))
    return F.linear(h.flatten(1), p["object_encoder.out.weight"], p["object_encoder.out.bias"])


def object_decoder(p, z, cfg: "OracleConfig"):
    """models.py:165 (MLP) / models.py:633-665 (conv sketch).  z [N,A] -> logits [N,P,P,C+1]."""
    px, C = cfg.object_shape[0], cfg.image_shape[0]
    if cfg.object_conv is None:
        return _mlp(p, "object_decoder", z).view(-1, px, px, C + 1)
    sizes, chans = [px], [C]
    for f, k, s in cfg.object_conv:
        sizes.append((sizes[-1] - k) // s + 1)
        chans.append(f)
    n = len(cfg.object_conv)
    h = F.linear(z, p["object_decoder.inp.weight"], p["object_decoder.inp.bias"]).view(-1, chans[-1], sizes[-1], sizes[-1])
    for i in range(n):
        _, k, s = cfg.object_conv[n - 1 - i]
        op = sizes[n - 1 - i] - ((sizes[n - i] - 1) * s + k)
        h = F.conv_transpose2d(h, p[f"object_decoder.conv.conv_transposed_{i}.weight"], p[f"object_decoder.conv.conv_transposed_{i}.bias"],
                               stride=s, output_padding=op)
        if i + 1 < n:
            h = F.relu(h)
    return h.permute(0, 2, 3, 1)


def latent_to_mean_std(lat):
    """modules.py:167-176."""
    mean, log_std = torch.chunk(lat, 2, dim=-1)
    return mean, torch.sigmoid(log_std.clamp(-10, 10)) * 2


def clamped_sigmoid(x, analytical=False):
    """modules.py:178-189."""
    if analytical:
        return 1 / ((-x).exp() + 1)
    return torch.sigmoid(torch.clamp(x, -10, 10))


def freeze(f, *ts):
    """models.py:413-429: value-preserving, scales the gradient by (1-f)."""
    out = [f * t.detach() + (1 - f) * t for t in ts]
    return out[0] if len(out) == 1 else out


# --------------------------------------------------------------------------------------
# Spatial transformer, restated explicitly (modules.py:216-273; torch affine_grid /
# grid_sample semantics)
# --------------------------------------------------------------------------------------
def _base_coords(n: int, align_corners: bool) -> torch.Tensor:
    i = torch.arange(n, dtype=torch.float32)
    if align_corners:
        return 2 * i / (n - 1) - 1 if n > 1 else torch.zeros(1)
    return (2 * i + 1) / n - 1


def _unnormalize(g, size: int, align_corners: bool):
    if align_corners:
        return (g + 1) / 2 * (size - 1)
    return ((g + 1) * size - 1) / 2


def stn_theta(z_where: torch.Tensor, inverse: bool, inverse_mode: str = "lu"):
    """theta rows as (a_x, t_x, a_y, t_y): src_x = a_x*X + t_x, src_y = a_y*Y + t_y."""
    xt, yt, xs, ys = z_where.unbind(-1)
    tx, ty = xt * 2 - 1, yt * 2 - 1
    if not inverse:
        return xs, tx, ys, ty
    if inverse_mode == "closed":
        return 1 / xs, -tx / xs, 1 / ys, -ty / ys
    n = z_where.shape[0]
    t = torch.zeros(n, 3, 3, dtype=z_where.dtype)
    t[:, 0, 0], t[:, 1, 1], t[:, 0, 2], t[:, 1, 2], t[:, 2, 2] = xs, ys, tx, ty, 1.0
    ti = t.inverse()
    return ti[:, 0, 0], ti[:, 0, 2], ti[:, 1, 1], ti[:, 1, 2]


def stn(image: torch.Tensor, z_where: torch.Tensor, out_hw: Tuple[int, int], inverse=False,
        align_corners=False, inverse_mode="lu", fast=False) -> torch.Tensor:
    """image [N,C,Hs,Ws], z_where [N,4]=(xt,yt,xs,ys) -> [N,C,Ho,Wo].
    forward: border padding; inverse: zeros padding (modules.py:268)."""
    N, C, Hs, Ws = image.shape
    Ho, Wo = out_hw
    ax, tx, ay, ty = stn_theta(z_where, inverse, inverse_mode)
    if fast:  # same maths through torch's fused ops (used only for the timed cpu_baseline)
        theta = torch.zeros(N, 2, 3, dtype=image.dtype)
        theta[:, 0, 0], theta[:, 0, 2], theta[:, 1, 1], theta[:, 1, 2] = ax, tx, ay, ty
        grid = F.affine_grid(theta, [N, C, Ho, Wo], align_corners=align_corners)
        return F.grid_sample(image, grid, padding_mode="zeros" if inverse else "border",
                             align_corners=align_corners)
    X = _base_coords(Wo, align_corners)
    Y = _base_coords(Ho, align_corners)
    gx = ax[:, None] * X[None, :] + tx[:, None]            # [N,Wo]
    gy = ay[:, None] * Y[None, :] + ty[:, None]            # [N,Ho]
    ix = _unnormalize(gx, Ws, align_corners)
    iy = _unnormalize(gy, Hs, align_corners)
    if not inverse:  # border: clip the coordinate (gradient 0 outside)
        ix = ix.clamp(0, Ws - 1)
        iy = iy.clamp(0, Hs - 1)
    x0 = torch.floor(ix.detach())
    y0 = torch.floor(iy.detach())
    wx1, wy1 = ix - x0, iy - y0
    wx0, wy0 = 1 - wx1, 1 - wy1
    x0l, y0l = x0.long(), y0.long()

    def tap(yl, xl):
        vy = (yl >= 0) & (yl < Hs)
        vx = (xl >= 0) & (xl < Ws)
        yc, xc = yl.clamp(0, Hs - 1), xl.clamp(0, Ws - 1)
        # gather rows then columns: image[n, c, yc[n,i], xc[n,j]]
        rows = image.gather(2, yc[:, None, :, None].expand(N, C, Ho, Ws))
        vals = rows.gather(3, xc[:, None, None, :].expand(N, C, Ho, Wo))
        mask = (vy[:, :, None] & vx[:, None, :]).to(image.dtype)
        return vals * mask[:, None]

    out = (tap(y0l, x0l) * (wy0[:, :, None] * wx0[:, None, :])[:, None]
           + tap(y0l, x0l + 1) * (wy0[:, :, None] * wx1[:, None, :])[:, None]
           + tap(y0l + 1, x0l) * (wy1[:, :, None] * wx0[:, None, :])[:, None]
           + tap(y0l + 1, x0l + 1) * (wy1[:, :, None] * wx1[:, None, :])[:, None])
    return out


# --------------------------------------------------------------------------------------
# The model
# --------------------------------------------------------------------------------------
def encode_cells(p, x, feat, noise, wheel, cfg: OracleConfig, fast=False, taps: Optional[dict] = None):
    """Sequential per-cell loop, models.py:68-117 with helpers :292-450.  ``taps`` (a dict) receives, per name, the list over cells (row-major)
    of the four networks' RAW outputs with ``retain_grad()`` set -- ``box_lat`` [B,8], ``enc_out`` [B,2A], ``depth_lat`` [B,2], ``pres_logit``
    [B,1] -- so that a test can read the reference's per-cell latent gradients after ``loss.backward()`` (the HIP backward keeps the same
    quantities per cell in its row buffers: spair_export_map 100.. / 200..)."""
    B = x.shape[0]
    _, I_h, I_w = cfg.image_shape
    G = feat.shape[-1]
    _, _, _, cell_px, _ = backbone_geometry(I_h, cfg.conv_kernels, cfg.conv_strides)
    A = cfg.n_attr
    edge = p["virtual_edge_element"][None, :].expand(B, -1)
    rec = {}
    names = ["cy_logit", "cx_logit", "height_logit", "width_logit", "attr", "depth_logit"]
    means = {n: [None] * (G * G) for n in names}
    sigmas = {n: [None] * (G * G) for n in names}
    z_where, z_attr, z_depth, z_pres = [], [], [], []
    for h in range(G):
        for w in range(G):
            ci = h * G + w
            cell_feat = feat[:, :, h, w]
            # context: UL, U, UR, L (models.py:292-320)
            # rows h-L..h, columns w-L..w+L in row-major order, without the current cell and its right-hand side (L = 1: UL, U, UR, L)
            Lb = cfg.n_lookback
            ctx = torch.cat([rec.get((h + dh, w + dw), edge) if 0 <= w + dw < G else edge
                             for dh in range(-Lb, 1) for dw in range(-Lb, (Lb if dh else -1) + 1)], dim=-1)
            # --- z_where (models.py:76-79,322-381)
            lat, passthru = _mlp(p, "box_network", torch.cat([cell_feat, ctx], -1), multi=True)
            if taps is not None and lat.requires_grad:
                lat.retain_grad(); taps.setdefault("box_lat", []).append(lat)
            mean, std = latent_to_mean_std(lat)
            mean, std = freeze(wheel, mean, std)
            eps = noise["eps_box"][:, :, h, w]
            zs = mean + std * eps                               # (cy, cx, height, width)
            for k, n in enumerate(names[:4]):
                means[n][ci], sigmas[n][ci] = mean[:, k:k + 1], std[:, k:k + 1]
            cell_y = (cfg.max_yx - cfg.min_yx) * clamped_sigmoid(zs[:, 0:1]) + cfg.min_yx
            cell_x = (cfg.max_yx - cfg.min_yx) * clamped_sigmoid(zs[:, 1:2]) + cfg.min_yx
            height = (cfg.max_hw - cfg.min_hw) * clamped_sigmoid(zs[:, 2:3]) + cfg.min_hw
            width = (cfg.max_hw - cfg.min_hw) * clamped_sigmoid(zs[:, 3:4]) + cfg.min_hw
            box = torch.cat([cell_x, cell_y, width, height], -1)
            ys = height * cfg.anchorbox[0] / I_h
            xs = width * cfg.anchorbox[0] / I_w
            yt = (cell_px / I_h) * (cell_y + h)
            xt = (cell_px / I_w) * (cell_x + w)
            nbox = torch.cat([xt, yt, xs, ys], -1)
            # --- z_what (models.py:82-85,383-391)
            glimpse = stn(x, nbox, tuple(cfg.object_shape), inverse=False,
                          align_corners=cfg.align_corners, fast=fast)
            enc = object_encoder(p, glimpse, cfg)
            if taps is not None and enc.requires_grad:
                enc.retain_grad(); taps.setdefault("enc_out", []).append(enc)
            a_mean, a_std = latent_to_mean_std(enc)
            attr = a_mean + a_std * noise["eps_attr"][:, :, h, w]
            means["attr"][ci], sigmas["attr"][ci] = a_mean, a_std
            # --- z_depth (models.py:88-97)
            dlat, passthru = _mlp(p, "z_network", torch.cat([cell_feat, ctx, passthru, box, attr], 1), multi=True)
            if taps is not None and dlat.requires_grad:
                dlat.retain_grad(); taps.setdefault("depth_lat", []).append(dlat)
            d_mean, d_std = latent_to_mean_std(dlat)
            d_mean, d_std = freeze(wheel, d_mean, d_std)
            d_logit = d_mean + d_std * noise["eps_depth"][:, :, h, w]
            means["depth_logit"][ci], sigmas["depth_logit"][ci] = d_mean, d_std
            depth = 4 * clamped_sigmoid(d_logit)
            # --- z_pres (models.py:100-102,393-411)
            logit = _mlp(p, "obj_network", torch.cat([cell_feat, ctx, passthru, box, attr, depth], 1))
            if taps is not None and logit.requires_grad:
                logit.retain_grad(); taps.setdefault("pres_logit", []).append(logit)
            logit = freeze(wheel, logit)
            log_odds = torch.clamp(logit, -10.0, 10.0)
            u = noise["u_pres"][:, :, h, w]
            pres = torch.sigmoid(log_odds + torch.log(u + 10e-10) - torch.log(1.0 - u + 10e-10))
            rec[(h, w)] = torch.cat([box, attr, depth, pres], -1)
            z_where.append(nbox), z_attr.append(attr), z_depth.append(depth), z_pres.append(pres)

    def to_map(lst):  # list over cells of [B,C] -> [B,C,G,G]
        return torch.stack(lst, dim=-1).view(B, -1, G, G)

    dist = {n: (to_map(means[n]), to_map(sigmas[n])) for n in names}
    return to_map(z_where), to_map(z_attr), to_map(z_depth), to_map(z_pres), dist


def compute_kl(dist, z_pres, global_step, cfg: OracleConfig, p_z_out: Optional[list] = None):
    """models.py:169-262.  ``p_z_out`` (a list) receives the [B,1,G,G] map of p(z_pres = 1 | counts so far) (models.py:217), which the
    reference does not keep: the count-prior tests compare the kernel's saved map with it cell by cell."""
    B, _, G, _ = z_pres.shape
    HW = G * G
    KL = {}
    for name, (mu, sigma) in dist.items():
        m, s = cfg.priors[name]
        var_ratio = (sigma / s) ** 2
        t1 = ((mu - m) / s) ** 2
        KL[name] = z_pres * (0.5 * (var_ratio + t1 - 1 - var_ratio.log()))
    support = torch.arange(HW + 1, dtype=z_pres.dtype)      # (fp32 as the reference; a float64 z_pres runs the recursion in float64)
    log_odds = exponential_decay(global_step, **cfg.count_prior)
    prob = 1 / ((-log_odds).exp() + 1)
    cd = (1 - prob) * (prob ** support)
    cd = (cd / cd.sum()).repeat(B, 1)
    count = torch.zeros(B, 1, dtype=z_pres.dtype)
    cells, pzs = [], []
    zp = z_pres.reshape(B, HW)
    for i in range(HW):
        q = torch.clamp(support - count, min=0.0, max=float(HW - i)) / (HW - i)
        p_z = (cd * q).sum(1, keepdim=True)
        pr = zp[:, i:i + 1]
        pzs.append(p_z.detach())
        cells.append(pr * (torch.log(pr + 1e-9) - torch.log(p_z + 1e-9))
                     + (1 - pr) * (torch.log(1 - pr + 1e-9) - torch.log(1 - p_z + 1e-9)))
        s = torch.round(pr.detach())
        cd1 = (s * q + (1 - s) * (1 - q)) * cd
        cd = cd1 / cd1.sum(1, keepdim=True).clamp(min=1e-6)
        count = count + s
    KL["pres_dist"] = torch.cat(cells, 1).view(B, 1, G, G)
    if p_z_out is not None:
        p_z_out.append(torch.cat(pzs, 1).view(B, 1, G, G))
    return KL


def decode_sprites(p, z_attr, z_depth, z_pres, cfg: OracleConfig):
    """models.py:468-504.  Returns objects [N,P,P,C+2] = (colour.., alpha*pres, importance)."""
    B, A, G, _ = z_attr.shape
    px = cfg.object_shape[0]
    C = cfg.image_shape[0]
    dec_in = z_attr.permute(0, 2, 3, 1).reshape(-1, A)
    logits = object_decoder(p, dec_in, cfg)
    colour = clamped_sigmoid(logits[..., :-1] * cfg.obj_logit_scale, analytical=True)
    alpha = clamped_sigmoid(logits[..., -1:] * cfg.alpha_logit_scale + cfg.alpha_logit_bias, analytical=True)
    alpha = alpha * z_pres.reshape(-1, 1, 1, 1)
    importance = torch.clamp(alpha * z_depth.reshape(-1, 1, 1, 1), min=0.01)
    return torch.cat([colour, alpha, importance], -1)


def render(p, z_attr, z_where, z_depth, z_pres, cfg: OracleConfig, fast=False):
    """models.py:452-542."""
    B, _, G, _ = z_where.shape
    C, I_h, I_w = cfg.image_shape
    objects = decode_sprites(p, z_attr, z_depth, z_pres, cfg).permute(0, 3, 1, 2)
    zw = z_where.permute(0, 2, 3, 1).reshape(-1, 4)
    t = stn(objects, zw, (I_h, I_w), inverse=True, align_corners=cfg.align_corners,
            inverse_mode=cfg.inverse_mode, fast=fast).view(B, G * G, C + 2, I_h, I_w)
    colour, alpha, imp = t[:, :, :C], t[:, :, C:C + 1], t[:, :, C + 1:C + 2] + 1e-9
    img = alpha * colour
    imp = imp / imp.sum(dim=1, keepdim=True)
    return torch.clamp((img * imp).sum(dim=1), min=0, max=1)


def forward(p: Dict[str, torch.Tensor], x: torch.Tensor, global_step: int,
            noise: Dict[str, torch.Tensor], cfg: OracleConfig, fast: bool = False,
            kl_scale: Optional[float] = None, taps: Optional[dict] = None):
    """models.py:35-131 + _build_loss :544-563.  ``kl_scale`` (default 1/B) is the factor
    on the summed KL maps -- 1/(B*world_size) reproduces the sharded loss of SURVEY §8(e)."""
    B = x.shape[0]
    feat = backbone_forward(p, x, cfg)
    wheel = exponential_decay(global_step, **cfg.wheel)
    z_where, z_attr, z_depth, z_pres, dist = encode_cells(p, x, feat, noise, wheel, cfg, fast=fast, taps=taps)
    kl = compute_kl(dist, z_pres, global_step, cfg)
    recon = render(p, z_attr, z_where, z_depth, z_pres, cfg, fast=fast)
    recon_loss = F.binary_cross_entropy(recon, x, reduction="sum")
    scale = (1.0 / B) if kl_scale is None else kl_scale
    terms = {"recon": recon_loss}
    kl_loss = 0
    for name, z_kl in kl.items():
        terms["kl_" + name] = z_kl.sum() * scale
        kl_loss = kl_loss + terms["kl_" + name]
    loss = recon_loss + cfg.vae_beta * kl_loss
    return dict(loss=loss, terms=terms, recon_x=recon, z_where=z_where, z_pres=z_pres,
                z_depth=z_depth, z_attr=z_attr, dist=dist, feat=feat, wheel=wheel, kl=kl)


def adam_step(params, grads, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (train.py:44): in-place on the given tensors."""
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    for k in params:
        if grads.get(k) is None:
            continue
        g = grads[k]
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m[k], denom, value=-lr / bc1)
