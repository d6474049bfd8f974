"""Deterministic (numpy-Generator) inputs shared by the golden-vector generator and the tests.

Nothing here touches the reference.  numpy's ``default_rng`` (PCG64) streams are stable
across numpy versions and platforms, so weights / noise / images can be regenerated on the
GPU box from a seed instead of being shipped as megabytes of fixture.

Shapes and key names follow the reference's ``state_dict`` (SURVEY.md §8(b);
/root/reference/spair/models.py:133-167, modules.py:43-66,124-165).
"""
import math

import numpy as np

N_BACKBONE_FEATURES = 100
N_PASSTHROUGH = 100
N_ATTR = 50
OBJ_PX = 28
CONTEXT_DIM = 4 * (4 + N_ATTR + 1 + 1)  # 224 (N_LOOKBACK = 1)


def context_dim(lookback=1):
    """models.py:26: ((2L+1)^2 // 2) records of 4 + A + 2."""
    return (2 * lookback + 1) ** 2 // 2 * (4 + N_ATTR + 1 + 1)


def param_shapes(in_chan=1, conv_kernels=(4, 4, 4, 1, 1, 1), filters=128, lookback=1):
    """Ordered {key: shape} for every tensor of the reference state_dict."""
    s = {}
    s["virtual_edge_element"] = (4 + N_ATTR + 2,)
    prev = in_chan
    for i, k in enumerate(conv_kernels):
        s[f"backbone.net.conv_{i}.weight"] = (filters, prev, k, k)
        s[f"backbone.net.conv_{i}.bias"] = (filters,)
        prev = filters
    s["backbone.net.conv_out.weight"] = (N_BACKBONE_FEATURES, prev, 1, 1)
    s["backbone.net.conv_out.bias"] = (N_BACKBONE_FEATURES,)

    def mlp(prefix, n_in, hidden, outs, multi):
        p = n_in
        body = prefix + (".body" if multi else "")
        for i, h in enumerate(hidden):
            s[f"{body}.dense{i}.weight"] = (h, p)
            s[f"{body}.dense{i}.bias"] = (h,)
            p = h
        if multi:
            for i, o in enumerate(outs):
                s[f"{prefix}.output_layers.{i}.weight"] = (o, p)
                s[f"{prefix}.output_layers.{i}.bias"] = (o,)
        else:
            s[f"{prefix}.out.weight"] = (outs, p)
            s[f"{prefix}.out.bias"] = (outs,)

    CONTEXT_DIM = context_dim(lookback)
    box_in = N_BACKBONE_FEATURES + CONTEXT_DIM
    mlp("box_network", box_in, (100, 100), (8, N_PASSTHROUGH), True)
    mlp("object_encoder", OBJ_PX * OBJ_PX * in_chan, (256, 128), 2 * N_ATTR, False)
    z_in = 4 + N_ATTR + N_PASSTHROUGH + CONTEXT_DIM + N_BACKBONE_FEATURES
    mlp("z_network", z_in, (100, 100), (2, N_PASSTHROUGH), True)
    mlp("obj_network", z_in + 1, (100, 100), 1, False)
    mlp("object_decoder", N_ATTR, (128, 256), OBJ_PX * OBJ_PX * (in_chan + 1), False)
    s["attn.gamma"] = (1,)
    for nm, o in (("query", 55 // 8), ("key", 55 // 8), ("value", 55)):
        s[f"attn.{nm}_conv.weight"] = (o, 55, 1, 1)
        s[f"attn.{nm}_conv.bias"] = (o,)
    return s


def make_weights(seed, scale=1.0, in_chan=1, lookback=1):
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) * scale per tensor (PyTorch-default-like
    magnitude), float32.  Returns {key: np.ndarray}."""
    rng = np.random.default_rng(seed)
    out = {}
    shapes = param_shapes(in_chan, lookback=lookback)
    for key, shp in shapes.items():
        if key == "virtual_edge_element":
            t = rng.standard_normal(shp).astype(np.float32)
            sig = lambda v: 1.0 / (1.0 + np.exp(-v))
            t[:4] = sig(t[:4])
            t[-2:] = sig(t[-2:])
            out[key] = t.astype(np.float32)
            continue
        if key == "attn.gamma":
            out[key] = np.zeros(shp, np.float32)
            continue
        if key.endswith(".weight"):
            fan_in = int(np.prod(shp[1:]))
        else:
            fan_in = int(np.prod(shapes[key[:-4] + "weight"][1:]))
        bound = scale / math.sqrt(fan_in)
        out[key] = rng.uniform(-bound, bound, size=shp).astype(np.float32)
    return out


def make_noise(seed, B, G):
    """The 7 per-cell draws of the reference (models.py:333-336,84,95,402-403), laid out
    as maps.  eps_box channel order = draw order (cy, cx, height, width)."""
    rng = np.random.default_rng(seed)
    return dict(
        eps_box=rng.standard_normal((B, 4, G, G)).astype(np.float32),
        eps_attr=rng.standard_normal((B, N_ATTR, G, G)).astype(np.float32),
        eps_depth=rng.standard_normal((B, 1, G, G)).astype(np.float32),
        u_pres=rng.uniform(0.0, 1.0, (B, 1, G, G)).astype(np.float32),
    )


def make_image(seed, B, I, max_objects, in_chan=1):
    """Small stand-in for scattered MNIST: k ~ U{0..max_objects} anti-aliased stroke
    blobs of 14..28 px on black, max-composited, values in [0,1]."""
    rng = np.random.default_rng(seed)
    img = np.zeros((B, in_chan, I, I), np.float32)
    yy, xx = np.mgrid[0:I, 0:I].astype(np.float32)
    for b in range(B):
        k = int(rng.integers(0, max_objects + 1))
        for _ in range(k):
            size = float(rng.uniform(14, 28))
            cy, cx = rng.uniform(size / 2, I - size / 2, 2)
            # a "stroke": a ring segment or a bar, softly anti-aliased
            if rng.uniform() < 0.5:
                r = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
                glyph = np.clip(1.5 - np.abs(r - size * 0.3) / 1.5, 0, 1)
            else:
                ang = rng.uniform(0, np.pi)
                d = np.abs((yy - cy) * np.cos(ang) - (xx - cx) * np.sin(ang))
                along = np.abs((yy - cy) * np.sin(ang) + (xx - cx) * np.cos(ang))
                glyph = np.clip(1.5 - d / 1.5, 0, 1) * (along < size * 0.45)
            # colour images: one random colour per object (drawn only then: the greyscale streams are unchanged)
            col = rng.uniform(0.25, 1.0, in_chan).astype(np.float32) if in_chan > 1 else np.ones(1, np.float32)
            for c in range(in_chan):
                img[b, c] = np.maximum(img[b, c], col[c] * glyph.astype(np.float32))
    return img


# name -> dict(I, strides, B, step, wseed, wscale, max_objects)
CASES = {
    # BASELINE.json configs[0]: 48x48, 6x6 grid, batch 16 (wheel on / off / sharp count prior)
    "c1_b16_step1": dict(I=48, strides=(2, 2, 2, 1, 1, 1), B=16, step=1, wseed=11, wscale=1.0, max_objects=3),
    "c1_b8_step1001": dict(I=48, strides=(2, 2, 2, 1, 1, 1), B=8, step=1001, wseed=11, wscale=1.0, max_objects=3),
    "c1_b8_step7001": dict(I=48, strides=(2, 2, 2, 1, 1, 1), B=8, step=7001, wseed=12, wscale=2.0, max_objects=3),
    # configs[1] geometry (128x128, 16x16 grid) as a B=2 slice
    "c2_b2_step1001": dict(I=128, strides=(2, 2, 2, 1, 1, 1), B=2, step=1001, wseed=13, wscale=1.0, max_objects=11),
    # the reference's own default topology (strides 3,2,2 -> 11x11 grid, config.py:7-14)
    "ref_default_b2_step1001": dict(I=128, strides=(3, 2, 2, 1, 1, 1), B=2, step=1001, wseed=14, wscale=1.5, max_objects=11),
    # configs[3] geometry (256x256, 32x32 grid) as a B=1 slice
    "c4_b1_step1001": dict(I=256, strides=(2, 2, 2, 1, 1, 1), B=1, step=1001, wseed=15, wscale=1.0, max_objects=11),
}
# N_LOOKBACK = 2 (config.py:31; 12 context neighbours, models.py:292-320): kept apart from CASES -- the fused bf16 kernels are built for
# N_LOOKBACK = 1, these run on the per-wavefront launches (tests/test_lookback_gpu.py)
LOOKBACK_CASES = {
    "lb2_c1_b4_step1001": dict(I=48, strides=(2, 2, 2, 1, 1, 1), B=4, step=1001, wseed=16, wscale=1.0, max_objects=3, lookback=2),
    "lb2_i80_b2_step1": dict(I=80, strides=(2, 2, 2, 1, 1, 1), B=2, step=1, wseed=17, wscale=1.5, max_objects=5, lookback=2),
    "lb3_c1_b3_step7001": dict(I=48, strides=(2, 2, 2, 1, 1, 1), B=3, step=7001, wseed=18, wscale=1.0, max_objects=3, lookback=3),
}
# colour images (config.py:4 INPUT_IMAGE_SHAPE[0] = 3; models.py:150,163,480,524, modules.py:24,239): the fp32 per-wavefront step with the
# generic-channel renderer (tests/test_rgb_gpu.py)
RGB_CASES = {
    "rgb_c1_b4_step1001": dict(I=48, strides=(2, 2, 2, 1, 1, 1), B=4, step=1001, wseed=21, wscale=1.0, max_objects=3, in_chan=3),
    "rgb_i80_b2_step1": dict(I=80, strides=(2, 2, 2, 1, 1, 1), B=2, step=1, wseed=22, wscale=1.5, max_objects=5, in_chan=3),
}


def all_cases():
    d = dict(CASES)
    d.update(LOOKBACK_CASES)
    d.update(RGB_CASES)
    return d


def grid_side(I, strides, kernels=(4, 4, 4, 1, 1, 1)):
    cell = 1
    for s in strides:
        cell *= s
    return int(math.ceil(I / cell))
