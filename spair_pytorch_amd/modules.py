"""Host-side mirror of /root/reference/spair/modules.py for the hot path.

Same public names (``Backbone``, ``build_MLP``, ``SequentialMultipleOutput``,
``latent_to_mean_std``, ``clamped_sigmoid``, ``exponential_decay``, ``stn``, ``to_C_H_W``,
``to_H_W_C``, ``safe_log``).  The nn.Modules here are *parameter containers* with the
reference's construction order (so ``torch.manual_seed`` reproduces its initial weights
bit for bit) and state_dict keys; the arithmetic of the training step runs
in libspair_hip.so (see models.py).  ``stn`` is the one helper that is itself a hot-path
operator and dispatches to the HIP kernels directly.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
from torch import nn
from torch.nn import Conv2d, Linear, Module, ModuleList, ReLU, Sequential

from . import config as cfg


def backbone_geometry(image_hw, topology):
    """Receptive-field padding arithmetic of modules.py:68-105.
    Returns (pad_pre, pad_post, n_grid_cells, grid_cell_size, rf_size)."""
    j, r = 1, 1
    for layer in topology:
        k, s = int(layer['kernel_size']), int(layer['stride'])
        r = r + (k - 1) * j
        j = j * s
    pre = int(math.floor(r / 2 - j / 2))
    n = int(math.ceil(image_hw / j))
    post = r + (n - 1) * j - image_hw - pre
    return pre, post, n, j, r


def _hip_linear(x, weight, bias, relu=False):
    """y = relu?(x @ weight.T + bias) through spair_gemm_nt (exact fp32 MFMA mode): [M,K] x [N,K] -> [M,N]."""
    import ctypes
    from . import _lib as L
    if not x.is_cuda:
        raise L.SpairHipError("tensors must live on the MI355X; there is no CPU path")
    M, K = x.shape
    N = weight.shape[0]
    Kp = (K + 3) // 4 * 4
    A = x.detach().float().contiguous()
    W = weight.detach().float().contiguous()
    if Kp != K:      # the GEMM wants K % 4 == 0: zero columns change nothing
        A = torch.nn.functional.pad(A, (0, Kp - K))
        W = torch.nn.functional.pad(W, (0, Kp - K))
    out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    b = bias.detach().float().contiguous() if bias is not None else None
    L.check(L.lib().spair_gemm_nt(L.ptr(A), Kp, L.ptr(W), Kp, L.ptr(out), N, M, N, Kp, L.ptr(b), None, 0, int(relu), 0, 0, L.stream()),
            "spair_gemm_nt")
    return out


def _hip_conv_nhwc(x, weight, bias, stride, relu):
    """NHWC valid convolution as the implicit GEMM of spair_gemm_nt_conv (fp32 mode): x [B,H,W,C], weight OIHW -> [B,Ho,Wo,O]."""
    import ctypes
    from . import _lib as L
    B, H, W_, C = x.shape
    O, Ci, kh, kw = weight.shape
    assert Ci == C
    Cp = (C + 3) // 4 * 4
    xw = x.detach().float()
    w = weight.detach().float().permute(0, 2, 3, 1)          # [O, kh, kw, Ci]: the K order of the gather
    if Cp != C:
        xw = torch.nn.functional.pad(xw, (0, Cp - C))
        w = torch.nn.functional.pad(w, (0, Cp - C))
    xw, w = xw.contiguous(), w.contiguous().view(O, kh * kw * Cp)
    Ho, Wo = (H - kh) // stride + 1, (W_ - kw) // stride + 1
    out = torch.empty(B, Ho, Wo, O, device=x.device, dtype=torch.float32)
    conv = (ctypes.c_int * 13)(H, W_, Cp, Ho, Wo, kh, kw, stride, stride, 1, 1, 0, 0)
    K = kh * kw * Cp
    b = bias.detach().float().contiguous()
    L.check(L.lib().spair_gemm_nt_conv(L.ptr(xw), conv, L.ptr(w), K, L.ptr(out), O, B * Ho * Wo, O, K, L.ptr(b), None, 0, int(relu), 0, None, 0,
                                       L.stream()), "spair_gemm_nt_conv")
    return out


class Backbone(Module):
    """modules.py:12-111.  Keeps ``net`` (conv_0..conv_out), ``padding``, ``n_grid_cells``,
    ``grid_cell_size``; ``forward`` is executed by the HIP engine, not by this module."""

    def __init__(self, input_shape, n_out_channels, topology=None, internal_activation=ReLU):
        super().__init__()
        self.topology = [dict(t) for t in (topology if topology is not None else cfg.DEFAULT_BACKBONE_TOPOLOGY)]
        self.input_shape = input_shape
        n_prev = input_shape[0]
        net = OrderedDict()
        f = n_prev
        for i, layer in enumerate(self.topology):
            f = layer.get('filters', layer.get('out_channels'))
            net['conv_%d' % i] = Conv2d(n_prev, f, kernel_size=layer['kernel_size'], stride=layer['stride'])
            net['act_%d' % i] = internal_activation()
            n_prev = f
        net['conv_out'] = Conv2d(in_channels=f, out_channels=n_out_channels, kernel_size=1, stride=1)
        self.net = Sequential(net)
        pre, post, n, cell, _ = backbone_geometry(input_shape[-1], self.topology)
        self.pad_pre, self.pad_post = pre, post
        self.padding = nn.ZeroPad2d((pre, post, pre, post))
        self.n_grid_cells = np.array([n, n])
        self.grid_cell_size = np.array([cell, cell])
        self.n_out_channels = n_out_channels

    def compute_output_shape(self):
        """modules.py:32-41 -- the reference probes with a random image, which consumes the
        global RNG; the draw is replicated so later initialisers see the same stream."""
        torch.rand(1, *cfg.INPUT_IMAGE_SHAPE)
        n = int(self.n_grid_cells[0])
        return torch.Size([self.n_out_channels, n, n])

    def forward(self, x):
        """modules.py:107-111 on its own: pad, then the conv stack as implicit GEMMs (spair_gemm_nt_conv, exact fp32 MFMA mode).
        Inference helper (no autograd graph): the training step runs the backbone inside the fused engine (SPAIR.forward)."""
        from . import _lib as L
        if not x.is_cuda:
            raise L.SpairHipError("Backbone.forward: the input must live on the MI355X; there is no CPU path")
        h = self.padding(x.detach().float()).permute(0, 2, 3, 1).contiguous()       # NHWC
        convs = [m for m in self.net if isinstance(m, Conv2d)]
        for i, conv in enumerate(convs):
            h = _hip_conv_nhwc(h, conv.weight, conv.bias, int(conv.stride[0]), relu=(i < len(convs) - 1))
        return h.permute(0, 3, 1, 2).contiguous()


class SequentialMultipleOutput(Module):
    """modules.py:276-284 (container only)."""

    def __init__(self, input, outputs):
        super().__init__()
        self.body = Sequential(input)
        self.output_layers = ModuleList(list(outputs.values()))

    def forward(self, x):
        """modules.py:282-284: the body, then a GENERATOR over the heads' outputs.  Linear layers run through spair_gemm_nt
        (fp32 mode); inference helper, no autograd graph (training runs these nets inside the fused per-cell kernels)."""
        h = hip_mlp_forward(self.body, x)
        return (_hip_linear(h, layer.weight, layer.bias) for layer in self.output_layers)


def hip_mlp_forward(seq, x):
    """Linear / ReLU stack of build_MLP through the HIP GEMM (a following ReLU is fused into the GEMM epilogue)."""
    mods = list(seq)
    h = x
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, Linear):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], ReLU)
            h = _hip_linear(h, m.weight, m.bias, relu=relu)
            i += 2 if relu else 1
        elif isinstance(m, ReLU):
            h = torch.relu(h)
            i += 1
        else:
            h = m(h)
            i += 1
    return h


def build_MLP(n_in, output=None, multiple_output=None, hidden_layers=None, activation=None, internal_activation=ReLU):
    """modules.py:124-165: same layer names and construction order."""
    hidden_layers = cfg.DEFAULT_MLP_TOPOLOGY if hidden_layers is None else hidden_layers
    n_prev = n_in
    net = OrderedDict()
    for i, h in enumerate(hidden_layers):
        net['dense%d' % i] = Linear(n_prev, h)
        net['relu%d' % i] = internal_activation()
        n_prev = h
    if output is not None:
        net['out'] = Linear(n_prev, output)
        if activation is not None:
            net['act'] = activation()
        return Sequential(net)
    elif multiple_output is not None:
        out_net = OrderedDict()
        for i, out in enumerate(multiple_output):
            out_net['out_%d' % i] = Linear(n_prev, out)
        return SequentialMultipleOutput(net, out_net)
    raise AssertionError('Unknown output type')


def latent_to_mean_std(latent_var):
    """modules.py:167-176."""
    mean, log_std = torch.chunk(latent_var, 2, dim=-1)
    return mean, torch.sigmoid(log_std.clamp(-10, 10)) * 2


def clamped_sigmoid(logit, use_analytical=False):
    """modules.py:178-189."""
    if use_analytical:
        return 1 / ((-logit).exp() + 1)
    return torch.sigmoid(torch.clamp(logit, -10, 10))


def exponential_decay(global_step, device=None, start=0.0, end=0.0, decay_rate=0.0, decay_step=1.0, staircase=False,
                      log_space=False):
    """modules.py:191-213 with the same fp32 tensor arithmetic, but on the host (CPU tensor) so a
    training step needs no device round trip; returns a python float."""
    gs = torch.tensor(global_step, dtype=torch.float32)
    t = gs // decay_step if staircase else gs / decay_step
    value = (start - end) * (decay_rate ** t) + end
    if log_space:
        value = (value + 1e-6).log()
    return float(value)


class _StnFn(torch.autograd.Function):
    """stn() as a differentiable operator over the HIP kernels: forward direction = border-padded glimpse gather (gradient wrt
    z_where; the image is data), inverse = zero-padded placement of sprites on the canvas (gradients wrt sprites and z_where)."""

    @staticmethod
    def forward(ctx, image, z_where, size, inverse):
        from . import _lib as L
        N, C, H, W = image.shape
        img = image.detach().contiguous().float()
        zw = z_where.detach().contiguous().float()
        ac = int(cfg.ALIGN_CORNERS)
        if inverse:
            out = torch.empty(N, C, size, size, device=img.device, dtype=torch.float32)
            L.check(L.lib().spair_stn_inverse_fwd(L.ptr(img), L.ptr(zw), L.ptr(out), N, C, H, size, ac, L.stream()), "spair_stn_inverse_fwd")
        else:
            out = torch.empty(N, C * size * size, device=img.device, dtype=torch.float32)
            L.check(L.lib().spair_stn_glimpse_fwd(L.ptr(img), L.ptr(zw), N, L.ptr(out), C * size * size, N, C, H, size, ac, L.stream()),
                    "spair_stn_glimpse_fwd")
            out = out.view(N, C, size, size)
        ctx.save_for_backward(img, zw)
        ctx.size, ctx.inverse = size, inverse
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib as L
        img, zw = ctx.saved_tensors
        N, C, H, W = img.shape
        g = g.contiguous().float()
        ac = int(cfg.ALIGN_CORNERS)
        dzw = torch.zeros(N, 4, device=img.device, dtype=torch.float32)
        if ctx.inverse:
            dimg = torch.zeros_like(img)
            L.check(L.lib().spair_stn_inverse_bwd(L.ptr(img), L.ptr(zw), L.ptr(g), L.ptr(dimg), L.ptr(dzw), N, C, H, ctx.size, ac, L.stream()),
                    "spair_stn_inverse_bwd")
            return dimg, dzw, None, None
        per = C * ctx.size * ctx.size
        L.check(L.lib().spair_stn_glimpse_bwd(L.ptr(img), L.ptr(zw), N, L.ptr(g.view(N, per)), per, L.ptr(dzw), N, C, H, ctx.size, ac,
                                              L.stream()), "spair_stn_glimpse_bwd")
        return None, dzw, None, None


def stn(image, z_where, output_dims, device=None, inverse=False):
    """modules.py:216-273.  ``image`` [N,C,H,W] float32 on the GPU, ``z_where`` [N,4] = (xt,yt,xs,ys), square sizes.
    Forward direction: glimpse extraction (border padding), differentiable wrt z_where.  ``inverse=True``: the sprites are placed
    on an ``output_dims`` canvas through the inverse affine (zeros padding), differentiable wrt both arguments; this materialises
    [N,C,I,I], which the training step itself never does (the renderer fuses it)."""
    from . import _lib as L
    if not image.is_cuda:
        raise L.SpairHipError("stn: tensors must live on the MI355X; there is no CPU path")
    N, C, H, W = image.shape
    assert H == W and int(output_dims[0]) == int(output_dims[1]), "square images / glimpses only"
    return _StnFn.apply(image, z_where, int(output_dims[0]), bool(inverse))


def _topology_conv_args(layer):
    return int(layer.get('filters', layer.get('out_channels'))), int(layer['kernel_size']), int(layer['stride'])


class ObjectConvEncoder(Module):
    """Convolutional glimpse encoder from ``cfg.CONV_OBJECT_ENCODER_TOPOLOGY`` (config.py:15-20), the variant the reference sketches in
    models.py:606-631 but never makes runnable (``Linear(123, ..)`` / ``self.linear`` undefined).  Opt-in and PARITY UNPINNED: the layer
    sizes follow the topology's own comments (28 -> 13 -> 6 -> 2 -> 2, 32 channels), flattened in (C, H, W) order into ``out``.
    ``SPAIR(..., object_encoder='conv')`` trains these parameters inside the HIP step (csrc/objconv.hip: direct fp32 convolutions per
    dependency wavefront, hand-written data / weight gradients).  ``forward`` here is a stand-alone inference helper (no autograd graph):
    every convolution as the implicit GEMM of spair_gemm_nt_conv (exact fp32 MFMA mode), the head through spair_gemm_nt."""

    def __init__(self, input_size, output_size, topology=None):
        super().__init__()
        n_prev, h, w = input_size
        topo = [dict(t) for t in (topology if topology is not None else cfg.CONV_OBJECT_ENCODER_TOPOLOGY)]
        net = OrderedDict()
        self.shapes = [(n_prev, h, w)]
        for i, layer in enumerate(topo):
            f, k, st = _topology_conv_args(layer)
            net['conv_%d' % i] = Conv2d(n_prev, f, kernel_size=k, stride=st)
            net['act_%d' % i] = ReLU()
            n_prev, h, w = f, (h - k) // st + 1, (w - k) // st + 1
            self.shapes.append((n_prev, h, w))
        self.conv = Sequential(net)
        self.out = Linear(n_prev * h * w, output_size)

    def forward(self, x):
        h = x.detach().float().permute(0, 2, 3, 1).contiguous()
        for m in self.conv:
            if isinstance(m, Conv2d):
                h = _hip_conv_nhwc(h, m.weight, m.bias, int(m.stride[0]), relu=True)       # every conv is followed by a ReLU
        flat = h.permute(0, 3, 1, 2).contiguous().flatten(start_dim=1)
        return _hip_linear(flat, self.out.weight, self.out.bias)


class ObjectConvDecoder(Module):
    """The mirrored decoder (models.py:633-665, equally non-functional there): ``Linear`` to the encoder's last feature map, then transposed
    convolutions of the reversed topology, ``output_padding`` chosen so that the encoder's sizes are retraced (2 -> 2 -> 6 -> 13 -> 28),
    no activation after the last one.  Each ConvTranspose2d runs on the HIP implicit-GEMM kernel as a stride-1 convolution of the
    zero-upsampled, (k-1)-padded input with the flipped kernel (stand-alone inference helper; inside the SPAIR step the layers run on
    csrc/objconv.hip and are trained, the output channels being the sprite's (colour.., alpha) logits per pixel).  Opt-in, parity unpinned."""

    def __init__(self, input_size, output_channel, encoder_shapes=None, topology=None):
        super().__init__()
        topo = [dict(t) for t in (topology if topology is not None else cfg.CONV_OBJECT_ENCODER_TOPOLOGY)]
        if encoder_shapes is None:
            encoder_shapes = ObjectConvEncoder([output_channel, cfg.OBJECT_SHAPE[0], cfg.OBJECT_SHAPE[1]], 1, topo).shapes
        self.top = encoder_shapes[-1]
        self.inp = Linear(input_size, self.top[0] * self.top[1] * self.top[2])
        net = OrderedDict()
        n_prev, h = self.top[0], self.top[1]
        for i, layer in enumerate(reversed(topo)):
            _, k, st = _topology_conv_args(layer)
            target = encoder_shapes[len(topo) - 1 - i]
            f = output_channel if i == len(topo) - 1 else target[0]     # (the encoder's own input has `chan` channels, the sprite chan + 1)
            op = target[1] - ((h - 1) * st + k)
            assert 0 <= op < max(st, 1) or (st == 1 and op == 0), "topology cannot be mirrored"
            net['conv_transposed_%d' % i] = nn.ConvTranspose2d(n_prev, f, kernel_size=k, stride=st, output_padding=op)
            if i < len(topo) - 1:
                net['act_%d' % i] = ReLU()
            n_prev, h = f, target[1]
        self.conv = Sequential(net)

    def forward(self, z):
        n = z.shape[0]
        h = _hip_linear(z.detach().float(), self.inp.weight, self.inp.bias).view(n, *self.top).permute(0, 2, 3, 1).contiguous()
        mods = list(self.conv)
        for i, m in enumerate(mods):
            if not isinstance(m, nn.ConvTranspose2d):
                continue
            k, st, op = int(m.kernel_size[0]), int(m.stride[0]), int(m.output_padding[0])
            B, H, W, C = h.shape
            up = torch.zeros(B, (H - 1) * st + 1 + 2 * (k - 1) + op, (W - 1) * st + 1 + 2 * (k - 1) + op, C, device=h.device)
            up[:, k - 1:k - 1 + (H - 1) * st + 1:st, k - 1:k - 1 + (W - 1) * st + 1:st, :] = h
            w = m.weight.detach().flip(2, 3).permute(1, 0, 2, 3).contiguous()          # [Cin,Cout,k,k] -> conv weight [Cout,Cin,k,k], flipped
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], ReLU)
            h = _hip_conv_nhwc(up, w, m.bias, 1, relu=relu)
        return h.permute(0, 3, 1, 2).contiguous()


def to_C_H_W(t):
    assert t.shape[1] == t.shape[2] and t.shape[3] != t.shape[2], 'are you sure this tensor is in [B, H, W, C] format?'
    return t.permute(0, 3, 1, 2)


def to_H_W_C(t):
    assert t.shape[2] == t.shape[3] and t.shape[1] != t.shape[2], 'are you sure this tensor is in [B, C, H, W] format?'
    return t.permute(0, 2, 3, 1)


def safe_log(t):
    return torch.log(t + 1e-9)
