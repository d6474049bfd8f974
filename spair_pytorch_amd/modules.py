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
        raise RuntimeError("Backbone.forward runs inside the HIP engine (SPAIR.forward); it has no PyTorch path")


class SequentialMultipleOutput(Module):
    """modules.py:276-284 (container only)."""

    def __init__(self, input, outputs):
        super().__init__()
        self.body = Sequential(input)
        self.output_layers = ModuleList(list(outputs.values()))

    def forward(self, x):
        raise RuntimeError("executed by the HIP engine")


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


def stn(image, z_where, output_dims, device=None, inverse=False):
    """modules.py:216-273.  Forward direction (glimpse extraction, border padding) runs the HIP
    gather kernel; ``image`` [N,C,H,W] float32 on the GPU, ``z_where`` [N,4]=(xt,yt,xs,ys).
    The inverse direction is only ever used inside the fused renderer (models.py:515) and is
    not materialised on its own."""
    from . import _lib as L
    if inverse:
        raise NotImplementedError("inverse stn is fused into the renderer (spair_render_fwd); it never materialises [N,C,H,W]")
    if not image.is_cuda:
        raise L.SpairHipError("stn: tensors must live on the MI355X; there is no CPU path")
    N, C, H, W = image.shape
    assert H == W and output_dims[0] == output_dims[1]
    P = int(output_dims[0])
    image = image.contiguous().float()
    zw = z_where.detach().contiguous().float()
    out = torch.empty(N, C * P * P, device=image.device, dtype=torch.float32)
    L.check(L.lib().spair_stn_glimpse_fwd(L.ptr(image), L.ptr(zw), N, L.ptr(out), C * P * P, N, C, H, P,
                                          int(cfg.ALIGN_CORNERS), L.stream()), "spair_stn_glimpse_fwd")
    return out.view(N, C, P, P)


def to_C_H_W(t):
    assert t.shape[1] == t.shape[2] and t.shape[3] != t.shape[2], 'are you sure this tensor is in [B, H, W, C] format?'
    return t.permute(0, 3, 1, 2)


def to_H_W_C(t):
    assert t.shape[2] == t.shape[3] and t.shape[1] != t.shape[2], 'are you sure this tensor is in [B, C, H, W] format?'
    return t.permute(0, 2, 3, 1)


def safe_log(t):
    return torch.log(t + 1e-9)
