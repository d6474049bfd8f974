"""``SPAIR`` -- drop-in for /root/reference/spair/models.py:15-131 on an MI355X.

Same constructor ``SPAIR(image_shape, writer, device)``, same ``forward(x, global_step=0) ->
(loss, recon_x, z_where, z_pres)``, same ``state_dict`` keys and ``spair.config``
hyper-parameters, so the reference's train.py loop (zero_grad / forward / backward / Adam
step, train.py:64-67) runs unchanged.  All arithmetic of the step is in libspair_hip.so
(hand-written gfx950 kernels behind the C ABI of include/spair_hip.h):

* parameters live in ONE flat fp32 device buffer (each nn.Parameter is a view into it), the
  gradients in a second one -- what the RCCL all-reduce and the fused Adam operate on;
* ``forward`` enqueues ``spair_forward`` (backbone -> 3G-2 dependency wavefronts of the
  per-cell encoder -> decoder -> fused inverse-STN compositor -> KL/loss); ``loss.backward()``
  enqueues ``spair_backward`` (hand-written reverse pass) which accumulates into ``p.grad``;
* there is no PyTorch/CPU fallback: without the HIP library or a GPU this module raises.
"""
import ctypes
import os
import math
import weakref

import numpy as np
import torch
from torch import nn

from . import _lib as L
from . import config as cfg
from .modules import Backbone, ObjectConvDecoder, ObjectConvEncoder, build_MLP, exponential_decay

DIST_NAMES = ['cy_logit', 'cx_logit', 'height_logit', 'width_logit', 'attr', 'depth_logit']


class SpairDims(ctypes.Structure):
    """include/spair_hip.h :: SpairDims"""
    _fields_ = [("B", ctypes.c_int), ("C", ctypes.c_int), ("I", ctypes.c_int), ("G", ctypes.c_int),
                ("P", ctypes.c_int), ("A", ctypes.c_int), ("F", ctypes.c_int), ("NP", ctypes.c_int),
                ("n_conv", ctypes.c_int), ("conv_k", ctypes.c_int * 8), ("conv_s", ctypes.c_int * 8),
                ("conv_c", ctypes.c_int * 8), ("pad_pre", ctypes.c_int), ("pad_post", ctypes.c_int),
                ("cell_px", ctypes.c_int), ("dtype", ctypes.c_int), ("align_corners", ctypes.c_int),
                ("anchor", ctypes.c_float), ("max_yx", ctypes.c_float), ("min_yx", ctypes.c_float),
                ("max_hw", ctypes.c_float), ("min_hw", ctypes.c_float), ("obj_logit_scale", ctypes.c_float),
                ("alpha_logit_scale", ctypes.c_float), ("alpha_logit_bias", ctypes.c_float),
                ("vae_beta", ctypes.c_float), ("prior_mean", ctypes.c_float * 6), ("prior_std", ctypes.c_float * 6),
                ("obj_conv", ctypes.c_int), ("oc_n", ctypes.c_int), ("oc_k", ctypes.c_int * 4), ("oc_s", ctypes.c_int * 4),
                ("oc_c", ctypes.c_int * 4), ("lookback", ctypes.c_int)]


class SpairStep(ctypes.Structure):
    """include/spair_hip.h :: SpairStep"""
    _fields_ = [("wheel", ctypes.c_float), ("count_prior_prob", ctypes.c_float), ("kl_scale", ctypes.c_float),
                ("train", ctypes.c_int), ("flags", ctypes.c_int), ("draw_noise", ctypes.c_int), ("noise_seed", ctypes.c_uint64),
                ("status", ctypes.c_void_p), ("status_host", ctypes.c_void_p)]


# bit 0: disable the fused persistent per-cell kernels (tests compare both paths); bit 1: stage stamps; bit 2: no helper stream
STEP_FLAGS = int(os.environ.get("SPAIR_STEP_FLAGS", "0"))

_DTYPES = {'f32': 0, 'fp32': 0, 'float32': 0, 'bf16': 1, 'bfloat16': 1}


def make_dims(batch, image_shape, topology, dtype=None, object_conv_topology=None, lookback=1):
    """``object_conv_topology``: the layer list of the convolutional object encoder / decoder variant (``None`` = the MLP pair)."""
    from .modules import backbone_geometry, _topology_conv_args
    d = SpairDims()
    C, I, I2 = image_shape
    assert I == I2, "square images only"
    pre, post, G, cell, _ = backbone_geometry(I, topology)
    d.B, d.C, d.I, d.G = int(batch), int(C), int(I), int(G)
    d.P, d.A, d.F, d.NP = int(cfg.OBJECT_SHAPE[0]), int(cfg.N_ATTRIBUTES), int(cfg.N_BACKBONE_FEATURES), int(cfg.N_PASSTHROUGH_FEATURES)
    d.n_conv = len(topology)
    for i, layer in enumerate(topology):
        d.conv_k[i], d.conv_s[i] = int(layer['kernel_size']), int(layer['stride'])
        d.conv_c[i] = int(layer.get('filters', layer.get('out_channels')))
    d.pad_pre, d.pad_post, d.cell_px = pre, post, cell
    d.dtype = _DTYPES[(dtype or cfg.COMPUTE_DTYPE).lower()]
    d.align_corners = int(bool(cfg.ALIGN_CORNERS))
    d.anchor = float(cfg.ANCHORBOX_SHAPE[0])
    d.max_yx, d.min_yx, d.max_hw, d.min_hw = cfg.MAX_YX, cfg.MIN_YX, cfg.MAX_HW, cfg.MIN_HW
    d.obj_logit_scale, d.alpha_logit_scale, d.alpha_logit_bias = cfg.OBJ_LOGIT_SCALE, cfg.ALPHA_LOGIT_SCALE, cfg.ALPHA_LOGIT_BIAS
    d.vae_beta = float(cfg.VAE_BETA)
    for i, n in enumerate(DIST_NAMES):
        d.prior_mean[i], d.prior_std[i] = float(cfg.PRIORS[n][0]), float(cfg.PRIORS[n][1])
    d.lookback = int(lookback)
    if object_conv_topology is not None:
        if len(object_conv_topology) > 4:
            raise L.SpairHipError("the convolutional object encoder takes at most 4 layers")
        d.obj_conv, d.oc_n = 1, len(object_conv_topology)
        for i, layer in enumerate(object_conv_topology):
            d.oc_c[i], d.oc_k[i], d.oc_s[i] = _topology_conv_args(layer)
    return d


def step_scalars(global_step, batch, world_size=1, train=True):
    """Host evaluation of the two schedules (models.py:59,186-188) in fp32."""
    st = SpairStep()
    st.wheel = exponential_decay(global_step, None, **cfg.LATENT_VAR_TRAINING_WHEEL_PARAM)
    lo = torch.tensor(exponential_decay(global_step, None, **cfg.OBJ_PRES_COUNT_LOG_PRIOR), dtype=torch.float32)
    st.count_prior_prob = float(1 / ((-lo).exp() + 1))
    st.kl_scale = 1.0 / (batch * world_size)
    st.train = int(train)
    st.flags = int(STEP_FLAGS)
    return st


class _NullWriter:
    def __getattr__(self, name):
        return lambda *a, **k: None


class _Engine(dict):
    """One batch size's workspace + noise maps + dims (a dict that can be weakly referenced)."""
    __slots__ = ("__weakref__",)


class _StepFn(torch.autograd.Function):
    """Ties the hand-written backward into autograd.  ``anchor`` is a dummy differentiable leaf:
    the parameter gradients are accumulated straight into the flat gradient buffer (the views
    behind every ``p.grad``) by ``spair_backward`` instead of being returned one tensor at a time."""

    @staticmethod
    def forward(ctx, anchor, model, x, step, noise):
        loss_terms, recon, z_where, z_pres = model._run_forward(x, step, noise, train=True)
        ctx.model, ctx.x, ctx.step, ctx.noise = model, x, step, noise
        # the saved activations live in the engine's workspace, not in autograd: remember WHICH forward filled it, so a backward through
        # a workspace that a later forward has overwritten (or that the engine cache has dropped) raises instead of being silently wrong.
        # A weak reference: the graph must not keep a multi-GB workspace alive after the cache evicted it.
        ctx.engine_ref = weakref.ref(model._last["engine"])
        ctx.generation = model._last["engine"]["generation"]
        ctx.mark_non_differentiable(recon, z_where, z_pres)
        ctx.set_materialize_grads(False)     # otherwise autograd zero-fills a gradient for each non-differentiable output (17 MB per step)
        model._loss_terms = loss_terms
        # a view, not a copy: `loss_terms` is a fresh buffer of this forward (a device copy here is 6 us + a launch gap between the loss
        # kernel and the backward's first kernel).  Slot 9 is the kernel's SECOND copy of the total: an in-place op on the returned loss
        # (`loss /= accum_steps`) does not change what loss_terms() and the logging helpers report (slots 0..8)
        return loss_terms[9], recon, z_where, z_pres

    @staticmethod
    def backward(ctx, g_loss, g_recon, g_zw, g_zp):
        if g_loss is None:
            return None, None, None, None, None
        engine = ctx.engine_ref()
        if engine is None or engine["generation"] != ctx.generation:
            raise L.SpairHipError(
                "backward() of a SPAIR forward whose saved activations were overwritten by a later forward of the same batch size, or "
                "whose workspace the engine cache has dropped (the engine keeps ONE set of activations per batch size and "
                "`max_engines` batch sizes; call backward before the next forward of that size)")
        ctx.model._run_backward(ctx.x, ctx.step, ctx.noise, g_loss.contiguous().float(), engine)
        return None, None, None, None, None      # the parameter gradients went straight into the flat buffer


class SPAIR(nn.Module):
    def __init__(self, image_shape, writer=None, device=None, compute_dtype=None, object_encoder=None):
        """``object_encoder``: 'mlp' (the reference's live configuration, models.py:152,165) or 'conv' -- the convolutional encoder /
        decoder pair of ``cfg.CONV_OBJECT_ENCODER_TOPOLOGY`` (config.py:15-20) that models.py:606-665 sketches but cannot run
        (parity unpinned); default ``cfg.OBJECT_ENCODER``.  The conv pair runs on the per-wavefront launches in either compute dtype (its
        own convolutions in fp32; the fused bf16 per-cell kernels are built for the MLP pair)."""
        super().__init__()
        self.object_encoder_kind = (object_encoder or cfg.OBJECT_ENCODER).lower()
        if self.object_encoder_kind not in ('mlp', 'conv'):
            raise ValueError("object_encoder must be 'mlp' or 'conv'")
        self.image_shape = list(image_shape)
        self.writer = writer if writer is not None else _NullWriter()
        self.B = 1
        self.device = torch.device(device) if device is not None else torch.device('cuda')
        self.compute_dtype = (compute_dtype or cfg.COMPUTE_DTYPE).lower()
        self.world_size = 1          # set by spair_pytorch_amd.ddp for the sharded loss
        self.context_dim = (cfg.N_LOOKBACK * 2 + 1) ** 2 // 2 * (4 + cfg.N_ATTRIBUTES + 1 + 1)
        self.lookback = int(cfg.N_LOOKBACK)
        if not 1 <= self.lookback <= 3:
            raise L.SpairHipError("N_LOOKBACK must be 1, 2 or 3 (4, 12 or 24 context neighbours)")
        self._build_networks()
        self._build_edge_element()
        self._build_indep_prior()
        self.pixels_per_cell = tuple(int(i) for i in self.backbone.grid_cell_size)
        self._flat = None
        self._flat_grad = None
        self._engines = {}
        # workspaces kept alive (one per batch size, least recently used dropped).  None = automatic: a second one (eval batch / last
        # partial batch beside the training batch) only while all live workspaces together stay under a quarter of the device's
        # memory, otherwise exactly one (they are ~5 GB each at B=256 / 128x128)
        self.max_engines = None
        self._grad_buckets = None    # set by spair_pytorch_amd.ddp.attach: readiness events for the overlapped all-reduce
        self._anchor = None
        self._loss_terms = None
        # failed / non-finite steps (SpairStep.status, status_host): two device ints [sticky bits, this step's bits] and one host word the
        # loss kernel stores to when a step fails -- owned by the model, so they outlive a workspace that is evicted or re-zeroed
        self._status_dev = None
        self._status_host = None
        self.raise_on_nonfinite = True     # forward() raises if an EARLIER step's loss was non-finite (no synchronisation: a host load)
        self.dist_param, self.dist = {}, {}
        self.training_wheel = None
        self.global_step = 0

    # ---- construction: same order of RNG consumption as models.py:133-167,273-290 -------------
    def _build_networks(self):
        self.backbone = Backbone(self.image_shape, cfg.N_BACKBONE_FEATURES)
        self.feature_space_dim = self.backbone.compute_output_shape()
        n_pass = cfg.N_PASSTHROUGH_FEATURES
        n_feat = self.feature_space_dim[0]
        self.box_network = build_MLP(n_feat + self.context_dim, multiple_output=(8, n_pass))
        obj_dim, chan = cfg.OBJECT_SHAPE[0], cfg.INPUT_IMAGE_SHAPE[0]
        if self.object_encoder_kind == 'conv':
            self.object_conv_topology = [dict(t) for t in cfg.CONV_OBJECT_ENCODER_TOPOLOGY]
            self.object_encoder = ObjectConvEncoder([chan, obj_dim, obj_dim], 2 * cfg.N_ATTRIBUTES, self.object_conv_topology)
        else:
            self.object_conv_topology = None
            self.object_encoder = build_MLP(obj_dim * obj_dim * chan, 2 * cfg.N_ATTRIBUTES, hidden_layers=[256, 128])
        z_in = 4 + cfg.N_ATTRIBUTES + n_pass + self.context_dim + cfg.N_BACKBONE_FEATURES
        self.z_network = build_MLP(z_in, multiple_output=(2, n_pass))
        self.obj_network = build_MLP(z_in + 1, 1)
        if self.object_encoder_kind == 'conv':
            self.object_decoder = ObjectConvDecoder(cfg.N_ATTRIBUTES, chan + 1, self.object_encoder.shapes, self.object_conv_topology)
        else:
            self.object_decoder = build_MLP(cfg.N_ATTRIBUTES, obj_dim * obj_dim * (chan + 1), hidden_layers=[128, 256])
        self.attn = _SelfAttnParams(55)   # dead in the reference (models.py:120,167); kept for state_dict parity

    def _build_edge_element(self):
        sizes = [4, cfg.N_ATTRIBUTES, 1, 1]
        t = torch.randn(sum(sizes))
        loc, attr, depth, pres = torch.split(t, sizes)
        elem = torch.cat((torch.sigmoid(loc), attr, torch.sigmoid(depth), torch.sigmoid(pres)))
        self.register_parameter('virtual_edge_element', nn.Parameter(elem))

    def _build_indep_prior(self):
        from torch.distributions import Normal
        self.kl_priors = {n: Normal(m, s) for n, (m, s) in cfg.PRIORS.items()}

    # ---- flat parameter / gradient buffers ---------------------------------------------------------
    def _dims(self, batch):
        return make_dims(batch, self.image_shape, self.backbone.topology, self.compute_dtype, self.object_conv_topology, self.lookback)

    def _flatten(self):
        """(Re)build the flat buffers on ``self.device`` and re-point every Parameter into them."""
        lib = L.lib()
        d = self._dims(1)
        lib.spair_param_total.restype = ctypes.c_int64
        total = lib.spair_param_total(ctypes.byref(d))
        n = lib.spair_param_count(ctypes.byref(d))
        named = dict(self.named_parameters())
        dev = self.device
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self._slices = {}
        name = ctypes.create_string_buffer(128)
        off, ndim = ctypes.c_int64(), ctypes.c_int()
        shape = (ctypes.c_int64 * 4)()
        seen = set()
        for i in range(n):
            L.check(lib.spair_param_info(ctypes.byref(d), i, name, 128, ctypes.byref(off), shape, ctypes.byref(ndim)), "spair_param_info")
            key = name.value.decode()
            shp = tuple(int(shape[k]) for k in range(ndim.value))
            p = named[key]
            if tuple(p.shape) != shp:
                raise L.SpairHipError("parameter %s: module shape %s != library layout %s" % (key, tuple(p.shape), shp))
            cnt = int(np.prod(shp))
            view = flat[off.value:off.value + cnt].view(shp)
            view.copy_(p.data.to(dev))
            p.data = view
            p.grad = None
            self._slices[key] = (off.value, cnt, shp)
            seen.add(key)
        missing = set(named) - seen
        if missing:
            raise L.SpairHipError("parameters without a slot in the flat layout: %s" % sorted(missing))
        self._flat, self._flat_grad = flat, flat_grad
        self._params_by_key = named
        self._anchor = torch.zeros((), device=dev, requires_grad=True)
        self._engines = {}
        self._status_dev = torch.zeros(2, dtype=torch.int32, device=dev)
        if self._status_host is None:
            word = ctypes.POINTER(ctypes.c_int)()
            L.check(lib.spair_host_word_alloc(ctypes.byref(word)), "spair_host_word_alloc")
            self._status_host = word
            weakref.finalize(self, lib.spair_host_word_free, word)

    def _apply(self, fn, recurse=True):
        # .to(device)/.cuda()/.float(): let nn.Module move the tensors, then re-flatten on the new device
        out = super()._apply(fn, recurse)
        p0 = next(self.parameters())
        if p0.is_cuda:
            self.device = p0.device
            self._flatten()
        else:
            self._flat = None
        return out

    def flat_parameters(self):
        self._ensure_ready()
        return self._flat

    def flat_gradients(self):
        self._ensure_ready()
        return self._flat_grad

    def _ensure_ready(self):
        if self._flat is None:
            if self.device.type != 'cuda':
                raise L.SpairHipError("SPAIR runs on an MI355X only: move the model with .to('cuda') (there is no CPU path)")
            self._flatten()
        # a load_state_dict / optimizer may have swapped .data: detect and re-point (cheap pointer check)
        p = self._params_by_key['virtual_edge_element']
        if p.data_ptr() != self._flat.data_ptr() + 4 * self._slices['virtual_edge_element'][0]:
            self._flatten()

    def _bind_grads(self):
        """Make every p.grad a view of the flat gradient buffer (zeroing slots that were None)."""
        all_none = True
        for key, p in self._params_by_key.items():
            if p.grad is not None:
                all_none = False
                break
        if all_none:
            self._flat_grad.zero_()
        base = self._flat_grad.data_ptr()
        for key, p in self._params_by_key.items():
            off, cnt, shp = self._slices[key]
            if key.startswith('attn.'):
                continue   # never receives a gradient in the reference (grad stays None)
            if p.grad is None:
                if not all_none:
                    self._flat_grad[off:off + cnt].zero_()
                p.grad = self._flat_grad[off:off + cnt].view(shp)
            elif p.grad.data_ptr() != base + 4 * off:
                v = self._flat_grad[off:off + cnt].view(shp)
                v.copy_(p.grad)
                p.grad = v

    # ---- engine ----------------------------------------------------------------------------------------
    def _engine(self, batch):
        e = self._engines.get(batch)
        if e is None:
            lib = L.lib()
            d = self._dims(batch)
            lib.spair_workspace_bytes.restype = ctypes.c_int64
            nbytes = lib.spair_workspace_bytes(ctypes.byref(d))
            if nbytes <= 0:
                raise L.SpairHipError("unsupported configuration for the HIP engine (B=%d, image=%s, topology=%s)" %
                                      (batch, self.image_shape, self.backbone.topology))
            G, A = d.G, d.A
            dev = self.device
            limit = self.max_engines
            if limit is None:
                live = sum(int(v["workspace"].numel()) for v in self._engines.values())
                limit = 2 if (live + nbytes) <= torch.cuda.get_device_properties(dev).total_memory // 4 else 1
            block = None
            while len(self._engines) >= max(1, limit):      # evict BEFORE allocating: two multi-GB workspaces never coexist needlessly
                old = self._engines.pop(next(iter(self._engines)))
                # `_last` holds a strong reference to the engine of the latest forward: drop it too, or the evicted workspace stays alive
                # until `_run_forward` reassigns `_last` -- i.e. across the allocation below (a pending backward through it raises, as for
                # any evicted engine: the step function only holds a weak reference; export_map and the logging helpers say so too)
                if getattr(self, "_last", None) is not None and self._last.get("engine") is old:
                    self._last = None
                # alternating batch sizes under limit 1 (the partial last batch of every epoch): the evicted block is re-used when the new
                # workspace fits in it -- re-zeroed, no hipFree / hipMalloc / device synchronisation per alternation
                if block is None and int(old["workspace"].numel()) >= nbytes:
                    block = old["workspace"]
                del old
            if block is None and max(1, limit) == 1:
                torch.cuda.empty_cache()                  # the limit-1 case exists because two workspaces do not fit comfortably: return the block
            ws = block.zero_() if block is not None else torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            e = _Engine(dims=d, generation=0,
                     workspace=ws,                                                   # zero-initialised ONCE per engine
                     noise=dict(eps_box=torch.empty(batch, 4, G, G, device=dev), eps_attr=torch.empty(batch, A, G, G, device=dev),
                                eps_depth=torch.empty(batch, 1, G, G, device=dev), u_pres=torch.empty(batch, 1, G, G, device=dev)))
        else:
            self._engines.pop(batch)
        self._engines[batch] = e         # most recently used last
        return e

    def _draw_noise(self, e):
        """The 7 per-cell draws (models.py:333-336,84,95,402-403) as whole maps, one launch; the seed
        comes from torch's CPU generator so torch.manual_seed controls it."""
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        # the maps are filled by spair_forward itself (SpairStep.draw_noise), on its helper stream beside the backbone
        return dict(e['noise'], _seed=seed)

    def _run_forward(self, x, step, noise, train):
        e = self._engine(x.shape[0])
        d = e['dims']
        dev = x.device
        B, G = x.shape[0], d.G
        loss_terms = torch.empty(16, device=dev, dtype=torch.float32)
        recon = torch.empty(B, d.C, d.I, d.I, device=dev, dtype=torch.float32)
        z_where = torch.empty(B, 4, G, G, device=dev, dtype=torch.float32)
        z_pres = torch.empty(B, 1, G, G, device=dev, dtype=torch.float32)
        st = step_scalars(step, B, self.world_size, train)
        if noise.get('_seed') is not None:
            st.draw_noise, st.noise_seed = 1, int(noise['_seed'])
        st.status = self._status_dev.data_ptr()
        st.status_host = ctypes.cast(self._status_host, ctypes.c_void_p).value
        e["generation"] += 1              # whatever this workspace held for an earlier forward is gone now
        self._last = dict(engine=e, st=st)
        L.check(L.lib().spair_forward(ctypes.byref(d), ctypes.byref(st), L.ptr(self._flat), L.ptr(x), L.ptr(noise['eps_box']),
                                      L.ptr(noise['eps_attr']), L.ptr(noise['eps_depth']), L.ptr(noise['u_pres']),
                                      L.ptr(e['workspace']), L.ptr(loss_terms), L.ptr(recon), L.ptr(z_where), L.ptr(z_pres),
                                      L.stream()), "spair_forward")
        return loss_terms, recon, z_where, z_pres

    def _run_backward(self, x, step, noise, g_loss, e=None):
        e = e if e is not None else self._engine(x.shape[0])
        st = step_scalars(step, x.shape[0], self.world_size, True)
        self._bind_grads()
        gb = self._grad_buckets
        ev = gb.handles() if gb is not None else [ctypes.c_void_p(0)] * 3
        L.check(L.lib().spair_backward_ev(ctypes.byref(e['dims']), ctypes.byref(st), L.ptr(self._flat), L.ptr(x), L.ptr(noise['eps_box']),
                                          L.ptr(noise['eps_attr']), L.ptr(noise['eps_depth']), L.ptr(noise['u_pres']),
                                          L.ptr(e['workspace']), L.ptr(g_loss), L.ptr(self._flat_grad), L.stream(), ev[0], ev[1], ev[2]),
                "spair_backward")
        if gb is not None:
            gb.pending = True             # ddp.allreduce_gradients(model) consumes the three events

    # ---- public API ------------------------------------------------------------------------------------
    def forward(self, x, global_step=0, noise=None):
        """models.py:35-131.  ``noise`` (optional dict eps_box/eps_attr/eps_depth/u_pres, NCHW maps)
        replaces the internal draws -- used by the parity tests."""
        self._ensure_ready()
        if not x.is_cuda:
            raise L.SpairHipError("input must be on the MI355X (got %s); there is no CPU path" % x.device)
        x = x.contiguous().float()
        if list(x.shape[1:]) != list(self.image_shape):
            raise AssertionError("expected input [B,%s], got %s" % (self.image_shape, tuple(x.shape)))
        if self.raise_on_nonfinite and self._status_host[0] != 0:
            raise L.SpairHipError(self._status_text(self._status_host[0]) + " -- seen by a later forward(); clear_step_status() to go on")
        self.global_step = global_step
        self.batch_size = x.shape[0]
        e = self._engine(x.shape[0])
        if noise is None:
            noise = self._draw_noise(e)
        else:
            noise = {k: noise[k].to(x.device).contiguous().float() for k in ('eps_box', 'eps_attr', 'eps_depth', 'u_pres')}
        self.training_wheel = exponential_decay(global_step, None, **cfg.LATENT_VAR_TRAINING_WHEEL_PARAM)
        if torch.is_grad_enabled():
            loss, recon, z_where, z_pres = _StepFn.apply(self._anchor, self, x, int(global_step), noise)
        else:
            terms, recon, z_where, z_pres = self._run_forward(x, int(global_step), noise, train=False)
            self._loss_terms = terms
            loss = terms[0].clone()
        self._log_scalars()
        self.dist_param, self.dist = _LazyDist(self), _LazyDist(self, as_normal=True)
        return loss, recon, z_where, z_pres

    def _log_scalars(self):
        w = self.writer
        if isinstance(w, _NullWriter):
            return
        t = self._loss_terms
        w.add_scalar('training_wheel', self.training_wheel, self.global_step)
        w.add_scalar('losses/reconst', t[1], self.global_step)
        for i, n in enumerate(DIST_NAMES + ['pres_dist']):
            w.add_scalar('losses/KL{}'.format(n), t[2 + i], self.global_step)
        w.add_scalar('losses/total', t[0], self.global_step)

    def loss_terms(self):
        """Device tensor [9]: total, reconstruction BCE, KL cy,cx,height,width,attr,depth,pres (no sync)."""
        return self._loss_terms[:9]

    def _last_engine(self):
        last = getattr(self, "_last", None)
        if last is None:
            raise L.SpairHipError("no live forward pass: the model has not run yet, or the workspace of its latest forward was evicted by a "
                                  "forward of another batch size (max_engines)")
        return last['engine']

    # ---- failed / non-finite steps -------------------------------------------------------------------------
    @staticmethod
    def _status_text(bits):
        what = []
        if bits & 2:
            what.append("a training step produced a non-finite loss (NaN / inf in a loss term; the reference raises at that point: "
                        "debug_tools.py:245-271)")
        if bits & 1:
            what.append("a band-split hand-off of the per-cell chain timed out (preempted / oversubscribed GPU?): the results of that "
                        "workspace are not to be used -- restart the process, do not retry in place")
        return "; ".join(what) or "ok"

    def step_status(self):
        """Bits of every forward of this model so far (sticky): 1 = a band-split hand-off timed out, 2 = a loss term was non-finite; 0 = clean.
        SYNCHRONISES -- call it where the host waits anyway (after ``loss.item()``, at the end of an epoch).  The same word is also
        kept in host memory the loss kernel writes to: ``forward()`` looks at it (a host load, no synchronisation) and raises once a
        failed step has completed, unless ``raise_on_nonfinite`` is off.  ``FusedAdam`` leaves a flagged step out (no NaN parameters)."""
        self._ensure_ready()
        return int(self._status_dev[0].item())

    def check_step_status(self):
        bits = self.step_status()
        if bits:
            raise L.SpairHipError(self._status_text(bits))

    def clear_step_status(self):
        """After a non-finite step that the caller has dealt with (the guarded optimizer skipped it): forget it.  A time-out (bit 1) stays in
        the workspace that saw it and comes back with its next forward."""
        self._ensure_ready()
        self._status_dev.zero_()
        torch.cuda.current_stream().synchronize()
        self._status_host[0] = 0

    def chain_status(self):
        """Band-split hand-off status of the latest forward's workspace (grids wider than 16 cells): -1 where the chain runs unsplit, 0 = every
        hand-off arrived, 1 = a wait timed out (sticky; that step's loss and every later one is NaN).  SYNCHRONISES -- call it where the
        host waits anyway (after ``loss.item()``, at the end of an epoch); ``check_chain_status()`` raises instead of returning 1."""
        e = self._last_engine()
        out = torch.zeros(1, dtype=torch.int32, device=self.device)
        L.check(L.lib().spair_chain_sync_status(ctypes.byref(e['dims']), L.ptr(e['workspace']), L.ptr(out), L.stream()), "spair_chain_sync_status")
        return int(out.item())

    def check_chain_status(self):
        if self.chain_status() == 1:
            raise L.SpairHipError("a band-split hand-off of the per-cell chain timed out (preempted / oversubscribed GPU?): the results of this "
                                  "workspace are not to be used -- restart the process, do not retry in place")

    def export_map(self, which):
        """Per-cell quantity of the last forward as an NCHW map (see spair_export_map)."""
        e = self._last_engine()
        d = e['dims']
        ch = d.A if which in (0, 6, 12) else {0: 8, 1: 2 * d.A, 2: 2, 3: 1}[which % 100] if which >= 100 else 1
        out = torch.empty(d.B, ch, d.G, d.G, device=self.device, dtype=torch.float32)
        L.check(L.lib().spair_export_map(ctypes.byref(d), L.ptr(e['workspace']), int(which), L.ptr(out), L.stream()), "spair_export_map")
        return out


class _LazyDist(dict):
    """``self.dist_param[name]['mean'|'sigma']`` / ``self.dist[name]`` of the reference
    (models.py:122-125,441-448), materialised from the engine's row buffers on first access."""

    def __init__(self, model, as_normal=False):
        super().__init__()
        self._m, self._n = model, as_normal

    def __missing__(self, name):
        i = DIST_NAMES.index(name)
        mean, sigma = self._m.export_map(2 + i), self._m.export_map(8 + i)
        if self._n:
            from torch.distributions import Normal
            v = Normal(loc=mean, scale=sigma)
        else:
            v = {'mean': mean, 'sigma': sigma}
        self[name] = v
        return v

    def keys(self):
        return DIST_NAMES

    def items(self):
        return [(n, self[n]) for n in DIST_NAMES]


class _SelfAttnParams(nn.Module):
    """Parameter container for the reference's Self_Attn(55) (models.py:667-699).  Its output is
    discarded there and its parameters never get gradients; only the state_dict entries matter."""

    def __init__(self, in_dim):
        super().__init__()
        self.chanel_in = in_dim
        self.query_conv = nn.Conv2d(in_dim, in_dim // 8, kernel_size=1)
        self.key_conv = nn.Conv2d(in_dim, in_dim // 8, kernel_size=1)
        self.value_conv = nn.Conv2d(in_dim, in_dim, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))
