"""CPU-side checks of the host layer: init parity with the reference (RNG order), state_dict
surface, schedules, the C-ABI library loads and exports what include/spair_hip.h declares, the
flat parameter layout agrees with the nn.Module, and the product path fails loudly off-GPU."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fresh_cfg(I=128, strides=(3, 2, 2, 1, 1, 1)):
    from spair_pytorch_amd import config as cfg
    cfg.set_grid(I, strides)
    return cfg


def test_init_reproduces_reference_seed3(golden_dir):
    """torch.manual_seed(3) + SPAIR(...) (train.py:39-41) gives the reference's initial weights
    bit for bit: same construction order, including the RNG draw in compute_output_shape."""
    _fresh_cfg()
    from spair_pytorch_amd.models import SPAIR
    z = np.load(os.path.join(golden_dir, "init_seed3.npz"))
    torch.manual_seed(3)
    m = SPAIR([1, 128, 128], None, torch.device("cpu"))
    sd = m.state_dict()
    keys = sorted(k[4:] for k in z.files if k.startswith("sha_"))
    assert sorted(sd.keys()) == keys
    assert len(keys) == 56 and sum(v.numel() for v in sd.values()) == 1462260
    for k in keys:
        h = np.frombuffer(hashlib.sha256(sd[k].numpy().tobytes()).digest(), np.uint8)
        assert (h == z["sha_" + k]).all(), k
    assert tuple(m.feature_space_dim) == tuple(z["feature_space_dim"])
    assert tuple(m.pixels_per_cell) == tuple(z["pixels_per_cell"])


def test_schedules_match_reference(golden_dir):
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.modules import exponential_decay
    u = np.load(os.path.join(golden_dir, "units.npz"))
    for s, wv, cv in zip(u["decay_steps"], u["decay_wheel"], u["decay_count_log"]):
        assert np.float32(exponential_decay(int(s), None, **cfg.LATENT_VAR_TRAINING_WHEEL_PARAM)) == wv
        assert np.float32(exponential_decay(int(s), None, **cfg.OBJ_PRES_COUNT_LOG_PRIOR)) == cv


def test_geometry_known_answers():
    from spair_pytorch_amd.modules import backbone_geometry
    topo = [dict(kernel_size=k, stride=s) for k, s in zip((4, 4, 4, 1, 1, 1), (3, 2, 2, 1, 1, 1))]
    assert backbone_geometry(128, topo) == (9, 14, 11, 12, 31)   # test_notebook.ipynb cell 10


def test_library_exports_every_declared_symbol():
    from spair_pytorch_amd import _build, _lib
    _build.build(verbose=False)
    hdr = open(os.path.join(ROOT, "include", "spair_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(spair_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 15
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name


def test_flat_layout_matches_module():
    _fresh_cfg()
    from spair_pytorch_amd import _lib as L
    from spair_pytorch_amd.models import SPAIR, make_dims
    m = SPAIR([1, 128, 128], None, torch.device("cpu"))
    d = make_dims(4, [1, 128, 128], m.backbone.topology, "f32")
    lib = L.lib()
    lib.spair_param_total.restype = ctypes.c_int64
    named = dict(m.named_parameters())
    n = lib.spair_param_count(ctypes.byref(d))
    assert n == len(named) == 56
    name = ctypes.create_string_buffer(128)
    off, ndim, shape = ctypes.c_int64(), ctypes.c_int(), (ctypes.c_int64 * 4)()
    end = 0
    for i in range(n):
        assert lib.spair_param_info(ctypes.byref(d), i, name, 128, ctypes.byref(off), shape, ctypes.byref(ndim)) == 0
        shp = tuple(shape[k] for k in range(ndim.value))
        assert tuple(named[name.value.decode()].shape) == shp
        assert off.value % 4 == 0 and off.value >= end     # 16-byte aligned, non-overlapping
        end = off.value + int(np.prod(shp))
    assert lib.spair_param_total(ctypes.byref(d)) >= end
    lib.spair_workspace_bytes.restype = ctypes.c_int64
    assert lib.spair_workspace_bytes(ctypes.byref(d)) > 0
    d.C = 4   # more than three colour channels: refused, not silently mis-rendered (1..3 are served: tests/test_rgb_gpu.py)
    assert lib.spair_workspace_bytes(ctypes.byref(d)) < 0


def test_flat_layout_of_colour_images():
    """cfg.INPUT_IMAGE_SHAPE[0] = 3 (config.py:4; models.py:150,163,480,524): the stem takes 3 input channels, the object encoder 3 x 28 x 28
    inputs, the decoder emits 4 channels per sprite texel -- every module parameter has a slot of its shape."""
    _fresh_cfg()
    from spair_pytorch_amd import _lib as L, config as cfg
    from spair_pytorch_amd.models import SPAIR, make_dims
    old = list(cfg.INPUT_IMAGE_SHAPE)
    try:
        cfg.INPUT_IMAGE_SHAPE[0] = 3
        m = SPAIR([3, 128, 128], None, torch.device("cpu"))
        d = make_dims(4, [3, 128, 128], m.backbone.topology, "f32")
        assert d.C == 3
        lib = L.lib()
        named = dict(m.named_parameters())
        assert tuple(named["backbone.net.conv_0.weight"].shape) == (128, 3, 4, 4)
        assert tuple(named["object_encoder.dense0.weight"].shape) == (256, 3 * 28 * 28)
        assert tuple(named["object_decoder.out.weight"].shape) == (4 * 28 * 28, 256)
        n = lib.spair_param_count(ctypes.byref(d))
        assert n == len(named) == 56
        name = ctypes.create_string_buffer(128)
        off, ndim, shape = ctypes.c_int64(), ctypes.c_int(), (ctypes.c_int64 * 4)()
        for i in range(n):
            assert lib.spair_param_info(ctypes.byref(d), i, name, 128, ctypes.byref(off), shape, ctypes.byref(ndim)) == 0
            assert tuple(named[name.value.decode()].shape) == tuple(shape[k] for k in range(ndim.value))
        lib.spair_workspace_bytes.restype = ctypes.c_int64
        assert lib.spair_workspace_bytes(ctypes.byref(d)) > 0
        for dt in ("f32", "bf16"):
            assert lib.spair_workspace_bytes(ctypes.byref(make_dims(4, [3, 128, 128], m.backbone.topology, dt))) > 0
    finally:
        cfg.INPUT_IMAGE_SHAPE[:] = old


def test_flat_layout_of_the_conv_object_variant():
    """SpairDims.obj_conv: object_encoder / object_decoder are the convolutional pair of CONV_OBJECT_ENCODER_TOPOLOGY (config.py:15-20;
    parity unpinned, models.py:606-665 cannot run) -- same naming contract (every module parameter has a slot of its shape), the decoder
    parameters in the first gradient bucket, the encoder's in the second."""
    _fresh_cfg()
    from spair_pytorch_amd import _lib as L
    from spair_pytorch_amd.models import SPAIR, make_dims
    m = SPAIR([1, 128, 128], None, torch.device("cpu"), compute_dtype="f32", object_encoder="conv")
    d = m._dims(4)
    assert d.obj_conv == 1 and d.oc_n == 4 and list(d.oc_k) == [4, 3, 3, 1] and list(d.oc_s) == [2, 2, 2, 1]
    lib = L.lib()
    named = dict(m.named_parameters())
    n = lib.spair_param_count(ctypes.byref(d))
    assert n == len(named)
    name = ctypes.create_string_buffer(128)
    off, ndim, shape = ctypes.c_int64(), ctypes.c_int(), (ctypes.c_int64 * 4)()
    offs, end = {}, 0
    for i in range(n):
        assert lib.spair_param_info(ctypes.byref(d), i, name, 128, ctypes.byref(off), shape, ctypes.byref(ndim)) == 0
        key = name.value.decode()
        shp = tuple(shape[k] for k in range(ndim.value))
        assert tuple(named[key].shape) == shp, key
        assert off.value % 4 == 0 and off.value >= end
        end = off.value + int(np.prod(shp))
        offs[key] = off.value
    assert tuple(named["object_encoder.out.weight"].shape) == (100, 128)
    assert tuple(named["object_decoder.conv.conv_transposed_3.weight"].shape) == (32, 2, 4, 4)
    lo, hi = (ctypes.c_int64 * 3)(), (ctypes.c_int64 * 3)()
    assert lib.spair_grad_buckets(ctypes.byref(d), lo, hi) == 0
    for key, o in offs.items():
        if key.startswith("object_decoder."):
            assert lo[0] <= o < hi[0], key
        if key.startswith("object_encoder."):
            assert lo[1] <= o < hi[1], key
    lib.spair_workspace_bytes.restype = ctypes.c_int64
    assert lib.spair_workspace_bytes(ctypes.byref(d)) > 0
    d.dtype = 1                     # the bf16 step takes the variant too (per-wavefront launches)
    assert lib.spair_workspace_bytes(ctypes.byref(d)) > 0
    d.oc_n = 5
    assert lib.spair_workspace_bytes(ctypes.byref(d)) < 0


def test_flat_layout_with_lookback_2():
    """N_LOOKBACK = 2 (config.py:31): 12 context records -> box network input 100 + 672, z / obj inputs wider by the same 448."""
    _fresh_cfg()
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd import _lib as L
    from spair_pytorch_amd.models import SPAIR
    cfg.N_LOOKBACK = 2
    try:
        m = SPAIR([1, 128, 128], None, torch.device("cpu"))
        d = m._dims(4)
        assert d.lookback == 2
        lib = L.lib()
        named = dict(m.named_parameters())
        name = ctypes.create_string_buffer(128)
        off, ndim, shape = ctypes.c_int64(), ctypes.c_int(), (ctypes.c_int64 * 4)()
        for i in range(lib.spair_param_count(ctypes.byref(d))):
            assert lib.spair_param_info(ctypes.byref(d), i, name, 128, ctypes.byref(off), shape, ctypes.byref(ndim)) == 0
            assert tuple(named[name.value.decode()].shape) == tuple(shape[k] for k in range(ndim.value)), name.value
        assert tuple(named["box_network.body.dense0.weight"].shape) == (100, 772)
        assert tuple(named["obj_network.dense0.weight"].shape) == (100, 4 + 50 + 100 + 672 + 100 + 1)
        lib.spair_workspace_bytes.restype = ctypes.c_int64
        assert lib.spair_workspace_bytes(ctypes.byref(d)) > 0
        d.lookback = 4
        assert lib.spair_workspace_bytes(ctypes.byref(d)) < 0
    finally:
        cfg.N_LOOKBACK = 1


def test_no_cpu_fallback():
    _fresh_cfg()
    from spair_pytorch_amd import _lib as L
    from spair_pytorch_amd.models import SPAIR
    m = SPAIR([1, 128, 128], None, torch.device("cpu"))
    with pytest.raises(L.SpairHipError):
        m(torch.zeros(2, 1, 128, 128), 0)


def test_experiment_patches_apply_to_the_product_sources():
    """tools/exp/patches/*.patch hold the measured-and-rejected variants and the stamp hooks that were moved out of the product kernels: they must
    keep applying to the sources they refer to (the round-2 band renderer is against round 2's render2.hip and is exempt)."""
    import glob
    import shutil
    import subprocess
    if shutil.which("patch") is None:
        pytest.skip("no patch(1) in this environment")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    patches = sorted(p for p in glob.glob(os.path.join(root, "tools", "exp", "patches", "*.patch")) if "render_fwd4_band" not in p)
    assert len(patches) >= 6
    for p in patches:
        r = subprocess.run(["patch", "-p1", "--dry-run", "-F0", "-i", p], cwd=root, capture_output=True, text=True)
        assert r.returncode == 0, (os.path.basename(p), r.stdout[-800:])
