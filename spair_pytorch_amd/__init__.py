"""spair_pytorch_amd -- MI355X-native SPAIR training step behind the reference's Python surface.

``from spair_pytorch_amd import config as cfg``; ``from spair_pytorch_amd.models import SPAIR``.
The compute lives in libspair_hip.so (hand-written gfx950 kernels, C ABI in include/spair_hip.h);
there is no PyTorch/CPU fallback.
"""
from . import config  # noqa: F401

__all__ = ["config"]
