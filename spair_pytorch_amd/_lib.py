"""ctypes binding of libspair_hip.so (C ABI in include/spair_hip.h).

There is NO fallback: if the library is missing or a call returns an error code, a
RuntimeError is raised.  ``import torch`` happens first so that the HIP runtime torch ships
(libamdhip64.so.7) is the one the library binds to -- streams and device pointers are then
shared between torch and these kernels.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPAIR_HIP_LIB") or os.path.join(_HERE, "libspair_hip.so")     # the override is for A/B builds of the same ABI

_ERR = {-1: "bad shape", -2: "unsupported dtype", -3: "kernel launch failed", -4: "unsupported configuration",
        -5: "misaligned leading dimension / size"}

_lib = None


class SpairHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SpairHipError(
                "libspair_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python -m spair_pytorch_amd._build`. There is no CPU/PyTorch fallback." % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def check(rc, what):
    if rc != 0:
        raise SpairHipError("%s failed: %s (code %d)" % (what, _ERR.get(rc, "unknown"), rc))


def ptr(t):
    """Device pointer of a (contiguous-enough) torch tensor, or NULL for None."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
