#!/usr/bin/env python3
"""The fused decoder forward (csrc/dec_fused.hip) alone at BASELINE config 2's size, through its C-ABI entry point (GPU box only;
developer tool).  SPAIR_HIP_LIB selects an A/B build (tools/build_variant.sh)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spair_pytorch_amd import _lib as L

N, A, LDZ, H1, H2, NO = 65536, 50, 56, 128, 256, 1568
lib = L.lib()
lib.spair_decoder_fwd16_scratch_bytes.restype = ctypes.c_int64
g = torch.Generator(device="cuda").manual_seed(0)
za = torch.randn(N, LDZ, device="cuda", generator=g).to(torch.bfloat16)
W0, b0 = torch.randn(H1, A, device="cuda", generator=g) * 0.2, torch.randn(H1, device="cuda", generator=g) * 0.1
W1, b1 = torch.randn(H2, H1, device="cuda", generator=g) * 0.1, torch.randn(H2, device="cuda", generator=g) * 0.1
W2, b2 = torch.randn(NO, H2, device="cuda", generator=g) * 0.08, torch.randn(NO, device="cuda", generator=g) * 0.1
scratch = torch.zeros(int(lib.spair_decoder_fwd16_scratch_bytes(NO)), dtype=torch.uint8, device="cuda")
H1d = torch.empty(N, H1, dtype=torch.bfloat16, device="cuda")
H2d = torch.empty(N, H2, dtype=torch.bfloat16, device="cuda")
S = torch.empty(N, NO, dtype=torch.float16, device="cuda")


def run():
    L.check(lib.spair_decoder_fwd16(L.ptr(za), LDZ, L.ptr(W0), L.ptr(b0), L.ptr(W1), L.ptr(b1), L.ptr(W2), L.ptr(b2), L.ptr(H1d), L.ptr(H2d),
                                    L.ptr(S), NO, ctypes.c_longlong(N), A, NO, ctypes.c_float(2.0), ctypes.c_float(0.1), ctypes.c_float(5.0),
                                    L.ptr(scratch), L.stream()), "dec")


for _ in range(3):
    run()
torch.cuda.synchronize()
reps = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flop = 2.0 * N * (A * H1 + H1 * H2 + H2 * NO)
print("%s: pack + fused decoder fwd %.3f ms per call  (%.0f TFLOP/s on 57.7 GFLOP, incl. the ~3 us pack launch)" % (os.environ.get("SPAIR_HIP_LIB", "default"), ms, flop / ms / 1e9))
