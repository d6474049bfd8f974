#!/usr/bin/env python3
"""The bf16 GEMMs of BASELINE config 2 through their C-ABI entry points, one line per shape (GPU box only; developer tool).
(The SPAIR_NT16_* A/B switches live in tools/exp/patches/gemm16_switches.patch.)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spair_pytorch_amd import _lib as L


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def _i(*a):
    return (ctypes.c_int * len(a))(*a)


lib = L.lib()
bf = torch.bfloat16
rows = []


def nt16(name, M, N, K, conv=None, Ain=None, c16=1, relu=1, mask=False):
    A = Ain if Ain is not None else torch.randn(M, K, device="cuda").to(bf)
    B = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
    C = torch.zeros(M, N, device="cuda", dtype=bf if c16 else torch.float32)
    bias = torch.zeros(N, device="cuda")
    mk = torch.randn(M, N, device="cuda").to(bf) if mask else None
    t = timeit(lambda: L.check(lib.spair_gemm_nt16(L.ptr(A), 0 if conv is not None else K, L.ptr(B), K, L.ptr(C), N, M, N, K, L.ptr(bias),
                                                   L.ptr(mk), N if mask else 0, 1, relu, c16, conv, None, L.stream()), name))
    rows.append((name, t, 2.0 * M * N * K / t / 1e9))


def tn16(name, M, N, R, conv=None, Bin=None, cw=(0, 0)):
    A = torch.randn(R, M, device="cuda").to(bf)
    Bm = Bin if Bin is not None else torch.randn(R, N, device="cuda").to(bf)
    C = torch.zeros(M, N, device="cuda")
    cs = torch.zeros(M, device="cuda")
    scratch = torch.zeros(26214400, device="cuda")
    t = timeit(lambda: L.check(lib.spair_gemm_tn16(L.ptr(A), M, L.ptr(Bm), 0 if conv is not None else N, 1, L.ptr(C), N, M, N, R, conv, cw[0], cw[1],
                                                   L.ptr(cs), L.ptr(scratch), ctypes.c_longlong(scratch.numel()), L.stream()), name))
    rows.append((name, t, 2.0 * M * N * R / t / 1e9))


Bsz = 256
x0 = torch.randn(Bsz, 70, 70, 128, device="cuda").to(bf)        # act0 NHWC
x1 = torch.randn(Bsz, 34, 34, 128, device="cuda").to(bf)
nt16("conv1 fwd  [295936 x 128 x 2048] gather", Bsz * 34 * 34, 128, 2048, conv=_i(70, 70, 128, 34, 34, 4, 4, 2, 2, 1, 1, 0, 0), Ain=x0)
nt16("conv2 fwd  [65536 x 128 x 2048] gather", Bsz * 16 * 16, 128, 2048, conv=_i(34, 34, 128, 16, 16, 4, 4, 2, 2, 1, 1, 0, 0), Ain=x1)
nt16("dec1 fwd   [65536 x 256 x 128]", 65536, 256, 128)
nt16("dec.out fwd [65536 x 1568 x 256] (no sigmoid)", 65536, 1568, 256, relu=0)
nt16("dec.out dgrad [65536 x 256 x 1568] + gate", 65536, 256, 1568, relu=0, mask=True)
nt16("plain [65536 x 128 x 1024]", 65536, 128, 1024)
tn16("dec.out wgrad TN [1568 x 256] R=65536", 1568, 256, 65536)
tn16("conv1 wgrad TN conv [128 x 2048] R=295936", 128, 2048, Bsz * 34 * 34, conv=_i(70, 70, 128, 34, 34, 4, 4, 2, 2, 1, 1, 0, 0), Bin=x0, cw=(128, 16))
for name, t, tf in rows:
    print("%-52s %8.3f ms  %7.0f TFLOP/s" % (name, t, tf))
