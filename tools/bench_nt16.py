#!/usr/bin/env python3
"""Per-tile overhead of gemm_nt16: time vs K at the decoder.out shape (GPU box only)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spair_pytorch_amd import _lib as L


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


lib = L.lib()
M, N = 65536, 1568
for K in (64, 128, 256, 512, 1024):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    for c16 in (1, 0):
        C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16 if c16 else torch.float32)
        t = timeit(lambda: L.check(lib.spair_gemm_nt16(L.ptr(A), K, L.ptr(B), K, L.ptr(C), N, M, N, K, None, None, 0, 0, 0, c16, None, None, L.stream()), "nt16"))
        tiles = (M // 128) * ((N + 127) // 128)
        print("K=%4d c16=%d: %.3f ms  %.0f TFLOP/s  %.2f us/tile (512 slots)" % (K, c16, t, 2.0 * M * N * K / t / 1e9, t * 1e3 / (tiles / 512)))
