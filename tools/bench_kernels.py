#!/usr/bin/env python3
"""Micro-benchmarks of individual C-ABI kernels at BASELINE config-2 shapes (GPU box only)."""
import ctypes
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spair_pytorch_amd import _lib as L


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def _i(*a):
    return (ctypes.c_int * len(a))(*a)


def main():
    lib = L.lib()
    B, Hin, Cin, Cout, k, s = 256, 70, 128, 128, 4, 2
    Hout = (Hin - k) // s + 1
    M = B * Hout * Hout
    K = k * k * Cin
    x = torch.randn(B, Hin, Hin, Cin, device="cuda")
    go = torch.randn(M, Cout, device="cuda")
    dW = torch.zeros(Cout, K, device="cuda")
    conv = _i(Hin, Hin, Cin, Hout, Hout, k, k, s, s, 1, 1, 0, 0)
    for dt in (1, 0):
        t = timeit(lambda: L.check(lib.spair_gemm_tn_conv(L.ptr(go), Cout, L.ptr(x), conv, L.ptr(dW), K, Cout, K, M, dt, L.stream()), "tn conv"))
        print("conv1 wgrad (TN conv) dtype=%d: %.3f ms  %.1f TFLOP/s" % (dt, t, 2.0 * M * Cout * K / t / 1e9))
    # plain TN at the same M/N with a smaller R that fits memory as an explicit matrix
    R2 = 65536
    A = torch.randn(R2, Cout, device="cuda")
    Bm = torch.randn(R2, K, device="cuda")
    for dt in (1, 0):
        t = timeit(lambda: L.check(lib.spair_gemm_tn(L.ptr(A), Cout, L.ptr(Bm), K, L.ptr(dW), K, Cout, K, R2, dt, L.stream()), "tn"))
        print("plain TN M=128 N=2048 R=65536 dtype=%d: %.3f ms  %.1f TFLOP/s" % (dt, t, 2.0 * R2 * Cout * K / t / 1e9))
    # NT conv forward
    w = (torch.randn(Cout, K, device="cuda") / K ** 0.5)
    wb = w.to(torch.bfloat16)
    out = torch.zeros(M, Cout, device="cuda")
    bias = torch.zeros(Cout, device="cuda")
    for dt, wt in ((1, wb), (0, w)):
        t = timeit(lambda: L.check(lib.spair_gemm_nt_conv(L.ptr(x), conv, L.ptr(wt), K, L.ptr(out), Cout, M, Cout, K, L.ptr(bias), None, 0, 1, 0, None, dt, L.stream()), "nt conv"))
        print("conv1 fwd (NT conv) dtype=%d: %.3f ms  %.1f TFLOP/s" % (dt, t, 2.0 * M * Cout * K / t / 1e9))
    # decoder-out shaped NT GEMM
    N, Kd, Nd = 65536, 256, 1568
    Hd = torch.randn(N, Kd, device="cuda")
    Wd = (torch.randn(Nd, Kd, device="cuda") / 16).to(torch.bfloat16)
    S = torch.zeros(N, Nd, device="cuda")
    t = timeit(lambda: L.check(lib.spair_gemm_nt(L.ptr(Hd), Kd, L.ptr(Wd), Kd, L.ptr(S), Nd, N, Nd, Kd, None, None, 0, 0, 0, 1, L.stream()), "nt"))
    print("decoder.out fwd NT [65536x256]x[1568x256]: %.3f ms  %.1f TFLOP/s" % (t, 2.0 * N * Nd * Kd / t / 1e9))


def main16():
    lib = L.lib()
    B, Hin, Cin, Cout, k, s = 256, 70, 128, 128, 4, 2
    Hout = (Hin - k) // s + 1
    M = B * Hout * Hout
    K = k * k * Cin
    x = torch.randn(B, Hin, Hin, Cin, device="cuda").to(torch.bfloat16)
    go = torch.randn(M, Cout, device="cuda").to(torch.bfloat16)
    dW = torch.zeros(Cout, K, device="cuda")
    conv = _i(Hin, Hin, Cin, Hout, Hout, k, k, s, s, 1, 1, 0, 0)
    t = timeit(lambda: L.check(lib.spair_gemm_tn16(L.ptr(go), Cout, L.ptr(x), 0, 1, L.ptr(dW), K, Cout, K, M, conv, Cin, k * k, None, L.stream()), "tn16"))
    print("conv1 wgrad tn16 (bf16 stored): %.3f ms  %.1f TFLOP/s" % (t, 2.0 * M * Cout * K / t / 1e9))
    w = (torch.randn(Cout, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    out = torch.zeros(M, Cout, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(Cout, device="cuda")
    t = timeit(lambda: L.check(lib.spair_gemm_nt16(L.ptr(x), 0, L.ptr(w), K, L.ptr(out), Cout, M, Cout, K, L.ptr(bias), None, 0, 0, 1, 1, conv, None, L.stream()), "nt16"))
    print("conv1 fwd nt16 (bf16 stored): %.3f ms  %.1f TFLOP/s" % (t, 2.0 * M * Cout * K / t / 1e9))
    N, Kd, Nd = 65536, 256, 1568
    Hd = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
    Wd = (torch.randn(Nd, Kd, device="cuda") / 16).to(torch.bfloat16)
    S = torch.zeros(N, Nd, device="cuda")
    t = timeit(lambda: L.check(lib.spair_gemm_nt16(L.ptr(Hd), Kd, L.ptr(Wd), Kd, L.ptr(S), Nd, N, Nd, Kd, None, None, 0, 0, 0, 0, None, None, L.stream()), "nt16"))
    print("decoder.out fwd nt16 [65536x256]x[1568x256] -> fp32: %.3f ms  %.1f TFLOP/s" % (t, 2.0 * N * Nd * Kd / t / 1e9))
    dL = torch.randn(N, Nd, device="cuda").to(torch.bfloat16)
    dWd = torch.zeros(Nd, Kd, device="cuda")
    t = timeit(lambda: L.check(lib.spair_gemm_tn16(L.ptr(dL), Nd, L.ptr(Hd), Kd, 1, L.ptr(dWd), Kd, Nd, Kd, N, None, 0, 0, None, L.stream()), "tn16"))
    print("decoder.out wgrad tn16: %.3f ms  %.1f TFLOP/s" % (t, 2.0 * N * Nd * Kd / t / 1e9))


if __name__ == "__main__":
    main16()
    main()
