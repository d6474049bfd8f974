#!/usr/bin/env python3
"""conv_1 / conv_2 forward of BASELINE config 2 (128 -> 128 channels, 4x4, stride 2, bf16 NHWC): the patch-resident kernel (conv_s2.hip)
against the implicit-GEMM kernel (gemm16.hip) through their C-ABI entry points (GPU box only; developer tool)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spair_pytorch_amd import _lib as L
lib = L.lib()
bf = torch.bfloat16


def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, B, Hout in (("conv_1", 256, 34), ("conv_2", 256, 16)):
    Hin, C = 2 * Hout + 2, 128
    x = torch.randn(B, Hin, Hin, C, device="cuda").to(bf)
    wf = (torch.randn(C, 2048, device="cuda") / 45).to(bf)
    bias = torch.zeros(C, device="cuda")
    M = B * Hout * Hout
    out = torch.empty(M, C, device="cuda", dtype=bf)
    t_new = timeit(lambda: L.check(lib.spair_conv_s2k4_fwd16(L.ptr(x), L.ptr(wf), L.ptr(bias), L.ptr(out), B, Hin, Hout, L.stream()), name))
    conv13 = (ctypes.c_int * 13)(Hin, Hin, C, Hout, Hout, 4, 4, 2, 2, 1, 1, 0, 0)
    out2 = torch.empty(M, C, device="cuda", dtype=bf)
    t_old = timeit(lambda: L.check(lib.spair_gemm_nt16(L.ptr(x), 0, L.ptr(wf), 2048, L.ptr(out2), C, M, C, 2048, L.ptr(bias), None, 0, 1, 1, 1, conv13, None,
                                                       L.stream()), name))
    fl = 2.0 * M * C * 2048
    print("%s: patch-resident %.3f ms (%.0f TFLOP/s) | implicit GEMM (natural K order) %.3f ms (%.0f TFLOP/s)" % (name, t_new, fl / t_new / 1e9, t_old, fl / t_old / 1e9))

# data gradients of the same layers: patch-resident kernel (plain, no stem fusion) vs the implicit-GEMM kernel (4 parity classes, one launch each here)
for name, B, Ho in (("conv_1 dgrad", 256, 34), ("conv_2 dgrad", 256, 16)):
    Hc, C = Ho + 1, 128
    Hi = 2 * Hc
    dout = torch.randn(B, Ho, Ho, C, device="cuda").to(bf)
    gate = torch.randn(B, Hi, Hi, C, device="cuda").to(bf)
    wd = [(torch.randn(C, 512, device="cuda") / 22).to(bf) for _ in range(4)]
    out = torch.empty(B * Hi * Hi, C, device="cuda", dtype=bf)
    t_new = timeit(lambda: L.check(lib.spair_conv_s2k4_dgrad16(L.ptr(dout), L.ptr(wd[0]), L.ptr(wd[1]), L.ptr(wd[2]), L.ptr(wd[3]), L.ptr(gate),
                                                               L.ptr(out), B, Ho, L.stream()), name))
    M = B * Hc * Hc
    conv13 = (ctypes.c_int * 13)(Ho, Ho, C, Hc, Hc, 2, 2, 1, 1, -1, -1, 0, 0)
    def old():
        for qq in range(4):
            cmap = (ctypes.c_int * 8)(Hc, Hc, Hi, Hi, 2, 2, qq // 2, qq % 2)
            L.check(lib.spair_gemm_nt16(L.ptr(dout), 0, L.ptr(wd[qq]), 512, L.ptr(out), C, M, C, 512, None, L.ptr(gate), C, 1, 0, 1, conv13, cmap,
                                        L.stream()), name)
    t_old = timeit(old)
    fl = 2.0 * M * C * 512 * 4
    print("%s: patch-resident %.3f ms (%.0f TFLOP/s) | implicit GEMM, 4 launches %.3f ms (%.0f TFLOP/s)" % (name, t_new, fl / t_new / 1e9, t_old, fl / t_old / 1e9))

# weight gradients: the DMA-staged split-K kernel (tn_ring.hip, taken when scratch is passed) vs gemm_tn16_kernel (no scratch: fp32 atomics)
scratch = torch.empty(1536 * 128 * 128, device="cuda")
for name, B, Hout in (("conv_1 wgrad", 256, 31), ("conv_2 wgrad", 256, 15)):
    Hin, C = 2 * Hout + 2, 128
    x = torch.randn(B, Hin, Hin, C, device="cuda").to(bf)
    M = B * Hout * Hout
    dy = torch.randn(M, C, device="cuda").to(bf)
    dW = torch.zeros(C, 2048, device="cuda")
    db = torch.zeros(C, device="cuda")
    conv13 = (ctypes.c_int * 13)(Hin, Hin, C, Hout, Hout, 4, 4, 2, 2, 1, 1, 0, 0)
    t_new = timeit(lambda: L.check(lib.spair_gemm_tn16(L.ptr(dy), C, L.ptr(x), 0, 1, L.ptr(dW), 2048, C, 2048, M, conv13, C, 16, L.ptr(db), L.ptr(scratch),
                                                       ctypes.c_longlong(scratch.numel()), L.stream()), name))
    t_old = timeit(lambda: L.check(lib.spair_gemm_tn16(L.ptr(dy), C, L.ptr(x), 0, 1, L.ptr(dW), 2048, C, 2048, M, conv13, C, 16, L.ptr(db), None,
                                                       ctypes.c_longlong(0), L.stream()), name))
    fl = 2.0 * M * C * 2048
    print("%s: ring (+reduce) %.3f ms (%.0f TFLOP/s) | register-staged, atomics %.3f ms (%.0f TFLOP/s)" % (name, t_new, fl / t_new / 1e9, t_old, fl / t_old / 1e9))
for name, R, M, N in (("decoder.out wgrad", 65536, 1568, 256), ("decoder.1 wgrad", 65536, 256, 128), ("enc.0 wgrad", 65536, 256, 784)):
    A = torch.randn(R, M, device="cuda").to(bf)
    Bm = torch.randn(R, (N + 7) // 8 * 8, device="cuda").to(bf)
    Cw = torch.zeros(M, N, device="cuda")
    cs = torch.zeros(M, device="cuda")
    t_new = timeit(lambda: L.check(lib.spair_gemm_tn16(L.ptr(A), M, L.ptr(Bm), Bm.shape[1], 1, L.ptr(Cw), N, M, N, R, None, 0, 0, L.ptr(cs), L.ptr(scratch),
                                                       ctypes.c_longlong(scratch.numel()), L.stream()), name))
    t_old = timeit(lambda: L.check(lib.spair_gemm_tn16(L.ptr(A), M, L.ptr(Bm), Bm.shape[1], 1, L.ptr(Cw), N, M, N, R, None, 0, 0, L.ptr(cs), None,
                                                       ctypes.c_longlong(0), L.stream()), name))
    fl = 2.0 * M * N * R
    print("%s: ring (+reduce) %.3f ms (%.0f TFLOP/s) | register-staged, atomics %.3f ms (%.0f TFLOP/s)" % (name, t_new, fl / t_new / 1e9, t_old, fl / t_old / 1e9))
