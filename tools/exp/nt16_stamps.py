#!/usr/bin/env python3
"""[needs `git apply tools/exp/patches/gemm16_switches.patch` first: the hooks are not in the product source]
K-loop phase cycles of gemm_nt16 (diagnostic build: tools/build_variant.sh nt16st gemm16.hip -DNT16_STAMP, SPAIR_HIP_LIB=build/libspair_nt16st.so)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spair_pytorch_amd import _lib as L
lib = L.lib()
bf = torch.bfloat16
def _i(*a):
    return (ctypes.c_int * len(a))(*a)
def run(name, M, N, K, conv=None, Ain=None, relu=1):
    A = Ain if Ain is not None else torch.randn(M, K, device="cuda").to(bf)
    B = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
    C = torch.zeros(M, N, device="cuda", dtype=bf)
    bias = torch.zeros(N, device="cuda")
    f = lambda: L.check(lib.spair_gemm_nt16(L.ptr(A), 0 if conv is not None else K, L.ptr(B), K, L.ptr(C), N, M, N, K, L.ptr(bias), None, 0, 1, relu, 1, conv, None, L.stream()), name)
    for _ in range(3): f()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    lib.spair_nt16_stamps(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    lib.spair_nt16_stamps(out, 1)
    n = max(out[4], 1)
    print("%-40s %.3f ms  per K tile per wave: issue %5.0f  mfma %5.0f  vmwait %5.0f  barrier %5.0f  (sum %5.0f cyc, %d tiles)" % (
        name, e0.elapsed_time(e1), out[0] / n, out[1] / n, out[2] / n, out[3] / n, sum(out[:4]) / n, n))
Bsz = 256
x0 = torch.randn(Bsz, 70, 70, 128, device="cuda").to(bf)
run("conv1 fwd gather K2048", Bsz * 34 * 34, 128, 2048, conv=_i(70, 70, 128, 34, 34, 4, 4, 2, 2, 1, 1, 0, 0), Ain=x0)
run("plain 65536x128x1024", 65536, 128, 1024)
run("plain 262144x128x2048", 262144, 128, 2048)
run("dec.out fwd 65536x1568x256", 65536, 1568, 256, relu=0)
