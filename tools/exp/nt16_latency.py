"""Is gemm_nt16's K loop bound by global-load latency?  Same kernel, operands resident in L2/MALL (small M) vs streamed from HBM."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spair_pytorch_amd import _lib as L
lib = L.lib()
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
N = 128
for M in (131072, 32768, 8192):
    for K in (2048, 4096):
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        B = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: L.check(lib.spair_gemm_nt16(L.ptr(A), K, L.ptr(B), K, L.ptr(C), N, M, N, K, None, None, 0, 0, 0, 1, None, None, L.stream()), "nt16"))
        tiles = M // 128
        print("M=%6d K=%5d (A %4d MB, %4d tiles): %.3f ms  %.0f TFLOP/s  %.2f us per K step per slot" % (M, K, M * K * 2 >> 20, tiles, t, 2.0 * M * N * K / t / 1e9,
              t * 1e3 / (max(1.0, tiles / 512.0) * (K / 64))))
