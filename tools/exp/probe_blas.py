"""What the vendor library reaches on the hot GEMM shapes (a yardstick for csrc/gemm16.hip, not part of the product)."""
import torch, time
dev = torch.device("cuda")
def run(name, M, N, K, tn=False):
    if tn:
        a = torch.randn(K, M, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
        f = lambda: a.t() @ b
    else:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        f = lambda: a @ b.t()
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-28s M=%7d N=%5d K=%6d  %8.1f us  %7.1f TFLOP/s" % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
run("conv1 fwd (explicit)", 262144, 128, 2048)
run("conv1 dgrad-like", 262144, 2048, 128)
run("conv1 wgrad TN", 128, 2048, 262144, tn=True)
run("conv2 fwd", 65536, 128, 2048)
run("dec_out fwd", 65536, 1568, 256)
run("dec_out dgrad", 65536, 256, 1568)
run("dec_out wgrad TN", 1568, 256, 65536, tn=True)
run("dec1 fwd", 65536, 256, 128)
run("square 8192", 8192, 8192, 8192)
