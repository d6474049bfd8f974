"""Measured HBM copy / fill / read rates on this box (report the measured copy peak beside the vendor 8 TB/s)."""
import torch
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (256, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    x.fill_(1.0)
    tc = t(lambda: y.copy_(x)); tf = t(lambda: y.fill_(2.0)); tr = t(lambda: x.sum())
    print("%5d MB: copy %.2f TB/s (read+write bytes), fill %.2f TB/s, read(sum) %.2f TB/s" % (mb, 2 * n * 4 / tc / 1e12, n * 4 / tf / 1e12, n * 4 / tr / 1e12))
