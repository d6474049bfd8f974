"""Experiment: does the chip run two half-batch steps on two streams faster than one full-batch step?
(the per-cell chain is latency bound with one workgroup per sample: 128 samples leave half the CUs to the other half's GEMMs)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spair_pytorch_amd import config as cfg
from spair_pytorch_amd.data import scattered_digits
from spair_pytorch_amd.models import SPAIR
from spair_pytorch_amd.optim import FusedAdam

dev = torch.device("cuda")
cfg.set_grid(128, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)

def make(B, seed):
    m = SPAIR([1, 128, 128], None, dev, compute_dtype="bf16").to(dev)
    opt = FusedAdam(m, lr=1e-4)
    x = torch.from_numpy(scattered_digits(seed, B, 128, 11)[0]).to(dev)
    return m, opt, x

def fwd(m, opt, x, gs):
    opt.zero_grad()
    return m(x, gs)[0]

def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

full = make(256, 1)
def step_full():
    l = fwd(*full, 2000); l.backward(); full[1].step()
print("one stream, B=256: %.3f ms" % timeit(step_full))
half = make(128, 1)
def step_half():
    l = fwd(*half, 2000); l.backward(); half[1].step()
print("one stream, B=128: %.3f ms" % timeit(step_half))

A, Bm = make(128, 1), make(128, 2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def step_two(stagger):
    with torch.cuda.stream(sa):
        la = fwd(*A, 2000)
    with torch.cuda.stream(sb):
        lb = fwd(*Bm, 2000)
    with torch.cuda.stream(sa):
        la.backward()
    with torch.cuda.stream(sb):
        lb.backward()
    with torch.cuda.stream(sa):
        A[1].step()
    with torch.cuda.stream(sb):
        Bm[1].step()
print("two streams, 2 x B=128: %.3f ms" % timeit(lambda: step_two(False)))
# host cost of issuing one step (no GPU wait)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step_half()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host issue time per B=128 step: %.3f ms" % ((t1 - t0) / 10 * 1e3))
for cyc in (500000, 1000000, 2000000, 3000000, 4000000):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sb):
        e0.record(sb); torch.cuda._sleep(cyc); e1.record(sb)
    torch.cuda.synchronize()
    off = e0.elapsed_time(e1)
    with torch.cuda.stream(sb):
        torch.cuda._sleep(cyc)
    n = 40
    t0 = time.perf_counter()
    for _ in range(n): step_two(False)
    torch.cuda.synchronize()
    print("offset %.2f ms: two streams %.3f ms/step" % (off, ((time.perf_counter() - t0) * 1e3 - off) / n))
