"""Timing-only emulation of "two samples per chain workgroup + two half-batches in flight"  with TODAY's kernels: in the
`chalf` variant (tools/exp/patches/chain_half_grid.patch) the chain kernels of a B = 128 engine are launched with 64 workgroups -- the
machine footprint and duration a two-sample launch over 128 samples would have if T2 = T1 (the optimistic end; tools/exp/chain_cells_slope.py
puts T2 >= 1.05 T1, the global-tap sampling it needs at +9 % on the forward kernel).  The other 64 samples' rows keep the values of the
warm-up steps, so the decoder / renderer / weight-gradient kernels see realistic objects.  Numerically meaningless; the question is only what
the chip makes of two such half-steps on two streams against one B = 256 step."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spair_pytorch_amd import _lib as L, config as cfg
from spair_pytorch_amd.data import scattered_digits
from spair_pytorch_amd.models import SPAIR
from spair_pytorch_amd.optim import FusedAdam
dev = torch.device("cuda")
cfg.set_grid(128, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)
lib = L.lib()
half = getattr(lib, "spair_exp_chain_half", None)
assert half is not None, "run with SPAIR_HIP_LIB=build/libspair_chalf.so"

def make(B, seed):
    m = SPAIR([1, 128, 128], None, dev, compute_dtype="bf16").to(dev)
    m.raise_on_nonfinite = False
    return m, FusedAdam(m, lr=1e-4), torch.from_numpy(scattered_digits(seed, B, 128, 11)[0]).to(dev)

def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

full = make(256, 1)
def step(m, opt, x):
    opt.zero_grad(); l = m(x, 2000)[0]; l.backward(); opt.step()
half(0)
print("one stream, B=256: %.3f ms" % timeit(lambda: step(*full)))
A, Bm = make(128, 1), make(128, 2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def step_two():
    with torch.cuda.stream(sa):
        A[1].zero_grad(); la = A[0](A[2], 2000)[0]
    with torch.cuda.stream(sb):
        Bm[1].zero_grad(); lb = Bm[0](Bm[2], 2000)[0]
    with torch.cuda.stream(sa):
        la.backward()
    with torch.cuda.stream(sb):
        lb.backward()
    with torch.cuda.stream(sa):
        A[1].step()
    with torch.cuda.stream(sb):
        Bm[1].step()
print("one stream, B=128: %.3f ms" % timeit(lambda: step(*A)))
print("two streams, 2 x B=128, chain as built: %.3f ms" % timeit(step_two))
half(1)
print("one stream, B=128, chain on 64 workgroups: %.3f ms" % timeit(lambda: step(*A)))
for rep in range(3):
    print("two streams, 2 x B=128, chain on 64 workgroups each: %.3f ms" % timeit(step_two))
# staggered start of the second stream
for cyc in (400000, 1000000, 2000000):
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
    for _ in range(n): step_two()
    torch.cuda.synchronize()
    print("  second stream %.2f ms behind: %.3f ms/step" % (off, ((time.perf_counter() - t0) * 1e3 - off) / n))
half(0)
