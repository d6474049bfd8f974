"""Debug aid for render3b.hip: d z_where of spair_render_bwd16r against spair_render_bwd16 (tap kernel), per component."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spair_pytorch_amd import _lib as L
B, G, I, smin, srange = [float(v) if "." in v else int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else "4 8 128 0.12 0.12".split())]
P, HW = 28, G * G
N = B * HW
g = torch.Generator().manual_seed(B + G + I + 1)
logits = torch.randn(N, P, P, 2, generator=g); logits[..., 1] += 1.0
S = torch.sigmoid(logits).half().float()
if os.environ.get("SMOOTH"):
    yy, xx = torch.meshgrid(torch.arange(28.), torch.arange(28.), indexing="ij")
    S[..., 0] = (0.5 + 0.4 * torch.sin(xx / 5 + 1)).half().float(); S[..., 1] = (0.5 + 0.4 * torch.cos(yy / 4)).half().float()
nbox = torch.stack([torch.rand(N, generator=g) * 1.2 - 0.1, torch.rand(N, generator=g) * 1.2 - 0.1,
                    torch.rand(N, generator=g) * srange + smin, torch.rand(N, generator=g) * srange + smin], 1)
pres = torch.rand(N, generator=g); depth = torch.rand(N, generator=g) * 4
x = (torch.rand(B, 1, I, I, generator=g) > 0.7).float() * torch.rand(B, 1, I, I, generator=g)
Sd = S.reshape(N, -1).half().contiguous().cuda()
nb, pr, dp, xd = nbox.cuda(), pres.cuda(), depth.cuda(), x.cuda()
ld = P * P * 2
f = lambda v: ctypes.c_float(v)
res = {}
for name in ("taps", "mma", "mix"):
    recon = torch.zeros(B, 1, I, I, device="cuda"); aux = torch.zeros(B, I, I, 2, device="cuda"); part = torch.zeros(B * ((I + 15) // 16) ** 2, device="cuda")
    gl = torch.ones((), device="cuda")
    dlog = torch.zeros(N, ld, device="cuda", dtype=torch.bfloat16)
    dnb, dpr, ddp = torch.zeros(N, 4, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    if name == "taps":
        L.check(L.lib().spair_render_fwd16(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(xd), L.ptr(recon), L.ptr(aux), L.ptr(part), B, HW, 1, I, P, 0, L.stream()), "t")
        L.check(L.lib().spair_render_bwd16(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(aux), L.ptr(gl), L.ptr(dlog), L.ptr(dnb), L.ptr(dpr), L.ptr(ddp), B, HW, 1, I, P, 0, f(2.0), f(0.1), L.stream()), "tb")
    elif name == "mix":      # matrix-core forward, tap backward
        recs = torch.zeros(N * 16, device="cuda", dtype=torch.int32)
        L.check(L.lib().spair_render_prep(L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(recs), B, HW, I, P, 0, L.stream()), "p")
        L.check(L.lib().spair_render_fwd16m(L.ptr(Sd), ld, L.ptr(recs), L.ptr(xd), L.ptr(recon), L.ptr(aux), L.ptr(part), B, HW, 1, I, P, 0, L.stream()), "m")
        L.check(L.lib().spair_render_bwd16(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(aux), L.ptr(gl), L.ptr(dlog), L.ptr(dnb), L.ptr(dpr), L.ptr(ddp), B, HW, 1, I, P, 0, f(2.0), f(0.1), L.stream()), "tb")
    else:
        recs = torch.zeros(N * 16, device="cuda", dtype=torch.int32)
        L.check(L.lib().spair_render_prep(L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(recs), B, HW, I, P, 0, L.stream()), "p")
        L.check(L.lib().spair_render_fwd16m(L.ptr(Sd), ld, L.ptr(recs), L.ptr(xd), L.ptr(recon), L.ptr(aux), L.ptr(part), B, HW, 1, I, P, 0, L.stream()), "m")
        L.check(L.lib().spair_render_bwd16m(L.ptr(Sd), ld, L.ptr(recs), L.ptr(aux), L.ptr(gl), L.ptr(dlog), L.ptr(dnb), L.ptr(dpr), L.ptr(ddp), B, HW, 1, I, P, 0, f(2.0), f(0.1), L.stream()), "mb")
    torch.cuda.synchronize()
    res[name] = dnb.cpu()
for key in ("mix", "mma"):
    print("=== %s vs taps" % key)
    a, b = res[key], res["taps"]
    for c, nm in enumerate(("tx", "ty", "xs", "ys")):
        d = (a[:, c] - b[:, c]).abs(); i = int(d.argmax())
        print("%s: max|taps| %.4g  max err %.4g (rel %.3g) at object %d | rms rel %.3g" % (
            nm, b[:, c].abs().max(), d.max(), d.max() / b[:, c].abs().max(), i, (d.pow(2).mean().sqrt() / b[:, c].pow(2).mean().sqrt())))
a, b = res["mma"], res["taps"]
for c, nm in enumerate(("tx", "ty", "xs", "ys")):
    sel = b[:, c].abs() > 0.02 * b[:, c].abs().max()
    ratio = a[sel, c] / b[sel, c]
    print(nm, "ratio mma/taps over %d objects: quantiles 5/25/50/75/95 %%:" % int(sel.sum()), [round(float(v), 4) for v in torch.quantile(ratio, torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95]))],
          " sum ratio %.5f" % (a[:, c].sum() / b[:, c].sum()), " norm ratio %.5f" % (a[:, c].norm() / b[:, c].norm()))
