"""Debug aid for render3.hip: worst pixels of spair_render_fwd16m against spair_render_fwd16 and the objects that cover them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spair_pytorch_amd import _lib as L
B, G, I, smin, srange = [float(v) if "." in v else int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else "8 4 64 0.08 0.5".split())]
P, HW = 28, G * G
N = B * HW
g = torch.Generator().manual_seed(B + G + I + 2)
logits = torch.randn(N, P, P, 2, generator=g); logits[..., 1] += 1.0
S = torch.sigmoid(logits).half().float()
if os.environ.get("CONST"): S[:] = 0.5
if os.environ.get("RAMP"):
    S[:] = 0.5; S[..., 0] = (torch.arange(28).view(1, 28, 1) * 28 + torch.arange(28).view(1, 1, 28)).float() / 1024
nbox = torch.stack([torch.rand(N, generator=g) * 1.2 - 0.1, torch.rand(N, generator=g) * 1.2 - 0.1,
                    torch.rand(N, generator=g) * srange + smin, torch.rand(N, generator=g) * srange + smin], 1)
pres = torch.rand(N, generator=g); depth = torch.rand(N, generator=g) * 4
if os.environ.get("ONE"):
    keep = int(os.environ["ONE"]); pres[:] = 0; pres[keep] = 1.0
x = (torch.rand(B, 1, I, I, generator=g) > 0.7).float() * torch.rand(B, 1, I, I, generator=g)
Sd = S.reshape(N, -1).half().contiguous().cuda()
nb, pr, dp, xd = nbox.cuda(), pres.cuda(), depth.cuda(), x.cuda()
ld = P * P * 2; nblk = B * ((I + 15) // 16) ** 2
res = {}
for name in ("taps", "mma"):
    recon = torch.zeros(B, 1, I, I, device="cuda"); aux = torch.zeros(B, I, I, 2, device="cuda"); part = torch.zeros(nblk, device="cuda")
    if name == "taps":
        L.check(L.lib().spair_render_fwd16(L.ptr(Sd), ld, L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(xd), L.ptr(recon), L.ptr(aux), L.ptr(part), B, HW, 1, I, P, 0, L.stream()), "t")
    else:
        recs = torch.zeros(N * 16, device="cuda", dtype=torch.int32)
        L.check(L.lib().spair_render_prep(L.ptr(nb), L.ptr(pr), L.ptr(dp), L.ptr(recs), B, HW, I, P, 0, L.stream()), "p")
        L.check(L.lib().spair_render_fwd16m(L.ptr(Sd), ld, L.ptr(recs), L.ptr(xd), L.ptr(recon), L.ptr(aux), L.ptr(part), B, HW, 1, I, P, 0, L.stream()), "m")
        torch.cuda.synchronize()
        R = recs.cpu().view(B, HW, 8)
    torch.cuda.synchronize()
    res[name] = (recon.cpu(), aux.cpu())
d = (res["mma"][1][..., 1] - res["taps"][1][..., 1])          # pre
print("max |d pre|", d.abs().max().item(), "mean", d.mean().item(), "n>1e-3:", int((d.abs() > 1e-3).sum()), "of", d.numel())
bad = (d.abs() > 1e-3).nonzero()
import collections
tiles = collections.Counter((int(b), int(y) // 16, int(xx) // 16) for b, y, xx in bad)
print("bad tiles:", list(tiles.items())[:20])
for b, y, xx in bad[:12]:
    b, y, xx = int(b), int(y), int(xx)
    cov = []
    for k in range(HW):
        xr, yr = int(R[b, k, 6]) & 0xffffffff, int(R[b, k, 7]) & 0xffffffff
        if (xr & 0xffff) <= xx <= (xr >> 16) and (yr & 0xffff) <= y <= (yr >> 16):
            cov.append((k, (xr & 0xffff, xr >> 16), (yr & 0xffff, yr >> 16), [round(float(v), 3) for v in nbox[k * B + b]]))
    print("pixel b%d y%d x%d: taps %.5f mma %.5f cover" % (b, y, xx, res["taps"][1][b, y, xx, 1], res["mma"][1][b, y, xx, 1]), cov)
if os.environ.get("ONE"):
    r = int(os.environ["ONE"]); b = r % B
    dd = d[b]
    ys, xs = dd.abs().gt(1e-4).nonzero(as_tuple=True)
    print("rows with errors:", sorted(set(ys.tolist())), "cols:", sorted(set(xs.tolist())))
    for y in sorted(set(ys.tolist())):
        print(y, [round(float(v), 4) for v in dd[y, 10:40]])
        print("  taps", [round(float(v), 4) for v in res["taps"][1][b, y, 10:40, 1]])
