"""How sparse are the glimpses of the bench workload?  For every (sample, dependency wavefront) and each of ENC0's 25 k-steps (32 consecutive
glimpse pixels), is the k-step zero on ALL cells of the wavefront (-> its weight fragments would not have to be streamed)?
usage (GPU box): python tools/exp/glimpse_sparsity.py [global_step]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from spair_pytorch_amd import config as cfg, models, _lib as L
from spair_pytorch_amd.data import scattered_digits
from spair_pytorch_amd.optim import FusedAdam
step = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg.set_grid(128, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)
B, I, G, P = 256, 128, 16, 28
m = models.SPAIR([1, I, I], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
opt = FusedAdam(m, lr=1e-4)
x = torch.from_numpy(scattered_digits(1, B, I, 11)[0]).cuda()
for it in range(60):      # the bench's own warm-up + timed steps move the boxes away from the init
    opt.zero_grad(); loss, recon, zw, zp = m(x, step); loss.backward(); opt.step()
loss, recon, zw, zp = m(x, step)
# z_where [B,4,G,G] -> normalised boxes as the engine forms them (models.py:364-377): (xt, yt, xs, ys)
zwn = zw.detach().permute(0, 2, 3, 1).reshape(-1, 4)           # rows (b, h, w): cell_x?, per the reference's box order [cell_x, cell_y, width, height]
hh, ww = torch.meshgrid(torch.arange(G), torch.arange(G), indexing="ij")
hh, ww = hh.reshape(-1).cuda().float(), ww.reshape(-1).cuda().float()
cx, cy, wd, ht = zwn[:, 0], zwn[:, 1], zwn[:, 2], zwn[:, 3]
ppc = I / G
xt = (ppc / I) * (cx + ww.repeat(B)); yt = (ppc / I) * (cy + hh.repeat(B))
xs = wd * cfg.ANCHORBOX_SHAPE[0] / I; ys = ht * cfg.ANCHORBOX_SHAPE[0] / I
nbox = torch.stack([xt, yt, xs, ys], 1).contiguous()
N = B * G * G
# unit STN entry: row r reads image r % Bn -> feed images repeated per row order (b-major rows here): use B = N images by index
img = x.repeat_interleave(G * G, 0).contiguous()                # [N,1,I,I]
out = torch.zeros(N, P * P, device="cuda")
L.check(L.lib().spair_stn_glimpse_fwd(L.ptr(img), L.ptr(nbox), N, L.ptr(out), P * P, N, 1, I, P, 0, L.stream()), "stn")
gl = out.view(B, G, G, P * P).to(torch.bfloat16).float()
nzk = torch.zeros(B, G, G, 25, dtype=torch.bool, device="cuda")
for k in range(25):
    nzk[..., k] = (gl[..., k * 32:min(784, (k + 1) * 32)] != 0).any(-1)
print("mean z_pres %.3f; nonzero glimpse pixels %.3f; nonzero (cell, k-step) pairs %.3f" % (zp.mean().item(), (gl != 0).float().mean().item(), nzk.float().mean().item()))
T = 3 * G - 2
tot = live = 0
per_wave = []
for t in range(T):
    cells = [(h, t - 2 * h) for h in range(G) if 0 <= t - 2 * h < G]
    if not cells: continue
    u = torch.zeros(B, 25, dtype=torch.bool, device="cuda")
    for (h, w) in cells: u |= nzk[:, h, w]
    tot += B * 18; live += u[:, :18].sum().item()
    per_wave.append(u[:, :18].float().sum(1).mean().item())
print("streamed k-steps (0..17) that some cell of the wavefront needs: %.3f of all; per wavefront mean %.1f of 18" % (live / tot, np.mean(per_wave)))
# per sample: fragment-load positions of the live-k-step form (n_live of k-steps 6..17 rounded up to even, per wavefront) against the dense 12
dyn = torch.zeros(B, device="cuda")
for t in range(T):
    cells = [(h, t - 2 * h) for h in range(G) if 0 <= t - 2 * h < G]
    if not cells: continue
    u = torch.zeros(B, 25, dtype=torch.bool, device="cuda")
    for (h, w) in cells: u |= nzk[:, h, w]
    n = u[:, 6:18].float().sum(1)
    dyn += torch.ceil(n / 2) * 2
nw = sum(1 for t in range(T) if any(0 <= t - 2 * h < G for h in range(G)))
print("requests per wavefront behind the prefilled six (dense: 12): mean %.2f, median sample %.2f, densest sample %.2f, 90th percentile %.2f" % (
    (dyn / nw).mean().item(), (dyn / nw).median().item(), (dyn / nw).max().item(), (dyn / nw).quantile(0.9).item()))
