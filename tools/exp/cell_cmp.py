"""Where do the fused and the per-wavefront backward differ most, cell by cell?  usage: python tools/exp/cell_cmp.py I B"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden")
import golden_inputs as gi
from spair_pytorch_amd import config as cfg, models
from spair_pytorch_amd.data import scattered_digits
I, B = int(sys.argv[1]), int(sys.argv[2])
strides = (2, 2, 2, 1, 1, 1)
cfg.set_grid(I, strides)
G = gi.grid_side(I, strides)
x = torch.from_numpy(scattered_digits(7 + B, B, I, 9)[0]).cuda()
noise = {k: torch.from_numpy(v).cuda() for k, v in gi.make_noise(3 + B, B, G).items()}
w = {k: torch.from_numpy(v) for k, v in gi.make_weights(21, 1.0).items()}
maps = {}
for flags in (0, 1):
    models.STEP_FLAGS = flags
    m = models.SPAIR([1, I, I], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
    m.load_state_dict(w); m.zero_grad()
    m(x, 2500, noise=noise)[0].backward()
    maps[flags] = [m.export_map((200 if flags == 0 else 100) + k).clone() for k in range(4)]
a, b = maps[0][1].double(), maps[1][1].double()
err, ref = (a - b).norm(dim=1), b.norm(dim=1)
ratio = err / (6e-2 * ref + 2e-3 * ref.max())
top = torch.topk(ratio.flatten(), 12)
for v, i in zip(top.values.tolist(), top.indices.tolist()):
    bb, h, ww = i // (G * G), (i // G) % G, i % G
    print("ratio %.2f  sample %d cell (%d, %d)  |ref| %.3e (max %.3e)  err %.3e  rel %.3f" % (v, bb, h, ww, ref[bb, h, ww], ref.max(), err[bb, h, ww], err[bb, h, ww] / ref[bb, h, ww]))
print("rows of the 40 worst cells:", sorted(((i // G) % G) for i in torch.topk(ratio.flatten(), 40).indices.tolist()))
