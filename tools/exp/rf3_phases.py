#!/usr/bin/env python3
"""[needs tools/exp/patches/render_fwd3_phases.patch + tools/build_variant.sh rfph render2.hip -DRF3_PHASE; SPAIR_HIP_LIB=build/libspair_rfph.so]
Forward renderer: where wave 0 of a tile's workgroup spends its cycles (s_memtime deltas accumulated per phase, 64 sampled tiles)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spair_pytorch_amd import _lib as L, config as cfg, models
from spair_pytorch_amd.data import scattered_digits
cfg.set_grid(128, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)
m = models.SPAIR([1, 128, 128], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
x = torch.from_numpy(scattered_digits(1, 256, 128, 11)[0]).cuda()
with torch.no_grad():
    for _ in range(3):
        m(x, 2000)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * (64 * 16))()
L.lib().spair_rf3_phases(out)
a = np.array(out[:], dtype=np.float64).reshape(64, 16)
a = a[a.sum(1) > 0]
names = ["launch->nbox+cull", "ballot+sync", "compact+sync", "params+sync", "tables+sync", "strip lists", "DMA issue", "DMA wait", "composite", "pass-end sync", "epilogue"]
tot = a[:, :11].sum(1).mean()
print("tiles sampled %d; s_memtime ticks per tile (100 MHz: 10 ns each), mean total %.0f" % (len(a), tot))
for i, n in enumerate(names):
    print("  %-20s %8.1f  (%.1f %%)" % (n, a[:, i].mean(), 100 * a[:, i].mean() / tot))
