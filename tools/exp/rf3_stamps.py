#!/usr/bin/env python3
"""[needs `git apply tools/exp/patches/render2_switches.patch` first: the hooks are not in the product source]
Forward renderer: cycles a strip's wave spends issuing sprite DMA, waiting for it, compositing.
Needs the diagnostic build: tools/build_variant.sh rfst render2.hip -DRF3_STAMP; SPAIR_HIP_LIB=build/libspair_rfst.so python tools/exp/rf3_stamps.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spair_pytorch_amd import _lib as L, config as cfg, models
from spair_pytorch_amd.data import scattered_digits
cfg.set_grid(128, (2, 2, 2, 1, 1, 1))
torch.manual_seed(3)
m = models.SPAIR([1, 128, 128], None, torch.device("cuda"), compute_dtype="bf16").to("cuda")
x = torch.from_numpy(scattered_digits(1, 256, 128, 11)[0]).cuda()
for _ in range(3):
    loss = m(x, 2000)[0]
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * (256 * 32))()
L.lib().spair_rf3_stamps(out)
a = np.array(out[:], dtype=np.float64).reshape(256 * 4, 8)
a = a[a[:, 3] > 0]
print("strips sampled %d: chunks/strip %.2f, objects/strip %.1f; cycles per strip: issue %.0f, wait %.0f, composite %.0f; per chunk: issue %.0f wait %.0f comp %.0f" % (
    len(a), a[:, 3].mean(), a[:, 4].mean(), a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(),
    a[:, 0].sum() / a[:, 3].sum(), a[:, 1].sum() / a[:, 3].sum(), a[:, 2].sum() / a[:, 3].sum()))
