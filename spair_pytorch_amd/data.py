"""Synthetic scattered-digits scenes (MNIST is not available offline).  Mirrors the 3-tuple of
/root/reference/spair/dataloader.py:33: (image [C,I,I] float in [0,1], bbox [max_objects,4] =
(x, y, w, h) px zero-padded, digit_count)."""
import numpy as np
import torch


def _glyph(rng, size):
    """A digit-like stroke pattern on a size x size canvas, anti-aliased, values in [0,1]."""
    n = int(np.ceil(size))
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float32)
    c = (n - 1) / 2.0
    g = np.zeros((n, n), np.float32)
    for _ in range(int(rng.integers(2, 4))):
        kind = rng.uniform()
        if kind < 0.45:   # arc
            r0 = rng.uniform(0.2, 0.42) * n
            cy, cx = c + rng.uniform(-0.15, 0.15) * n, c + rng.uniform(-0.15, 0.15) * n
            r = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
            ang = np.arctan2(yy - cy, xx - cx)
            a0, span = rng.uniform(-np.pi, np.pi), rng.uniform(0.8, 2.0) * np.pi
            inarc = ((ang - a0) % (2 * np.pi)) < span
            g = np.maximum(g, np.clip(1.4 - np.abs(r - r0) / 1.2, 0, 1) * inarc)
        else:             # bar
            ang = rng.uniform(0, np.pi)
            cy, cx = c + rng.uniform(-0.2, 0.2) * n, c + rng.uniform(-0.2, 0.2) * n
            d = np.abs((yy - cy) * np.cos(ang) - (xx - cx) * np.sin(ang))
            along = np.abs((yy - cy) * np.sin(ang) + (xx - cx) * np.cos(ang))
            g = np.maximum(g, np.clip(1.4 - d / 1.2, 0, 1) * (along < rng.uniform(0.25, 0.45) * n))
    return g


def scattered_digits(seed, B, I, max_objects, obj_px=(14, 28), channels=1):
    rng = np.random.default_rng(seed)
    img = np.zeros((B, channels, I, I), np.float32)
    bbox = np.zeros((B, max_objects, 4), np.float32)
    count = np.zeros((B,), np.int64)
    for b in range(B):
        k = int(rng.integers(0, max_objects + 1))
        count[b] = k
        for j in range(k):
            size = int(rng.integers(obj_px[0], obj_px[1] + 1))
            size = min(size, I)
            y0, x0 = int(rng.integers(0, I - size + 1)), int(rng.integers(0, I - size + 1))
            g = _glyph(rng, size)[:size, :size]
            for c in range(channels):
                img[b, c, y0:y0 + size, x0:x0 + size] = np.maximum(img[b, c, y0:y0 + size, x0:x0 + size], g)
            bbox[b, j] = (x0, y0, size, size)
    return img, bbox, count


class SyntheticScatteredDigits(torch.utils.data.Dataset):
    """Same item contract as SimpleScatteredMNISTDataset (dataloader.py:10-36), generated up front."""

    def __init__(self, n, image_side=128, max_objects=11, seed=1234):
        self.image, self.bbox, self.count = scattered_digits(seed, n, image_side, max_objects)

    def __getitem__(self, i):
        return self.image[i], self.bbox[i], self.count[i]

    def __len__(self):
        return self.image.shape[0]


class DeviceScatteredDigits:
    """Batches of synthetic scenes generated on the GPU (csrc/scenes.hip): same item contract as SimpleScatteredMNISTDataset
    (dataloader.py:10-36) -- ``(image [B,1,I,I] float in [0,1], bbox [B,K,4] = (x, y, w, h) px zero padded, digit_count [B])`` -- but as
    device tensors, one launch pair per batch, deterministic in (seed, sample index).  Iterating yields ``len(self)`` batches per epoch;
    ``batch(i)`` regenerates batch i of the stream directly (any rank, any epoch: rank r of W reads batches r, r+W, ...)."""

    def __init__(self, n_samples, batch_size, image_side=128, max_objects=11, seed=1234, obj_px=(14, 28), device="cuda", rank=0, world=1):
        self.n, self.B, self.I, self.K, self.seed, self.obj_px = int(n_samples), int(batch_size), int(image_side), int(max_objects), int(seed), obj_px
        self.device, self.rank, self.world = torch.device(device), int(rank), int(world)
        self._scratch = None

    def __len__(self):
        return self.n // (self.B * self.world)

    def batch(self, i, epoch=0):
        from . import _lib as L
        import ctypes
        if self.device.type != "cuda":
            raise L.SpairHipError("DeviceScatteredDigits generates on the GPU (no CPU fallback)")
        first = (epoch * len(self) * self.world + i * self.world + self.rank) * self.B
        img = torch.empty(self.B, 1, self.I, self.I, device=self.device)
        bbox = torch.empty(self.B, self.K, 4, device=self.device)
        cnt = torch.empty(self.B, dtype=torch.int64, device=self.device)
        if self._scratch is None:
            self._scratch = torch.empty(self.B * self.K * 28, device=self.device)
        L.check(L.lib().spair_scenes_generate(ctypes.c_uint64(self.seed), ctypes.c_longlong(first), self.B, self.I, self.K, int(self.obj_px[0]),
                                              int(self.obj_px[1]), L.ptr(img), L.ptr(bbox), L.ptr(cnt), L.ptr(self._scratch), L.stream()),
                "spair_scenes_generate")
        return img, bbox, cnt

    def __iter__(self):
        for i in range(len(self)):
            yield self.batch(i)
