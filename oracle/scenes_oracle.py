"""CPU oracle for the device scene generator -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as spair_oracle.py).

numpy restatement of spair_pytorch_amd/csrc/scenes.hip.  PARITY UNPINNED against the reference: the reference has no generator, it
reads scattered-MNIST scenes from an HDF5 file (spair/dataloader.py:10-36) that is not available; what is kept is the item contract
(image [1,I,I] in [0,1], bbox [K,4] = (x, y, w, h) px zero padded, digit_count).  Integer outputs (count, bbox) must match the device
bit for bit; pixels to fp32 rounding (the device contracts a*b+c into fma).
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def philox(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3, k0, k1 = (np.uint64(int(v) & 0xFFFFFFFF) for v in (c0, c1, c2, c3, k0, k1))
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M32
        n1 = p1 & M32
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M32
        n3 = p0 & M32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return [int(c0), int(c1), int(c2), int(c3)]


def uni(x):
    return np.float32(x >> 8) * np.float32(1.0 / 16777216.0)


def generate(seed, first, B, I, K, smin=14, smax=28):
    f = np.float32
    img = np.zeros((B, 1, I, I), np.float32)
    bbox = np.zeros((B, K, 4), np.float32)
    count = np.zeros((B,), np.int64)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for b in range(B):
        gi = first + b
        g0, g1 = gi & 0xFFFFFFFF, (gi >> 32) & 0xFFFFFFFF
        k = philox(g0, g1, 0, 0, k0, k1)[0] % (K + 1)
        count[b] = k
        for j in range(k):
            r = philox(g0, g1, 1 + j, 0, k0, k1)
            size = min(smin + r[0] % (smax - smin + 1), I)
            y0, x0 = r[1] % (I - size + 1), r[2] % (I - size + 1)
            ns = 2 + (r[3] & 1)
            bbox[b, j] = (x0, y0, size, size)
            n = f(size)
            c = (n - f(1)) * f(0.5)
            yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
            g = np.zeros((size, size), np.float32)
            for s in range(ns):
                a = philox(g0, g1, 1 + j, 1 + 2 * s, k0, k1)
                q = philox(g0, g1, 1 + j, 2 + 2 * s, k0, k1)
                arc = uni(a[0]) < f(0.45)
                vx, vy = f(2) * uni(q[0]) - f(1), f(2) * uni(q[1]) - f(1)
                vn = np.sqrt(vx * vx + vy * vy, dtype=np.float32)
                if vn < f(1e-3):
                    vx, vy = f(1), f(0)
                else:
                    vx, vy = vx / vn, vy / vn
                if arc:
                    cy = c + (f(0.3) * uni(a[1]) - f(0.15)) * n
                    cx = c + (f(0.3) * uni(a[2]) - f(0.15)) * n
                    r0 = (f(0.2) + f(0.22) * uni(a[3])) * n
                    cth = f(1.3) * uni(q[2]) - f(1)
                    dy, dx = yy - cy, xx - cx
                    rr = np.sqrt(dy * dy + dx * dx, dtype=np.float32)
                    v = np.clip(f(1.4) - np.abs(rr - r0) / f(1.2), 0, 1).astype(np.float32)
                    v = np.where(dx * vx + dy * vy >= cth * rr, v, f(0))
                else:
                    cy = c + (f(0.4) * uni(a[1]) - f(0.2)) * n
                    cx = c + (f(0.4) * uni(a[2]) - f(0.2)) * n
                    hl = (f(0.25) + f(0.2) * uni(a[3])) * n
                    dy, dx = yy - cy, xx - cx
                    across, along = np.abs(dy * vx - dx * vy), np.abs(dy * vy + dx * vx)
                    v = np.clip(f(1.4) - across / f(1.2), 0, 1).astype(np.float32)
                    v = np.where(along < hl, v, f(0))
                g = np.maximum(g, v.astype(np.float32))
            img[b, 0, y0:y0 + size, x0:x0 + size] = np.maximum(img[b, 0, y0:y0 + size, x0:x0 + size], g)
    return img, bbox, count
