#!/usr/bin/env python3
"""Golden vectors for the evaluation metrics from the reference's own spair/metric.py.

Runs only in the build container (imports /root/reference); tests read the .npz it writes.  The reference functions mutate their
arguments in place and read cfg.BATCH_SIZE / cfg.INPUT_IMAGE_SHAPE (metric.py:11-12,15,21-22,51): inputs are cloned and the config
is set per case.  Usage: python tests/golden/make_golden_metrics.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def cases():
    rng = np.random.default_rng(20240607)
    out = {}
    for name, (B, G, I, K) in {"m_b4_g6": (4, 6, 48, 5), "m_b3_g16": (3, 16, 128, 11), "m_b2_g11": (2, 11, 128, 7)}.items():
        z_where = rng.uniform(0.0, 0.9, (B, 4, G, G)).astype(np.float32)
        z_where[:, 2:] = rng.uniform(0.05, 0.4, (B, 2, G, G)).astype(np.float32)
        z_pres = rng.uniform(0.0, 1.0, (B, 1, G, G)).astype(np.float32)
        z_pres[0, 0, 0, 0] = 0.5                                      # torch.round: half to even
        z_pres[0, 0, 0, 1] = 1.5 - 1.0
        count = rng.integers(1, K + 1, (B, 1)).astype(np.float32)
        bbox = np.zeros((B, K, 4), np.float32)                         # (x, y, w, h) px, zero padded
        for b in range(B):
            for j in range(int(count[b, 0])):
                w, h = rng.uniform(8, 28, 2)
                bbox[b, j] = (rng.uniform(0, I - w), rng.uniform(0, I - h), w, h)
        out[name] = dict(B=B, G=G, I=I, K=K, z_where=z_where, z_pres=z_pres, bbox=bbox, count=count)
    return out


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import torch
    from spair import config as cfg
    from spair import metric
    res = {}
    for name, c in cases().items():
        cfg.BATCH_SIZE = c["B"]
        cfg.INPUT_IMAGE_SHAPE[:] = [1, c["I"], c["I"]]
        t = {k: torch.from_numpy(c[k].copy()) for k in ("z_where", "z_pres", "bbox", "count")}
        m = metric.mAP(t["z_where"].clone(), t["z_pres"].clone(), t["bbox"].clone(), t["count"].clone())
        acc = metric.object_count_accuracy(t["z_pres"].clone(), t["count"].clone())
        # batch_jaccard on corner-format boxes exactly as mAP builds them (metric.py:14-22)
        zw = (t["z_where"].clone() * c["I"]).permute(0, 2, 3, 1).contiguous().view(c["B"], -1, 4)
        zw[..., 2:] += zw[..., :2]
        gt = t["bbox"].clone()
        gt[..., 2:] += gt[..., :2]
        iou = metric.batch_jaccard(zw, gt)
        for k in ("z_where", "z_pres", "bbox", "count"):
            res["%s/%s" % (name, k)] = c[k]
        res["%s/dims" % name] = np.array([c["B"], c["G"], c["I"], c["K"]], np.int64)
        res["%s/mAP" % name] = np.array(float(m), np.float64)
        res["%s/count_accuracy" % name] = np.array(float(acc), np.float64)
        res["%s/iou" % name] = iou.numpy()
        print(name, "mAP", float(m), "count_accuracy", float(acc))
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **res)


if __name__ == "__main__":
    main()
