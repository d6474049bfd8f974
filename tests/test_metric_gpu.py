"""Device metrics (spair_pytorch_amd/metric.py -> spair_metrics / spair_batch_jaccard) against the reference's own numbers
(tests/golden/metrics.npz, produced by spair/metric.py) and the CPU oracle at BASELINE sizes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["m_b4_g6", "m_b3_g16", "m_b2_g11"])
def test_metrics_match_reference(name, golden_dir):
    from spair_pytorch_amd import metric
    z = np.load(os.path.join(golden_dir, "metrics.npz"))
    B, G, I, K = (int(v) for v in z[name + "/dims"])
    zw, zp = torch.from_numpy(z[name + "/z_where"]).cuda(), torch.from_numpy(z[name + "/z_pres"]).cuda()
    bb, cnt = torch.from_numpy(z[name + "/bbox"]).cuda(), torch.from_numpy(z[name + "/count"]).cuda()
    keep = [t.clone() for t in (zw, zp, bb, cnt)]
    m = metric.mAP(zw, zp, bb, cnt, image_side=I)
    acc = metric.object_count_accuracy(zp, cnt)
    assert m.is_cuda and m.dim() == 0
    for t, k in zip((zw, zp, bb, cnt), keep):
        assert torch.equal(t, k)
    assert abs(m.item() - float(z[name + "/mAP"])) <= 2e-6 * max(1.0, abs(float(z[name + "/mAP"])))
    assert abs(acc.item() - float(z[name + "/count_accuracy"])) <= 1e-5 * max(1.0, abs(float(z[name + "/count_accuracy"])))
    # batch_jaccard on the corner boxes the reference builds
    a = (zw * I).permute(0, 2, 3, 1).reshape(B, -1, 4).clone()
    a[..., 2:] += a[..., :2]
    g = bb.clone()
    g[..., 2:] += g[..., :2]
    iou = metric.batch_jaccard(a, g)
    assert np.allclose(iou.cpu().numpy(), z[name + "/iou"], rtol=2e-6, atol=1e-7, equal_nan=True)


def test_metrics_full_size_vs_oracle():
    """BASELINE config-2 size (B=256, 16x16 grid, 11 label boxes): device result vs the CPU oracle on the same inputs."""
    from oracle import metric_oracle as mo
    from spair_pytorch_amd import metric
    g = torch.Generator().manual_seed(5)
    B, G, I, K = 256, 16, 128, 11
    zw = torch.rand(B, 4, G, G, generator=g) * 0.8
    zw[:, 2:] = torch.rand(B, 2, G, G, generator=g) * 0.3 + 0.05
    zp = torch.rand(B, 1, G, G, generator=g)
    cnt = torch.randint(1, K + 1, (B, 1), generator=g).float()
    bb = torch.zeros(B, K, 4)
    for b in range(B):
        n = int(cnt[b, 0])
        wh = torch.rand(n, 2, generator=g) * 20 + 8
        bb[b, :n, 2:] = wh
        bb[b, :n, :2] = torch.rand(n, 2, generator=g) * (I - wh)
    ref_m, ref_a = mo.mAP(zw, zp, bb, cnt, I), mo.object_count_accuracy(zp, cnt)
    m = metric.mAP(zw.cuda(), zp.cuda(), bb.cuda(), cnt.cuda(), image_side=I)
    a = metric.object_count_accuracy(zp.cuda(), cnt.cuda())
    assert abs(m.item() - ref_m.item()) <= 5e-6 * max(1.0, abs(ref_m.item()))
    assert abs(a.item() - ref_a.item()) <= 1e-5 * max(1.0, abs(ref_a.item()))


def test_map_skips_empty_scenes_and_accepts_flat_counts():
    """The reference divides each sample's AP sum by its object count (metric.py:45): an empty scene is 0/0 there.  The device metric
    leaves empty scenes out of the batch mean; a count tensor of shape [B] means the same as [B,1]."""
    from oracle import metric_oracle as mo
    from spair_pytorch_amd import metric
    g = torch.Generator().manual_seed(9)
    B, G, I, K = 8, 6, 48, 3
    zw = torch.rand(B, 4, G, G, generator=g) * 0.8
    zw[:, 2:] = torch.rand(B, 2, G, G, generator=g) * 0.3 + 0.05
    zp = torch.rand(B, 1, G, G, generator=g)
    cnt = torch.tensor([1, 2, 0, 3, 1, 0, 2, 3]).float()
    bb = torch.zeros(B, K, 4)
    for b in range(B):
        n = int(cnt[b])
        wh = torch.rand(n, 2, generator=g) * 10 + 6
        bb[b, :n, 2:] = wh
        bb[b, :n, :2] = torch.rand(n, 2, generator=g) * (I - wh)
    keep = cnt > 0
    ref = mo.mAP(zw[keep], zp[keep], bb[keep], cnt[keep].view(-1, 1), I)
    m = metric.mAP(zw.cuda(), zp.cuda(), bb.cuda(), cnt.cuda(), image_side=I)            # counts of shape [B]
    assert torch.isfinite(m).all()
    assert abs(m.item() - ref.item()) <= 5e-6 * max(1.0, abs(ref.item()))
    m2 = metric.mAP(zw.cuda(), zp.cuda(), bb.cuda(), cnt.view(-1, 1).cuda(), image_side=I)
    assert m2.item() == m.item()
    # a batch of ONLY empty scenes has no mAP: NaN, as the reference's 0/0 (metric.py:45) -- distinguishable from a true 0
    none = metric.mAP(zw[:2].cuda(), zp[:2].cuda(), torch.zeros(2, K, 4).cuda(), torch.zeros(2).cuda(), image_side=I)
    assert torch.isnan(none).item()
