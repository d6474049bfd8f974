"""CPU oracle for the evaluation metrics -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as spair_oracle.py).

torch-CPU restatement of the reference's spair/metric.py:5-99 without its side effects: the reference scales ``z_where`` and turns
both box sets into corner format IN PLACE (metric.py:15,21-22) and reads the batch size from ``cfg.BATCH_SIZE`` (:12,51); here inputs
are left untouched and the batch size is taken from the tensors.  Parity pin: ``tests/test_oracle_golden.py`` checks these functions
against ``tests/golden/metrics.npz``, produced by the reference's own functions (``tests/golden/make_golden_metrics.py``).
"""
import torch


def intersect(box_a, box_b):
    """metric.py:59-79: intersection areas [B, A, Bn] of corner-format boxes."""
    max_xy = torch.min(box_a[..., 2:].unsqueeze(2), box_b[..., 2:].unsqueeze(1))
    min_xy = torch.max(box_a[..., :2].unsqueeze(2), box_b[..., :2].unsqueeze(1))
    inter = torch.clamp(max_xy - min_xy, min=0)
    return inter[..., 0] * inter[..., 1]


def batch_jaccard(box_a, box_b):
    """metric.py:82-99."""
    inter = intersect(box_a, box_b)
    area_a = ((box_a[..., 2] - box_a[..., 0]) * (box_a[..., 3] - box_a[..., 1])).unsqueeze(2)
    area_b = ((box_b[..., 2] - box_b[..., 0]) * (box_b[..., 3] - box_b[..., 1])).unsqueeze(1)
    return inter / (area_a + area_b - inter)


def corners(z_where, bbox, image_side):
    """metric.py:14-22: [B,4,G,G] image fractions -> [B,HW,4] corner boxes in px; (x,y,w,h) labels -> corners."""
    B = z_where.shape[0]
    zw = (z_where * image_side).permute(0, 2, 3, 1).contiguous().view(B, -1, 4).clone()
    zw[..., 2:] = zw[..., 2:] + zw[..., :2]
    gt = bbox.clone()
    gt[..., 2:] = gt[..., 2:] + gt[..., :2]
    return zw, gt


def mAP(z_where, z_pres, ground_truth_bbox, truth_bbox_digit_count, image_side):
    """metric.py:5-47 (z_pres only enters through the unused masked copy, :27-28)."""
    zw, gt = corners(z_where, ground_truth_bbox, image_side)
    iou = batch_jaccard(zw, gt)                                  # [B, HW, K]
    best = torch.max(iou, dim=1)[0].unsqueeze(-1)                # best prediction per label box
    ap_scale = torch.arange(0.1, 1.0, 0.1)
    scaled = torch.clamp((best - ap_scale) / (1 - ap_scale), min=0, max=1)
    ap = scaled.mean(dim=-1)
    mean_ap = ap.sum(dim=-1, keepdim=True) / truth_bbox_digit_count.view(-1, 1)
    return mean_ap.mean()


def object_count_accuracy(z_pres, truth_bbox_digit_count):
    """metric.py:49-56."""
    B = z_pres.shape[0]
    n = z_pres.permute(0, 2, 3, 1).contiguous().view(B, -1, 1).round().sum(dim=-2)
    return (truth_bbox_digit_count.view(-1, 1) - n).mean()
