"""Device-side evaluation metrics with the reference's function names (spair/metric.py:5-99).

Differences from the reference, on purpose: nothing is mutated in place, the batch size comes from the
tensors instead of ``cfg.BATCH_SIZE``, results stay on the device (0-dim tensors).  The reference's box convention (z_where taken as
top-left x, y + width, height) is kept so that numbers are comparable with it.
"""
import ctypes

import torch

from . import _lib as L
from . import config as cfg


def _f32(t):
    return t.detach().to(dtype=torch.float32).contiguous()


def _both(z_where, z_pres, ground_truth_bbox, truth_bbox_digit_count, image_side=None):
    if not z_where.is_cuda:
        raise L.SpairHipError("metrics run on the GPU (no CPU fallback)")
    B, _, G, _ = z_where.shape
    K = ground_truth_bbox.shape[1]
    I = int(image_side if image_side is not None else cfg.INPUT_IMAGE_SHAPE[-1])
    zw, zp, bb = _f32(z_where), _f32(z_pres), _f32(ground_truth_bbox)
    cnt = _f32(truth_bbox_digit_count.to(z_where.device)).reshape(-1)
    scratch = torch.empty(2 * B, device=z_where.device, dtype=torch.float32)
    out = torch.empty(2, device=z_where.device, dtype=torch.float32)
    L.check(L.lib().spair_metrics(L.ptr(zw), L.ptr(zp), L.ptr(bb), L.ptr(cnt), B, G, I, K, L.ptr(scratch), L.ptr(out), L.stream()), "spair_metrics")
    return out


def mAP(z_where, z_pres, ground_truth_bbox, truth_bbox_digit_count, image_side=None):
    """Mean average precision @ IoU [0.1:0.1:0.9] of the best predicted box per label box (metric.py:5-47).  Scenes without objects
    (0/0 in the reference) are left out of the batch mean; a batch of only such scenes returns NaN."""
    return _both(z_where, z_pres, ground_truth_bbox, truth_bbox_digit_count, image_side)[0]


def object_count_accuracy(z_pres, truth_bbox_digit_count):
    """Mean of (label count - number of cells with round(z_pres) = 1) (metric.py:49-56)."""
    B, _, G, _ = z_pres.shape
    dummy_w = torch.zeros(B, 4, G, G, device=z_pres.device)
    dummy_b = torch.zeros(B, 1, 4, device=z_pres.device)
    return _both(dummy_w, z_pres, dummy_b, truth_bbox_digit_count, 1)[1]


def batch_jaccard(box_a, box_b):
    """IoU of corner-format boxes: [B,A,4] x [B,Bn,4] -> [B,A,Bn] (metric.py:82-99)."""
    if not box_a.is_cuda:
        raise L.SpairHipError("metrics run on the GPU (no CPU fallback)")
    a, b = _f32(box_a), _f32(box_b)
    out = torch.empty(a.shape[0], a.shape[1], b.shape[1], device=a.device, dtype=torch.float32)
    L.check(L.lib().spair_batch_jaccard(L.ptr(a), L.ptr(b), a.shape[0], a.shape[1], b.shape[1], L.ptr(out), L.stream()), "spair_batch_jaccard")
    return out
