"""Data-parallel SPAIR over the GPUs of one node: one process per GPU, replicated 1.46 M-parameter
model, a SUM all-reduce (RCCL over xGMI; gloo in the CPU tests / one-GPU rehearsals) of the flat fp32
gradient buffer, issued in THREE BUCKETS in the order the hand-written backward completes them and
overlapped with the backward kernels that are still running.

Loss scaling: the reference's loss is  sum_batch BCE + mean_batch KL
(models.py:547,553,558).  For S ranks to equal one process on the global batch, each rank
back-propagates  BCE_sum_local + KL_sum_local / (B_local * S)  and the gradients are SUMMED --
plain DDP averaging would silently divide the reconstruction gradient by S.  The model does this
when ``world_size`` is set; ``global_loss`` rebuilds the reported ELBO the same way.

Overlap: ``spair_backward_ev`` (include/spair_hip.h) records one event per gradient range as soon as
that range is final -- decoder first (its weight gradients run on the engine's helper stream under
the per-cell backward chain), then the box / encoder / z / obj nets, last the backbone with the edge
element.  ``allreduce_gradients(model)`` makes a communication stream wait on each event and starts
that range's all-reduce there, so the decoder's 1.7 MB travel while the chain and the backbone
backward still compute; Adam (on the caller's stream) waits on the last collective.  The reference
has no counterpart (train.py:27-30 is single-device).
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib as L

BUCKET_NAMES = ("decoder", "cell_nets", "backbone+edge")


class GradBuckets:
    """The three readiness events + flat-buffer ranges of a model (created by ``attach``)."""

    def __init__(self, model, timing=False):
        lib = L.lib()
        d = model._dims(1)
        lo, hi = (ctypes.c_int64 * 3)(), (ctypes.c_int64 * 3)()
        L.check(lib.spair_grad_buckets(ctypes.byref(d), lo, hi), "spair_grad_buckets")
        self.ranges = [(int(lo[i]), int(hi[i])) for i in range(3)]
        dev = model.device
        with torch.cuda.device(dev):
            self.events = [torch.cuda.Event(enable_timing=bool(timing)) for _ in range(3)]      # timing: the overlap-window measurement only
            for e in self.events:
                e.record()                 # materialises the hipEvent_t behind the torch object
            self.comm = torch.cuda.Stream(device=dev)
            # timing: per bucket, when its all-reduce could start (its range was final) and when it had completed, on the communication stream
            self.t_start = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timing else None
            self.t_done = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timing else None
        self.timing = bool(timing)
        self.pending = False

    def handles(self):
        return [ctypes.c_void_p(e.cuda_event) for e in self.events]


def attach(model, world_size=None, overlap=True, timing=False):
    """Marks ``model`` as one of ``world_size`` replicas (sharded-loss scaling) and, with ``overlap``,
    arms the bucket events its backward records.  ``overlap=False`` (bench.py: SPAIR_DDP_OVERLAP=0) keeps the
    plain path: one all-reduce of the whole flat gradient on the caller's stream after the backward."""
    model.world_size = int(world_size if world_size is not None else dist.get_world_size())
    model._grad_buckets = GradBuckets(model, timing) if (overlap and model.device.type == "cuda" and model.world_size > 1) else None
    return model


def broadcast_parameters(flat_params, src=0):
    dist.broadcast(flat_params, src)


def allreduce_gradients(target):
    """SUM (not mean) over ranks, in place.  ``target`` is a model prepared by ``attach`` (bucketed, overlapped
    with the tail of its backward) or a flat gradient tensor (one bucket, on the current stream)."""
    if isinstance(target, torch.Tensor):
        dist.all_reduce(target, op=dist.ReduceOp.SUM)
        return
    model = target
    flat = model.flat_gradients()
    gb = getattr(model, "_grad_buckets", None)
    if gb is None or not gb.pending:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return
    gb.pending = False
    works = []
    with torch.cuda.stream(gb.comm):
        for i, (ev, (lo, hi)) in enumerate(zip(gb.events, gb.ranges)):
            gb.comm.wait_event(ev)                       # this range's gradients are final
            if gb.timing:
                gb.t_start[i].record(gb.comm)
            w = dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
            if gb.timing:
                w.wait()                                 # the communication stream waits for the collective: `t_done` is its completion
                gb.t_done[i].record(gb.comm)
            works.append(w)
    for w in works:
        w.wait()                                         # the CALLER's stream (Adam) waits for the collectives
    torch.cuda.current_stream().wait_stream(gb.comm)


def bucket_timings(model, backward_end=None):
    """After a synchronised step of a model attached with ``timing=True``: per gradient bucket, the bytes, the time from "range final" to
    "all-reduce complete" on the communication stream and (given the event recorded behind ``loss.backward()``) how long before the end of the
    backward the range was final -- whether the window hides the collective is then one comparison."""
    gb = getattr(model, "_grad_buckets", None)
    if gb is None or not gb.timing:
        return None
    out = []
    for i, (lo, hi) in enumerate(gb.ranges):
        rec = dict(bucket=BUCKET_NAMES[i], bytes=4 * (hi - lo), allreduce_ms=gb.t_start[i].elapsed_time(gb.t_done[i]))
        if backward_end is not None:
            rec["ready_before_backward_end_ms"] = gb.events[i].elapsed_time(backward_end)
            rec["done_after_backward_end_ms"] = backward_end.elapsed_time(gb.t_done[i])
        out.append(rec)
    return out


def global_loss(loss_terms):
    """loss_terms[0..8] of each rank (total, BCE_sum_local, KL_k_local/(B*S)) -> global ELBO terms."""
    t = loss_terms.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
