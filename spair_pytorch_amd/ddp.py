"""Data-parallel SPAIR over the GPUs of one node: one process per GPU, replicated 1.46 M-parameter
model, ONE collective per step -- a SUM all-reduce (RCCL over xGMI; gloo in the CPU tests) of the
flat fp32 gradient buffer.

Loss scaling: the reference's loss is  sum_batch BCE + mean_batch KL
(models.py:547,553,558).  For S ranks to equal one process on the global batch, each rank
back-propagates  BCE_sum_local + KL_sum_local / (B_local * S)  and the gradients are SUMMED --
plain DDP averaging would silently divide the reconstruction gradient by S.  The model does this
when ``world_size`` is set; ``global_loss`` rebuilds the reported ELBO the same way.
"""
import torch
import torch.distributed as dist


def attach(model, world_size=None):
    model.world_size = int(world_size if world_size is not None else dist.get_world_size())
    return model


def broadcast_parameters(flat_params, src=0):
    dist.broadcast(flat_params, src)


def allreduce_gradients(flat_grad):
    """SUM (not mean) over ranks, in place, one bucket (5.85 MB: latency-bound on xGMI)."""
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)


def global_loss(loss_terms):
    """loss_terms[0..8] of each rank (total, BCE_sum_local, KL_k_local/(B*S)) -> global ELBO terms."""
    t = loss_terms.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
