"""One rank of tests/test_ddp_gpu.py: the HIP training step on this rank's shard of a golden batch, bucketed + overlapped SUM
all-reduce (spair_pytorch_amd.ddp) through gloo (all ranks share cuda:0 on the one-GPU box).  Rank 0 saves the reduced gradients."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    out_path, case_name, dtype = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import golden_inputs as gi
    from helpers import load_case
    from spair_pytorch_amd import config as cfg, ddp
    from spair_pytorch_amd.models import SPAIR
    from spair_pytorch_amd.optim import FusedAdam
    backend = os.environ.get("SPAIR_DIST_BACKEND", "gloo")
    # gloo: every rank on cuda:0 (one-GPU box); nccl (= RCCL): one GPU per rank, only where the box has that many
    dev_index = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    z, case = load_case(case_name)
    cfg.set_grid(case["I"], case["strides"])
    m = SPAIR([1, case["I"], case["I"]], None, torch.device("cuda"), compute_dtype=dtype).to("cuda")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gi.make_weights(case["wseed"], case["wscale"]).items()})
    overlap = os.environ.get("SPAIR_DDP_OVERLAP", "1") != "0"
    ddp.attach(m, world, overlap=overlap)      # sharded-loss scaling + bucket events
    assert (m._grad_buckets is not None) == overlap
    ddp.broadcast_parameters(m.flat_parameters())
    B = z["x"].shape[0]
    lo, hi = rank * B // world, (rank + 1) * B // world
    x = torch.from_numpy(z["x"][lo:hi]).cuda()
    noise = {k: torch.from_numpy(z[k][lo:hi]).cuda() for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}
    opt = FusedAdam(m, lr=1e-4)
    opt.zero_grad()
    loss = m(x, int(z["global_step"]), noise=noise)[0]
    loss.backward()
    assert not overlap or m._grad_buckets.pending
    ddp.allreduce_gradients(m)                 # three buckets, each behind its readiness event, on the communication stream (or one, plain)
    terms = ddp.global_loss(m.loss_terms().clone())
    grads = m.flat_gradients().clone()
    opt.step()
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out_path, grads=grads.cpu().numpy(), terms=terms.cpu().numpy(), params=m.flat_parameters().cpu().numpy(),
                 ranges=np.array(ddp.GradBuckets(m).ranges))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
