"""world_size-2 gloo test of the data-parallel contract: each rank back-propagates
BCE_sum_local + KL_sum_local/(B_local*S) on its shard, gradients are SUM-all-reduced, and the result
equals the single-process step on the global batch.  The per-rank step here is the CPU oracle (no GPU
in this container); the collective / loss-scaling code under test is spair_pytorch_amd.ddp."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_inputs as gi
    from helpers import case_noise, case_weights, load_case, oracle_cfg
    from oracle import spair_oracle as orc
    from spair_pytorch_amd import ddp
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z, case = load_case("c1_b8_step1001")
    cfg = oracle_cfg(case)
    B = z["x"].shape[0]
    lo, hi = rank * B // world, (rank + 1) * B // world
    p = case_weights(case, requires_grad=True)
    noise = {k: v[lo:hi] for k, v in case_noise(z).items()}
    out = orc.forward(p, torch.from_numpy(z["x"][lo:hi]), int(z["global_step"]), noise, cfg, kl_scale=1.0 / ((hi - lo) * world))
    out["loss"].backward()
    keys = [k for k in p if p[k].grad is not None]
    flat = torch.cat([p[k].grad.reshape(-1) for k in keys])
    ddp.allreduce_gradients(flat)
    terms = torch.stack([out["loss"].detach(), out["terms"]["recon"].detach()] + [out["terms"]["kl_" + n].detach() for n in
                        ("cy_logit", "cx_logit", "height_logit", "width_logit", "attr", "depth_logit", "pres_dist")])
    g = ddp.global_loss(terms)
    if rank == 0:
        torch.save(dict(flat=flat, keys=keys, sizes=[p[k].numel() for k in keys], terms=g), out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_step_equals_global_step(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_case
    out_path = str(tmp_path / "ddp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    r = torch.load(out_path)
    z, _ = load_case("c1_b8_step1001")
    assert abs(r["terms"][0].item() - float(z["loss"])) <= 1e-5 * float(z["loss"])
    assert abs(r["terms"][1].item() - float(z["recon_loss"])) <= 1e-5 * float(z["recon_loss"])
    off = 0
    for k, n in zip(r["keys"], r["sizes"]):
        g = r["flat"][off:off + n].numpy()
        off += n
        gn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        ref = float(z["gradnorm_" + k])
        assert abs(gn - ref) <= 5e-4 * ref + 1e-7, (k, gn, ref)
