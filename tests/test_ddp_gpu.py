"""Two real processes (gloo, both on the one GPU of the box) run the HIP step on the two halves of a golden batch with the bucketed,
overlapped gradient all-reduce of spair_pytorch_amd.ddp; the reduced gradients, the global loss terms and the Adam update must equal
the single-process step on the whole batch.  The N = 8 RCCL run itself is the driver's (bench.py --gpus N)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_inputs as gi
from helpers import assert_adam_updates_close, load_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# (backend, overlapped buckets): gloo with both ranks on the box's one GPU always; RCCL ("nccl") only where the box has two GPUs -- the
# collective then runs on RCCL's internal stream behind the communication stream's events; the plain single-bucket path is the fallback
# bench.py selects with SPAIR_DDP_OVERLAP=0
_MODES = [("gloo", "1"), ("gloo", "0")] + ([("nccl", "1"), ("nccl", "0")] if torch.cuda.device_count() >= 2 else [])


@pytest.mark.parametrize("backend,overlap", _MODES)
@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("bf16", 2e-2)])
def test_two_ranks_overlapped_allreduce_equals_global_batch(tmp_path, dtype, tol, backend, overlap):
    if backend == "gloo" and overlap == "0" and dtype == "f32":
        pytest.skip("covered by the bf16 case")
    name = "c1_b8_step1001"
    out = str(tmp_path / "rank0.npz")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPAIR_DIST_BACKEND=backend,
                   SPAIR_DDP_OVERLAP=overlap, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_gpu_worker.py"), out, name, dtype], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors="replace")[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    r = np.load(out)

    # single process, whole batch, same weights / image / noise
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.models import SPAIR
    from spair_pytorch_amd.optim import FusedAdam
    z, case = load_case(name)
    cfg.set_grid(case["I"], case["strides"])
    m = SPAIR([1, case["I"], case["I"]], None, torch.device("cuda"), compute_dtype=dtype).to("cuda")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gi.make_weights(case["wseed"], case["wscale"]).items()})
    opt = FusedAdam(m, lr=1e-4)
    opt.zero_grad()
    x = torch.from_numpy(z["x"]).cuda()
    noise = {k: torch.from_numpy(z[k]).cuda() for k in ("eps_box", "eps_attr", "eps_depth", "u_pres")}
    loss = m(x, int(z["global_step"]), noise=noise)[0]
    loss.backward()
    g = m.flat_gradients().cpu().numpy().astype(np.float64)
    t = m.loss_terms().cpu().numpy()
    opt.step()
    p = m.flat_parameters().cpu().numpy()
    assert np.allclose(r["terms"][:9], t[:9], rtol=max(tol, 1e-5), atol=1e-4), (r["terms"][:9], t[:9])
    covered = np.zeros(g.size, bool)
    for lo, hi in r["ranges"]:
        assert not covered[lo:hi].any()
        covered[lo:hi] = True
        ref = g[lo:hi]
        d = np.abs(r["grads"][lo:hi] - ref).max()
        assert d <= tol * np.abs(ref).max() + 1e-7, (int(lo), int(hi), d, np.abs(ref).max())
    assert np.abs(g[~covered]).max() == 0.0            # only the grad-less attn.* slots are outside the buckets
    # after Adam the replicas' parameters equal the single-process ones (the update is sign-like: compare loosely in bf16)
    if dtype == "f32":
        assert_adam_updates_close(r["params"], p, 1e-4, tight=1e-6)
    else:
        assert np.abs(r["params"] - p).max() <= 2.1e-4


def test_bench_runs_under_torchrun_with_two_ranks():
    """The driver's multi-GPU launch form of bench.py (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`), rehearsed with
    two ranks on the box's one GPU (SPAIR_DIST_BACKEND=gloo; a smaller batch keeps it to seconds): rank 0 must print ONE JSON line with the
    whole-job throughput, and every rank must exit -- after the timed region rank 0 alone finishes the record (kernel table, the STN stage
    stamps), so nothing there may contain a collective."""
    import json
    env = dict(os.environ, SPAIR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["config"]["parallelism"] == "dp2"
    assert rec["config"]["global_batch"] == 64 and rec["value"] > 0 and np.isfinite(rec["elbo"])
    assert abs(rec["value"] - 64 * 3 / (rec["ms_per_step"] * 3e-3)) < 1e-6 * rec["value"]          # whole-job images/s = world * B * K / time
    assert "stn_fwd" in rec["kernels"] and rec["roofline"]["kernel"] in rec["kernels"]
    # the per-bucket all-reduce record the first real scaling run is read by: three buckets in readiness order, each with its bytes, the
    # range-final -> collective-complete time and its position against the end of the backward
    d = rec["ddp"]
    assert d["overlap"] and [b["bucket"] for b in d["buckets"]] == ["decoder", "cell_nets", "backbone+edge"]
    assert sum(b["bytes"] for b in d["buckets"]) > 5.8e6 and all(b["allreduce_ms"] > 0 for b in d["buckets"])
    assert d["buckets"][0]["ready_before_backward_end_ms"] >= d["buckets"][1]["ready_before_backward_end_ms"] >= 0.0
