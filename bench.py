#!/usr/bin/env python3
"""SPAIR train-step benchmark on MI355X (contract: see the driver's bench.py spec).

Workload = BASELINE.json configs[1]: 128x128 scattered-digit scenes (<= 11 objects), 16x16 cell grid
(backbone strides 2,2,2 -> 8-px cells), batch 256 per GPU, bf16 GEMM/conv operands, fp32 elsewhere.
A "step" = zero_grad + forward + backward + (gradient all-reduce if N>1) + Adam, input resident in HBM,
global_step >= 2000 (training wheel off: every gradient path live, nothing skipped).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}
SLOT_NAMES = ["prep", "backbone_fwd", "cells_fwd", "decoder_fwd", "count_kl", "render_fwd", "kl_loss", "render_bwd",
              "decoder_bwd", "cells_bwd", "cells_wgrad", "backbone_bwd", "conv1_fwd", "dec_out_fwd", "stn_fwd", "adam",
              "dec_out_wgrad", "dec_out_dgrad"]


# Speed of the oracle relative to the REAL reference, measured in the build container (8 vCPU, torch 2.10 CPU) where both run:
# reference 55.7 img/s (48x48, 6x6 grid, B=16) and 3.9 img/s (128x128, 16x16 grid, B=16) -- BASELINE.md section 2 -- against the oracle's
# 64.7 and 4.55 img/s on the same machine and thread count.  The reference itself never travels to the GPU box.
ORACLE_OVER_REFERENCE = {"config1_48px_b16": 64.7 / 55.7, "config2_128px_b16": 4.55 / 3.9}


def _oracle_steps(image_side, strides, batch, kmax, steps, threads):
    from oracle import spair_oracle as orc
    import golden_inputs as gi
    from spair_pytorch_amd.data import scattered_digits
    torch.set_num_threads(threads)
    cfg = orc.OracleConfig(image_shape=(1, image_side, image_side), conv_strides=tuple(strides))
    p = {k: torch.from_numpy(v).clone().requires_grad_(not k.startswith("attn.")) for k, v in gi.make_weights(3, 1.0).items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    x = torch.from_numpy(scattered_digits(99, batch, image_side, kmax)[0])
    G = gi.grid_side(image_side, strides)
    times = []
    for it in range(steps + 1):
        noise = {k: torch.from_numpy(a) for k, a in gi.make_noise(it, batch, G).items()}
        t0 = time.perf_counter()
        for t in p.values():
            t.grad = None
        out = orc.forward(p, x, 2000 + it, noise, cfg, fast=True)
        out["loss"].backward()
        with torch.no_grad():
            orc.adam_step({k: t for k, t in p.items()}, {k: t.grad for k, t in p.items()}, m, v2, it + 1)
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / steps
    return dict(images_per_sec=batch / dt, s_per_step=dt, batch=batch, timed_steps=steps, threads=torch.get_num_threads())


def cpu_baseline(strides):
    """The CPU oracle (oracle/spair_oracle.py, a restatement of the reference's PyTorch path, pinned to it by tests/golden) timed on
    this box's host cores, as SURVEY 8(d) / BASELINE.md section 3 prescribe: BASELINE config 1 (48x48, 6x6 grid, B=16) and config 2's
    geometry (128x128, 16x16 grid) at B=16, 5 timed steps after 1 warm-up on all cores, plus 1-core runs (bounded: the 128x128
    1-core run uses B=4 and 2 timed steps).  Reported baseline only -- never the product, never the optimisation target."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    ncpu = min(16, len(os.sched_getaffinity(0)))       # the GPU box's CPU share (16 per GPU)
    runs = {
        "config1_48px_b16": _oracle_steps(48, strides, 16, 3, 5, ncpu),
        "config1_48px_b16_1core": _oracle_steps(48, strides, 16, 3, 5, 1),
        "config2_128px_b16": _oracle_steps(128, strides, 16, 11, 5, ncpu),
        "config2_128px_b4_1core": _oracle_steps(128, strides, 4, 11, 2, 1),
    }
    main = runs["config2_128px_b16"]
    ratio = {k: ORACLE_OVER_REFERENCE[k] for k in ORACLE_OVER_REFERENCE}
    return dict(value=main["images_per_sec"], unit="images/sec", cores=main["threads"], kind="port",
                sample="oracle fwd+bwd+Adam on the bench workload's geometry (128x128, 16x16 grid), batch 16, 5 timed steps after 1 warm-up "
                       "(%.2f s/step) on %d threads; nproc=%d" % (main["s_per_step"], main["threads"], len(os.sched_getaffinity(0))),
                runs=runs, ratio_to_reference=ratio,
                ratio_note="oracle img/s divided by the real reference's img/s, both measured in the build container (8 vCPU); "
                           "divide a `runs` figure by it to estimate the reference on this box")


def hbm_copy_rates(dev, mb=1024):
    """Stream rates of this box (the measured copy peak beside the vendor figure): a 1 GiB device copy counted as
    read + written bytes, and a fill (writes only)."""
    n = mb * (1 << 20) // 4
    a, b = torch.empty(n, device=dev), torch.empty(n, device=dev)
    a.fill_(1.0)

    def rate(fn, nbytes, reps=10):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9

    return dict(copy_GBs=rate(lambda: b.copy_(a), 2 * n * 4), fill_GBs=rate(lambda: b.fill_(2.0), n * 4), unit="GB/s", bytes=n * 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--image", type=int, default=128)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--global-step", type=int, default=2000, help="global_step of the first timed step (count-prior / wheel schedules)")
    ap.add_argument("--sweep", action="store_true",
                    help="BASELINE configs[4]: after the main measurement, time the step at global_step in {0,2000,4000,6000,7000,8000,10000}")
    ap.add_argument("--sweep-steps", type=int, default=15)
    ap.add_argument("--prof-every", type=int, default=4, help="record the per-kernel HIP event pairs on every n-th timed step")
    ap.add_argument("--prof-mask", type=lambda v: int(v, 0), default=-1, help="bit mask of the engine's event-pair slots to record (-1 = all)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # SPAIR_DIST_BACKEND=gloo: rehearsal of the N > 1 code path on a one-GPU box (all ranks on cuda:0, collectives through gloo)
    backend = os.environ.get("SPAIR_DIST_BACKEND", "nccl")
    if backend != "nccl" and torch.cuda.device_count() < world:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from spair_pytorch_amd import _lib as L
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd import ddp
    from spair_pytorch_amd.data import scattered_digits
    from spair_pytorch_amd.models import SPAIR
    from spair_pytorch_amd.optim import FusedAdam

    strides = (2, 2, 2, 1, 1, 1)
    cfg.set_grid(args.image, strides)
    torch.manual_seed(3)                                   # train.py:39
    model = SPAIR([1, args.image, args.image], None, dev, compute_dtype=args.dtype).to(dev)
    ddp.attach(model, world)
    if world > 1:
        ddp.broadcast_parameters(model.flat_parameters())
    opt = FusedAdam(model, lr=1e-4)
    B = args.batch
    x = torch.from_numpy(scattered_digits(1234 + rank, B, args.image, 11)[0]).to(dev)   # resident in HBM
    torch.manual_seed(7 + rank)                            # noise seed, per rank
    gstep = [args.global_step]
    last = {}

    def step():
        opt.zero_grad()
        loss, recon, z_where, z_pres = model(x, gstep[0])
        last["z_pres"] = z_pres
        loss.backward()
        if world > 1:
            ddp.allreduce_gradients(model)       # three buckets behind the backward's readiness events, on a communication stream
        opt.step()
        gstep[0] += 1
        return loss

    for _ in range(args.warmup):
        step()
    lib = L.lib()
    L.check(lib.spair_prof_select(ctypes.c_ulonglong(args.prof_mask & 0xFFFFFFFFFFFFFFFF)), "prof_select")
    L.check(lib.spair_prof_enable(1), "prof_enable")
    lib.spair_prof_enable(0)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # the kernels' event pairs are sampled on every n-th step of the timed region (each pair costs ~3 us of queue time)
        lib.spair_prof_enable(2 if i % args.prof_every == 0 else 0)
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    nslots = len(SLOT_NAMES)
    ms = (ctypes.c_float * nslots)()
    cnt = (ctypes.c_int * nslots)()
    L.check(lib.spair_prof_read(ms, cnt, nslots), "prof_read")
    lib.spair_prof_enable(0)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    terms = model.loss_terms().clone()
    if world > 1:
        terms = ddp.global_loss(terms)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    K = args.steps
    Ks = len(range(0, K, args.prof_every))                                              # steps on which the event pairs were recorded
    per_step_ms = {SLOT_NAMES[i]: ms[i] / Ks for i in range(nslots) if cnt[i] > 0}
    avg = {SLOT_NAMES[i]: ms[i] / cnt[i] for i in range(nslots) if cnt[i] > 0}       # per launch / region
    d = model._last["engine"]["dims"]
    N = B * d.G * d.G
    P2 = d.P * d.P
    sprite_b = N * P2 * 2 * (2 if args.dtype == "bf16" else 4)    # (grey, alpha) sprites: bf16 pairs in the bf16 step, fp32 otherwise
    render_fwd_bytes = sprite_b + N * 6 * 4 + B * d.I * d.I * 4          # 429.4 MB at config 2
    render_bwd_bytes = render_fwd_bytes + N * P2 * 2 * (2 if args.dtype == "bf16" else 4)   # + the d-logits write
    peak_f = MFMA_PEAK_TFLOPS["bf16" if args.dtype == "bf16" else "f32"]
    c1 = [c for c in [(d.conv_k[1], d.conv_s[1], d.conv_c[1])]][0]
    h0 = (d.I + d.pad_pre + d.pad_post - d.conv_k[0]) // d.conv_s[0] + 1
    h1 = (h0 - c1[0]) // c1[1] + 1
    conv1_flop = 2.0 * B * h1 * h1 * c1[2] * (c1[0] * c1[0] * d.conv_c[0])
    dec_out_flop = 2.0 * N * 256 * (P2 * 2)
    # per-cell chain (k_chain_fwd / k_chain_bwd): arithmetic intensity ~50 flop/B, far left of the ridge (312 flop/B),
    # so HBM is the roofline that bounds it.  Algorithmic bytes per row: what the kernel MUST move --
    # forward: every layer input it has to keep for the weight-gradient GEMMs + latents/records + the glimpse derivative pairs.
    A, NPc, Fc = d.A, d.NP, d.F
    REC = 4 + A + 2                                                        # record [box4 | attr A | depth | pres]
    box_in = Fc + 4 * REC                                                  # features + 4 neighbour records
    z_in, glim = box_in + NPc + 4 + A, P2
    o_in = z_in + 1
    hid = 2 * 100 + (256 + 128) + 2 * 100 + 2 * 100                       # relu outputs of the four nets
    if args.dtype == "bf16":
        # the fused chain stores every GEMM operand as bf16, once: [features|context|box|attr|depth] (o_in columns, shared by the
        # box/z/obj first layers), the glimpse, the relu outputs; the glimpse derivative pairs as bf16x2; fp32: the 308-float
        # per-row bundle the elementwise backward reads and the record; relu sign bits; + (features, noise) loads
        fwd_row = 2 * (o_in + glim + hid + A) + 4 * glim + 4 * (308 + REC) + 66 * 4 * 8 // 8 + 4 * (Fc + REC + 2)
    else:
        fwd_row = 4 * (box_in + glim + z_in + o_in + hid + REC + glim) + 4 * (Fc + REC + 2)    # fp32 stores + (features, noise) loads
    outs = (100 + 100 + 8 + NPc) + (256 + 128 + 2 * A) + (100 + 100 + 2 + NPc) + (100 + 100 + 1)
    # backward: the 308-float per-row bundle + the glimpse derivative pairs + relu sign bits (66 tiles x 4 x 8 B per <= 8 rows) in;
    # layer-output gradients and d feat out as bf16
    bwd_row = 4 * (308 + glim) + 66 * 4 * 8 // 8 + 2 * (outs + Fc)
    chain_flop = 2.0 * N * (box_in * 100 + 100 * 100 + 100 * (8 + NPc) + glim * 256 + 256 * 128 + 128 * 2 * A
                            + z_in * 100 + 100 * 100 + 100 * (2 + NPc) + o_in * 100 + 100 * 100 + 100)
    kernels = {}

    def add(name, bound, work, unit_scale, peak, unit, slot=None, **extra):
        slot = slot or name
        if slot in avg and avg[slot] > 0:
            ach = work / (avg[slot] * 1e-3) / unit_scale
            kernels[name] = dict(bound=bound, achieved=ach, peak=peak, unit=unit, frac=ach / peak, avg_ms=avg[slot], traffic=None, **extra)

    add("chain_bwd", "hbm", N * bwd_row, 1e9, HBM_PEAK_GBS, "GB/s", slot="cells_bwd", mfma_tflops=chain_flop / (avg.get("cells_bwd", 1) * 1e-3) / 1e12)
    add("chain_fwd", "hbm", N * fwd_row, 1e9, HBM_PEAK_GBS, "GB/s", slot="cells_fwd", mfma_tflops=chain_flop / (avg.get("cells_fwd", 1) * 1e-3) / 1e12)
    add("render_fwd", "hbm", render_fwd_bytes, 1e9, HBM_PEAK_GBS, "GB/s")
    add("render_bwd", "hbm", render_bwd_bytes, 1e9, HBM_PEAK_GBS, "GB/s")
    add("conv1_fwd", "mfma", conv1_flop, 1e12, peak_f, "TFLOP/s")
    add("dec_out_fwd", "mfma", dec_out_flop, 1e12, peak_f, "TFLOP/s")
    add("dec_out_dgrad", "mfma", dec_out_flop, 1e12, peak_f, "TFLOP/s")
    # (decoder / per-cell weight gradients run on the helper stream beside the chain: their event times include the
    #  overlap and are not per-kernel durations -- see profiles/ for the rocprofv3 kernel stats)
    # HBM traffic per launch from the committed PMC passes of this same command (two separate rocprofv3 --pmc runs; FETCH_SIZE doubled
    # as MI355X_MICROARCH.md prescribes for gfx950).  Only valid for the default workload it was collected on.
    pmc_path = os.path.join(ROOT, "profiles", "r02_d_pmc_traffic.json")
    if os.path.exists(pmc_path) and B == 256 and args.image == 128 and args.dtype == "bf16":
        pmc = json.load(open(pmc_path))
        for name, rec in kernels.items():
            if name in pmc:
                rec["traffic"] = (pmc[name]["read_MB"] + pmc[name]["write_MB"]) * 1e6
                rec["traffic_source"] = "profiles/r02_d_pmc_traffic.json"
    dominant = max(kernels, key=lambda k: kernels[k]["avg_ms"]) if kernels else None
    roof = dict(kernels[dominant], kernel=dominant) if dominant else None

    if args.image == 128 and B == 256 and d.G == 16:
        which = "BASELINE configs[1]" if world == 1 else "BASELINE configs[2] (configs[1] per GPU, DDP over %d GPUs)" % world
    elif args.image == 256 and B == 64 and d.G == 32:
        which = "BASELINE configs[3] (STN gather stress)"
    else:
        which = "custom (not a BASELINE config)"
    workload = ("%s: %dx%d synthetic scattered digits (<=11), %dx%d grid, batch %d/GPU, fwd+bwd+Adam, global_step %d+ (wheel %s)"
                % (which, args.image, args.image, d.G, d.G, B, args.global_step, "off" if args.global_step >= 1000 else "on"))
    out = dict(metric="SPAIR train images/sec + ELBO, 128x128 scattered-MNIST, batch 256", value=world * B * K / dt, unit="images/sec",
               n_gpus=world, steps=K, warmup=args.warmup, ms_per_step=dt / K * 1e3, higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype=args.dtype, data="synthetic",
               config=dict(workload=workload, global_batch=world * B, image=args.image, grid=d.G, global_step=args.global_step,
                           parallelism="dp%d" % world),
               elbo=float(terms[0].item()), elbo_terms=[float(v) for v in terms[:9].tolist()],
               roofline=roof, kernels=kernels, step_breakdown_ms=per_step_ms)
    if args.sweep and world == 1:
        # BASELINE configs[4] on one GPU: the count-prior schedule (config.py:65-69, models.py:186-188) changes z_pres and with it the
        # renderer's active-cell density; same model state, only global_step differs between the points
        from spair_pytorch_amd.models import step_scalars
        sweep = []
        snap = (model.flat_parameters().clone(), opt.state_dict())
        snap = (snap[0], {k: (v.clone() if torch.is_tensor(v) else v) for k, v in snap[1].items()})
        for gs in (0, 2000, 4000, 6000, 7000, 8000, 10000):
            model.flat_parameters().copy_(snap[0])         # every point starts from the same model and optimizer state
            opt.load_state_dict(snap[1])
            gstep[0] = gs
            step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.sweep_steps):
                step()
            torch.cuda.synchronize()
            ms_s = (time.perf_counter() - t1) / args.sweep_steps * 1e3
            st_ = step_scalars(gs, B)
            sweep.append(dict(global_step=gs, count_prior_prob=float(st_.count_prior_prob), wheel=float(st_.wheel), ms_per_step=ms_s,
                              images_per_sec=B / ms_s * 1e3, mean_z_pres=float(last["z_pres"].mean().item())))
        out["sweep"] = sweep
    out["hbm_measured"] = hbm_copy_rates(dev)      # what this box sustains, beside the vendor 8 TB/s the roofline divides by
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(strides)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
