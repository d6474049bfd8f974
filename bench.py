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
import math
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


# fragment packs the chain kernels stream from L2 on EVERY wavefront (engine.hip carve(): tiles x k-steps x 1 KiB per layer)
# (forward: the box network's three layers stream a hi AND a lo pack -- split-bf16 products --, the one-column obj output layer none: it is
#  folded into OBJ1's epilogue; 7 of encoder layer 0's 25 k-steps are register-resident for the whole kernel (chain.hip ENC0_RES) and are
#  not streamed per wavefront.  backward: real tiles only -- surplus waves of a layer's last round stream nothing)
CHAIN_PACK_KIB = dict(fwd=sum(a * b for a, b in zip((14, 14, 14, 16, 8, 7, 7, 7, 7, 7, 7), (11, 4, 4, 25 - 7, 8, 4, 16, 4, 4, 16, 4))),
                      bwd=sum(a * b for a, b in zip((21, 7, 7, 49, 16, 8, 30, 7, 7, 30, 7, 0), (4, 4, 4, 8, 4, 4, 4, 4, 4, 4, 4, 0))))
L2_GATHER_PEAK_TBS = 17.8      # MI355X_MICROARCH.md "Indexed rows: gather into LDS": 16.8-18.8 TB/s chip-wide for rows served by the XCDs' L2


def kernel_table(d, B, dtype, ms, cnt, n_sampled):
    """Per-kernel roofline records from the engine's HIP-event slots.  Algorithmic work = SURVEY.md section 8(d)'s per-unit figures."""
    nslots = len(SLOT_NAMES)
    per_step_ms = {SLOT_NAMES[i]: ms[i] / n_sampled for i in range(nslots) if cnt[i] > 0}
    avg = {SLOT_NAMES[i]: ms[i] / cnt[i] for i in range(nslots) if cnt[i] > 0}       # per launch / region
    N = B * d.G * d.G
    P2 = d.P * d.P
    es = 2 if dtype == "bf16" else 4
    sprite_b = N * P2 * 2 * es                                           # (grey, alpha) sprites: 16-bit pairs in the bf16 step
    render_fwd_bytes = sprite_b + N * 6 * 4 + B * d.I * d.I * 4          # 223.9 MB (bf16) at config 2
    render_bwd_bytes = render_fwd_bytes + N * P2 * 2 * es                # + the d-logits write
    peak_f = MFMA_PEAK_TFLOPS["bf16" if dtype == "bf16" else "f32"]
    h0 = (d.I + d.pad_pre + d.pad_post - d.conv_k[0]) // d.conv_s[0] + 1
    h1 = (h0 - d.conv_k[1]) // d.conv_s[1] + 1
    conv1_flop = 2.0 * B * h1 * h1 * d.conv_c[1] * (d.conv_k[1] * d.conv_k[1] * d.conv_c[0])
    dec_out_flop = 2.0 * N * 256 * (P2 * 2)
    A, NPc, Fc = d.A, d.NP, d.F
    REC = 4 + A + 2                                                        # record [box4 | attr A | depth | pres]
    box_in = Fc + 4 * REC                                                  # features + 4 neighbour records
    z_in, glim = box_in + NPc + 4 + A, P2
    o_in = z_in + 1
    # K2 per-cell chain: SURVEY 8(d) prices it as MFMA work, 2 * N * 425,472 flop ("latency-limited by the 3G-2 dependent diagonals")
    chain_flop = 2.0 * N * (box_in * 100 + 100 * 100 + 100 * (8 + NPc) + glim * 256 + 256 * 128 + 128 * 2 * A
                            + z_in * 100 + 100 * 100 + 100 * (2 + NPc) + o_in * 100 + 100 * 100 + 100)
    # builder-side byte model of what the chain kernels must move through HBM per row -- reported as `hbm_row_model`,
    # NOT the roofline: forward = every GEMM operand once as bf16 + glimpse derivative pairs + the fp32 bundle/record + sign bits + inputs;
    # backward = bundle + derivative pairs + sign bits in, layer-output gradients and d feat out as bf16
    hid = 2 * 100 + (256 + 128) + 2 * 100 + 2 * 100
    outs = (100 + 100 + 8 + NPc) + (256 + 128 + 2 * A) + (100 + 100 + 2 + NPc) + (100 + 100 + 1)
    if dtype == "bf16":
        fwd_row = 2 * (o_in + glim + hid + A) + 4 * glim + 4 * (308 + REC) + 66 * 4 * 8 // 8 + 4 * (Fc + REC + 2)
    else:
        fwd_row = 4 * (box_in + glim + z_in + o_in + hid + REC + glim) + 4 * (Fc + REC + 2)
    bwd_row = 4 * (308 + glim) + 66 * 4 * 8 // 8 + 2 * (outs + Fc)
    T = 3 * d.G - 2
    kernels = {}

    def add(name, bound, work, unit_scale, peak, unit, slot=None, **extra):
        slot = slot or name
        if slot in avg and avg[slot] > 0:
            ach = work / (avg[slot] * 1e-3) / unit_scale
            kernels[name] = dict(bound=bound, achieved=ach, peak=peak, unit=unit, frac=ach / peak, avg_ms=avg[slot], traffic=None, **extra)

    def chain_extra(slot, pack_kib, row_bytes):
        if slot not in avg or avg[slot] <= 0:
            return {}
        t = avg[slot] * 1e-3
        # every workgroup re-streams the whole pack on each wavefront it walks: one workgroup per sample and 3G-2 wavefronts up to 16 x 16
        # cells; beyond, ceil(G/8) bands per sample (one workgroup each), a band of hb grid rows walking 2(hb-1)+G wavefronts
        nb = 1 if d.G <= 16 else (d.G + 7) // 8
        hb = (d.G + nb - 1) // nb
        l2_bytes = float(B) * nb * (2 * (hb - 1) + d.G) * pack_kib * 1024
        return dict(l2_stream=dict(bytes=l2_bytes, achieved_TBs=l2_bytes / t / 1e12, peak_TBs=L2_GATHER_PEAK_TBS,
                                   frac=l2_bytes / t / 1e12 / L2_GATHER_PEAK_TBS,
                                   note="weight fragments L2 -> registers: workgroups x wavefronts walked x pack (a companion figure: round-4 ablations put the "
                                        "weight loads at ~12 % of the kernel, the dependent stage chain at ~70 %)"),
                    hbm_row_model=dict(bytes=float(N) * row_bytes, achieved_GBs=N * row_bytes / t / 1e9, frac_of_8TBs=N * row_bytes / t / 1e9 / HBM_PEAK_GBS,
                                       note="builder's byte model of the row buffers (DESIGN.md section 4), not SURVEY 8(d)'s figure"))

    add("chain_bwd", "mfma", chain_flop, 1e12, peak_f, "TFLOP/s", slot="cells_bwd", **chain_extra("cells_bwd", CHAIN_PACK_KIB["bwd"], bwd_row))
    add("chain_fwd", "mfma", chain_flop, 1e12, peak_f, "TFLOP/s", slot="cells_fwd", **chain_extra("cells_fwd", CHAIN_PACK_KIB["fwd"], fwd_row))
    add("render_fwd", "hbm", render_fwd_bytes, 1e9, HBM_PEAK_GBS, "GB/s")
    add("render_bwd", "hbm", render_bwd_bytes, 1e9, HBM_PEAK_GBS, "GB/s")
    add("conv1_fwd", "mfma", conv1_flop, 1e12, peak_f, "TFLOP/s")
    dec_flop = 2.0 * N * (A * 128 + 128 * 256 + 256 * P2 * 2)                                                   # K5 whole: 57.7 GFLOP
    from spair_pytorch_amd import models as _m
    # the engine's dec_out_* event scopes bracket the whole three-layer launch when the fused decoder kernels run (SpairStep.flags bits 4 / 6
    # clear, bf16 step): price them with the whole decoder's flops under a name that says so; decoder.out alone otherwise
    if dtype == "bf16" and not (_m.STEP_FLAGS & 16):
        add("decoder_fwd_fused", "mfma", dec_flop, 1e12, peak_f, "TFLOP/s", slot="dec_out_fwd")
    else:
        add("dec_out_fwd", "mfma", dec_out_flop, 1e12, peak_f, "TFLOP/s")
    add("decoder_fwd", "mfma", dec_flop, 1e12, peak_f, "TFLOP/s")
    if dtype == "bf16" and not (_m.STEP_FLAGS & 64):
        add("decoder_dgrad_fused", "mfma", dec_flop, 1e12, peak_f, "TFLOP/s", slot="dec_out_dgrad")
    else:
        add("dec_out_dgrad", "mfma", dec_out_flop, 1e12, peak_f, "TFLOP/s")
    # (decoder / per-cell weight gradients run on the helper stream beside the chain: their event times include the
    #  overlap and are not per-kernel durations -- see profiles/ for the rocprofv3 kernel stats)
    return kernels, per_step_ms


def stn_fwd_from_stamps(model, step_fn, d, B, dtype, chain_fwd_ms):
    """K4 = the glimpse stage of k_chain_fwd.  SpairStep.flags bit 1 makes sample 0's workgroup stamp s_memtime after every stage; the
    stage's share of the stamped cycles x the kernel's event time = its duration (clock-free).  One extra, untimed FORWARD pass (step_fn must not
    contain a collective: under torchrun only rank 0 gets here)."""
    if not chain_fwd_ms:
        return None
    from spair_pytorch_amd import _lib as L
    from spair_pytorch_amd import models
    import numpy as np
    ns_, gl_, nb_ = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    L.check(L.lib().spair_chain_stamp_layout(ctypes.byref(ns_), ctypes.byref(gl_), ctypes.byref(nb_)), "stamp_layout")
    T, NS, GL = L.lib().spair_chain_stamp_wavefronts(ctypes.byref(d)), ns_.value, gl_.value       # stamps per wavefront; interval GL = glimpse sampling
    if T * NS > 2048:
        return None
    old = models.STEP_FLAGS
    models.STEP_FLAGS = old | 2
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        models.STEP_FLAGS = old
    e = model._last["engine"]
    out = torch.zeros(4096, dtype=torch.int64, device=model.device)
    L.check(L.lib().spair_chain_stamps(ctypes.byref(e["dims"]), L.ptr(e["workspace"]), L.ptr(out), T * NS, L.stream()), "stamps")
    st = out.cpu().numpy()[:T * NS].reshape(T, NS).astype(np.float64)
    span = st[-1, NS - 1] - st[0, 0]
    if not span > 0:
        return None
    share = float(np.diff(st, axis=1)[:, GL].sum() / span)
    ms_ = share * chain_fwd_ms
    N, P2 = B * d.G * d.G, d.P * d.P
    work = B * d.C * d.I * d.I * 4 + N * P2 * d.C * (2 if dtype == "bf16" else 4) + N * 4 * 4       # 120.6 MB at config 2 (bf16)
    ach = work / (ms_ * 1e-3) / 1e9
    return dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, avg_ms=None, stage_ms=ms_, traffic=None,
                share_of_chain_fwd=share,
                note="fused stage of k_chain_fwd (no launch of its own): stage share from in-kernel s_memtime stamps of sample 0 x the kernel's "
                     "event time; the stamping workgroup runs this stage %d times.  Since round 5 the stage also hosts the early products of "
                     "Z0 / OBJ0 on images up to 128 px (22 MFMAs per wave, ~0.8 of its 3.6 us by sub-stamps, DESIGN 4.1): stage_ms is an upper "
                     "bound on the sampling's own time" % T)


def _sha16(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def attach_pmc_traffic(kernels, B, image, dtype):
    """roofline.traffic = HBM bytes per launch from the committed PMC passes of this same command (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Attached only when the file was collected on
    THESE kernels: every kernel name in it must exist in the built library and the sources it records must hash to the current ones."""
    import glob
    if not (B == 256 and image == 128 and dtype == "bf16"):
        return
    # configs[1] sets only: the 256x256 (configs[3]) sets are named r*_c3_* / r*_c4_*
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")) if "_c3_" not in f and "_c4_" not in f)
    if not files:
        return
    path = files[-1]
    rel = os.path.relpath(path, ROOT)
    pmc = json.load(open(path))
    lib_bytes = open(os.path.join(ROOT, "spair_pytorch_amd", "libspair_hip.so"), "rb").read()
    import re
    stale = None
    for src, sha in (pmc.get("src_sha16") or {}).items():
        cur = os.path.join(ROOT, src)
        if not os.path.exists(cur) or _sha16(cur) != sha:
            stale = "%s changed since %s was collected" % (src, rel)
    if not pmc.get("src_sha16"):
        stale = "%s records no source hashes" % rel
    for name, rec in pmc.items():
        if isinstance(rec, dict) and "kernel" in rec:
            m = re.search(r"k_[A-Za-z0-9_]+", rec["kernel"])
            if m and m.group(0).encode() not in lib_bytes:
                stale = "kernel %s of %s is not in the built library" % (m.group(0), rel)
    for name, rec in kernels.items():
        if stale:
            rec["traffic_source"] = "refused: " + stale
        elif name in pmc:
            rec["traffic"] = (pmc[name]["read_MB"] + pmc[name]["write_MB"]) * 1e6
            rec["traffic_source"] = rel


def config3_record(dev, args, strides):
    """BASELINE configs[3]: 256x256 scenes, 32x32 grid, batch 64 (same N = B*G*G as configs[1]) -- ms per step and the chain / renderer
    roofline fractions, bounded to a few seconds."""
    from spair_pytorch_amd import _lib as L
    from spair_pytorch_amd import config as cfg
    from spair_pytorch_amd.data import scattered_digits
    from spair_pytorch_amd.models import SPAIR
    from spair_pytorch_amd.optim import FusedAdam
    I, B = 256, 64
    cfg.set_grid(I, strides)
    torch.manual_seed(3)
    model = SPAIR([1, I, I], None, dev, compute_dtype=args.dtype).to(dev)
    opt = FusedAdam(model, lr=1e-4)
    x = torch.from_numpy(scattered_digits(1234, B, I, 11)[0]).to(dev)
    torch.manual_seed(7)
    gs = [args.global_step]

    def step():
        opt.zero_grad()
        loss = model(x, gs[0])[0]
        loss.backward()
        opt.step()
        gs[0] += 1
        return loss

    for _ in range(5):
        step()
    lib = L.lib()
    L.check(lib.spair_prof_enable(1), "prof_enable")
    lib.spair_prof_enable(0)
    torch.cuda.synchronize()
    K, every = args.config3_steps, 2
    t0 = time.perf_counter()
    for i in range(K):
        lib.spair_prof_enable(2 if i % every == 0 else 0)
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nslots = len(SLOT_NAMES)
    ms = (ctypes.c_float * nslots)()
    cnt = (ctypes.c_int * nslots)()
    L.check(lib.spair_prof_read(ms, cnt, nslots), "prof_read")
    lib.spair_prof_enable(0)
    d = model._last["engine"]["dims"]
    kernels, per_step = kernel_table(d, B, args.dtype, ms, cnt, len(range(0, K, every)))
    stn = stn_fwd_from_stamps(model, lambda: model(x, gs[0]), d, B, args.dtype, kernels.get("chain_fwd", {}).get("avg_ms"))
    if stn:
        kernels["stn_fwd"] = stn
    keep = ("chain_fwd", "chain_bwd", "render_fwd", "render_bwd", "stn_fwd")
    rec = dict(workload="BASELINE configs[3]: 256x256 synthetic scattered digits (<=11), 32x32 grid, batch 64, fwd+bwd+Adam, global_step %d+"
                        % args.global_step,
               ms_per_step=dt / K * 1e3, images_per_sec=B * K / dt, steps=K, warmup=5, elbo=float(loss.item()),
               kernels={k: {f: v for f, v in kernels[k].items() if f in ("bound", "achieved", "peak", "unit", "frac", "avg_ms", "stage_ms", "l2_stream")}
                        for k in keep if k in kernels},
               step_breakdown_ms=per_step,
               chain_status=model.chain_status(),      # band-split hand-offs: 0 = all arrived, 1 = a wait timed out (then the ELBO above is NaN)
               step_status=model.step_status())        # sticky bits over all its steps: 1 = hand-off time-out, 2 = a loss term was NaN / inf
    rec["finite"] = bool(rec["step_status"] == 0 and math.isfinite(rec["elbo"]))
    cfg.set_grid(args.image, strides)
    return rec


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
    ap.add_argument("--sweep", action="store_true", help="(kept for compatibility: the sweep is part of the default single-GPU line)")
    ap.add_argument("--no-sweep", action="store_true",
                    help="skip BASELINE configs[4]: after the main measurement, the step at global_step in {0,2000,4000,6000,7000,8000,10000}")
    ap.add_argument("--sweep-steps", type=int, default=10)
    ap.add_argument("--no-config3", action="store_true",
                    help="skip the bounded BASELINE configs[3] sub-record (256x256, 32x32 grid, batch 64) of the default single-GPU line")
    ap.add_argument("--config3-steps", type=int, default=20)
    ap.add_argument("--dense-input", action="store_true", help="diagnostic: uniform-noise images (no zero pixels) instead of scattered digits; "
                                                                "config.workload says so")
    ap.add_argument("--repeat", type=int, default=3, help="repeat the K-step timed region this many times; the MEDIAN repeat is reported")
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
    # SPAIR_DDP_OVERLAP=0: one all-reduce of the whole flat gradient after the backward instead of the three bucketed, overlapped ones.
    # The timed region runs WITHOUT bucket instrumentation (no extra event records, no wait on the communication stream per bucket); the
    # buckets' issue -> complete times of the N > 1 line's `ddp` record come from one extra step behind it (re-attached with timing=True)
    ddp_overlap = os.environ.get("SPAIR_DDP_OVERLAP", "1") != "0"
    ddp.attach(model, world, overlap=ddp_overlap, timing=False)
    bwd_end = [None]
    if world > 1:
        ddp.broadcast_parameters(model.flat_parameters())
    opt = FusedAdam(model, lr=1e-4)
    B = args.batch
    x = torch.from_numpy(scattered_digits(1234 + rank, B, args.image, 11)[0]).to(dev)   # resident in HBM
    if args.dense_input:      # diagnostic: images without a single zero pixel (what the kernels do on data that is not scattered digits on black)
        x = torch.rand(x.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1234 + rank)) * 0.9 + 0.05
    torch.manual_seed(7 + rank)                            # noise seed, per rank
    gstep = [args.global_step]
    last = {}

    def step():
        opt.zero_grad()
        loss, recon, z_where, z_pres = model(x, gstep[0])
        last["z_pres"] = z_pres
        last["z_where"] = z_where
        last["loss"] = loss
        loss.backward()
        if world > 1:
            if bwd_end[0] is not None:
                bwd_end[0].record()
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
    # EXACTLY K steps bracketed by barrier + synchronize on both sides; the bracket is repeated `--repeat` times (default 3) and the
    # MEDIAN repeat is the line's ms_per_step / value (`ms_per_step_repeats` lists all of them, `ms_per_step_min` the fastest): one
    # host-side hiccup (allocator, Python GC, a cold page) in a 0.1 s region does not decide the line, and neither does the luckiest
    # repeat.  Per-kernel event pairs are sampled at the same cadence in EVERY repeat, so the reported repeat carries them too.
    rep_dt = []
    nslots = len(SLOT_NAMES)
    ms = (ctypes.c_float * nslots)()
    cnt = (ctypes.c_int * nslots)()
    for rep in range(max(1, args.repeat)):
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
        rep_dt.append(time.perf_counter() - t0)
        L.check(lib.spair_prof_read(ms, cnt, nslots), "prof_read")        # drains the event pool between repeats (outside the bracket)
        lib.spair_prof_enable(0)
    if world > 1:
        tt = torch.tensor(rep_dt, device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)          # MAX over ranks, per repeat
        rep_dt = [float(v) for v in tt.tolist()]
    srt = sorted(rep_dt)
    dt = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])     # the median repeat
    terms = model.loss_terms().clone()
    ddp_rec = None
    if world > 1:
        terms = ddp.global_loss(terms)
        # one extra, untimed step with the bucket instrumentation (rank 0's view): does the window each bucket opens before the end of
        # the backward hide its collective?
        ddp.attach(model, world, overlap=ddp_overlap, timing=True)
        bwd_end[0] = torch.cuda.Event(enable_timing=True)
        step()
        torch.cuda.synchronize()
        bt = ddp.bucket_timings(model, bwd_end[0])
        if bt is not None:
            ddp_rec = dict(backend=backend, overlap=True, buckets=bt,
                           note="one instrumented step behind the timed region, rank 0: allreduce_ms = range final -> collective complete on the communication stream; "
                                "done_after_backward_end_ms > 0 is the exposed part of that bucket")
        else:
            ddp_rec = dict(backend=backend, overlap=False, note="one all-reduce of the whole flat gradient behind the backward (SPAIR_DDP_OVERLAP=0)")
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    K = args.steps
    d = model._last["engine"]["dims"]
    kernels, per_step_ms = kernel_table(d, B, args.dtype, ms, cnt, len(range(0, K, args.prof_every)) * len(rep_dt))
    # K4 (the STN-forward gather is a stage of k_chain_fwd, not a launch): its share of the kernel from the in-kernel stage stamps of
    # one extra step outside the timed region
    # (rank 0 alone runs it, the other ranks have left: a forward pass only -- nothing collective)
    stn = stn_fwd_from_stamps(model, lambda: model(x, gstep[0]), d, B, args.dtype, kernels.get("chain_fwd", {}).get("avg_ms"))
    if stn:
        kernels["stn_fwd"] = stn
    attach_pmc_traffic(kernels, B, args.image, args.dtype)
    timed = [k for k in kernels if kernels[k].get("avg_ms")]
    dominant = max(timed, key=lambda k: kernels[k]["avg_ms"]) if timed else None
    roof = dict(kernels[dominant], kernel=dominant) if dominant else None

    if args.image == 128 and B == 256 and d.G == 16:
        which = "BASELINE configs[1]" if world == 1 else "BASELINE configs[2] (configs[1] per GPU, DDP over %d GPUs)" % world
    elif args.image == 256 and B == 64 and d.G == 32:
        which = "BASELINE configs[3] (STN gather stress)"
    else:
        which = "custom (not a BASELINE config)"
    if args.dense_input:
        which = "custom (not a BASELINE config; --dense-input: uniform-noise images without zero pixels)"
    workload = ("%s: %dx%d synthetic scattered digits (<=11), %dx%d grid, batch %d/GPU, fwd+bwd+Adam, global_step %d+ (wheel %s)"
                % (which, args.image, args.image, d.G, d.G, B, args.global_step, "off" if args.global_step >= 1000 else "on"))
    out = dict(metric="SPAIR train images/sec + ELBO, 128x128 scattered-MNIST, batch 256", value=world * B * K / dt, unit="images/sec",
               n_gpus=world, steps=K, warmup=args.warmup, ms_per_step=dt / K * 1e3, higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype=args.dtype, data="synthetic",
               config=dict(workload=workload, global_batch=world * B, image=args.image, grid=d.G, global_step=args.global_step,
                           parallelism="dp%d" % world),
               repeat=len(rep_dt), ms_per_step_repeats=[v / K * 1e3 for v in rep_dt], ms_per_step_min=min(rep_dt) / K * 1e3,
               ms_per_step_note="ms_per_step / value = the MEDIAN of the repeats",
               elbo=float(terms[0].item()), elbo_terms=[float(v) for v in terms[:9].tolist()],
               roofline=roof, kernels=kernels, step_breakdown_ms=per_step_ms)
    # SpairStep.status: sticky bits over every step this model ran (1 = a band-split hand-off timed out, 2 = a loss term was NaN / inf)
    out["step_status"] = model.step_status()
    out["finite"] = bool(out["step_status"] == 0 and all(math.isfinite(v) for v in out["elbo_terms"]))
    if ddp_rec is not None:
        out["ddp"] = ddp_rec
    default_workload = args.image == 128 and B == 256 and d.G == 16 and args.dtype == "bf16" and not args.dense_input
    if world == 1 and not args.no_sweep and (args.sweep or default_workload):
        # BASELINE configs[4] on one GPU: "z_pres discovery-prior curriculum sweep (max_objects 1 -> 11) ... sequential compositing kernel
        # under varying active-cell density".  Three axes, every point from the same model and optimizer state:
        #   schedule : global_step of the count-prior / training-wheel schedules (config.py:65-69, models.py:186-188), the bench batch;
        #   objects  : scenes with at most k = 1, 3, 6, 11 digits (the dataset's max_objects);
        #   density  : the state itself moved to where training takes it -- the presence logit's bias (obj_network.out.bias) bisected to
        #              mean z_pres 0.05 / 0.3 / 0.7 (training-mode forwards), the box-size means' bias (box_network.output_layers.0.bias[2:4])
        #              shifted by -0.79 / 0 / +1.33: what the renderer's cost actually follows (active cells x footprint).
        # Every point runs whole steps with the learning rate at 0: 160 steps into training the state moves fast (12 steps change the
        # mean presence by a third), and a point must measure the state it names.
        from spair_pytorch_amd.models import step_scalars
        sd = model.state_dict()
        snap = (model.flat_parameters().clone(), opt.state_dict())
        snap = (snap[0], {k: (v.clone() if torch.is_tensor(v) else v) for k, v in snap[1].items()})
        x_main = x
        nsl = len(SLOT_NAMES)
        R_F, R_B = SLOT_NAMES.index("render_fwd"), SLOT_NAMES.index("render_bwd")

        model.raise_on_nonfinite = False                   # a dead point is RECORDED (finite: false, sweep_ok: false, exit code 4), not raised

        def restore():
            model.clear_step_status()
            model.flat_parameters().copy_(snap[0])         # every point starts from the same model and optimizer state ...
            opt.load_state_dict(snap[1])
            opt.lr = 0.0                                   # ... and keeps it: the step runs whole (fwd + bwd + the Adam kernel), the
                                                           # parameters do not move, so a point measures the state it names

        def time_point(gs, xin, label):
            nonlocal x
            x = xin
            gstep[0] = gs
            step()
            step()                                         # two warm steps behind the state restore
            torch.cuda.synchronize()
            # per-step device time from event pairs on the launch stream; the point's figure is the MEDIAN step (a host-side stall inside
            # one step -- the restore's allocator traffic, a GC pause -- does not move it), the wall-clock mean is kept beside it; the
            # renderer's two kernels from the engine's own event slots over the same steps
            pm, pc = (ctypes.c_float * nsl)(), (ctypes.c_int * nsl)()
            lib.spair_prof_select(ctypes.c_ulonglong((1 << R_F) | (1 << R_B)))
            lib.spair_prof_enable(1)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.sweep_steps)]
            t1 = time.perf_counter()
            for e0, e1 in evs:
                e0.record()
                step()
                e1.record()
            torch.cuda.synchronize()
            wall_ms = (time.perf_counter() - t1) / args.sweep_steps * 1e3
            lib.spair_prof_read(pm, pc, nsl)
            lib.spair_prof_enable(0)
            per = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
            ms_s = per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2])
            st_ = step_scalars(gs, B)
            zp, zw = last["z_pres"], last["z_where"]
            # a point that went non-finite is not a measurement: the last step's loss, the model's sticky status word over the point's steps
            # (SpairStep.status: bit 1 = a loss term was NaN / inf) and the parameters themselves
            status = model.step_status()
            loss_v = float(last["loss"].item())
            finite = bool(status == 0 and math.isfinite(loss_v) and torch.isfinite(model.flat_parameters()).all().item()
                          and torch.isfinite(zw).all().item())
            side_px = float(zw[:, 2:4].mean().item()) * args.image       # z_where = (xt, yt, xs, ys) in image units (models.py:375-381)
            # presence-weighted footprint: what the compositing kernels walk
            rec = dict(axis=label, global_step=gs, count_prior_prob=float(st_.count_prior_prob), wheel=float(st_.wheel), ms_per_step=ms_s,
                       ms_per_step_wall_mean=wall_ms, ms_per_step_max=per[-1], images_per_sec=B / ms_s * 1e3,
                       mean_z_pres=float(zp.mean().item()), mean_box_side_px=side_px, loss=loss_v, finite=finite, step_status=status,
                       render_fwd_ms=pm[R_F] / max(pc[R_F], 1), render_bwd_ms=pm[R_B] / max(pc[R_B], 1))
            x = x_main
            return rec

        sweep = []
        for gs in (0, 2000, 4000, 6000, 7000, 8000, 10000):
            restore()
            sweep.append(time_point(gs, x_main, "schedule"))
        for k in (1, 3, 6, 11):
            restore()
            xk = torch.from_numpy(scattered_digits(4321, B, args.image, k)[0]).to(dev)
            r = time_point(args.global_step, xk, "objects")
            r["max_objects"] = k
            sweep.append(r)
        pres_b, box_b = sd["obj_network.out.bias"], sd["box_network.output_layers.0.bias"]

        def mean_pres():
            with torch.enable_grad():                      # a TRAINING-mode forward (relaxed-Bernoulli samples, as in the timed steps), no backward
                return float(model(x_main, args.global_step)[3].mean().item())

        for target_pres in (0.05, 0.3, 0.7):
            for size_bias in (-0.79, 0.0, 1.33):          # x 0.6 / x 1 / x 1.9 on the state's mean object side (15 / 24 / 38 px at initialisation)
                restore()
                with torch.no_grad():
                    box_b[2:4] += size_bias
                    base = pres_b.clone()
                    lo_, hi_ = -12.0, 12.0                  # the mean presence is monotone in the logit's bias
                    for _ in range(14):
                        mid = 0.5 * (lo_ + hi_)
                        pres_b.copy_(base + mid)
                        if mean_pres() < target_pres:
                            lo_ = mid
                        else:
                            hi_ = mid
                    pres_b.copy_(base + 0.5 * (lo_ + hi_))
                r = time_point(args.global_step, x_main, "density")
                r["target_mean_z_pres"] = target_pres
                r["box_size_logit_bias"] = size_bias
                # the calibration must have landed: the timed steps' mean presence (fresh noise per step) within 10 % of the target
                r["presence_miss"] = abs(r["mean_z_pres"] - target_pres) / target_pres
                r["on_target"] = bool(r["finite"] and r["presence_miss"] <= 0.10)
                sweep.append(r)
        restore()
        model.raise_on_nonfinite = True
        opt.lr = float(snap[1]["lr"])
        lib.spair_prof_select(ctypes.c_ulonglong(args.prof_mask & 0xFFFFFFFFFFFFFFFF))
        out["sweep"] = sweep
        out["sweep_ok"] = bool(all(r["finite"] and r.get("on_target", True) and r["mean_box_side_px"] > 1.0 for r in sweep))
        out["sweep_note"] = ("BASELINE configs[4] on ONE GPU (the 8-GPU form is the driver's): schedule / object-count / density axes, %d timed steps per "
                             "point behind 2 warm steps, every point from the same model and optimizer state with the learning rate at 0 (whole "
                             "steps, frozen parameters: a point measures the state it names); ms_per_step = median of the per-step "
                             "event times; render_*_ms = the renderer's two kernels over the same steps" % args.sweep_steps)
    if world == 1 and default_workload and not args.no_config3:
        # BASELINE configs[3] (256x256, 32x32 grid, batch 64) as a bounded sub-record of the default line: its own model, a few steps
        out["config3"] = config3_record(dev, args, strides)      # (the main model stays alive: two ~5 GB workspaces of 288 GB)
    out["hbm_measured"] = hbm_copy_rates(dev)      # what this box sustains, beside the vendor 8 TB/s the roofline divides by
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(strides)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not out["finite"]:
        sys.stderr.write("bench.py: the timed steps produced a non-finite loss (step_status %d): not a measurement\n" % out["step_status"])
        sys.exit(5)
    if out.get("sweep_ok") is False:      # a sweep point went non-finite or missed the state it names: the record says which, the exit code says so
        dead = [(r["axis"], r.get("global_step"), r.get("target_mean_z_pres"), r.get("box_size_logit_bias")) for r in out["sweep"]
                if not (r["finite"] and r.get("on_target", True) and r["mean_box_side_px"] > 1.0)]
        sys.stderr.write("bench.py: configs[4] sweep points non-finite or off their target state: %s\n" % dead)
        sys.exit(4)
    if out.get("config3", {}).get("finite") is False and out["config3"].get("chain_status") != 1:
        sys.stderr.write("bench.py: the configs[3] sub-record's steps produced a non-finite loss\n")
        sys.exit(5)
    if out.get("config3", {}).get("chain_status") == 1:      # a band-split hand-off timed out: that sub-record is not a measurement
        sys.stderr.write("bench.py: the configs[3] sub-record's per-cell chain reported a hand-off time-out (chain_status 1)\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
