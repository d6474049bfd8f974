"""The driver's command, `python bench.py` with its default sub-records, end to end on the GPU box (shortened loops): ONE JSON line with the
contract's keys, the roofline / kernels objects, the configs[4] sweep, the configs[3] sub-record and the fused-stage K4 entry."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_has_every_sub_record():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--sweep-steps", "1", "--config3-steps", "2",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "kernels", "sweep", "config3"):
        assert k in rec, k
    assert rec["n_gpus"] == 1 and rec["steps"] == 4 and rec["unit"] == "images/sec" and rec["dtype"] == "bf16" and rec["data"] == "synthetic"
    assert "configs[1]" in rec["config"]["workload"] and rec["value"] > 0
    roof = rec["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["kernel"].startswith("chain") and 0 < roof["l2_stream"]["frac"] < 1
    assert {"chain_fwd", "chain_bwd", "render_fwd", "render_bwd", "conv1_fwd", "decoder_fwd", "stn_fwd"} <= set(rec["kernels"])
    assert rec["kernels"]["stn_fwd"]["stage_ms"] > 0
    sw = rec["sweep"]
    assert [p["global_step"] for p in sw if p["axis"] == "schedule"] == [0, 2000, 4000, 6000, 7000, 8000, 10000]
    assert [p["max_objects"] for p in sw if p["axis"] == "objects"] == [1, 3, 6, 11]
    dens = [p for p in sw if p["axis"] == "density"]
    # no dead points: every point finite with boxes of at least a pixel (a collapsed state reads sigmoid(-10) * 48 = 0.002 px), the
    # nine density points each within 10 % of the mean presence they name (round 5's count-prior KL went NaN on the 0.7 row and the
    # line printed it)
    assert rec["finite"] is True and rec["step_status"] == 0 and rec["sweep_ok"] is True
    assert len(dens) == 9 and all(p["render_fwd_ms"] > 0 and p["render_bwd_ms"] > 0 and p["mean_box_side_px"] > 1.0 and p["finite"] for p in sw)
    assert sorted(set(p["target_mean_z_pres"] for p in dens)) == [0.05, 0.3, 0.7]
    for p in dens:
        assert abs(p["mean_z_pres"] - p["target_mean_z_pres"]) <= 0.10 * p["target_mean_z_pres"] and p["on_target"], p
    # the density axis moves what it is named for: presence and object size
    assert max(p["mean_z_pres"] for p in dens) > 3 * min(p["mean_z_pres"] for p in dens)
    assert max(p["mean_box_side_px"] for p in dens) > 1.5 * min(p["mean_box_side_px"] for p in dens)
    assert rec["ms_per_step_min"] <= rec["ms_per_step"] <= max(rec["ms_per_step_repeats"])
    c3 = rec["config3"]
    assert "configs[3]" in c3["workload"] and c3["ms_per_step"] > 0 and "stn_fwd" in c3["kernels"]
    assert c3["finite"] is True and c3["chain_status"] == 0
