#!/usr/bin/env python3
"""Regenerate the measured tables of DESIGN.md from a profile set in profiles/ -- so that the numbers in the document cannot drift from the
files they quote.

    python tools/design_tables.py r06_b            # print the blocks
    python tools/design_tables.py r06_b --write    # replace the text between the <!-- GENERATED:<name> BEGIN/END --> markers of DESIGN.md

Inputs (all under profiles/, produced on the GPU box by tools/profile_set.sh <tag> and `python bench.py`):
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --no-sweep --no-config3` (average duration per kernel name)
  <tag>_step_timeline.txt  the last traced step, launch by launch (tools/step_trace.py): separates the k_tn_ring launches that share a name
  <tag>_bench.json         the default `python bench.py` line (event times, sweep, config3, step breakdown)
  <tag>_pmc_traffic.txt    HBM bytes per launch (FETCH_SIZE / WRITE_SIZE passes, tools/pmc_summary.py)
  <tag>_sq_counters.txt    SQ counter ratios (tools/sq_summary.py);  <tag>_l2_requests.txt  L1 -> L2 requests (tools/l2_summary.py)
  <tag>_gputest.txt        tail of `pytest tests -m gpu` at that state (optional: the test count)
Algorithmic work per launch: SURVEY.md 8(d) at BASELINE configs[1] (B = 256, I = 128, G = 16, N = 65,536), bf16 rows.
Peaks: 8 TB/s HBM, 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
HBM_PEAK, MFMA_PEAK = 8.0e12, 2.5e15

# (row label, reference lines, name substring in kernel_stats / timeline, grid filter in the timeline or None, bench.json `kernels` key, bound, work)
ROWS = [
    ("`k_chain_fwd<true>` (K2 forward)", "models.py:68-117", "k_chain_fwd<true>", None, "chain_fwd", "mfma", 55.8e9),
    ("`k_chain_bwd` (K2 backward)", "autograd of models.py:68-117", "k_chain_bwd", None, "chain_bwd", "mfma", 55.8e9),
    ("`k_render_fwd_mma<1,4>` (K6 forward)", "modules.py:256-269, models.py:511-540", "k_render_fwd_mma<1, 4>", None, "render_fwd", "hbm", 223.9e6),
    ("`k_render_bwd2<28,0,1,REC>` (K6 backward)", "autograd of the same", "k_render_bwd2ILi28ELi0ELi1ELb1", None, "render_bwd", "hbm", 429.4e6),
    ("`k_dec_fwd` (K5 forward)", "models.py:474-504", "k_dec_fwd", None, "decoder_fwd_fused", "mfma", 57.7e9),
    ("`k_dec_bwd` (K5 data gradients)", "autograd of the same", "k_dec_bwd", None, "decoder_dgrad_fused", "mfma", 58.6e9),
    ("`k_conv0_fwd_c1k4_mfma` (stem)", "modules.py:59-64 (conv_0)", "k_conv0_fwd_c1k4_mfma", None, None, "hbm", 321.0e6),
    ("`k_conv_s2k4_patch` conv_1 forward", "modules.py:59-64 (conv_1)", "k_conv_s2k4_patch", 887808, "conv1_fwd", "mfma", 155.2e9),
    ("`k_conv_s2k4_patch` conv_2 forward", "modules.py:59-64 (conv_2)", "k_conv_s2k4_patch", 196608, None, "mfma", 34.4e9),
    ("`k_conv_s2k4_dgrad<true,true>` conv_1 dgrad + stem wgrad", "autograd of conv_1 / conv_0", "k_conv_s2k4_dgrad<true, true>", None, None, "mfma", 155.2e9),
    ("`k_conv_s2k4_dgrad<false,true>` conv_2 dgrad", "autograd of conv_2", "k_conv_s2k4_dgrad<false, true>", None, None, "mfma", 34.4e9),
    ("`k_tn_ring<256,true,3>` conv_1 wgrad", "autograd of conv_1", "k_tn_ring<256, true, 3>", "max", None, "mfma", 155.2e9),
    ("`k_tn_ring<256,true,3>` conv_2 wgrad", "autograd of conv_2", "k_tn_ring<256, true, 3>", "min", None, "mfma", 34.4e9),
    ("`k_tn_ring<128,false,3>` per-cell wgrads, grouped (13 layers)", "autograd of modules.py:124-165", "k_tn_ring<128, false, 3>", 98304, None, "mfma", 29.0e9),
    ("`k_tn_ring<256,false,3>` encoder layer 0 wgrad", "autograd of object_encoder.dense0", "k_tn_ring<256, false, 3>", 131072, None, "mfma", 26.8e9),
    ("`k_tn_ring<256,false,3>` decoder.out wgrad (behind `k_chain_bwd` since round 6)", "autograd of object_decoder.out", "k_tn_ring<256, false, 3>", 106496, None, "mfma", 52.6e9),
    ("`k_pw_stack<false>` 1x1 stack forward", "modules.py:59-64 (conv_3..out)", "k_pw_stack<false>", None, None, "mfma", 8.4e9),
    ("`k_pw_stack<true>` 1x1 stack dgrad", "autograd of the same", "k_pw_stack<true>", None, None, "mfma", 8.4e9),
    ("`k_count_kl<5,4>` (K8, beside decoder + renderer)", "models.py:186-257", "k_count_kl<5, 4>", None, None, "latency", 0.0),
    ("`k_gauss_kl` (K7)", "models.py:169-185", "k_gauss_kl", None, None, "hbm", 16.0e6),
]


def load_stats(tag):
    out = {}
    with open(os.path.join(PROF, tag + "_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            out[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
    return out


def load_timeline(tag):
    """[(us, name, grid)] of the last traced step."""
    rows = []
    p = os.path.join(PROF, tag + "_step_timeline.txt")
    if not os.path.exists(p):
        return rows, None
    span = None
    for ln in open(p):
        m = re.match(r"\s*([\d.]+) us\s+@\s+([\d.]+)\s+(.*) grid=(\d+)", ln)
        if m:
            rows.append((float(m.group(1)), m.group(3), int(m.group(4)), float(m.group(2))))
        m = re.match(r"sum of kernel time ([\d.]+) us, step span ([\d.]+) us, (\d+) launches", ln)
        if m:
            span = (float(m.group(1)), float(m.group(2)), int(m.group(3)))
    return rows, span


def load_table(path):
    """whitespace tables of tools/sq_summary.py / l2_summary.py: {kernel-name-prefix: [columns]}"""
    out = {}
    if not os.path.exists(path):
        return out
    lines = open(path).read().splitlines()
    for ln in lines[1:]:
        parts = ln.split()
        if len(parts) < 4:
            continue
        k = 0
        while k < len(parts) and not re.fullmatch(r"[-+0-9.e]+", parts[k]):
            k += 1
        out[" ".join(parts[:k])] = parts[k:]
    return out


def find(table, sub, grid=None):
    for k, v in table.items():
        if sub.replace(" ", "") in k.replace(" ", "") and (grid is None or ("grid=%d" % grid) in k.replace(" ", "")):
            return v
    return None


def kernels_block(tag):
    stats, (tl, span) = load_stats(tag), load_timeline(tag)
    bench = json.load(open(os.path.join(PROF, tag + "_bench.json"))) if os.path.exists(os.path.join(PROF, tag + "_bench.json")) else {}
    pmc = load_table(os.path.join(PROF, tag + "_pmc_traffic.txt"))      # kernel | calls | read MB | write MB | total MB (per launch)
    sq = load_table(os.path.join(PROF, tag + "_sq_counters.txt"))
    l2 = load_table(os.path.join(PROF, tag + "_l2_requests.txt"))
    out = ["| kernel | reference | avg per launch, rocprofv3 (bench event) | SURVEY 8(d) work | achieved | **frac of peak** | HBM bytes, PMC | MFMA busy / wait_any | L2 hit |",
           "|---|---|---|---|---|---|---|---|---|"]
    for label, ref, sub, grid, bkey, bound, work in ROWS:
        ms = None
        cands = [(us, g) for us, name, g, _ in tl if sub.replace(" ", "") in name.replace(" ", "")]
        if grid in ("max", "min") and cands:
            ms = (max if grid == "max" else min)(cands)[0] / 1e3
            src = "timeline"
        elif isinstance(grid, int):
            hit = [us for us, g in cands if g == grid]
            if hit:
                ms, src = hit[0] / 1e3, "timeline"
        if ms is None and grid is None:
            for name, (calls, avg) in stats.items():
                if sub.replace(" ", "") in name.replace(" ", ""):
                    ms, src = avg, "stats"
                    break
        if ms is None:
            continue
        ev = bench.get("kernels", {}).get(bkey, {}).get("avg_ms") if bkey else None
        if bound == "mfma":
            ach, frac, unit = work / (ms * 1e-3) / 1e12, work / (ms * 1e-3) / MFMA_PEAK, "TFLOP/s"
            wtxt = "%.1f GFLOP" % (work / 1e9)
        elif bound == "hbm":
            ach, frac, unit = work / (ms * 1e-3) / 1e12, work / (ms * 1e-3) / HBM_PEAK, "TB/s"
            wtxt = "%.1f MB" % (work / 1e6)
        else:
            ach, frac, unit, wtxt = None, None, "", "B·HW·(HW+1)/2 element-steps"
        pv = find(pmc, sub if "ILi" not in sub else sub[:24])
        traffic = float(pv[-1]) if pv else None
        sqv = find(sq, sub if "ILi" not in sub else sub[:24], grid if isinstance(grid, int) else None)
        l2v = find(l2, sub if "ILi" not in sub else sub[:24])
        out.append("| %s | `%s` | %.4f ms%s%s | %s | %s | %s | %s | %s | %s |" % (
            label, ref, ms, " (last traced step)" if src == "timeline" else "", (" (%.4f)" % ev) if ev else "", wtxt,
            ("%.0f %s" % (ach, unit)) if ach is not None and unit == "TFLOP/s" else ("%.2f %s" % (ach, unit)) if ach is not None else "—",
            ("**%.3f**" % frac) if frac is not None else "—",
            ("%.0f MB%s" % (traffic, " (mean of the launches sharing the name)" if isinstance(grid, (int, str)) else "")) if traffic else "—",
            ("%s / %s" % (sqv[0], sqv[3])) if sqv and len(sqv) >= 4 else "—",
            l2v[-1] if l2v else "—"))
    if span:
        out.append("")
        out.append("Last traced step: %d launches, %.1f µs of kernel time inside a %.1f µs span (`profiles/%s_step_timeline.txt`)." % (span[2], span[0], span[1], tag))
    return "\n".join(out)


def results_block(tag):
    b = json.load(open(os.path.join(PROF, tag + "_bench.json")))
    out = []
    out.append("`profiles/%s_bench.json` (default `python bench.py`): **%.3f ms/step = %.0f images/s** (median of repeats %s; fastest %.3f); ELBO after the run %.4g; "
               "`step_status` %s, `finite` %s, `sweep_ok` %s." % (
                   tag, b["ms_per_step"], b["value"], ", ".join("%.3f" % v for v in b["ms_per_step_repeats"]), b["ms_per_step_min"], b["elbo"],
                   b.get("step_status", "n/a"), b.get("finite", "n/a"), b.get("sweep_ok", "n/a")))
    r = b["roofline"]
    out.append("`roofline` (dominant kernel `%s`): %.1f %s of %.0f = **%.4f**; HBM bytes per launch (PMC) %s." % (
        r["kernel"], r["achieved"], r["unit"], r["peak"], r["frac"], ("%.0f MB" % (r["traffic"] / 1e6)) if r.get("traffic") else "null"))
    sb = b.get("step_breakdown_ms", {})
    if sb:
        out.append("Event-scope breakdown of the step (ms; scopes on the helper stream overlap the others): " + ", ".join("%s %.3f" % (k, v) for k, v in sb.items()) + ".")
    c3 = b.get("config3")
    if c3:
        out.append("configs[3] sub-record: **%.3f ms/step = %.0f images/s**, chain %.3f + %.3f ms, renderer %.3f + %.3f ms, `chain_status` %s, `finite` %s." % (
            c3["ms_per_step"], c3["images_per_sec"], c3["kernels"]["chain_fwd"]["avg_ms"], c3["kernels"]["chain_bwd"]["avg_ms"],
            c3["kernels"]["render_fwd"]["avg_ms"], c3["kernels"]["render_bwd"]["avg_ms"], c3.get("chain_status"), c3.get("finite", "n/a")))
    cb = b.get("cpu_baseline")
    if cb:
        out.append("`cpu_baseline`: %.2f %s on %s threads (`%s`; %s)." % (cb["value"], cb["unit"], cb["cores"], cb["kind"], cb["sample"]))
    hm = b.get("hbm_measured")
    if hm:
        out.append("`hbm_measured` on that box: " + json.dumps(hm) + ".")
    return "\n\n".join(out)


def sweep_block(tag):
    b = json.load(open(os.path.join(PROF, tag + "_bench.json")))
    sw = b.get("sweep", [])
    out = ["| axis | point | mean z_pres (target) | mean object side (px) | ms / step | renderer fwd / bwd (ms) | finite |", "|---|---|---|---|---|---|---|"]
    for p in sw:
        if p["axis"] == "schedule":
            pt = "global_step %d (count-prior p %.4g)" % (p["global_step"], p["count_prior_prob"])
        elif p["axis"] == "objects":
            pt = "max_objects %d" % p["max_objects"]
        else:
            pt = "size-logit bias %+.2f" % p["box_size_logit_bias"]
        tgt = (" (%.2f)" % p["target_mean_z_pres"]) if "target_mean_z_pres" in p else ""
        out.append("| %s | %s | %.4f%s | %.1f | %.3f | %.3f / %.3f | %s |" % (p["axis"], pt, p["mean_z_pres"], tgt, p["mean_box_side_px"], p["ms_per_step"],
                                                                      p["render_fwd_ms"], p["render_bwd_ms"], p.get("finite", "n/a")))
    return "\n".join(out)


def tests_block(tag):
    p = os.path.join(PROF, tag + "_gputest.txt")
    if not os.path.exists(p):
        return "(no `profiles/%s_gputest.txt`)" % tag
    last = [ln for ln in open(p).read().splitlines() if "passed" in ln or "failed" in ln]
    return "`pytest tests -m gpu` at this state (`profiles/%s_gputest.txt`): **%s**." % (tag, last[-1].strip("= ") if last else "?")


BLOCKS = {"kernels": kernels_block, "results": results_block, "sweep": sweep_block, "tests": tests_block}


def main():
    tag = sys.argv[1]
    write = "--write" in sys.argv
    blocks = {}
    for name, fn in BLOCKS.items():
        try:
            blocks[name] = fn(tag)
        except FileNotFoundError as e:
            blocks[name] = "(missing input: %s)" % e.filename
    if not write:
        for name, text in blocks.items():
            print("<!-- %s -->\n%s\n" % (name, text))
        return
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    for name, text in blocks.items():
        pat = re.compile(r"(<!-- GENERATED:%s BEGIN[^>]*-->)(.*?)(<!-- GENERATED:%s END -->)" % (name, name), re.S)
        if not pat.search(s):
            print("no marker for", name)
            continue
        s = pat.sub(lambda m: "<!-- GENERATED:%s BEGIN (tools/design_tables.py %s; do not edit by hand) -->\n%s\n%s" % (name, tag, text, m.group(3)), s)
    open(path, "w").write(s)
    print("DESIGN.md updated from profiles/%s_*" % tag)


if __name__ == "__main__":
    main()
