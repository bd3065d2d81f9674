#!/usr/bin/env python3
"""Rewrite the measured-numbers table of DESIGN.md section 6a from the committed artefacts
(profiles/rNN_bench_n1.json, rNN_bench_emulated_rank3of8.json, rNN_pmc_hbm_traffic.json, rNN_mfma_utilisation.json).
Usage: update_design_numbers.py r02"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
j = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench_n1.json")))
e = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench_emulated_rank3of8.json")))
pm = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_hbm_traffic.json")))
mf = json.load(open(os.path.join(ROOT, "profiles", tag + "_mfma_utilisation.json")))
cv, cb = j["convergence"], j["cpu_baseline"]
K, r = j["kernels"], j["roofline"]


def rocprof_avg_us(kernel):
    """average duration of the family in the committed rocprofv3 --stats summary (the k_spd_level template is
    reported as k_spd_fwd / k_spd_bwd by its MODE argument)"""
    import csv
    import re
    calls = tot = 0
    for row in csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_kernel_stats_bench_default.csv"))):
        m = re.search(r"k_spd_level<([^>]*)>", row["Name"])
        if m:
            name = "k_spd_fwd" if m.group(1).split(",")[3].strip() in ("0", "2") else "k_spd_bwd"   # MODE 0 / 2 (roots) / 1
        else:
            m2 = re.search(r"\b(k_[a-z_0-9]+)", row["Name"])
            name = m2.group(1) if m2 else row["Name"]
        if name == kernel:
            calls += int(row["Calls"])
            tot += float(row["TotalDurationNs"])
    return tot / calls / 1e3 if calls else float("nan")

tr = pm["kernels"][r["kernel"]]["hbm_bytes_per_launch"]
spd = K["k_spd_fwd"]["ms_per_step"] + K["k_spd_bwd"]["ms_per_step"]
ops = K["k_bsr"]["ms_per_step"] + K.get("k_bsr_tcol", {"ms_per_step": 0.0})["ms_per_step"]   # (the translation-column passes are a family of their own since round 4)
other = sum(v["ms_per_step"] for k, v in K.items() if not k.startswith("k_spd") and k not in ("k_bsr", "k_bsr_tcol"))
rows = [
    "| quantity | value | source |", "|---|---|---|",
    "| throughput | **%.1f outer iterations / s**, %.2f ms / iteration | `profiles/%s_bench_n1.json` |" % (j["value"], j["ms_per_step"], tag),
    "| to the reference objective (within 1e-6 of the objective the CPU path reaches, `2F = %.2f`) | **%d iterations, %.2f s** (mean %.1f ms / iteration); the CPU path: %d iterations, %.0f s on %d cores; whole 400-iteration run %.1f ms / iteration; last 20 iterations (no node refines any more) %.2f ms / iteration | same file, `convergence`; `profiles/%s_cpu_convergence.json` |" % (
        cv["target_2F"], cv["iterations_to_1e-6"], cv["seconds_to_1e-6"], cv["mean_ms_per_iter_to_1e-6"],
        cv["cpu_reference"]["iterations_to_1e-6"], cv["cpu_reference"]["seconds_to_1e-6"], cv["cpu_reference"]["cores"],
        cv["mean_ms_per_iter_whole_run"], cv["last20_ms_per_iter"], tag),
    "| CPU baseline (C++ restatement, all 8 nodes, 3 iterations; %s) | %.2f iterations / s on %d cores, %.2f on 1 thread | same file, `cpu_baseline` |" % (
        cb.get("cpu_model", "?"), cb["value"], cb["cores"], cb["value_1_thread"]),
    "| set-up (untimed) | %.1f s graph + chordal init, %.1f s operators + both factorizations | same file |" % (
        j["setup_s"]["graph+chordal_init"], j["setup_s"]["operators+factorizations"]),
    "| one rank of an 8-GPU run emulated on one GPU (1 node, frozen neighbours, no exchange) | %.2f ms / iteration | `profiles/%s_bench_emulated_rank3of8.json` (diagnostic, not a metric) |" % (e["ms_per_step"], tag),
    "| dominant kernel family | `%s`: %.0f launches / iteration, %.1f µs average (HIP events around each sweep, launch gaps included) vs %.1f µs (rocprofv3 --stats, kernel time only) | `%s_bench_n1.json`, `%s_kernel_stats_bench_default.csv` |" % (
        r["kernel"], r["launches_per_step"], r["avg_launch_us"], rocprof_avg_us(r["kernel"]), tag, tag),
    "| its algorithmic bytes | %.1f MB / launch ⇒ %.2f TB/s = **%.2f of the 8 TB/s HBM roofline** | §3 table |" % (
        r["algorithmic_bytes_per_launch"] / 1e6, r["achieved"] / 1e3, r["frac"]),
    "| its measured HBM traffic | %.1f MB / launch (2×FETCH_SIZE + WRITE_SIZE) = %.2f × algorithmic; the bench line's own PMC passes: %.1f MB | `profiles/%s_pmc_hbm_traffic.json`, `roofline.traffic` |" % (
        tr / 1e6, tr / r["algorithmic_bytes_per_launch"], (r["traffic"] or 0) / 1e6, tag),
    "| time split per iteration (instrumented pass) | SPD solves %.2f ms (1 `G_RR+λI` solve + 3 `G_tt` solves), operator applies %.2f ms, everything else %.2f ms | `%s_bench_n1.json` `kernels` |" % (
        spd, ops, other, tag),
    "| factorisation on the GPU (MFMA tile kernel `k_fa_abt`) | %s | `profiles/%s_mfma_utilisation.json` |" % (
        "; ".join("%.1f GFLOP in %.1f ms = %.1f TFLOP/s (%.2f of the 78.6 TFLOP/s FP64 matrix peak)" % (r_["GFLOP"], r_["ms"], r_["TFLOPs"], r_["fraction_of_peak"])
                  for r_ in mf["hip_event_rates_G_tt_then_G_RR"]) +
        "; matrix pipe busy %.2f of the CU-busy cycles" % mf["kernels"]["k_fa_abt"].get("mfma_busy_over_cu_busy", float("nan")), tag),
    "| solver | %.1f M / %.1f M factor entries, %d / %d tree levels (`G_tt` / `G_RR+λI`); one node per GPU: %d / %d levels | `solver` in both files |" % (
        j["solver"]["nnz_tt"] / 1e6, j["solver"]["nnz_rr"] / 1e6, j["solver"]["levels_tt"], j["solver"]["levels_rr"],
        e["solver"]["levels_tt"], e["solver"]["levels_rr"]),
]
path = os.path.join(ROOT, "DESIGN.md")
d = open(path).read()
a = d.index("| quantity | value | source |")
b = d.index("\n\n", a)
open(path, "w").write(d[:a] + "\n".join(rows) + d[b:])
print("\n".join(rows))
