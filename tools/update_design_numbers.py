#!/usr/bin/env python3
"""Rewrite the measured-numbers table of DESIGN.md section 6a from the committed artefacts
(profiles/rNN_bench_n1.json, rNN_bench_emulated_rank3of8.json, pmc_hbm_traffic.json).  Usage: update_design_numbers.py r01"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
j = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench_n1.json")))
e = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench_emulated_rank3of8.json")))
pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")))
K, r = j["kernels"], j["roofline"]


def rocprof_avg_us(kernel):
    """average duration of the family in the committed rocprofv3 --stats summary (the k_spd_level template is
    reported as k_spd_fwd / k_spd_bwd by its FWD argument)"""
    import csv
    import re
    calls = tot = 0
    for row in csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_kernel_stats_bench_default.csv"))):
        m = re.search(r"k_spd_level<([^>]*)>", row["Name"])
        if m:
            name = "k_spd_fwd" if m.group(1).split(",")[3].strip() == "true" else "k_spd_bwd"
        else:
            m2 = re.search(r"\b(k_[a-z_0-9]+)", row["Name"])
            name = m2.group(1) if m2 else row["Name"]
        if name == kernel:
            calls += int(row["Calls"])
            tot += float(row["TotalDurationNs"])
    return tot / calls / 1e3 if calls else float("nan")

tr = pm["kernels"][r["kernel"]]["hbm_bytes_per_launch"]
spd = K["k_spd_fwd"]["ms_per_step"] + K["k_spd_bwd"]["ms_per_step"]
other = sum(v["ms_per_step"] for k, v in K.items() if not k.startswith("k_spd") and k != "k_bsr")
rows = [
    "| quantity | value | source |", "|---|---|---|",
    "| throughput | **%.1f outer iterations / s**, %.2f ms / iteration | `profiles/%s_bench_n1.json` |" % (j["value"], j["ms_per_step"], tag),
    "| CPU baseline (oracle, 1 core, bounded sample: 1 of 8 nodes) | %.2f iterations / s | same file, `cpu_baseline` |" % j["cpu_baseline"]["value"],
    "| set-up (untimed) | %.1f s graph + chordal init, %.1f s operators + both factorizations | same file |" % (
        j["setup_s"]["graph+chordal_init"], j["setup_s"]["operators+factorizations"]),
    "| one rank of an 8-GPU run emulated on one GPU (1 node, frozen neighbours, no exchange) | %.2f ms / iteration | `profiles/%s_bench_emulated_rank3of8.json` (diagnostic, not a metric) |" % (e["ms_per_step"], tag),
    "| dominant kernel family | `%s`: %.0f launches / iteration, %.1f µs average (HIP events, events included) vs %.1f µs (rocprofv3 --stats) | `%s_bench_n1.json`, `%s_kernel_stats_bench_default.csv` |" % (
        r["kernel"], r["launches_per_step"], r["avg_launch_us"], rocprof_avg_us(r["kernel"]), tag, tag),
    "| its algorithmic bytes | %.1f MB / launch ⇒ %.2f TB/s = **%.2f of the 8 TB/s HBM roofline** | §3 table |" % (
        r["algorithmic_bytes_per_launch"] / 1e6, r["achieved"] / 1e3, r["frac"]),
    "| its measured HBM traffic | %.1f MB / launch (2×FETCH_SIZE + WRITE_SIZE) = %.2f × algorithmic | `profiles/pmc_hbm_traffic.json` |" % (
        tr / 1e6, tr / r["algorithmic_bytes_per_launch"]),
    "| time split per iteration | SPD solves %.2f ms (1 `G_RR+λI` solve + 3 `G_tt` solves), operator applies %.2f ms, everything else %.2f ms | `%s_bench_n1.json` `kernels` |" % (
        spd, K["k_bsr"]["ms_per_step"], other, tag),
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
