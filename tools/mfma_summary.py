#!/usr/bin/env python3
"""MFMA utilisation of the device factorisation's tile kernel (k_fa_abt, spd_dev.hip) from a rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F64) and the HIP-event flop rate printed by
DPGO_SPD_DUMP.  Usage: mfma_summary.py <pmc dir> <rate txt> <out json>"""
import collections
import csv
import glob
import json
import os
import re
import sys

d, rate_txt, out = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_fa_[a-z_]+)", r["Kernel_Name"])
        if not m:
            continue
        agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            calls[m.group(1)] += 1
res = {"kernels": {}, "peak_fp64_mfma_TFLOPs": 78.6,
       "peak_source": "AMD Instinct MI355X data sheet, FP64 matrix 78.6 TFLOP/s (the microarchitecture guide lists no FP64 MFMA peak)"}
for k, c in agg.items():
    e = dict(c)
    e["launches"] = calls[k]
    busy, cu = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CU_CYCLES", 0.0)
    if cu > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES sums the matrix pipes of all four SIMDs of a CU in shader cycles (check: it is 64 cycles per
        # v_mfma_f64_16x16x4, 4 "MOPS" of 512 flops each), SQ_BUSY_CU_CYCLES counts a CU once: the pipes' share of the
        # cycles in which their CU has work is busy / (4 cu)
        e["mfma_pipe_busy_fraction_while_cu_busy"] = busy / (4.0 * cu)
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
    if mops > 0:
        e["mfma_busy_cycles_per_instruction"] = busy / (mops / 4.0)
        e["executed_GFLOP"] = mops * 512 / 1e9
    res["kernels"][k] = e
rates = []
for line in open(rate_txt):
    m = re.search(r"([0-9.]+) GFLOP in the MFMA tile kernel, ([0-9.]+) ms there = ([0-9.]+) TFLOP/s", line)
    if m:
        rates.append({"GFLOP": float(m.group(1)), "ms": float(m.group(2)), "TFLOPs": float(m.group(3)),
                      "fraction_of_peak": float(m.group(3)) / 78.6})
res["hip_event_rates_G_tt_then_G_RR"] = rates
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])
