#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into per-kernel-family HBM traffic per launch.

Usage (on the GPU box, separate passes as MI355X_MICROARCH.md prescribes -- FETCH_SIZE and
WRITE_SIZE do not fit one pass):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/pmc_write -- python3 bench.py ...
    python3 tools/pmc_summary.py out/pmc_fetch out/pmc_write profiles/rNN_pmc_hbm_traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (128-byte requests
tallied at 64 B), so the read side is doubled; WRITE_SIZE is taken as is.  Kernel families are the
bench.py profiler's names (template arguments stripped).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def family(name):
    m = re.search(r"\b(k_[a-z_0-9]+)", name)
    n = m.group(1) if m else name
    if n == "k_spd_level":   # template <D, DOF, ROWS, MODE, NT>: MODE 0 forward level, 1 backward level, 2 the fused roots
        a = re.search(r"k_spd_level<([^>]*)>", name)   # (the bench's profiler counts the roots with the forward sweep)
        fwd = a is not None and a.group(1).split(",")[3].strip() in ("0", "2")
        return "k_spd_fwd" if fwd else "k_spd_bwd"
    # (the roots stored as one triangle and their combine pass are part of the forward sweep's scope in the bench's profiler)
    return {"k_cg_init": "k_axpby", "k_extrapolate": "k_axpby", "k_axpby_node": "k_axpby", "k_dots": "k_dot",
            "k_tangent_full": "k_rot_op", "k_root_sym": "k_spd_fwd", "k_root_combine": "k_spd_fwd"}.get(n, n)


def collect(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = family(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "50,50,40,400000"
    n_gpus = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    fa, wa = collect(fetch, "FETCH_SIZE"), collect(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fa) | set(wa)):
        if not k.startswith("k_"):
            continue
        nf, sf = fa.get(k, [0, 0.0])
        nw, sw = wa.get(k, [0, 0.0])
        rd = 2.0 * 1024.0 * sf / max(nf, 1)     # gfx950: FETCH_SIZE counts half the bytes
        wr = 1024.0 * sw / max(nw, 1)
        res[k] = {"launches_fetch_pass": nf, "launches_write_pass": nw,
                  "fetch_size_raw_KiB_per_launch": sf / max(nf, 1), "write_size_raw_KiB_per_launch": sw / max(nw, 1),
                  "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr}
    json.dump({"method": __doc__.strip().splitlines()[0], "corrections": "read = 2 x FETCH_SIZE x 1024; write = WRITE_SIZE x 1024",
               "command": " ".join(sys.argv), "workload": workload, "n_gpus": n_gpus, "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_fetch_pass"]):
        print("%-16s launches %6d  HBM MB/launch %9.3f" % (k, v["launches_fetch_pass"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
