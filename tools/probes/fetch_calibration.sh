# On the GPU box: what FETCH_SIZE reports for the access shapes of the solve kernels (8 / 16 B per lane, plain / NT)
# against the bytes the stream probe really reads.  Usage: bash tools/probes/fetch_calibration.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o /tmp/stream_probe
rm -rf gpurun_out/fetch_cal
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/fetch_cal -- /tmp/stream_probe > gpurun_out/fetch_cal.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/fetch_cal/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[(r["Kernel_Name"][:60], int(r["Grid_Size"]) // 512 if "Grid_Size" in r else 0)].append(float(r["Counter_Value"]))
n = (256 << 20)
for (k, blocks), v in sorted(agg.items()):
    if blocks == 0: continue
    slab = n // blocks // 8192 * 8192
    true = slab * blocks * 8
    print("%-62s blocks %5d  FETCH_SIZE %10.1f KiB/launch  true %10.1f KiB  ratio true/reported %.3f" % (k, blocks, sum(v) / len(v), true / 1024, true / 1024 / (sum(v) / len(v))))
PY
