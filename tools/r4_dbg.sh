mkdir -p gpurun_out/r4
python tools/probes/star_trace.py 2>&1 | md5sum > gpurun_out/r4/star_det.txt
DPGO_STAR_BATCH=0 python tools/probes/star_trace.py 2>&1 | md5sum >> gpurun_out/r4/star_det.txt
