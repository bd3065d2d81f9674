# what is different on the boxes where the emulated one-node rank runs at 0.59-0.60 ms instead of 0.36-0.40?
mkdir -p gpurun_out/r4
ms=$(python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 2>/dev/null | python3 -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
echo "emu $ms" > gpurun_out/r4/slowbox.txt
rocm-smi --showclocks --showperflevel --showmemuse 2>&1 | head -40 >> gpurun_out/r4/slowbox.txt
cat /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/current_memory_partition 2>/dev/null >> gpurun_out/r4/slowbox.txt
lscpu | grep -E "Model name|MHz|^CPU\(s\)" >> gpurun_out/r4/slowbox.txt
nproc >> gpurun_out/r4/slowbox.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/r4/slowbox.txt
if python3 -c "import sys; sys.exit(0 if float('$ms') > 0.5 else 1)"; then
  bash tools/trace_levels.sh slowbox --emulate-world 8 --emulate-rank 3
  cp gpurun_out/timeline_slowbox.txt gpurun_out/r4/slowbox_timeline.txt
  DPGO_SPD_KEEP_MB=0 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 2>/dev/null | python3 -c "import json,sys; print('KEEP_MB=0 %.4f' % json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/r4/slowbox.txt
  python3 tools/trace_busy.py gpurun_out/trace_slowbox 4000 >> gpurun_out/r4/slowbox.txt 2>&1
fi
