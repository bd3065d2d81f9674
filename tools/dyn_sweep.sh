python -m pytest tests/test_gpu_factor.py -m gpu -x -q 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "switches or dynamic or rescale" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
DYN_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/dyntr -- python3 $R/tools/probes/dynamic_headline.py 50,50,40,400000 12 > $R/gpurun_out/dyntr.log 2>&1
cd $R
python3 tools/probes/dyn_trace.py gpurun_out/dyntr > gpurun_out/dyn_launches_new.txt
rm -rf gpurun_out/dyntr
