# The auxiliary round artefacts (run on the GPU box after tools/final_profile.sh): per-level solve tables, the one-node
# timeline, the Dynamic-rescale probe and the launches of one refactorisation.  Usage: bash tools/final_extras.sh r03
tag=${1:-r04}
out=gpurun_out/final
mkdir -p $out
bash tools/spd_profile.sh $tag
cp gpurun_out/spd_${tag}_n1.txt $out/${tag}_spd_levels.txt
cp gpurun_out/spd_${tag}_emu8.txt $out/${tag}_spd_levels_one_node.txt
bash tools/trace_levels.sh one_node --emulate-world 8 --emulate-rank 3
tail -60 gpurun_out/timeline_one_node.txt > $out/${tag}_timeline_last_step_one_node.txt
python3 tools/probes/dynamic_headline.py 50,50,40,400000 40 > $out/${tag}_dynamic_headline.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dyntr -- python3 tools/probes/dynamic_headline.py 50,50,40,400000 12 > gpurun_out/dyntr.log 2>&1
python3 tools/probes/dyn_trace.py gpurun_out/dyntr > $out/${tag}_dynamic_refactorisation_launches.txt
rm -rf gpurun_out/dyntr
tail -3 $out/${tag}_dynamic_headline.txt; tail -1 $out/${tag}_dynamic_refactorisation_launches.txt
