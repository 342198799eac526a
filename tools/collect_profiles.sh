#!/bin/bash
# Run on the GPU box (gpurun): collects everything the numbers in DESIGN.md / bench.py's roofline object are
# checked against into gpurun_out/<tag>/; tools/save_profiles.py then condenses it into profiles/.
#   usage: bash tools/collect_profiles.sh <tag>
# Separate rocprofv3 passes: kernel trace + stats, then one --pmc pass per HBM counter (never combined with traces).
set -u
tag=${1:-r06a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin --no-other-configs"
python3 $R/bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_pipeline_b32.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --no-clock-sample > $O/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --no-roofline > /dev/null 2>&1
python3 $R/tools/layer_table.py f16x3 32 2>&1 | grep -v amdgpu.ids > $O/layer_table_b32.txt
python3 $R/bench.py --workload a2j --no-cpu-baseline --steps 50 --warmup 10 2>&1 | tail -1 > $O/bench_a2j_b64.json
python3 $R/bench.py --workload fcos --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 > $O/bench_fcos_b16.json
python3 $R/bench.py --precision f32 --no-cpu-baseline --no-dropin --no-other-configs --steps 5 --warmup 2 2>&1 | tail -1 > $O/bench_pipeline_b32_f32.json
python3 $R/bench.py --batch 64 --no-cpu-baseline --no-roofline --no-dropin --no-other-configs --steps 10 --warmup 3 2>&1 | tail -1 > $O/bench_pipeline_b64.json
python3 $R/bench.py --batch 1 --graph --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>&1 | tail -1 > $O/bench_pipeline_b1_graph.json
python3 $R/bench.py --batch 1 --native --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>&1 | tail -1 > $O/bench_pipeline_b1_native.json
bash $R/tools/probes/exp/b1trace.sh ${tag}_b1trace --graph > /dev/null 2>&1; cp $R/gpurun_out/${tag}_b1trace/step_timeline.txt $O/b1_timeline.txt
bash $R/tools/probes/exp/clocks.sh > $O/clocks_power.txt 2>&1
# round 6: the lifter alone (kernel stats of 50 replayed forwards at batch 1) and the live chain's launch timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2m -- python3 $R/bench.py --workload pose2mesh --batch 1 --graph --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_pose2mesh_b1.json 2>/dev/null
python3 $R/bench.py --workload pose2mesh --batch 1 --graph --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_pose2mesh_b1.json
python3 $R/bench.py --workload pose2mesh --batch 32 --graph --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_pose2mesh_b32.json
python3 $R/bench.py --workload live --batch 1 --graph --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_live_b1.json
bash $R/tools/probes/exp/b1trace.sh ${tag}_livetrace --graph --workload live > /dev/null 2>&1; cp $R/gpurun_out/${tag}_livetrace/step_timeline.txt $O/live_b1_timeline.txt
echo collected into $O
