set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2m -- python3 $R/bench.py --workload pose2mesh --batch 1 --graph --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python3 $R/bench.py --workload pose2mesh --batch 1 --graph --steps 300 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_pose2mesh_b1.json
python3 $R/bench.py --workload pose2mesh --batch 32 --graph --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_pose2mesh_b32.json
python3 $R/bench.py --workload live --batch 1 --graph --steps 300 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_live_b1.json
python3 $R/bench.py --batch 1 --graph --no-cpu-baseline --no-roofline --steps 300 --warmup 20 2>/dev/null | tail -1 > $O/bench_pipeline_b1_graph.json
bash $R/tools/probes/exp/b1trace.sh r06b_livetrace --graph --workload live > /dev/null 2>&1; cp $R/gpurun_out/r06b_livetrace/step_timeline.txt $O/live_b1_timeline.txt
python3 $R/bench.py --workload a2j --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tail -1 > $O/bench_a2j_b64.json
python3 $R/tools/layer_table_a2j.py 2>/dev/null > $O/layer_table_a2j_b64.txt
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_default.json 2>/dev/null
echo collected
