#!/bin/bash
# batch-1 (ros_demo live case) timeline: rocprofv3 kernel trace of bench.py --batch 1 [--graph], one steady-state
# step printed launch by launch (start offset, duration, gap to the previous kernel's end)
#   usage: bash tools/probes/exp/b1trace.sh <tag> [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-b1trace}; shift
O=$R/gpurun_out/$tag
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --batch 1 --steps 30 --warmup 10 --no-cpu-baseline --no-roofline "$@" > $O/bench.log 2>&1
tail -1 $O/bench.log
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/trace/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the preprocess kernel
starts = [i for i, r in enumerate(rows) if "fcos_preprocess" in r["Kernel_Name"]]
a, b = starts[-3], starts[-2]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
busy = 0
out = open(O + "/step_timeline.txt", "w")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]
    out.write(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  grid {r.get('Grid_Size_X','?'):>8s} wg {r.get('Workgroup_Size_X','?'):>4s}  {name}\n")
    prev_end = e
span = int(step[-1]["End_Timestamp"]) - t0
out.write(f"# {len(step)} launches, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us\n")
out.close()
print(open(O + "/step_timeline.txt").read()[-400:])
PY
