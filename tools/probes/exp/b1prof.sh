#!/bin/bash
# batch-1 (ros_demo live case) kernel breakdown: rocprofv3 stats of bench.py --batch 1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b1prof -- python3 $R/bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > $R/gpurun_out/b1prof.log 2>&1
tail -1 $R/gpurun_out/b1prof.log
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/b1prof/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"total kernel time {tot/60/1e3:.1f} us/step over {calls/60:.0f} launches/step")
for r in rows[:22]:
    print(f"{r['Name'][:86]:86s} {int(r['Calls'])/60:6.1f}/step avg {float(r['AverageNs'])/1e3:7.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
