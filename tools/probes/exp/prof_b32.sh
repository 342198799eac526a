cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof_b32 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-clock-sample --no-dropin > $R/gpurun_out/r03_prof_b32.log 2>&1
python3 - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/r03_prof_b32/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = 17
print(f"total kernel time {tot/steps/1e6:.2f} ms/step")
for r in rows[:24]:
    print(f"{r['Name'].replace('(anonymous namespace)::','')[:80]:80s} {int(r['Calls'])/steps:6.1f}/step avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/steps/1e6:7.3f} ms/step {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
