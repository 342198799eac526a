cd $GRAFT_REPO_ROOT
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'
P=$GRAFT_REPO_ROOT/handnet-pipeline_amd/csrc/build/libhn_prev.so
for b in 1 2 4; do
  for rep in 1 2 3; do
    echo "pipeline batch $b prev: $(HN_LIB_PATH=$P python bench.py --batch $b --graph --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c "$J")"
    echo "pipeline batch $b new:  $(python bench.py --batch $b --graph --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c "$J")"
  done
done
echo "a2j b64 prev: $(HN_LIB_PATH=$P python bench.py --workload a2j --graph --no-cpu-baseline --no-roofline --steps 50 --warmup 10 2>/dev/null | python -c "$J")"
echo "a2j b64 new:  $(python bench.py --workload a2j --graph --no-cpu-baseline --no-roofline --steps 50 --warmup 10 2>/dev/null | python -c "$J")"
echo "pipeline b32 prev: $(HN_LIB_PATH=$P python bench.py --no-cpu-baseline --no-roofline --no-dropin --no-other-configs --steps 10 --warmup 3 2>/dev/null | python -c "$J")"
echo "pipeline b32 new:  $(python bench.py --no-cpu-baseline --no-roofline --no-dropin --no-other-configs --steps 10 --warmup 3 2>/dev/null | python -c "$J")"
