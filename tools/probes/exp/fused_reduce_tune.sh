# With the reduction inside the last workgroup, do the constants of the split-K cost model (fitted with the separate launch) still
# sit at their optimum?  Batch-1 frame under graph replay, two rounds.
cd $GRAFT_REPO_ROOT
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'
for rep in 1 2; do
  for f in "" "HN_TUNE_SPLITK_RED0=0.25" "HN_TUNE_SPLITK_RED0=0.5" "HN_TUNE_SPLITK_RED0=2" "HN_TUNE_SPLITK_PLANE=0.5" "HN_TUNE_SPLITK_PLANE=2" "HN_TUNE_SPLITK_PLANE=4" "HN_TUNE_SPLITK_RED0=0.5 HN_TUNE_SPLITK_PLANE=2" "HN_TUNE_SPLITK_FIX=0.75" "HN_TUNE_SPLITK_FIX=1.5"; do
    echo "pipeline batch 1 ${f:-default}: $(env $f python bench.py --batch 1 --graph --no-cpu-baseline --no-roofline --steps 300 --warmup 30 2>/dev/null | python -c "$J")"
  done
done
