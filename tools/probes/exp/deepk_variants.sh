# in-frame A/B at batch 1 of the 64x64 deep-k ring shapes (HN_TUNE_DEEPK_VARIANT: 0 = 2 tiles x 3 stages, 1 = 2 x 4, 2 = 3 x 3, 3 = 4 x 2, 4 = 2 x 2, 5 = 3 x 2)
# and of "all three deep-k forms" (HN_TUNE_DEEPK_ALL=2) against the pinned loop (HN_CONV_NO_DEEPK=1)
cd $GRAFT_REPO_ROOT
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'
for rep in 1 2 3; do
  for f in "HN_CONV_NO_DEEPK=1" "HN_TUNE_DEEPK_VARIANT=0.000001" "HN_TUNE_DEEPK_VARIANT=4" "HN_TUNE_DEEPK_VARIANT=5"; do
    echo "batch 1 $f: $(env $f python bench.py --batch 1 --graph --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c "$J")"
  done
done
