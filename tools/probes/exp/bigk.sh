# in-frame A/B at batch 1: the 64x64 tile instead of 32x64 for tiny maps with at least HN_TUNE_BIGK_KTILES k tiles
cd $GRAFT_REPO_ROOT
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'
for rep in 1 2 3; do
  for f in "" "HN_TUNE_BIGK_KTILES=512" "HN_TUNE_BIGK_KTILES=256" "HN_TUNE_BIGK_KTILES=128"; do
    echo "batch 1 ${f:-off}: $(env $f python bench.py --batch 1 --graph --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c "$J")"
  done
done
