# in-frame A/B of the mixed-tile grouped launch (conv_igemm_f16x3_mixed_kernel) at small batches; HN_CONV_NO_MIXED=1 = one tile shape
cd $GRAFT_REPO_ROOT
for b in 1 2 3 4; do
  B="python bench.py --batch $b --graph --no-cpu-baseline --no-roofline --steps 200 --warmup 20"
  for rep in 1 2 3; do
    echo "batch $b plain: $(HN_CONV_NO_MIXED=1 $B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
    echo "batch $b mixed: $($B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
  done
done
echo "batch 1 head_streams=2: $(HN_HEAD_STREAMS=2 python bench.py --batch 1 --graph --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
