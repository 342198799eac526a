# A/B of the split-K reduction inside the last workgroup of a tile against the separate reduction launch (HN_CONV_NO_FUSED_REDUCE=1):
# isolated split layers at batch 1, the batch-1 / 2 / 4 frames, A2J alone at 1 and 64 crops, the batch-32 frame
cd $GRAFT_REPO_ROOT
for f in "HN_CONV_NO_FUSED_REDUCE=1" ""; do
  echo "== ${f:-fused}"
  for shp in "1 25 34 512 512 3" "1 11 11 512 512 3" "1 11 11 1024 256 1" "1 11 11 2048 512 3" "1 11 11 256 256 3"; do
    env $f python tools/perf_conv.py f16x3 0 $shp 1 1 300 0 1 2>&1 | tail -1
  done
done
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'
for b in 1 2 4; do
  for rep in 1 2 3; do
    for f in "HN_CONV_NO_FUSED_REDUCE=1" ""; do
      echo "pipeline batch $b ${f:-fused}: $(env $f python bench.py --batch $b --graph --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c "$J")"
    done
  done
done
for f in "HN_CONV_NO_FUSED_REDUCE=1" ""; do
  echo "a2j batch 1 ${f:-fused}: $(env $f python bench.py --workload a2j --batch 1 --no-cpu-baseline --no-roofline --steps 300 --warmup 30 2>/dev/null | python -c "$J")"
  echo "a2j batch 64 ${f:-fused}: $(env $f python bench.py --workload a2j --no-cpu-baseline --no-roofline --steps 50 --warmup 10 2>/dev/null | python -c "$J")"
  echo "pipeline batch 32 ${f:-fused}: $(env $f python bench.py --no-cpu-baseline --no-roofline --no-dropin --no-other-configs --steps 10 --warmup 3 2>/dev/null | python -c "$J")"
done
