# The deep-k loop with its stage re-ordered (tile-0 fragment reads -> refill DMA issue -> tile-1 reads -> MFMAs, one basic block
# per stage): isolated ResNet-34 layer3 at batch 1 in both the deep-k form and its parent tile, then the batch-1 / 2 / 4 frames.
# Run once per library build (there is no form switch for the schedule): compare with profiles/r05_deepk_ab.txt / the previous run.
cd $GRAFT_REPO_ROOT
for t in 12 3; do
  for shp in "1 50 68 256 256 3" "1 50 68 256 256 1" "2 50 68 256 256 3"; do
    python tools/perf_conv.py f16x3 $t $shp 1 1 300 0 1 2>&1 | tail -1
  done
done
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'
for b in 1 2 4; do
  for rep in 1 2 3; do
    for f in "HN_CONV_NO_DEEPK=1" ""; do
      echo "pipeline batch $b ${f:-deepk}: $(env $f python bench.py --batch $b --graph --no-cpu-baseline --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c "$J")"
    done
  done
done
