# in-frame A/B of the tile pick for grids of 256..767 128x128 tiles (hn_conv2d_f16x3_pick_tile): HN_TILE_SMALL_128x128=1 = rounds 1-4
cd $GRAFT_REPO_ROOT
for b in 1 2 3; do
  B="python bench.py --batch $b --graph --no-cpu-baseline --no-roofline --steps 200 --warmup 20"
  for rep in 1 2 3; do
    echo "batch $b old (128x128): $(HN_TILE_SMALL_128x128=1 $B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
    echo "batch $b new (128x64):  $($B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
  done
done
