cd $GRAFT_REPO_ROOT
P="python tools/perf_conv.py f16x3 0 32 200 272 64 64 3 1 1 100 0 1"
for rep in 1 2; do
  echo "base:  $($P 2>&1 | tail -1)"
  echo "range: $(HN_HALO_STAGGER_RANGE=1 $P 2>&1 | tail -1)"
  echo "odd:   $(HN_HALO_STAGGER_ODD=1 $P 2>&1 | tail -1)"
done
B="python bench.py --no-dropin --no-cpu-baseline --no-other-configs --steps 10 --warmup 3"
for rep in 1 2; do
  echo "bench base:  $($B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["stages"]["resnet34_body"])')"
  echo "bench range: $(HN_HALO_STAGGER_RANGE=1 $B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["stages"]["resnet34_body"])')"
  echo "bench odd:   $(HN_HALO_STAGGER_ODD=1 $B 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["stages"]["resnet34_body"])')"
done
