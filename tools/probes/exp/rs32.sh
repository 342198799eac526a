#!/bin/bash
# row-shared A on the 128x32 tile (the 5-channel FCOS output convs): HN_CONV_NO_RS32=1 keeps the 3-stage per-tap kernel
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in classic rs32; do
  export HN_CONV_NO_RS32=1; [ $v = rs32 ] && unset HN_CONV_NO_RS32
  echo "== $v"
  python tools/perf_conv.py f16x3 0 32 100 136 256 5 3 1 1 400 0 0 2>&1 | grep -v amdgpu.ids
  python tools/perf_conv.py f16x3 0 32 50 68 256 5 3 1 1 400 0 0 2>&1 | grep -v amdgpu.ids
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipeline b32', d['value'], d['ms_per_step'])"
done
done
