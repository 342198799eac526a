#!/bin/bash
# same-process-environment A/B of the row-shared A operand (RS) kernels: HN_CONV_NO_RS=1 selects the classic per-tap form
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in classic rs; do
  export HN_CONV_NO_RS=1; [ $v = rs ] && unset HN_CONV_NO_RS
  echo "== $v"
  python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 600 0 1 2>&1 | grep -v amdgpu.ids
  python tools/perf_conv.py f16x3 0 32 200 272 64 64 3 1 1 400 0 1 2>&1 | grep -v amdgpu.ids
  python tools/perf_conv.py f16x3 0 32 100 136 128 128 3 1 1 400 0 1 2>&1 | grep -v amdgpu.ids
  python tools/perf_conv.py f16x3 0 32 50 68 256 256 3 1 1 400 0 1 2>&1 | grep -v amdgpu.ids
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipeline b32', d['value'], d['ms_per_step'])"
done
done
