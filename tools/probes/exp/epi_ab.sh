#!/bin/bash
# same-box A/B of the working-tree library against tools/probes/exp/lib_old.so (HN_LIB_PATH)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in old new; do
  lib=""; [ $v = old ] && lib=$GRAFT_REPO_ROOT/tools/probes/exp/lib_old.so
  echo "== $v"
  HN_LIB_PATH=$lib python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 600 0 1 2>&1 | grep -v amdgpu.ids
  HN_LIB_PATH=$lib python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 600 0 0 2>&1 | grep -v amdgpu.ids
  HN_LIB_PATH=$lib python tools/perf_conv.py f16x3 0 32 200 272 64 64 3 1 1 400 0 1 2>&1 | grep -v amdgpu.ids
  HN_LIB_PATH=$lib python tools/perf_conv.py f16x3 0 32 100 136 128 256 1 1 1 400 0 1 2>&1 | grep -v amdgpu.ids
  HN_LIB_PATH=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipeline b32', d['value'], d['ms_per_step'])"
  
done
done
