#!/bin/bash
# row-shared A: two 4-wave 128x128 workgroups per CU (tile 1) against one 8-wave 256x128 workgroup (tile 9, half the W traffic per CU)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv_gpu.py -x -q -k "row_shared" 2>&1 | tail -2
for rep in 1 2; do
for t in 1 9; do
  python tools/perf_conv.py f16x3 $t 32 100 136 256 256 3 1 1 600 0 1 2>&1 | grep -v amdgpu.ids
  python tools/perf_conv.py f16x3 $t 32 50 68 256 256 3 1 1 600 0 1 2>&1 | grep -v amdgpu.ids
  python tools/perf_conv.py f16x3 $t 32 100 136 128 128 3 1 1 600 0 1 2>&1 | grep -v amdgpu.ids
done
done
