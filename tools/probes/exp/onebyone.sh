#!/bin/bash
# 1x1 convolutions with short k loops: 128x128 (2 workgroups/CU) against 128x64 (3/CU) and 64x128
cd $GRAFT_REPO_ROOT
for shape in "32 100 136 128 256" "32 50 68 256 256" "32 44 44 64 256" "32 22 22 128 512" "32 200 272 64 128"; do
  for t in 1 2 6; do
    python tools/perf_conv.py f16x3 $t $shape 1 1 1 400 0 1 2>&1 | grep -v amdgpu.ids
  done
done
