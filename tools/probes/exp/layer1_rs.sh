#!/bin/bash
# ResNet-34 layer1 (64 -> 64, 200x272, batch 32): per-tap 256x64 (tile 8), row-shared 128x64 (tile 2), row-shared 256x64 with 8 waves (tile 10)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv_gpu.py -x -q -k "row_shared" 2>&1 | tail -2
for rep in 1 2; do
for t in 8 2 10; do
  python tools/perf_conv.py f16x3 $t 32 200 272 64 64 3 1 1 400 0 1 2>&1 | grep -v amdgpu.ids
done


done
