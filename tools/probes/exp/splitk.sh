#!/bin/bash
# split-K on/off inside the full pipeline (isolated micro-runs of 20 us kernels leave the GPU in a low DPM state)
cd $GRAFT_REPO_ROOT
for b in 1 32; do for sk in 0 1; do
  echo "== batch $b splitk=$sk"
  HN_SPLITK=$sk python tools/layer_table.py f16x3 $b 2>&1 | grep -v amdgpu.ids | grep -E "^#|, 11, 11,|, 25, 34,|, 22, 22,"  | head -${1:-14}
  HN_SPLITK=$sk python bench.py --batch $b --no-cpu-baseline --no-roofline --steps 50 --warmup 10 2>&1 | tail -1 | cut -c1-150
done; done
