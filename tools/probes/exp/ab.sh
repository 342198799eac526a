#!/bin/bash
# A/B: library built from the previous commit's conv kernel vs the working tree, long runs (power-capped chip).
# Build the comparison library first (not tracked):
#   git show <rev>:handnet-pipeline_amd/csrc/conv_igemm_f16x3.hip > /tmp/conv_igemm_f16x3_old.hip
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -Ihandnet-pipeline_amd/csrc -c /tmp/conv_igemm_f16x3_old.hip -o /tmp/conv_old.o
#   hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/exp/lib_old.so /tmp/conv_old.o handnet-pipeline_amd/csrc/build/{conv_igemm_f32,hn_common,split_ops,a2j_ops,groupnorm,fcos_post}.o
# The variant is selected through HN_LIB_PATH (hn_amd/_lib.py); the product library is never overwritten.
cd $GRAFT_REPO_ROOT
shapes=("1 32 100 136 256 256 3 1 1 600 0 0" "1 32 50 68 256 256 3 1 1 2000 0 0" "2 32 200 272 64 64 3 1 1 1500 0 1" "1 32 100 136 128 128 3 1 1 1500 0 1" "3 32 11 11 256 256 3 1 1 5000 0 1" "1 32 25 34 512 512 3 1 1 3000 0 1")
for v in old new old new; do
  lib=""; [ $v = old ] && lib=$GRAFT_REPO_ROOT/tools/probes/exp/lib_old.so
  echo "== $v"
  for s in "${shapes[@]}"; do HN_LIB_PATH=$lib python tools/perf_conv.py f16x3 $s 2>&1 | grep -v amdgpu.ids; done
done
