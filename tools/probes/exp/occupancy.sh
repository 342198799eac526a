#!/bin/bash
# Are two 256x64 workgroups (80 KB LDS each) resident per CU?  Time a grid of 256 workgroups (one per CU) against 512
# and 768: if 512 takes about as long as 256, two are resident; if twice as long, one.
cd $GRAFT_REPO_ROOT
for tile in 8 2; do
  for h in 256 512 768 1024; do
    python tools/perf_conv.py f16x3 $tile 1 $h 256 64 64 3 1 1 300 0 1 2>&1 | grep -v amdgpu.ids
  done
done
