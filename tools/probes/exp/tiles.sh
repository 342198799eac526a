#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 1 9 5 1 9; do python tools/perf_conv.py f16x3 $t 32 100 136 256 256 3 1 1 800 0 0 2>&1 | grep -v amdgpu.ids; done
for t in 1 9; do python tools/perf_conv.py f16x3 $t 32 50 68 256 256 3 1 1 2000 0 0 2>&1 | grep -v amdgpu.ids; done
