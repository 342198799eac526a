#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 2 8 1 2 8; do python tools/perf_conv.py f16x3 $t 32 200 272 64 64 3 1 1 1200 0 1 2>&1 | grep -v amdgpu.ids; done
for t in 2 8; do python tools/perf_conv.py f16x3 $t 32 44 44 64 64 3 1 1 3000 0 1 2>&1 | grep -v amdgpu.ids; done
