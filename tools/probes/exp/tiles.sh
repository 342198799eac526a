#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 3 7 2 6 1; do python tools/perf_conv.py f16x3 $t 32 11 11 256 256 3 1 1 3000 0 1 2>&1 | grep -v amdgpu.ids; done
for t in 3 7 2 6 1; do python tools/perf_conv.py f16x3 $t 32 11 11 1024 256 1 1 1 3000 0 1 2>&1 | grep -v amdgpu.ids; done
for t in 3 7 2 6 1; do python tools/perf_conv.py f16x3 $t 32 11 11 256 1024 1 1 1 3000 0 1 2>&1 | grep -v amdgpu.ids; done
for t in 3 7 2 6 1; do python tools/perf_conv.py f16x3 $t 32 11 11 2048 512 3 1 1 1000 0 1 2>&1 | grep -v amdgpu.ids; done
