#!/bin/bash
# fixed per-workgroup cost (prologue + epilogue) vs per-k-step cost: same M and Cout, Cin = 32 / 64 / 128 / 256
# (9 / 18 / 36 / 72 k steps), 256x64 tile and 128x128 tile
cd $GRAFT_REPO_ROOT
for cin in 32 64 128 256; do python tools/perf_conv.py f16x3 8 32 200 272 $cin 64 3 1 1 300 0 1 2>&1 | grep -v amdgpu.ids; done
for cin in 32 64 128 256; do python tools/perf_conv.py f16x3 1 32 100 136 $cin 256 3 1 1 300 0 1 2>&1 | grep -v amdgpu.ids; done
for cin in 32 64 128 256; do python tools/perf_conv.py f16x3 8 32 200 272 $cin 64 1 1 1 300 0 1 2>&1 | grep -v amdgpu.ids; done
