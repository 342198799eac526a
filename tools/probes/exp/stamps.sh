#!/bin/bash
# build first (container): python tools/probes/exp/build_stamps.py
cd $GRAFT_REPO_ROOT
export HN_LIB_PATH=$GRAFT_REPO_ROOT/tools/probes/exp/lib_stamps.so
python tools/probes/exp/stamps.py 1 32 100 136 256 256 3 2>&1 | grep -v amdgpu
python tools/probes/exp/stamps.py 8 32 200 272 64 64 3 2>&1 | grep -v amdgpu
python tools/probes/exp/stamps.py 8 32 200 272 64 64 3 res 2>&1 | grep -v amdgpu
python tools/probes/exp/stamps.py 7 1 11 11 1024 256 1 2>&1 | grep -v amdgpu
