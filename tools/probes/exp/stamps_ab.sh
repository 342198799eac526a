#!/bin/bash
# phase stamps of one conv shape with two diagnostic builds (lib_stamps_old.so = HEAD~ epilogue, lib_stamps.so = working tree)
cd $GRAFT_REPO_ROOT
for lib in lib_stamps_old.so lib_stamps.so; do
  echo "== $lib"
  export HN_LIB_PATH=$GRAFT_REPO_ROOT/tools/probes/exp/$lib
  python tools/probes/exp/stamps.py "$@" 2>&1 | grep -v amdgpu
done
