#!/bin/bash
# upper bound for producing the taps' A fragments from each other by lane shifts (DPP row_shl) instead of re-reading them
# from LDS: a timing-only library (lib_AFRAG.so, results wrong by design) whose row-shared kernels read A fragments for
# tap 0 only and keep them in registers for taps 1 and 2 -- realistic operand values, a third of the A fragment reads.
cd $GRAFT_REPO_ROOT
for v in orig AFRAG orig AFRAG; do
  lib=""; [ $v = AFRAG ] && lib=$GRAFT_REPO_ROOT/tools/probes/exp/lib_AFRAG.so
  echo -n "$v: "; HN_LIB_PATH=$lib python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 600 0 1 2>&1 | grep -v amdgpu.ids
done
