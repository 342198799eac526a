#!/bin/bash
# round-2 rerun of halo.sh on the v6 kernel: the A-operand DMA of taps s != 0 is made out-of-range (descriptor zero fill:
# the instruction still issues and still writes LDS, but no L2 traffic) -- what would fetching A once per filter ROW buy?
# lib_HALO2.so = conv_igemm_f16x3.hip with `| (cur_s != 0 ? 0x80000000u : 0u)` on the A piece offset.  Timing only.
cd $GRAFT_REPO_ROOT
for v in orig HALO2 orig HALO2; do
  lib=""; [ $v = HALO2 ] && lib=$GRAFT_REPO_ROOT/tools/probes/exp/lib_HALO2.so
  echo "== $v"
  HN_LIB_PATH=$lib python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 600 0 1 2>&1 | grep -v amdgpu.ids
  HN_LIB_PATH=$lib python tools/perf_conv.py f16x3 0 32 200 272 64 64 3 1 1 400 0 1 2>&1 | grep -v amdgpu.ids
done
