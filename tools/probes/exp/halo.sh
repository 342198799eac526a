#!/bin/bash
# what would fetching the A operand once per filter ROW (instead of once per tap) buy?  Ablation: skip the A
# DMA on two of three taps (results wrong by design, timing only).  lib_HALO.so = conv_igemm_f16x3.hip built with
# `if (cur_s == 0)` in front of dma_a_piece in the steady-state step.
# The variant is selected through HN_LIB_PATH (hn_amd/_lib.py); the product library is never overwritten.
cd $GRAFT_REPO_ROOT
for v in orig HALO orig HALO; do
  lib=""; [ $v = HALO ] && lib=$GRAFT_REPO_ROOT/tools/probes/exp/lib_HALO.so
  echo -n "$v: "; HN_LIB_PATH=$lib python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 600 0 0 2>&1 | grep -v amdgpu.ids
done
