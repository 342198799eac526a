#!/bin/bash
# what would fetching the A operand once per filter ROW (instead of once per tap) buy?  Ablation: skip the A
# DMA on two of three taps (results wrong by design, timing only).  lib_HALO.so = conv_igemm_f16x3.hip built with
# `if (cur_s == 0)` in front of dma_a_piece in the steady-state step.
cd $GRAFT_REPO_ROOT
L=handnet-pipeline_amd/csrc/libhandnet_hip.so
cp $L /tmp/lib_orig.so
for v in orig HALO orig HALO; do
  if [ $v = orig ]; then cp /tmp/lib_orig.so $L; else cp tools/probes/exp/lib_$v.so $L; fi
  echo -n "$v: "; python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 600 0 0 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_orig.so $L
