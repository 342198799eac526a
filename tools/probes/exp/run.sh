#!/bin/bash
# ablation timing: swap the library for variants with pieces of the hot loop removed (results are wrong by design).
# The variants are built by compiling conv_igemm_f16x3.hip with the dma / lds-read / mfma statements of the k loop
# wrapped in #ifndef EXP_NO_DMA_A / EXP_NO_DMA_B / EXP_NO_LDSREAD (see the commit that added this file) into
# tools/probes/exp/lib_<variant>.so; do NOT build a no-MFMA variant: it faulted on the GPU.
# The variants are selected through HN_LIB_PATH (hn_amd/_lib.py); the product library is never overwritten.
cd $GRAFT_REPO_ROOT
echo "== full"; python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 20 0 0
for v in NO_DMA_A NO_DMA_B NO_DMA_ADEXP_NO_DMA_B NO_LDSREAD; do
  [ -f tools/probes/exp/lib_$v.so ] || continue
  echo "== $v"; HN_LIB_PATH=$GRAFT_REPO_ROOT/tools/probes/exp/lib_$v.so python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 20 0 0
done
