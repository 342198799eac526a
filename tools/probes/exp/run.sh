#!/bin/bash
# ablation timing: swap the library for variants with pieces of the hot loop removed (results are wrong by design)
cd $GRAFT_REPO_ROOT
L=handnet-pipeline_amd/csrc/libhandnet_hip.so
cp $L /tmp/lib_orig.so
echo "== full"; python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 20 0 0
for v in NO_DMA_A NO_DMA_B NO_DMA_ADEXP_NO_DMA_B NO_MFMA NO_LDSREAD; do
  cp tools/probes/exp/lib_$v.so $L
  echo "== $v"; python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 20 0 0
done
cp /tmp/lib_orig.so $L
