#!/bin/bash
# MFMA-count elasticity of the f16x3 conv (timing only, results wrong by design): a variant of the library that issues
# 2 of the 3 MFMA terms per MAC, same DMA / LDS traffic, selected through HN_LIB_PATH (the product .so is not touched).
# How the time falls with the MFMA count bounds what a Winograd F(2x2,3x3) kernel (1.33 MFMA per MAC) can gain.
#   build (container):  bash tools/probes/exp/terms.sh build      run (GPU box):  bash tools/probes/exp/terms.sh
# Do NOT build a 1-term variant by dropping terms: fragment registers whose pinned ds_read result is never consumed
# are re-used by the compiler while the asynchronous read is still in flight -> corrupted addresses -> GPU memory
# fault (seen once, r02; same cause as the round-1 "no-MFMA" fault).
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../../.. && pwd)}
cd $R
if [ "$1" = build ]; then
  mkdir -p /tmp/exp
  sed -e 's|#include "hn_common.h"|#include "'$R'/handnet-pipeline_amd/csrc/hn_common.h"|' \
      -e 's|mfma_pinned(acc\[i\]\[jj\], term|if constexpr (term < 2) mfma_pinned(acc[i][jj], term|' \
      -e 's|mfma_pinned(acc\[i\]\[TH + jj\], term|if constexpr (term < 2) mfma_pinned(acc[i][TH + jj], term|' \
      handnet-pipeline_amd/csrc/conv_igemm_f16x3.hip > /tmp/exp/x2.hip
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I$R/include -c /tmp/exp/x2.hip -o /tmp/exp/x2.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/exp/lib_terms2.so /tmp/exp/x2.o \
      $(ls handnet-pipeline_amd/csrc/build/*.o | grep -v conv_igemm_f16x3.o)
  exit
fi
for lib in "" tools/probes/exp/lib_terms2.so ""; do
  echo "== ${lib:-product (3 terms)}"
  HN_LIB_PATH=${lib:+$R/$lib} python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 1500 0 1 2>&1 | grep -v amdgpu.ids
done
