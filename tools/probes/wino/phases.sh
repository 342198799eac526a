#!/bin/bash
# phase ablation of the v2 prototype (timing only, results wrong by design): without the input transform / the MFMAs /
# the global B-fragment loads / the patch DMA.
#   build (container):  bash tools/probes/wino/phases.sh build      run (GPU box):  bash tools/probes/wino/phases.sh
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../../.. && pwd)}
cd $R
VARIANTS="NO_TRANSFORM NO_MFMA NO_BLOAD NO_DMA"
if [ "$1" = build ]; then
  bash tools/probes/wino/build.sh
  for v in $VARIANTS; do WINO_TAG=$v bash tools/probes/wino/build.sh -DWINO_$v >/dev/null; rm -f tools/probes/wino/libwino_v1_$v.so; done
  exit
fi
for v in "" $VARIANTS; do
  echo "== v2 ${v:-complete}"
  WINO_LIB=$R/tools/probes/wino/libwino_v2${v:+_$v}.so python tools/probes/wino/wino.py tower 2>&1 | grep "^bench"
done
