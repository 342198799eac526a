#!/bin/bash
# phase ablation of the Winograd prototype (timing only, results wrong by design): variants of the library without the
# input transform / the MFMAs / the global B-fragment loads, selected through HN_LIB_PATH.
#   build (container):  bash tools/probes/exp/wino_phases.sh build      run (GPU box):  bash tools/probes/exp/wino_phases.sh
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../../.. && pwd)}
cd $R
VARIANTS="NO_TRANSFORM NO_MFMA NO_BLOAD"
if [ "$1" = build ]; then
  python -m hn_amd.build >/dev/null 2>&1 || (cd handnet-pipeline_amd && python -m hn_amd.build >/dev/null)
  for v in $VARIANTS; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I$R/include -DWINO_$v -c handnet-pipeline_amd/csrc/wino_f16x3.hip -o /tmp/wino_$v.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/exp/lib_wino_$v.so /tmp/wino_$v.o \
        $(ls handnet-pipeline_amd/csrc/build/*.o | grep -v wino_f16x3.o)
  done
  exit
fi
for v in "" $VARIANTS; do
  echo "== ${v:-product}"
  HN_LIB_PATH=${v:+$R/tools/probes/exp/lib_wino_$v.so} python tools/probes/exp/wino.py tower 2>&1 | grep "^bench"
done
