#!/bin/bash
# Winograd F(2x2,3x3) prototypes (not part of the product library): the product objects + one prototype kernel -> a
# library selected through HN_LIB_PATH.   bash tools/probes/wino/build.sh [extra -D flags for the v2 ablations]
#   libwino_v1.so  wino_f16x3.hip     S32 input, 128 output channels per workgroup, phases serialised, B prefetched
#   libwino_v2.so  wino_f16x3_v2.hip  fp32 input, 64 output channels per workgroup, transform woven into the MFMA phases
R=$(cd $(dirname $0)/../../.. && pwd)
cd $R
(cd handnet-pipeline_amd && python -m hn_amd.build >/dev/null) || exit 1
for v in v1 v2; do
  src=tools/probes/wino/wino_f16x3$([ $v = v2 ] && echo _v2).hip
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I$R/include "$@" -c $src -o /tmp/wino_$v.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/wino/libwino_$v${WINO_TAG:+_$WINO_TAG}.so /tmp/wino_$v.o handnet-pipeline_amd/csrc/build/*.o || exit 1
done
ls -la tools/probes/wino/*.so
