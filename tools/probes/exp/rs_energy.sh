#!/bin/bash
# where the row-shared kernel's remaining data-movement cost sits (timing-only libraries, results wrong by design, operands
# keep realistic values because the skipped DMA leaves stale tiles in LDS): no W-operand DMA in the loop (NOB), no A DMA (NOA)
cd $GRAFT_REPO_ROOT
for v in orig NOB NOA orig NOB NOA; do
  lib=""; [ $v != orig ] && lib=$GRAFT_REPO_ROOT/tools/probes/exp/lib_RS_$v.so
  echo -n "$v: "; HN_LIB_PATH=$lib python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 600 0 1 2>&1 | grep -v amdgpu.ids
done
