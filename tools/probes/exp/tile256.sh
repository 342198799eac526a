#!/bin/bash
# 256x256 / 8-wave tiles (ids 10, 11) against the 128x128 tile on the big 3x3 layers; >= 600 launches each
# NOTE: ids 10 / 11 are not in the product library.  To repeat the measurement add to conv16_run's switch
#   case 10: return launch16<256, 256, 2, 4, 2>(p, st);   case 11: return launch16<256, 256, 4, 2, 2>(p, st);
# (both spill ~66 VGPRs with 2 scratch reloads per k step, and id 11 failed test_conv_f16x3_matches_fp64_reference).
cd $GRAFT_REPO_ROOT
for t in 1 10 11 1 10 11; do
  python tools/perf_conv.py f16x3 $t 32 100 136 256 256 3 1 1 600 0 1
done
for t in 1 10 11; do
  python tools/perf_conv.py f16x3 $t 32 100 136 128 128 3 1 1 600 0 1
  python tools/perf_conv.py f16x3 $t 32 50 68 256 256 3 1 1 600 0 1
done
