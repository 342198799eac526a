#!/bin/bash
# 8-wave variants of the big tile on the tower layer: ids 10 (128x128, 32x64 wave tiles), 11 (128x128, 64x32), 12 (256x128, 3 stages), 9 (256x128 w8)
# NOTE: ids 10 - 12 are not in the product library.  To repeat the measurement add to conv16_run's switch
#   case 10: return launch16<128, 128, 4, 2, 2>(p, st);  case 11: return launch16<128, 128, 2, 4, 2>(p, st);
#   case 12: return launch16<256, 128, 4, 2, 3>(p, st);
# Results: profiles/r02_tile_variants_power_wall.txt.
cd $GRAFT_REPO_ROOT
for t in 1 10 11 12 9 1; do
  python tools/perf_conv.py f16x3 $t 32 100 136 256 256 3 1 1 800 0 1 2>&1 | grep -v amdgpu.ids
done
