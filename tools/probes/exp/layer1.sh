#!/bin/bash
# layer1 (64->64 3x3 on 200x272, 18 k steps per workgroup): LDS stage count / residency variants
# ids 10-12 are experimental (not in the product library): add to conv16_run's switch
#   case 10: launch16<128, 64, 2, 2, 3>   case 11: launch16<256, 64, 4, 1, 3>   case 12: launch16<128, 64, 4, 1, 3>
# r02 result (us): 256x64x2 467, 128x64x2 470, 128x64x3 504, 256x64x3 628 (one workgroup per CU), 128x64 4x1 x3 498
cd $GRAFT_REPO_ROOT
for t in 8 2 10 11 12 8; do
  python tools/perf_conv.py f16x3 $t 32 200 272 64 64 3 1 1 400 0 1 2>&1 | grep -v amdgpu.ids
done
