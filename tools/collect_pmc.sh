#!/bin/bash
# Run on the GPU box (gpurun): the three PMC passes of tools/pmc_conv.sh on the tower shape (100 x 136 x 256 -> 256, 3x3, batch
# 32: the dominant kernel of the bench line) + the summary, stamped with the hash of the kernel source it was collected for
# (the same stamp bench.py pairs the HBM-traffic file by).   usage: bash tools/collect_pmc.sh <tag>
tag=${1:-r05}
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_conv.sh ${tag}_pmc f16x3 1 32 100 136 256 256 3 1 1 30 0 1
out=$R/gpurun_out/${tag}_pmc_tower_conv.txt
python3 $R/tools/pmc_summary.py $R/gpurun_out/${tag}_pmc > $out
sha=$(sha256sum $R/handnet-pipeline_amd/csrc/conv_igemm_f16x3_kernel.h | cut -c1-16)
echo "# collected by tools/collect_pmc.sh $tag: bash tools/pmc_conv.sh ${tag}_pmc f16x3 1 32 100 136 256 256 3 1 1 30 0 1" >> $out
echo "# kernel_source_sha16 $sha (sha256 of handnet-pipeline_amd/csrc/conv_igemm_f16x3_kernel.h, first 16 hex digits)" >> $out
cat $out
