#!/bin/bash
# usage: tools/pmc_conv.sh <outdir> <perf_conv.py args...>   (run on the GPU box; 3 counter passes)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/perf_conv.py $*"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out/p1 -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out/p2 -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out/p3 -- $P > /dev/null 2>&1
