#!/bin/bash
# instruction-cache behaviour of the conv kernels (code size: ~23k instructions per instantiation, mostly the unrolled epilogue)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH" | head -20
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d $R/gpurun_out/icache -- python3 $R/tools/perf_conv.py f16x3 8 32 200 272 64 64 3 1 1 10 0 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/icache/*/*counter_collection.csv")
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "conv_igemm" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, sum(v) / len(v))
PY
