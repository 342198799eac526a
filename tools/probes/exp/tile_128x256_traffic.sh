#!/bin/bash
# HBM traffic of the 128x128 and the 128x256w8 tile on the P3 tower shape: separate --pmc passes (never combined with traces)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tile256_pmc; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/probes/exp/tile_128x256.py 5 tower > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/probes/exp/tile_128x256.py 5 tower > $O/write.log 2>&1
python3 - <<PY
import csv, glob, collections
for name in ("fetch", "write"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % name, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "conv_igemm_f16x3_kernel" in k:
                agg[k[:110]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        print(name, k, "launches", len(v), "avg KB", round(sum(v) / len(v), 1))
PY
