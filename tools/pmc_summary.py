import csv, glob, collections, sys
d = sys.argv[1]
vals = {}
dur = None
for f in glob.glob(f"{d}/p*/*/*counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        vals[k] = sum(v) / len(v)
for f in glob.glob(f"{d}/p1/*/*kernel_trace.csv"):
    ds = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "conv_igemm" in r["Kernel_Name"]]
    dur = sum(ds) / len(ds)
mf = vals.get("SQ_INSTS_MFMA", 1)
print(f"kernel avg {dur/1e3:.1f} us")
wc = vals["SQ_WAVE_CYCLES"] * 4
print(f"wave-cycles {wc/1e6:.0f}M  MFMA busy {vals['SQ_VALU_MFMA_BUSY_CYCLES']/1e6:.0f}M  "
      f"VALU/MFMA {vals['SQ_INSTS_VALU']/mf:.2f}  SALU/MFMA {vals.get('SQ_INSTS_SALU',0)/mf:.2f}  LDS/MFMA {vals.get('SQ_INSTS_LDS',0)/mf:.2f} VMEM/MFMA {vals.get('SQ_INSTS_VMEM',0)/mf:.2f} CVT/MFMA {vals.get('SQ_INSTS_VALU_CVT',0)/mf:.2f}")
for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
    if k in vals:
        print(f"  {k:24s} {100*vals[k]*4/wc:5.1f} % of wave-cycles")
print(f"  MFMA busy / wave-cycles  {100*vals['SQ_VALU_MFMA_BUSY_CYCLES']/wc:5.1f} %   coexec/MFMA busy {100*vals.get('SQ_VALU_MFMA_COEXEC_CYCLES',0)/vals['SQ_VALU_MFMA_BUSY_CYCLES']:5.1f} %")
print(f"  LDS idx active {vals.get('SQ_LDS_IDX_ACTIVE',0)/1e6:.0f}M  bank conflict {vals.get('SQ_LDS_BANK_CONFLICT',0)/1e6:.0f}M  data fifo full {vals.get('SQ_LDS_DATA_FIFO_FULL',0)/1e6:.0f}M cmd fifo full {vals.get('SQ_LDS_CMD_FIFO_FULL',0)/1e6:.0f}M")
