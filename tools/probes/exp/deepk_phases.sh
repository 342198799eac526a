# Where does the deep-k kernel's time go?  Five builds of the library with parts of the stage loop compiled out
# (-DHN_DK_VARIANT: 0 complete, 1 no MFMAs, 2 no refill DMA, 3 no fragment reads, 4 no barrier; results are wrong by design in
# 1-4), the ResNet-34 layer3 convolution at batch 1, device-side duration from rocprofv3.  The variant libraries were built on the
# development host from a scratch edit of the stage loop (#if HN_DK_VARIANT ... around mfma_tile / dma_tile / read_frags / the
# barrier; NOT in the tree: the product has one loop) into handnet-pipeline_amd/csrc/build/libhn_dk_v<n>.so:
#   hipcc ... -DHN_DK_VARIANT=n -c conv_igemm_f16x3.hip -o v.o && hipcc -shared -o build/libhn_dk_v<n>.so <the other objects> v.o
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 0 1 2 3 4; do
  export HN_LIB_PATH=$R/handnet-pipeline_amd/csrc/build/libhn_dk_v$n.so
  rm -rf /tmp/dkv_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dkv_$n -- python3 $R/tools/perf_conv.py f16x3 12 1 50 68 256 256 3 1 1 300 0 1 > /dev/null 2>&1
  python3 - $n <<'PY'
import csv, glob, sys
n = sys.argv[1]
f = glob.glob(f"/tmp/dkv_{n}/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "deepk" in r["Name"]:
        print(f"variant {n}: {r['Calls']} calls, avg {float(r['AverageNs']) / 1e3:.2f} us, min {float(r['MinNs']) / 1e3:.2f} us")
PY
done
