# A/B of the streaming 1x1 kernel (conv1x1_stream.hip) against the implicit-GEMM form (HN_CONV_NO_STREAM=1): isolated layers, the
# batch-32 frame and A2J at 64 crops
cd $GRAFT_REPO_ROOT
for f in "" "HN_CONV_NO_STREAM=1"; do
  echo "== ${f:-stream}"
  env $f python tools/perf_conv.py f16x3 0 32 100 136 128 256 1 1 1 200 0 1 2>&1 | tail -1
  env $f python tools/perf_conv.py f16x3 0 64 44 44 64 256 1 1 1 200 0 1 2>&1 | tail -1
  env $f python tools/perf_conv.py f16x3 0 32 44 44 64 256 1 1 1 200 0 1 2>&1 | tail -1
done
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in (d["roofline"] or {}).get("stages", {}).items()})'
for rep in 1 2; do
  for f in "" "HN_CONV_NO_STREAM=1"; do
    echo "pipeline b32 ${f:-stream}: $(env $f python bench.py --no-dropin --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 2>/dev/null | python -c "$J")"
    echo "a2j b64 ${f:-stream}: $(env $f python bench.py --workload a2j --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "$J")"
  done
done
