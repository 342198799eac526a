cd $GRAFT_REPO_ROOT
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); oc=d.get("other_configs") or {}; print(d["ms_per_step"], {k:v.get("ms_per_step") for k,v in oc.items() if isinstance(v,dict)})'
for f in "HN_CONV_NO_FUSED_REDUCE=1" ""; do
  echo "default bench ${f:-fused}: $(env $f python bench.py --no-cpu-baseline --no-dropin --steps 5 --warmup 2 2>/dev/null | python -c "$J")"
  echo "a2j b64 graph ${f:-fused}: $(env $f python bench.py --workload a2j --graph --no-cpu-baseline --no-roofline --steps 50 --warmup 10 2>/dev/null | python -c "$J")"
  echo "a2j b64 eager ${f:-fused}: $(env $f python bench.py --workload a2j --no-cpu-baseline --no-roofline --steps 50 --warmup 10 2>/dev/null | python -c "$J")"
done
