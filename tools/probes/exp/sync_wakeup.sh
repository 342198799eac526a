# Does the host's wait policy matter for the drop-in's one sync per call at batch 1?  (dropin.batch1 ms per call, engine ms per step)
# NOT in the list any more: ROC_SYSTEM_SCOPE_SIGNAL=0 -- the run stopped answering with it and was killed after 7 silent minutes.
cd $GRAFT_REPO_ROOT
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["dropin"]["batch1"]["ms_per_call"], d["ms_per_step"])'
for rep in 1 2; do
  for f in "" "ROC_ACTIVE_WAIT_TIMEOUT=5000" "ROC_ACTIVE_WAIT_TIMEOUT=100000" "HIP_LAUNCH_BLOCKING=0 ROC_CPU_WAIT_FOR_SIGNAL=0"; do
    echo "dropin batch 1 ${f:-default}: $(env $f python bench.py --batch 1 --no-cpu-baseline --no-roofline --no-other-configs --steps 100 --warmup 10 2>/dev/null | python -c "$J")"
  done
done
