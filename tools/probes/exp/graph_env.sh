# Does any runtime switch of the HIP graph path change the launch-to-launch gap of the batch-1 replay (148 launches)?
# Three alternating rounds; each line is ms per step of `bench.py --batch 1 --graph`.
cd $GRAFT_REPO_ROOT
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'
for rep in 1 2 3; do
  for f in "" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "AMD_OPT_FLUSH=0" "DEBUG_HIP_GRAPH_BATCH_SIZE=256" "DEBUG_HIP_GRAPH_BATCH_SIZE=16" "DEBUG_HIP_KERNARG_COPY_OPT=0" "HIP_FORCE_DEV_KERNARG=0" "ROC_ACTIVE_WAIT_TIMEOUT=100" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "GPU_MAX_HW_QUEUES=1"; do
    echo "pipeline batch 1 ${f:-default}: $(env $f python bench.py --batch 1 --graph --no-cpu-baseline --no-roofline --steps 300 --warmup 30 2>/dev/null | python -c "$J")"
  done
done
