#!/bin/bash
# in-frame scan of the split-K cost model's constants (csrc/conv_igemm_f16x3.hip plan_splits; hn_set_tuning through HN_TUNE_*):
# batch-1 frame under graph replay and A2J alone at 64 crops for one-at-a-time scale factors
cd $GRAFT_REPO_ROOT
run() {
  b1=$(python bench.py --batch 1 --graph --steps 300 --warmup 20 --no-cpu-baseline --no-roofline --no-dropin 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  a64=$(python bench.py --workload a2j --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$1: batch-1 frame $b1 ms, A2J 64 crops $a64 ms"
}
run "model as committed"
for k in SPLITK_RED0 SPLITK_FIX SPLITK_TK SPLITK_PLANE; do
  for v in 0.5 0.75 1.5 2.0; do
    export HN_TUNE_$k=$v
    run "$k x $v"
    unset HN_TUNE_$k
  done
done
run "model as committed (again)"
