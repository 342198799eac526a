#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in ${BATCHES:-1 4 8 32}; do for hs in 1 6; do for g in "" "--graph"; do
  echo -n "batch $b head_streams=$hs $g: "
  HN_HEAD_STREAMS=$hs python bench.py --batch $b --no-cpu-baseline --no-roofline --steps 60 --warmup 12 $g 2>&1 | tail -1 | cut -c60-150
done; done; done
