#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 1 4; do for sk in 0 1; do for g in "" "--graph"; do
  echo -n "batch $b splitk=$sk $g: "
  HN_SPLITK=$sk python bench.py --batch $b --no-cpu-baseline --no-roofline --steps 100 --warmup 20 $g 2>&1 | tail -1 | cut -c60-150
done; done; done
