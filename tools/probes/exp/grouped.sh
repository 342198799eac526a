#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 1 4 32; do for gc in 0 1; do for g in "" "--graph"; do
  [ $b = 32 ] && [ "$g" = "--graph" ] && continue
  echo -n "batch $b grouped=$gc $g: "
  HN_GROUP_CONVS=$gc python bench.py --batch $b --no-cpu-baseline --no-roofline --steps 60 --warmup 12 $g 2>&1 | tail -1 | cut -c60-150
done; done; done
