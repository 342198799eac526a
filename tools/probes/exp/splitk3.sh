#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 1 2 4 8; do for e in 0 1; do
  echo -n "batch $b eager-aggressive=$e: "
  HN_SPLITK_EAGER=$e python bench.py --batch $b --no-cpu-baseline --no-roofline --steps 100 --warmup 20 2>&1 | tail -1 | cut -c60-150
done; done
