#!/bin/bash
cd $GRAFT_REPO_ROOT
for hs in 1 6 1 6 2 3; do
  echo "== HN_HEAD_STREAMS=$hs"
  HN_HEAD_STREAMS=$hs python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 8 2>&1 | tail -1 | cut -c1-140
done
echo "== graph, streams 6"
HN_HEAD_STREAMS=6 python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 8 --graph 2>&1 | tail -1 | cut -c1-140
