#!/bin/bash
# pipeline A/B of the GroupNorm-apply pass: generic kernel (HN_SPLIT_GENERIC=1) against the channel-group-stationary one
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in generic rule; do
  [ $v = generic ] && export HN_SPLIT_GENERIC=1
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v pipeline b32', d['value'], d['ms_per_step'])"
done; done
