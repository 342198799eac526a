#!/bin/bash
# sample sclk / power while a kernel loops (is the chip power- or clock-limited under this load?)
cd $GRAFT_REPO_ROOT
run() {
  "$@" > /tmp/run.log 2>&1 &
  pid=$!
  sleep 2.5
  for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' '
    echo
    sleep 0.4
  done
  wait $pid
  grep -v amdgpu.ids /tmp/run.log | tail -2
}
echo "== conv 128x128 tower layer"; run python tools/perf_conv.py f16x3 1 32 100 136 256 256 3 1 1 4000 0 0
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
