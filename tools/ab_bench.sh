#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by a few per cent): alternates spot_amd/libspot_hip.so.A
# and spot_amd/libspot_hip.so.B under the bench command and prints ms per step of every run.
#   tools/ab_bench.sh [rounds] [bench args...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
N=${1:-2}; shift
for i in $(seq $N); do
  for v in A B; do
    cp spot_amd/libspot_hip.so.$v spot_amd/libspot_hip.so || exit 1
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-kernel-timing --exclusive-steps 0 "$@" > gpurun_out/ab_$v.log 2>&1 || { tail -5 gpurun_out/ab_$v.log; exit 1; }
    echo "$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$v.log)"
  done
done
