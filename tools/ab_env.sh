#!/bin/bash
# A/B of a host-side switch on ONE box: alternates SPOT_EXP=0 and SPOT_EXP=1 (or the values listed in
# SPOT_EXP_VALUES) under the bench command.
#   tools/ab_env.sh [rounds] [bench args...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
N=${1:-2}; shift
for i in $(seq $N); do
  for v in ${SPOT_EXP_VALUES:-0 1}; do
    SPOT_EXP=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-kernel-timing --exclusive-steps 0 "$@" > gpurun_out/abe_$v.log 2>&1 || { tail -5 gpurun_out/abe_$v.log; exit 1; }
    echo "SPOT_EXP=$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abe_$v.log)"
  done
done
