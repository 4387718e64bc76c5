#!/bin/bash
# Two counter passes over bench.py (FETCH_SIZE and WRITE_SIZE do not fit one pass), then the per-kernel summary.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 500 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
f=$(ls gpurun_out/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1)
w=$(ls gpurun_out/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1)
python3 tools/traffic_from_pmc.py $f $w gpurun_out/traffic.json
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
