#!/bin/bash
# rocprofv3 kernel trace of the bench command, reduced to the timed steps (gpurun_out/steady.{txt,csv})
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/prof.log 2>&1 || exit 1
cd $R
f=$(ls $O/prof/*/*kernel_trace.csv | head -1)
python3 profiles/summarize_trace.py $f 10 $O/steady.csv --skip-last 3 > $O/steady.txt || exit 1
rm -rf $O/prof
head -45 $O/steady.txt
