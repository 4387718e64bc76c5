#!/bin/bash
# rocprofv3 kernel trace of the bench command; keeps the raw dispatch rows of the last timed steps
# (gpurun_out/trace_window.csv) for tools/step_timeline.py, and the steady-state per-kernel summary.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/prof.log 2>&1 || exit 1
cd $R
f=$(ls $O/prof/*/*kernel_trace.csv | head -1)
python3 profiles/summarize_trace.py $f 10 $O/steady.csv --skip-last 3 > $O/steady.txt || exit 1
python3 tools/step_timeline.py --extract $f $O/trace_window.csv --steps 4 --skip-last 3 || exit 1
python3 tools/step_timeline.py $O/trace_window.csv > $O/step_timeline.txt || exit 1
rm -rf $O/prof
head -5 $O/steady.txt
tail -c 400 $O/prof.log
