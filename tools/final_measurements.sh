#!/bin/bash
# The measurement set behind the numbers in the documentation, in two stages that each fit one GPU call:
#   stage1: bench line, all-kernel timing, rocprofv3 kernel trace of the same command (steady-state summary +
#           whole-process stats), the two counter passes for HBM traffic
#   stage2: kernel micro-benchmarks, the other configurations, the Siamese variant
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
if [ "$1" = "stage1" ]; then
timeout -k 10 400 python3 bench.py > $O/bench.log 2>&1 || exit 1
tail -c 600 $O/bench.log
timeout -k 10 400 python3 bench.py --no-cpu-baseline --time-kernels all > $O/bench_all_kernels.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline > $O/prof.log 2>&1 || exit 1
cd $R
f=$(ls $O/prof/*/*kernel_trace.csv | head -1)
python3 profiles/summarize_trace.py $f 10 $O/steady.csv --skip-last 3 > $O/steady.txt || exit 1
head -4 $O/steady.txt
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/whole_process_kernel_stats.csv
rm -rf $O/prof
bash tools/collect_traffic.sh > $O/traffic.txt 2>&1 || exit 1
cp gpurun_out/traffic.json $O/traffic.json
echo stage1 done
fi
if [ "$1" = "stage2" ]; then
rm -f $O/microbench.txt
for mode in "0 0" "0 0.1" "1 0.1"; do set -- $mode; SPOT_MB_BF16=$1 SPOT_MB_DROP=$2 timeout -k 10 100 python3 tools/microbench.py mha >> $O/microbench.txt 2>&1 || exit 1; done
timeout -k 10 200 python3 tools/microbench.py gemm >> $O/microbench.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/microbench.py gemm_tiles >> $O/microbench.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/microbench.py gemm_bf16x3 >> $O/microbench.txt 2>&1 || exit 1
timeout -k 10 100 python3 tools/microbench.py xcorr >> $O/microbench.txt 2>&1 || exit 1
timeout -k 10 100 python3 tools/microbench.py norm >> $O/microbench.txt 2>&1 || exit 1
echo microbench done
timeout -k 10 400 python3 tools/measure_configs.py > $O/configs.log 2>&1 || exit 1
timeout -k 10 300 python3 bench.py --attention bf16 --clips 48 --sup 16 --no-cpu-baseline > $O/bench_cfg4.log 2>&1 || exit 1
grep "^{" $O/configs.log | cut -c1-200
# the Siamese variant: head at both search sizes, then tracker rate and one training iteration
timeout -k 10 300 python3 tools/ocean_bench.py --batch 32 --search 255 > $O/ocean.log 2>&1 || exit 1
timeout -k 10 400 python3 tools/ocean_bench.py --batch 128 --search 384 --steps 3 >> $O/ocean.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/ocean_bench.py variant 8 >> $O/ocean.log 2>&1 || exit 1
grep "^{" $O/ocean.log | cut -c1-160
echo stage2 done
fi
