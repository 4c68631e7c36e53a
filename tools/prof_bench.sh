set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/bp
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 bench.py --no-cpu > $O/trace.log 2>&1 || exit 1
python3 tools/prof_summary.py $O/trace/bench_results.db 30 > $O/kernel_stats.csv || exit 1
rm -rf $O/trace
cat $O/kernel_stats.csv
