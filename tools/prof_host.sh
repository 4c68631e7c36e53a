set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/hp
rm -rf $O && mkdir -p $O
NO_VERDICT_WAIT=1 timeout -k 10 120 rocprofv3 --hip-trace --stats -d $O/t -o r --output-format csv -- python3 tools/layer_bench.py 64 56 56 1 1 64 0 1 50 > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
tail -1 $O/log.txt
ls $O/t
f=$(ls $O/t/*hip_api_stats.csv | head -1)
head -25 $f
