set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/lp
rm -rf $O && mkdir -p $O
for L in "64 56 56 1 1 256 0 1" "128 28 28 1 1 512 0 1" "256 14 14 1 1 1024 0 1" "512 7 7 1 1 2048 0 1" "256 14 14 3 3 256 1 1"; do
  tag=$(echo $L | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/t_$tag -o r -- python3 tools/layer_bench.py $L 10 > $O/$tag.log 2>&1 || exit 1
  tail -1 $O/$tag.log
  python3 tools/prof_summary.py $O/t_$tag/r_results.db 25 > $O/$tag.csv || exit 1
  rm -rf $O/t_$tag
done
