# per-kernel split of single layers: prof_layers.sh "C H W fh fw M same stride" ...
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/lp
rm -rf $O && mkdir -p $O
for L in "$@"; do
  tag=$(echo $L | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/t_$tag -o r -- python3 tools/layer_bench.py $L 10 > $O/$tag.log 2>&1 || exit 1
  grep "ms/layer" $O/$tag.log
  python3 tools/prof_summary.py $O/t_$tag/r_results.db 12 > $O/$tag.csv || exit 1
  cat $O/$tag.csv
  rm -rf $O/t_$tag
done
