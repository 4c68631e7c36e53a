# rocprofv3 --kernel-trace --stats of one layer's server call (tools/layer_bench) under VAR=a and VAR=b: the per-kernel
# averages side by side.   usage: layer_ab_trace.sh VAR a b "C H W fh fw M same stride"
set -o pipefail
export TMPDIR=/tmp
VAR="$1"; A="$2"; B="$3"; L="$4"
O=gpurun_out/lab
rm -rf $O && mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmm_amd:$LD_LIBRARY_PATH
for V in $A $B; do
  export $VAR=$V
  timeout -k 10 60 tools/layer_bench $L 50 20 > $O/plain_$V.log 2>&1 || { cat $O/plain_$V.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/t_$V -o r -- tools/layer_bench $L 50 20 > $O/trace_$V.log 2>&1 || { tail -5 $O/trace_$V.log; exit 1; }
  python3 tools/prof_summary.py $O/t_$V/r_results.db 10 > $O/kernels_$V.csv || exit 1
  rm -rf $O/t_$V
  echo "== $VAR=$V: $(cat $O/plain_$V.log)"; cat $O/kernels_$V.csv
done
