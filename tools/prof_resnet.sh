set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/rp
rm -rf $O && mkdir -p $O
timeout -k 10 300 python3 tools/resnet_prof.py > $O/plain.log 2>&1 || { tail -5 $O/plain.log; exit 1; }
cat $O/plain.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/t -o r -- python3 tools/resnet_prof.py > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
python3 tools/prof_summary.py $O/t/r_results.db 30 > $O/kernels.csv || exit 1
cat $O/kernels.csv
rm -rf $O/t
