# Kernel-trace stats and two PMC passes of the COT post-processing kernels (tools/cot_bench.py)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/cot
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/trace -o cot --output-format csv -- python3 tools/cot_bench.py > $O/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/p1 -o p --output-format csv -- python3 tools/cot_bench.py > $O/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $O/p2 -o p --output-format csv -- python3 tools/cot_bench.py > $O/p2.log 2>&1 || exit 1
for k in "cot_send_kernel<true>" "cot_send_kernel<false>" "cot_recv_kernel<true>" "cot_recv_kernel<false>"; do
  echo "== $k"; python3 tools/pmc_summary.py "$k" $O/p1 $O/p2 || exit 1
done > $O/cot_pmc.txt
cat $O/cot_pmc.txt
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
grep cot_ $O/kernel_stats.csv
