#!/bin/bash
# Round evidence on the GPU box: bench line, kernel-trace summary of the same command, PMC passes on
# the NTT micro-bench.  Everything lands under gpurun_out/ev/.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/ev
rm -rf $O && mkdir -p $O
timeout -k 10 400 python3 -u bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 bench.py --no-cpu > $O/trace.log 2>&1 || exit 1
python3 tools/prof_summary.py $O/trace/bench_results.db 40 > $O/kernel_stats.csv || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_h -o bench -- python3 bench.py --headline-only > $O/trace_h.log 2>&1 || exit 1
python3 tools/prof_summary.py $O/trace_h/bench_results.db 10 > $O/kernel_stats_headline.csv || exit 1
grep '"metric"' $O/trace_h.log > $O/bench_headline.json
cd tools && export LD_LIBRARY_PATH=../openmm_amd
B="./ntt_bench 12 4096 5"
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE -d ../$O/pmc_fetch -o p --output-format csv -- $B > ../$O/pmc.log 2>&1 || exit 1
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE -d ../$O/pmc_write -o p --output-format csv -- $B >> ../$O/pmc.log 2>&1 || exit 1
timeout -s KILL 60 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d ../$O/pmc_sq1 -o p --output-format csv -- $B >> ../$O/pmc.log 2>&1 || exit 1
timeout -s KILL 60 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_VMEM -d ../$O/pmc_sq2 -o p --output-format csv -- $B >> ../$O/pmc.log 2>&1 || exit 1
cd .. && python3 tools/pmc_summary.py ntt_fwd_kernel $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2 --json $O/ntt_pmc.json > $O/ntt_pmc.txt || exit 1
./tools/ntt_bench 12 4096 20 >> $O/ntt_pmc.txt
rm -rf $O/trace $O/trace_h $O/pmc_*   # keep the summaries only
cat $O/kernel_stats_headline.csv $O/bench_headline.json; cat $O/ntt_pmc.txt
