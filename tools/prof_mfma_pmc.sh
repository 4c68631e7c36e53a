# PMC passes on the direct-route product of one layer: prof_mfma_pmc.sh "C H W fh fw M same stride"
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/mp
rm -rf $O && mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_]*MFMA[A-Z_]*" | sort -u > $O/mfma_counters.txt
cat $O/mfma_counters.txt
L="$1"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/p1 -o p --output-format csv -- python3 tools/layer_bench.py $L 5 > $O/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_I SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM -d $O/p2 -o p --output-format csv -- python3 tools/layer_bench.py $L 5 > $O/p2.log 2>&1 || exit 1
python3 tools/pmc_summary.py pw_mfma_kernel $O/p1 $O/p2 > $O/mfma_pmc.txt || exit 1
cat $O/mfma_pmc.txt
