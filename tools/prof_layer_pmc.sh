# PMC passes over the kernels of one convolution layer's server call, driven by the stand-alone tools/layer_bench
# (placed directly after `--`).  Separate rocprofv3 runs per counter group (MI355X_MICROARCH.md; at most 8 SQ_, 4 TCC_
# per run); output: gpurun_out/lp_<tag>/pmc.txt with per-kernel averages.
#   usage: prof_layer_pmc.sh TAG "C H W fh fw M same stride"
set -o pipefail
export TMPDIR=/tmp
TAG="$1"; L="$2"
O=gpurun_out/lp_$TAG
rm -rf $O && mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmm_amd:$LD_LIBRARY_PATH
timeout -k 10 60 tools/layer_bench $L 20 5 > $O/plain.log 2>&1 || { cat $O/plain.log; exit 1; }
cat $O/plain.log
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/t -o r -- tools/layer_bench $L 20 5 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python3 tools/prof_summary.py $O/t/r_results.db 12 > $O/kernels.csv || exit 1
rm -rf $O/t
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_I SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $G -d $O/p$i -o p --output-format csv -- tools/layer_bench $L 5 2 > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; exit 1; }
done
{
  echo "# layer: $L"; cat $O/plain.log; echo "# kernel times (rocprofv3 --kernel-trace --stats)"; cat $O/kernels.csv
  for K in pw_mfma_kernel pw_narrow_kernel ntt_inv_kernel pw_expand_kernel conv_front_kernel reshare_elem_kernel; do
    echo "## $K"; python3 tools/pmc_summary.py $K $O/p1 $O/p2 $O/p3 $O/p4
  done
} > $O/pmc.txt
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
cat $O/pmc.txt
