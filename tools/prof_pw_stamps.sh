# In-kernel clock stamps of the product kernel for one layer (tools/layer_bench with MMH_PW_STAMPS set; the stamped
# instantiation runs in place of the plain one) -> gpurun_out/stamps_<tag>.txt (summary) ; usage: prof_pw_stamps.sh TAG "C H W fh fw M same stride"
set -o pipefail
TAG="$1"; L="$2"
O=gpurun_out
mkdir -p $O
export LD_LIBRARY_PATH=$PWD/openmm_amd:$LD_LIBRARY_PATH
rm -f $O/stamps_raw_$TAG.txt
MMH_PW_STAMPS=$O/stamps_raw_$TAG.txt timeout -k 10 60 tools/layer_bench $L 3 3 > $O/stamps_run_$TAG.log 2>&1 || { cat $O/stamps_run_$TAG.log; exit 1; }
{ echo "# layer: $L"; python3 tools/pw_stamps_summary.py $O/stamps_raw_$TAG.txt; } > $O/stamps_$TAG.txt
cat $O/stamps_$TAG.txt
