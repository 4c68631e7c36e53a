# rocprofv3 --kernel-trace --stats over one network's in-order linear calls (tools/resnet_prof.py): the per-kernel
# share of the leg -> gpurun_out/rp/kernels_<net>.csv and gpurun_out/rp/kernel_share.json (copied to profiles/ by hand)
#   usage: prof_resnet.sh [resnet50 [densenet121 ...]]
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/rp
rm -rf $O && mkdir -p $O
for NET in ${@:-resnet50}; do
  timeout -k 10 300 python3 tools/resnet_prof.py $NET 5 > $O/plain_$NET.log 2>&1 || { tail -5 $O/plain_$NET.log; exit 1; }
  cat $O/plain_$NET.log
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/t -o r -- python3 tools/resnet_prof.py $NET 5 > $O/prof_$NET.log 2>&1 || { tail -5 $O/prof_$NET.log; exit 1; }
  python3 tools/prof_summary.py $O/t/r_results.db 40 > $O/kernels_$NET.csv || exit 1
  cat $O/kernels_$NET.csv
  rm -rf $O/t
done
python3 - "$O" ${@:-resnet50} <<'PY'
import csv, json, sys
out = {}
for net in sys.argv[2:]:
    rows = list(csv.DictReader(open(f"{sys.argv[1]}/kernels_{net}.csv")))
    out[net] = {r["kernel"]: round(float(r["percent"]), 2) for r in rows[:14]}
json.dump(out, open(f"{sys.argv[1]}/kernel_share.json", "w"), indent=1)
PY
