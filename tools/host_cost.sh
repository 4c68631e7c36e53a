for L in "64 56 56 1 1 64 0 1" "256 56 56 1 1 64 0 1" "256 14 14 1 1 1024 0 1"; do
  echo "$L"
  NO_VERDICT_WAIT=1 timeout -k 10 120 python3 tools/layer_bench.py $L 200 2>&1 | grep -o 'n_out.*'
  timeout -k 10 120 python3 tools/layer_bench.py $L 200 2>&1 | grep -o 'n_out.*'
done
