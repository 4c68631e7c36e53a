# one layer through the transform route (0), the direct route (1) and the plan's choice: route_cmp.sh "C H W fh fw M same stride" ...
for L in "$@"; do
  for r in 0 1 m; do
    if [ $r = m ]; then unset MMH_CONV_POINTWISE; else export MMH_CONV_POINTWISE=$r; fi
    echo "route $r: $(timeout -k 10 120 python3 tools/layer_bench.py $L 10 | grep -o 'n_in.*ms/layer')"
  done
done
