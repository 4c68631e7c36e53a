#!/bin/bash
# A/B of libmm_hip variants on the NTT micro-bench: ntt_ab.sh variant...
cd "$(dirname "$0")/.."
for v in "$@"; do
  export LD_LIBRARY_PATH=exp/$v
  for args in "12 16384 20 0" "12 16384 20 0 0" "12 16384 20 0 1" "12 16384 20 1" "13 2048 20 0"; do
    echo -n "$v [$args] "; timeout -k 10 60 ./ntt_bench $args || exit 1
  done
done
