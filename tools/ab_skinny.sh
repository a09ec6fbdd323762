#!/bin/bash
# A/B of the 16 x 16 tile weight-streaming kernel: tools/ab_skinny.sh <lib.so or ""> ...
for v in "$@"; do
  echo "##### ${v:-shipped}"
  for d in "3072 512" "4096 1024" "6144 1024"; do set -- $d
    NVK_LIB_PATH=$v timeout -k 10 60 python tools/bench_linear.py --no-norm --q-dim $1 --kv-dim $2 --rows 16 --iters 50 | sed "s/^/cols $(( $1 + 2 * $2 )) /" | cut -c1-45,105-160
  done
done
