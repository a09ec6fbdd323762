#!/bin/bash
# usage (on the GPU box): tools/run_prefill_variants.sh "<bench_prefill args>" variant1 variant2 ...
args="$1"; shift
for v in "$@"; do
  echo -n "$v: "
  NVK_LIB_PATH=build/variants/$v.so timeout -k 10 120 python -u tools/bench_prefill.py $args 2>&1 | tail -1
done
