#!/bin/bash
# A/B of the split-K tiled kernel (nvk_gemm_nt_kernel<64, 64, 64, kNtSplit>) against the kernels it replaces, on an
# -DNVK_LINEAR_ENV build: NVK_NT_SPLIT_MIN_ROWS=100000 switches it off.
cd "$(dirname "$0")/.."
export NVK_LIB_PATH=build/variants/linenv.so
rows=${1:-17,24,32,33,48,64,96,97,128,192,256,384,512}
echo "== split-K off"; NVK_NT_SPLIT_MIN_ROWS=100000 timeout -k 10 300 python tools/bench_matmul.py --rows $rows || exit 1
echo "== split-K from 17 rows"; NVK_NT_SPLIT_MIN_ROWS=17 timeout -k 10 300 python tools/bench_matmul.py --rows $rows || exit 1
