# per-shape sweep of the GEMV geometry (columns per workgroup x ring depth); variants built by tools/build_variants.sh
for v in ${VARIANTS:-c16d2 c8d2 c8d4 c4d4 c4d8}; do
  echo "== $v"
  for shape in "--dim 2048 --q-dim 2048 --kv-dim 256" "--dim 2048 --q-dim 1024 --kv-dim 512" "--dim 2048 --q-dim 5632 --kv-dim 2816" "--dim 5632 --q-dim 1024 --kv-dim 512" "--dim 4096 --q-dim 4096 --kv-dim 1024" "--dim 2048 --q-dim 16000 --kv-dim 8000" "--dim 288 --q-dim 288 --kv-dim 288" "--dim 768 --q-dim 768 --kv-dim 768"; do
    NVK_LIB_PATH=build/variants/$v.so timeout -k 10 100 python -u tools/bench_linear.py --no-norm --rows 1,2 --iters 200 $shape | awk -v s="$shape" '{print s, $1, $2, $3, $4, $11, $12}' || exit 1
  done
done
