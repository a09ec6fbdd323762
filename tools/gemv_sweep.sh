for v in c16d2 c16d4 c8d4 c8d8; do
  echo "== $v"
  for shape in "--dim 2048 --q-dim 2048 --kv-dim 256" "--dim 2048 --q-dim 5632 --kv-dim 2816" "--dim 5632 --q-dim 1024 --kv-dim 512" "--dim 4096 --q-dim 4096 --kv-dim 1024" "--dim 2048 --q-dim 16000 --kv-dim 8000"; do
    NVK_LIB_PATH=build/variants/$v.so timeout -k 10 100 python -u tools/bench_linear.py --no-norm --rows 1,2 --iters 200 $shape | awk -v s="$shape" '{print s, $1, $2, $3, $4, $11, $12}' || exit 1
  done
done
