# engine-level effect of the merge form and of AUTO's shortest chunk with few requests (variants: tools/build_variants.sh)
for v in ${VARIANTS:-nowide default w64 w32}; do
  lib=build/variants/$v.so; [ $v = default ] && lib=nano_vllm_amd/libnvk_paged_attn.so
  for seqs in 1 8 64; do
    echo -n "$v seqs=$seqs: "
    NVK_LIB_PATH=$lib timeout -k 10 200 python -u tools/bench_model.py --preset tinyllama --seqs $seqs --ctx 1024 --steps 32 | tail -1 || exit 1
  done
done
