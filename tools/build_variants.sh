#!/bin/bash
# Build tuning variants of libnvk_paged_attn.so into build/variants/<name>.so
# usage: tools/build_variants.sh name1:"-DNVK_X=1 -DNVK_Y=2" name2:"..."
# Select one at run time with NVK_LIB_PATH=build/variants/<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
CSRC=nano_vllm_amd/csrc
pids=()
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -I$CSRC -fno-gpu-rdc $flags \
      -o build/variants/$name.so $CSRC/nvk_paged_attn.hip $CSRC/nvk_batch.hip $CSRC/nvk_linear.hip $CSRC/nvk_model.hip $CSRC/nvk_shard.hip $CSRC/nvk_block_manager.cpp $CSRC/nvk_tokenizer.cpp $CSRC/nvk_model_group.cpp -ldl -lpthread && echo "built $name [$flags]" ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
