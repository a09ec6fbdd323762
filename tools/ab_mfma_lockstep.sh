#!/bin/bash
# A/B of the lockstep barrier in nvk_decode_mfma_kernel: build/variants/mfmafree.so (-DNVK_MFMA_LOCKSTEP=0) against the shipped library.
cd "$(dirname "$0")/.."
for spec in "config2 gqa8" "config2 tinyllama" "config2 gqa8x8" "config4 gqa8" "config4 tinyllama" "config4 gqa8x8" "5b gqa8" "5b tinyllama" "5a gqa8" "config3 gqa8"; do set -- $spec
for lib in build/variants/mfmafree.so nano_vllm_amd/libnvk_paged_attn.so; do
  NVK_LIB_PATH=$lib timeout -k 10 120 python bench.py --workload $1 --geom $2 --steps 40 --warmup 5 --no-cpu --no-extra --no-frows 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$1 $2', '$lib'.split('/')[-1][:12], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['kernel'], d['checked']['ok'], d['checked']['max_abs_err'])" || exit 1
done; done
