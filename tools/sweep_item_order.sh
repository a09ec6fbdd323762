#!/bin/bash
# usage (GPU box): tools/sweep_item_order.sh "<skews>" "<workload:geom ...>" [rotate]
# uniform batches: each XCD a contiguous eighth of the items (default) vs item = workgroup id, at several K/V slab distances
export NVK_LIB_PATH=build/variants/skew.so
for sk in $1; do
  export NVK_POOL_SKEW=$sk
  unset NVK_XCD_RANGES_OFF; echo "== xcd ranges, skew $sk B, pool layers ${3:-1}"
  tools/sweep_geoms.sh "--no-extra --rotate ${3:-1}" $2 || exit 1
  export NVK_XCD_RANGES_OFF=1; echo "== item = workgroup id, skew $sk B, pool layers ${3:-1}"
  tools/sweep_geoms.sh "--no-extra --rotate ${3:-1}" $2 || exit 1
done
