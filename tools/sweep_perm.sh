#!/bin/bash
# usage (GPU box): tools/sweep_perm.sh "<perms>" "<skews>" "<workload:geom ...>"
export NVK_LIB_PATH=build/variants/skew.so
for sk in $2; do for pm in $1; do
  export NVK_POOL_SKEW=$sk NVK_ITEM_PERM=$pm
  echo "== perm $pm skew $sk B"
  tools/sweep_geoms.sh "--no-extra --rotate 1" $3 || exit 1
done; done
