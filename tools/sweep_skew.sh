#!/bin/bash
# usage (GPU box): tools/sweep_skew.sh "<skews in bytes>" "<workload:geom ...>" [rotate]
export NVK_LIB_PATH=build/variants/skew.so
for sk in $1; do
  export NVK_POOL_SKEW=$sk
  echo "== skew $sk B, rotate ${3:-1}"
  tools/sweep_geoms.sh "--no-extra --rotate ${3:-1}" $2 || exit 1
done
