#!/bin/bash
# usage (GPU box): tools/run_variants2.sh "workload:geom ..." variant1 variant2 ...   (variant "base" = the product library)
shapes="$1"; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset NVK_LIB_PATH; else export NVK_LIB_PATH=build/variants/$v.so; fi
  echo "== $v"
  tools/sweep_geoms.sh "" $shapes || exit 1
done
