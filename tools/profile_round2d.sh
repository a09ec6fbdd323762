#!/bin/bash
# usage (GPU box): tools/profile_round2d.sh -- every profiled shape under the final round-2 library (lockstep kernels,
# token-part merge over all threads): the five BASELINE configs and the four other head geometries at two sizes
set -o pipefail
for spec in "config2:--workload config2" "config3:--workload config3" "config4:--workload config4" \
            "config5a:--workload 5a" "config5b:--workload 5b" \
            "5b_gqa8:--workload 5b --geom gqa8" "config2_gqa8:--workload config2 --geom gqa8" \
            "5b_tinyllama:--workload 5b --geom tinyllama" "config2_tinyllama:--workload config2 --geom tinyllama" \
            "5b_shard8:--workload 5b --geom shard8" "config2_shard8:--workload config2 --geom shard8" \
            "5b_tiny:--workload 5b --geom tiny" "config2_tiny:--workload config2 --geom tiny"; do
  tag="${spec%%:*}"; args="${spec#*:}"
  tools/profile_workload.sh "$tag" "$args" || exit 1
done
