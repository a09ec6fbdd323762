#!/bin/bash
# usage (GPU box): tools/profile_round2b.sh -- re-profile the shapes whose kernels changed late in round 2
set -o pipefail
for spec in "5b_tiny:--workload 5b --geom tiny" "config2_tiny:--workload config2 --geom tiny" \
            "5b_gqa8:--workload 5b --geom gqa8" "config2_gqa8:--workload config2 --geom gqa8" \
            "5b_tinyllama:--workload 5b --geom tinyllama" "config2_tinyllama:--workload config2 --geom tinyllama" \
            "config2:--workload config2" "config5b:--workload 5b"; do
  tag="${spec%%:*}"; args="${spec#*:}"
  tools/profile_workload.sh "$tag" "$args" || exit 1
done
