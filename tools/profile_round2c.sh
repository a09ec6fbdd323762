#!/bin/bash
# usage (GPU box): tools/profile_round2c.sh -- the five BASELINE configs under the final round-2 library
set -o pipefail
for spec in "config2:--workload config2" "config3:--workload config3" "config4:--workload config4" \
            "config5a:--workload 5a" "config5b:--workload 5b"; do
  tag="${spec%%:*}"; args="${spec#*:}"
  tools/profile_workload.sh "$tag" "$args" || exit 1
done
