#!/bin/bash
# Per-XCD streaming time of the decode kernel, launch by launch (tools/xcd_balance.py on the -DNVK_STAMPS -DNVK_XCD_RANGES_ENV build).
cd "$(dirname "$0")/.."
export NVK_LIB_PATH=build/variants/stamps.so
mkdir -p gpurun_out
run() { echo "== $1"; shift; timeout -k 10 120 env "$@" || exit 1; }
{
run "config2, planned launch (item = workgroup id: a sequence's chunks go round the XCDs)" python tools/xcd_balance.py --workload config2 --launches 6
run "config2, uniform launch: each XCD a contiguous eighth of the items (shipped)" python tools/xcd_balance.py --workload config2 --launches 6 --uniform
run "config2, uniform, another placement of the blocks (seed 7)" python tools/xcd_balance.py --workload config2 --launches 4 --uniform --seed 7
run "config4, uniform, no skew: eighths" NVK_XCD_SKEW=0 python tools/xcd_balance.py --workload config4 --launches 4 --uniform
run "config4, uniform, XCD pairs + 4 % skew (shipped)" python tools/xcd_balance.py --workload config4 --launches 4 --uniform
run "config4, uniform, XCD pairs + 6 % skew" NVK_XCD_SKEW=60 python tools/xcd_balance.py --workload config4 --launches 4 --uniform
run "5b, uniform" python tools/xcd_balance.py --workload 5b --launches 4 --uniform
run "5a, planned" python tools/xcd_balance.py --workload 5a --launches 4
} > gpurun_out/xcd_balance.log 2>&1
