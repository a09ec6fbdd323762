#!/bin/bash
# Round-3 profiles under the shipped library (GPU box): per workload three rocprofv3 runs (kernel stats, FETCH_SIZE,
# WRITE_SIZE; tools/profile_workload.sh), then the f-rows' kernel stats (tools/profile_round3.sh).
#   tools/profile_round3_all.sh     -> gpurun_out/prof_r02/<tag>/... (the script's directory name is historical)
set -u
run() { tools/profile_workload.sh "$1" "$2" > gpurun_out/prof_$1.log 2>&1 || { echo "FAILED $1"; tail -5 gpurun_out/prof_$1.log; return 1; }; tail -6 gpurun_out/prof_$1.log; }
run config2 "--workload config2" || exit 1
run config3 "--workload config3" || exit 1
run config4 "--workload config4" || exit 1
run config5a "--workload 5a" || exit 1
run config5b "--workload 5b" || exit 1
run config2_shard8 "--workload config2 --geom shard8" || exit 1
run config2_tinyllama "--workload config2 --geom tinyllama" || exit 1
run config2_gqa8 "--workload config2 --geom gqa8" || exit 1
