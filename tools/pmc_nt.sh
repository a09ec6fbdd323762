#!/bin/bash
# SQ counters of the tiled projection kernel, one shape per pass (run on the GPU box from the repo root):
#   tools/pmc_nt.sh > gpurun_out/pmc_nt_sq.log
# MFMA pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel time x clock); the wave's time splits into
# SQ_WAIT_ANY (parked: s_waitcnt, barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (quad-cycles).
R=$PWD
cd /tmp && export TMPDIR=/tmp
for spec in "qkv:4096:6144 512" "qkv:4096:6144 4096" "gateup:2048:5632x2 64" "gateup:2048:5632x2 512"; do
    set -- $spec
    rm -rf $R/gpurun_out/pmc_nt_pass
    timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
        --output-format csv -d $R/gpurun_out/pmc_nt_pass -o nt -- python3 $R/tools/bench_matmul.py --shapes $1 --rows $2 --iters 5 > $R/gpurun_out/pmc_nt_pass.log 2>&1 || { echo "pass failed: $spec"; tail -3 $R/gpurun_out/pmc_nt_pass.log; exit 1; }
    echo "== $1 at $2 rows (profiled pass): $(tail -1 $R/gpurun_out/pmc_nt_pass.log)"
    python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_nt_pass gemm_nt
    timeout -k 10 120 python3 $R/tools/bench_matmul.py --shapes $1 --rows $2 --iters 50 | sed 's/^/   unprofiled: /' || exit 1
done
