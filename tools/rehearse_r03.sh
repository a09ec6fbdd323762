#!/bin/bash
# Round-3 rehearsal of `python bench.py --gpus N` typed directly (no torch.distributed.run in the command): bench.py
# starts its own ranks, which share the one GPU over gloo; the N > 1 extras run in a fresh child per rank.
#   tools/rehearse_r03.sh > log     (4 ranks with extras = 8 GPU processes; 8 ranks without = 8)
set -o pipefail
echo "## python bench.py --gpus 4 --backend gloo (extras in fresh children)"
timeout -k 10 400 python bench.py --gpus 4 --steps 10 --warmup 3 --backend gloo --prewarm-ms 20 2>gpurun_out/rehearse4.err | grep '^{' || { echo FAILED; tail -30 gpurun_out/rehearse4.err; exit 1; }
echo "## the same with a 0.3 s budget for the extras: children ended, extras_failed true, status 0"
timeout -k 10 200 python bench.py --gpus 2 --steps 10 --warmup 3 --backend gloo --prewarm-ms 20 --extra-timeout 0.3 2>gpurun_out/rehearse2.err | grep '^{' || { echo FAILED; tail -30 gpurun_out/rehearse2.err; exit 1; }
echo "## python bench.py --gpus 8 --backend gloo --no-extra"
timeout -k 10 300 python bench.py --gpus 8 --steps 10 --warmup 3 --backend gloo --prewarm-ms 20 --no-extra 2>gpurun_out/rehearse8.err | grep '^{' || { echo FAILED; tail -30 gpurun_out/rehearse8.err; exit 1; }
