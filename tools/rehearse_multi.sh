#!/bin/bash
# Rehearsal of the multi-process launch paths on a one-GPU box (ranks share the GPU; gloo carries the barrier, the
# timing reduction and -- for the head-sharded variant -- the collective on host copies).  The driver's real runs use
# --backend nccl (RCCL) on an 8-GPU node; this only proves that every launch path starts, runs and prints its line.
#   tools/rehearse_multi.sh <ranks> > log
n=${1:-8}
run() {
  echo "## python -m torch.distributed.run --nnodes=1 --nproc-per-node $n bench.py --gpus $n --backend gloo $*"
  timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) \
      bench.py --gpus $n --steps 10 --warmup 3 --backend gloo --prewarm-ms 20 "$@" 2>&1 | grep '^{' || { echo "FAILED: $*"; return 1; }
}
run || exit 1
run --strong --workload config3 || exit 1
run --workload 5a || exit 1
run --shard heads || exit 1
echo "## python bench.py --shard heads (one rank, RCCL through nvk_shard_*, gather overlapped)"
timeout -k 10 120 python bench.py --shard heads --steps 10 --warmup 3 --no-cpu 2>&1 | grep '^{' || exit 1
