#!/bin/bash
# Rehearsal of bench.py's N>1 extras (configs[4]: 5a / 5b sequence-sharded, 5b head-sharded) and of their watchdog on
# a one-GPU box: ranks share the GPU, gloo carries barriers / reductions / the head-sharded collective.
#   tools/rehearse_extras.sh <ranks> > log
n=${1:-2}
launch() {
  echo "## torch.distributed.run --nproc-per-node $n bench.py --gpus $n --backend gloo $*"
  timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) \
      bench.py --gpus $n --steps 10 --warmup 3 --backend gloo --prewarm-ms 20 "$@" 2>&1 | grep '^{' || { echo "FAILED: $*"; return 1; }
}
launch || exit 1
echo "## the same with a 0.05 s budget for the extras: the watchdog prints the headline alone"
launch --extra-timeout 0.05 || exit 1
