#!/bin/bash
# usage (on the GPU box): tools/ab_pairs.sh  -- head_dim 48 (G-tiny): a pair of kv-heads per wave vs one
set -o pipefail
echo "== pairs (default)" && tools/sweep_geoms.sh "--no-extra" 5b:tiny config2:tiny config3:tiny 5a:tiny &&
echo "== one kv-head per wave (--no-pairs)" && tools/sweep_geoms.sh "--no-extra --no-pairs" 5b:tiny config2:tiny config3:tiny 5a:tiny
