#!/bin/bash
# Round-3 profiles of the f-rows under the shipped library: kernel durations by rocprofv3 --kernel-trace --stats.
#   tools/profile_round3.sh <what>   what = qkv | qkv2048 | prefill | model
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
what=$1
out=gpurun_out/r03_$what
case $what in
  qkv)     cmd="python tools/bench_linear.py --no-norm --rows 1,2,3,8,16,17,24,32,48,64,96,512,4096 --iters 30" ;;
  qkv2048) cmd="python tools/bench_linear.py --no-norm --dim 2048 --q-dim 2048 --kv-dim 256 --rows 1,2,3,8,16,17,24,32,64 --iters 30" ;;
  wo2048)  cmd="python tools/bench_linear.py --no-norm --dim 2048 --q-dim 2048 --kv-dim 4 --rows 1,2,3,8,16,17,24,32,64 --iters 30" ;;
  prefill) cmd="python tools/bench_prefill.py --seqs 2 --chunk 8192 --iters 5" ;;
  model)   cmd="python tools/bench_model.py --preset tinyllama --seqs 1 --ctx 1024 --steps 16" ;;
  *) echo "unknown $what"; exit 1 ;;
esac
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- $cmd > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
grep -v "^\[" $out.log | tail -20
f=$(ls $out/*/*kernel_stats.csv | head -1)
cp "$f" gpurun_out/r03_${what}_kernel_stats.csv
head -12 "$f" | cut -c1-200
