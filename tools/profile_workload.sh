#!/bin/bash
# usage (GPU box): tools/profile_workload.sh <tag> "<bench args, e.g. --workload config3 --geom llama>"
# Three rocprofv3 runs of the same bench command (one kind of collection per run, as the pool requires):
#   1. --kernel-trace --stats  -> gpurun_out/prof_r02/<tag>/stats/*_kernel_stats.csv
#   2. --pmc FETCH_SIZE        -> .../fetch/*_counter_collection.csv
#   3. --pmc WRITE_SIZE        -> .../write/*_counter_collection.csv
# then tools/pmc_traffic.py -> .../<tag>_pmc_traffic.json (gfx950 corrections of MI355X_MICROARCH.md).
# Copy what is to be judged into profiles/ afterwards.
set -u
tag="$1"; args="$2"
out=gpurun_out/prof_r02/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
common="--no-cpu --no-extra --no-frows --rotate 1 --prewarm-ms 0 $args"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o stats -- python3 bench.py --steps 20 --warmup 5 $common > "$out/stats.log" 2>&1 || { echo "stats pass failed"; tail -5 "$out/stats.log"; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o fetch -- python3 bench.py --steps 5 --warmup 2 $common > "$out/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$out/fetch.log"; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o write -- python3 bench.py --steps 5 --warmup 2 $common > "$out/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$out/write.log"; exit 1; }
name=$(grep -o '"workload": "[^"]*"' "$out/stats.log" | head -1 | sed 's/.*: "//; s/"//')
f=$(find "$out/fetch" -name '*counter_collection.csv' | head -1)
w=$(find "$out/write" -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py "$f" "$w" "$name" 3 > "$out/${tag}_pmc_traffic.json"
k=$(find "$out/stats" -name '*kernel_stats.csv' | head -1)
cp "$k" "$out/${tag}_kernel_stats.csv"
grep '^{' "$out/stats.log" | tail -1 > "$out/${tag}_bench_under_rocprof.json"
echo "== $tag ($name)"
head -4 "$out/${tag}_kernel_stats.csv"
python3 - "$out/${tag}_pmc_traffic.json" "$out/${tag}_bench_under_rocprof.json" <<'PY'
import json,sys
p=json.load(open(sys.argv[1])); b=json.loads(open(sys.argv[2]).read())
alg=b["roofline"]["algorithmic_bytes"]
for k,v in p["kernels"].items():
    if "decode" in k:
        print(f"   {k}: HBM bytes/launch {v['hbm_bytes_per_launch_corrected']}  = {v['hbm_bytes_per_launch_corrected']/alg:.4f} x algorithmic ({alg})")
PY
