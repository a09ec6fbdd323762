#!/bin/bash
# usage (GPU box): tools/sweep_steps.sh "<bench args>" "<variant args 1>" "<variant args 2>" ...
# step time only (no per-kernel events inside the region), one line per variant
base="$1"; shift
for v in "$@"; do
  out=$(timeout -k 10 150 python -u bench.py --no-cpu --no-kernel-events --steps 50 --warmup 10 $base $v 2>&1 | tail -1)
  python - "$base $v" "$out" <<'PY'
import json,sys
a,out=sys.argv[1:3]
try:
    d=json.loads(out)
    print(f"{a:60s} step {d['value']:7.1f} GB/s {d['value']/8000:.3f} ({d['ms_per_step']*1e3:7.1f} us; events {d['event_ms_per_step']*1e3:7.1f} us)", flush=True)
except Exception as e:
    print(a,"FAILED",out[-300:], flush=True); sys.exit(1)
PY
  [ $? -ne 0 ] && exit 1
done
exit 0
