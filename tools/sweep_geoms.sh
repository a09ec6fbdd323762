#!/bin/bash
# usage (on the GPU box): tools/sweep_geoms.sh "<extra bench args>" workload:geom [workload:geom ...]
# One compact line per (workload, geometry): kernel GB/s (HIP events around the decode kernel), step GB/s.
# NVK_LIB_PATH selects a variant build.  Stops at the first failing run (a faulting kernel must not be retried).
extra="$1"; shift
for wg in "$@"; do
  w="${wg%%:*}"; g="${wg#*:}"
  out=$(timeout -k 10 150 python -u bench.py --no-cpu --steps 32 --warmup 4 --workload "$w" --geom "$g" $extra 2>&1 | tail -1)
  rc=$?
  python - "$w" "$g" "$out" <<'PY'
import json,sys
w,g,out=sys.argv[1:4]
try:
    d=json.loads(out)
    r=d['roofline']
    print(f"{w:8s} {g:10s} kernel {r['achieved']:7.1f} GB/s {r['frac']:.3f} ({r['kernel_ms']*1e3:7.1f} us)  step {d['value']:7.1f} GB/s {d['value']/8000:.3f} ({d['ms_per_step']*1e3:7.1f} us)", flush=True)
except Exception as e:
    print(w,g,"FAILED",out[-300:], flush=True)
    sys.exit(1)
PY
  [ $? -ne 0 ] && exit 1
done
exit 0
