#!/bin/bash
# usage (on the GPU box): tools/run_variants.sh "<bench args>" variant1 variant2 ...
# prints one compact line per variant: kernel GB/s, step GB/s
args="$1"; shift
for v in "$@"; do
  out=$(NVK_LIB_PATH=build/variants/$v.so timeout -k 10 120 python -u bench.py --no-cpu $args 2>/dev/null | tail -1)
  python - "$v" "$args" "$out" <<'PY'
import json,sys
v,args,out=sys.argv[1:4]
try:
    d=json.loads(out)
    print(f"{v:14s} {args:34s} kernel {d['roofline']['achieved']:7.1f} GB/s ({d['roofline']['kernel_ms']*1e3:6.1f} us)  step {d['value']:7.1f} GB/s ({d['ms_per_step']*1e3:6.1f} us)")
except Exception as e:
    print(v,args,"FAILED",out[:200])
PY
done
