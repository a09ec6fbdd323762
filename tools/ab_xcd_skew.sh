#!/bin/bash
# A/B of the XCD dealing of uniform batches on an -DNVK_XCD_RANGES_ENV build: NVK_XCD_GROUP = XCDs that share a contiguous
# range of the items (1 = an eighth each, 2 = pairs, 4 = fours, 8 = item = workgroup id), NVK_XCD_SKEW = thousandths of a
# chunk handed from the even XCDs' chunks to the odd XCDs' (pairs only).
# A group other than 1 / 2 / 4 / 8 and a skew the library does not know leave the library's own choice (e.g. 9:9).
# usage: tools/ab_xcd_skew.sh "group:skew ..." "workloads" [bench flags]
cd "$(dirname "$0")/.."
export NVK_LIB_PATH=${NVK_LIB_PATH:-build/variants/skewenv.so}
for w in ${2:-config2 config3 config4 5b}; do for gk in ${1:-1:0 2:0 4:0 8:0 2:20 2:40}; do
  g=${gk%%:*}; k=${gk#*:}; off=""; [ "$g" = 8 ] && off=1
  env ${off:+NVK_XCD_RANGES_OFF=1} NVK_XCD_GROUP=$g NVK_XCD_SKEW=$k timeout -k 10 120 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu --no-extra --no-frows $3 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$w group $g skew $k/1000', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['kernel'], d['checked']['ok'], d['checked']['max_abs_err'])" || exit 1
done; done
