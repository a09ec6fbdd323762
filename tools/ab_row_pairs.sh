for w in 5a config2 config3 5b; do for f in "" "--row-pairs"; do
  timeout -k 10 120 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu --no-extra --no-frows --rotate 1 $f 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$w $f', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['kernel'], d['checked']['ok'], d['checked']['max_abs_err'])"
done; done
