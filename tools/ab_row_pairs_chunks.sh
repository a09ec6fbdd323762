for c in 48 64 96 128; do for f in "" "--row-pairs"; do
  timeout -k 10 120 python bench.py --workload 5a --mode split --chunk $c --steps 40 --warmup 5 --no-cpu --no-extra --no-frows --rotate 1 $f 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('5a chunk $c $f', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['kernel'], d['checked']['ok'])"
done; done
