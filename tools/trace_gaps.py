"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (on one stream the device runs them in order):
    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [skip_first_n]
Prints per kernel name: calls, mean duration, mean idle time before it (start - previous end)."""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
dur, gap, n = defaultdict(float), defaultdict(float), defaultdict(int)
for (s0, e0, _), (s1, e1, name) in zip(rows, rows[1:]):
    dur[name] += e1 - s1
    gap[name] += max(0, s1 - e0)
    n[name] += 1
for name in sorted(n, key=lambda k: -dur[k]):
    print(f"{name:50s} calls {n[name]:5d}  mean {dur[name] / n[name] / 1e3:9.2f} us  idle before {gap[name] / n[name] / 1e3:7.2f} us")
