"""Mean of every counter per kernel from rocprofv3 --pmc CSV output:  python tools/pmc_summary.py <dir> [name filter]"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for (k, c), v in sorted(acc.items()):
    if flt in k:
        print(f"{k:42s} {c:28s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
