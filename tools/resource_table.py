#!/usr/bin/env python3
"""Condense hipcc's -Rpass-analysis=kernel-resource-usage remarks (make asm -> build/resource_usage.txt) into one
line per kernel:  python tools/resource_table.py build/resource_usage.txt [name filter]"""
import re
import subprocess
import sys

rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass-analysis", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
except OSError:
    dem = names
print(f"{'kernel':58s} {'VGPR':>4s} {'AGPR':>4s} {'SGPR':>4s} {'spillV':>6s} {'scratch B/lane':>14s} {'LDS B':>6s} {'waves/SIMD':>10s}")
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("void ", "")
    if flt in d:
        print(f"{d:58s} {r.get('VGPRs','?'):>4s} {r.get('AGPRs','?'):>4s} {r.get('TotalSGPRs','?'):>4s} {r.get('VGPRs Spill','?'):>6s} "
              f"{r.get('ScratchSize [bytes/lane]','?'):>14s} {r.get('LDS Size [bytes/block]','?'):>6s} {r.get('Occupancy [waves/SIMD]','?'):>10s}")
