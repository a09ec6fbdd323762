#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, CSV output) into the
per-kernel HBM-bytes summary bench.py reads (profiles/rNN_*_pmc_traffic.json).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <workload> <round> > out.json

Units and corrections as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes: both counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced streaming reads, so it is doubled;
WRITE_SIZE is taken as it is."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch_csv, write_csv, workload, rnd = sys.argv[1:5]
    fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        if not name.startswith(("nvk::", "void nvk::")):
            continue
        f, w = fetch.get(name, []), write.get(name, [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        key = name.replace("void ", "").split("(")[0]
        kernels[key] = {"FETCH_SIZE_KiB_mean": fm, "launches_FETCH_SIZE": len(f), "WRITE_SIZE_KiB_mean": wm,
                        "launches_WRITE_SIZE": len(w), "hbm_bytes_per_launch_corrected": int(round((2 * fm + wm) * 1024))}
    print(json.dumps({
        "round": int(rnd),
        "command": "tools/profile_workload.sh: rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-extra --no-frows --rotate 1 --prewarm-ms 0 <workload args> ; "
                   "same with --pmc WRITE_SIZE (one counter per pass)",
        "workload": workload,
        "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB; gfx950 correction (MI355X_MICROARCH.md, HBM): "
                 "FETCH_SIZE x2 for 16 B/lane streaming reads, WRITE_SIZE exact",
        "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
