"""Paged chunked-prefill attention timing on one MI355X (not the headline bench).

    python tools/bench_prefill.py [--seqs 8] [--start 0] [--chunk 2048] [--mode tiled|rowwise] [--iters 10]
Reports algorithmic TFLOP/s (4 * keys_seen * n_heads * head_dim per query token: QK^T + PV) against the
fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md) and the HBM bytes a one-pass read of K/V would move."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import nano_vllm_amd as nv  # noqa: E402
from hipmem import DeviceArray, _hip  # noqa: E402
from nano_vllm_amd import workloads as wl  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seqs", type=int, default=8)
ap.add_argument("--start", type=int, default=0)
ap.add_argument("--chunk", type=int, default=2048)
ap.add_argument("--mode", default="tiled", choices=["tiled", "rowwise", "auto"])
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--geom", default="llama", choices=["llama", "mha", "tiny"])
args = ap.parse_args()

g = {"llama": wl.G_LLAMA, "mha": wl.G_MHA, "tiny": wl.G_TINY}[args.geom]
work = wl.uniform_workload("prefill", args.seqs, args.start + args.chunk, g)
flags = {"tiled": nv.PREFILL_TILED, "rowwise": nv.PREFILL_ROWWISE, "auto": nv.PREFILL_AUTO}[args.mode]
total = args.seqs * args.chunk
starts = np.full(args.seqs, args.start, np.int32)
lens = np.full(args.seqs, args.chunk, np.int32)
with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
    pool.fill_uniform(0, 1, 2)
    q = DeviceArray(wl.uniform(3, 0, total * g.n_heads * g.head_dim).reshape(total, g.n_heads, g.head_dim))
    out = DeviceArray(np.zeros((total, g.n_heads, g.head_dim), np.float32))
    tabs = DeviceArray(work.block_tables)
    run = lambda: nv.prefill_device(pool, 0, g.n_heads, q, tabs, args.seqs, work.max_blocks, starts, lens, out, flags)
    for _ in range(2):
        run()
    pool.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        run()
    pool.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
keys = args.seqs * sum(args.start + b + 1 for b in range(args.chunk))
flops = 4.0 * keys * g.n_heads * g.head_dim
kv_bytes = 2.0 * args.seqs * (args.start + args.chunk) * g.n_kv_heads * g.head_dim * 4
print(f"{args.mode:8s} geom={args.geom} seqs={args.seqs} start={args.start} chunk={args.chunk}: {dt * 1e3:9.3f} ms  "
      f"{flops / dt / 1e12:7.2f} TFLOP/s ({100 * flops / dt / 157.3e12:5.1f}% of fp32 MFMA peak)  "
      f"{total / dt:12.0f} query tokens/s  one-pass KV {kv_bytes / 1e6:.0f} MB")
