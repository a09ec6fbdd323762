"""Latency of ONE drop-in call, Attention::paged_attention's signature with a device pool bound (INTEGRATION.md
level 1): host q in, host out back, synchronous -- what the reference engine's own call site (model.hpp:405-415)
would see per layer and token.  (The reference's CPU function on one host core: bench.py's cpu_baseline leg,
103.6 ms per 64 sequences of 2048 tokens = 1.6 ms per call at ctx 2048.)  python tools/bench_dropin.py [--geom llama|tiny]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import nano_vllm_amd as nv  # noqa: E402
from nano_vllm_amd import workloads as wl  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--geom", default="llama", choices=["llama", "tiny"])
ap.add_argument("--ctx", default="16,256,2048,8192,32768")
args = ap.parse_args()
H, KV, hd, bs = (32, 8, 128, 16) if args.geom == "llama" else (6, 6, 48, 16)
ctxs = [int(c) for c in args.ctx.split(",")]
nb = max(ctxs) // bs + 2
with nv.PagedKVPool(1, nb, bs, KV, hd) as pool:
    pool.fill_uniform(0, 1, 2)
    nv.bind_pool(pool, 0)
    q = wl.uniform(3, 0, H * hd)
    out = np.empty(H * hd, np.float32)
    for T in ctxs:
        table = np.arange((T + bs - 1) // bs, dtype=np.int32)
        for _ in range(5):
            nv.paged_attention(out, q, None, None, table, None, T, bs, hd, H, KV)
        n = 200 if T <= 8192 else 50
        t0 = time.perf_counter()
        for _ in range(n):
            nv.paged_attention(out, q, None, None, table, None, T, bs, hd, H, KV)
        dt = (time.perf_counter() - t0) / n
        line = f"ctx {T:6d}: {dt * 1e6:8.1f} us per call ({2 * T * KV * hd * 4 / dt / 1e9:7.1f} GB/s of K/V)"
        print(line, flush=True)
    nv.bind_pool(None, 0)
