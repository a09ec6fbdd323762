"""One projection (nvk_batch_matmul_device) or the gate | up pair (nvk_batch_matmul2_device) timed on one MI355X:
    python tools/bench_matmul.py --shapes wo:2048:2048,down:5632:2048,gateup:2048:5632x2,qkv:4096:6144 --rows 1,3,8,16,24,32
name:in_dim:out_dim[x2]; per shape and row count us per back-to-back call and the weight stream's GB/s."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import nano_vllm_amd as nv  # noqa: E402
from hipmem import DeviceArray, _hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="wo:2048:2048,down:5632:2048,gateup:2048:5632x2,qkv2560:2048:2560,wide:4096:6144")
ap.add_argument("--rows", default="1,3,8,16,24,32")
ap.add_argument("--iters", type=int, default=50)
args = ap.parse_args()
rng = np.random.default_rng(1)
for spec in args.shapes.split(","):
    name, k, n = spec.split(":")
    two = n.endswith("x2")
    K, N = int(k), int(n[:-2] if two else n)
    w = [DeviceArray((rng.random((N, K), dtype=np.float32) - 0.5) / np.float32(np.sqrt(K))) for _ in range(2 if two else 1)]
    wbytes = (2 if two else 1) * N * K * 4
    line = f"{name:8s} {K:5d} -> {N:5d}{' x2' if two else '   '}:"
    for B in (int(r) for r in args.rows.split(",")):
        x = DeviceArray(rng.random((B, K), dtype=np.float32) - 0.5)
        o = [DeviceArray(np.zeros((B, N), np.float32)) for _ in w]
        call = (lambda: nv.batch_matmul2_device(o[0], o[1], x, w[0], w[1], B, K, N, N)) if two else \
            (lambda: nv.batch_matmul_device(o[0], x, w[0], B, K, N))
        for _ in range(5):
            call()
        _hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            call()
        _hip.hipDeviceSynchronize()
        dt = (time.perf_counter() - t0) / args.iters
        line += f"  {B:2d} rows {dt * 1e6:6.1f} us {wbytes / dt / 1e12:4.2f} TB/s |"
    print(line, flush=True)
