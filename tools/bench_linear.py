"""The q/k/v projection (nvk_qkv_project_device) timed on one MI355X.

    python tools/bench_linear.py [--dim 4096] [--q-dim 4096] [--kv-dim 1024] [--rows 1,8,32,64,128,512,2048] [--no-norm]
Per row count: ms per call (wall clock over back-to-back launches; run under `rocprofv3 --kernel-trace --stats` for the
kernels' own durations), TFLOP/s of 2*rows*(q_dim+2*kv_dim)*dim against the fp32 MFMA peak (157.3), and GB/s of the
weight bytes against HBM (8000): <= 32 rows is the weight-streaming kernel, more is the tiled MFMA kernel."""
import argparse
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
ap.add_argument("--dim", type=int, default=4096)
ap.add_argument("--q-dim", type=int, default=4096)
ap.add_argument("--kv-dim", type=int, default=1024)
ap.add_argument("--rows", default="1,8,32,64,128,256,512,2048")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--no-norm", action="store_true", help="projections only (x is used as it is)")
args = ap.parse_args()

D, QD, KD = args.dim, args.q_dim, args.kv_dim
sc = np.float32(1.0 / np.sqrt(D))
wq = DeviceArray((wl.uniform(1, 0, QD * D) * sc).reshape(QD, D))
wk = DeviceArray((wl.uniform(2, 0, KD * D) * sc).reshape(KD, D))
wv = DeviceArray((wl.uniform(3, 0, KD * D) * sc).reshape(KD, D))
g = None if args.no_norm else DeviceArray(wl.uniform(4, 0, D) + np.float32(1.5))
wbytes = (QD + 2 * KD) * D * 4
for B in (int(r) for r in args.rows.split(",")):
    x = DeviceArray(wl.uniform(5, 0, B * D).reshape(B, D))
    xb = None if args.no_norm else DeviceArray(np.zeros((B, D), np.float32))
    q, k, v = (DeviceArray(np.zeros((B, n), np.float32)) for n in (QD, KD, KD))

    def call():
        nv.qkv_project_device(q, k, v, x, xb, g, wq, wk, wv, B, D, QD, KD)

    for _ in range(3):
        call()
    _hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        call()
    _hip.hipDeviceSynchronize()
    dt = (time.perf_counter() - t0) / args.iters
    flops = 2.0 * B * (QD + 2 * KD) * D
    print(f"rows {B:5d}: {dt * 1e6:9.1f} us/call  {flops / dt / 1e12:7.2f} TFLOP/s ({100 * flops / dt / 157.3e12:5.1f}% of fp32 MFMA peak)  "
          f"weights {wbytes / dt / 1e9:7.1f} GB/s ({100 * wbytes / dt / 8e12:5.1f}% of HBM peak)", flush=True)
