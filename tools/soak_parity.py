"""Seeds beyond the ones `pytest -m gpu` runs, for the differential sweeps of tests/ (random head geometry, block size,
ragged lengths, fragmented tables; decode in every mode, the in-kernel merge and plan-free launches, skewed long chunks,
chunked prefill), each against the CPU checker at the path's tolerance, and random products through the projection
entry points against float64:
    python tools/soak_parity.py --first 100 --minutes 8
One line per sweep: seeds run, failures (with the seed and the assertion's first line).  Exit status 1 on any failure."""
import argparse
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

ap = argparse.ArgumentParser()
ap.add_argument("--first", type=int, default=100, help="first seed (the tests run the seeds below 36)")
ap.add_argument("--minutes", type=float, default=8.0)
ap.add_argument("--only", default="", help="run the sweeps whose name contains this")
args = ap.parse_args()

import test_gpu_attention as ta  # noqa: E402
import test_gpu_fused_merge as tm  # noqa: E402
import test_gpu_prefill as tp  # noqa: E402

import numpy as np  # noqa: E402
import nano_vllm_amd as nv  # noqa: E402
from hipmem import DeviceArray  # noqa: E402


def projection_case(seed):
    """One random product through the C ABI's projection entry points (every kernel family and its thresholds are a
    function of rows, columns, contraction and the number of products sharing the activations) against float64."""
    rng = np.random.default_rng(seed)
    B = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 24, 32, 33, 48, 64, 65, 96, 97, 128, 129, 200, 256, 300, 512, 700])
            if rng.random() < 0.7 else rng.integers(1, 600))
    K = int(rng.choice([64, 256, 288, 768, 2048, 4096, 5632]) if rng.random() < 0.6 else 4 * rng.integers(1, 1500))
    kind = int(rng.integers(0, 4))  # matmul, matmul2, swiglu_ffn, q|k|v (with or without the norm)
    top = max(4, min(6000, int(3e9 / (B * K))))
    dims = [int(rng.choice([32, 64, 256, 2048, 2560, 5632]) if rng.random() < 0.4 else rng.integers(1, top)) for _ in range(3)]
    dims = [min(d, top) for d in dims]
    x = (rng.random((B, K), dtype=np.float32) - 0.5)
    mk = lambda n: ((rng.random((n, K), dtype=np.float32) - 0.5) / np.float32(np.sqrt(K) / 2))  # noqa: E731
    xd = DeviceArray(x)
    x64 = x.astype(np.float64)
    what = f"seed {seed}: kind {kind}, {B} rows, {K} -> {dims}"

    def close(got, want, tag):
        err = np.abs(got.astype(np.float64) - want)
        tol = 1e-4 + 1e-5 * np.abs(want)
        assert np.isfinite(got).all() and (err <= tol).all(), f"{what} {tag}: max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"

    if kind == 0:
        w = mk(dims[0])
        o = DeviceArray(np.full((B, dims[0]), np.nan, np.float32))
        nv.batch_matmul_device(o, xd, DeviceArray(w), B, K, dims[0])
        close(o.numpy(), x64 @ w.astype(np.float64).T, "matmul")
    elif kind == 1:
        wa, wb = mk(dims[0]), mk(dims[1])
        oa, ob = DeviceArray(np.full((B, dims[0]), np.nan, np.float32)), DeviceArray(np.full((B, dims[1]), np.nan, np.float32))
        nv.batch_matmul2_device(oa, ob, xd, DeviceArray(wa), DeviceArray(wb), B, K, dims[0], dims[1])
        close(oa.numpy(), x64 @ wa.astype(np.float64).T, "matmul2 a")
        close(ob.numpy(), x64 @ wb.astype(np.float64).T, "matmul2 b")
    elif kind == 2:
        wg, wu = mk(dims[0]), mk(dims[0])
        hb, hb2 = DeviceArray(np.full((B, dims[0]), np.nan, np.float32)), DeviceArray(np.zeros((B, dims[0]), np.float32))
        nv.batch_swiglu_ffn_device(hb, hb2, xd, DeviceArray(wg), DeviceArray(wu), B, K, dims[0])
        g = x64 @ wg.astype(np.float64).T
        close(hb.numpy(), g / (1.0 + np.exp(-g)) * (x64 @ wu.astype(np.float64).T), "swiglu")
    else:
        qd, kvd = dims[0], dims[1]
        wq, wk, wv = mk(qd), mk(kvd), mk(kvd)
        normed = bool(rng.integers(0, 2))
        gain = (1.0 + 0.1 * (rng.random(K, dtype=np.float32) - 0.5)).astype(np.float32)
        q, k, v = (DeviceArray(np.full((B, n), np.nan, np.float32)) for n in (qd, kvd, kvd))
        xb = DeviceArray(np.zeros((B, K), np.float32))
        nv.qkv_project_device(q, k, v, xd, xb if normed else None, DeviceArray(gain) if normed else None, DeviceArray(wq),
                              DeviceArray(wk), DeviceArray(wv), B, K, qd, kvd)
        a = x64
        if normed:
            a = x64 / np.sqrt((x64 * x64).mean(axis=1, keepdims=True) + 1e-5) * gain.astype(np.float64)
        for o, w, tag in ((q, wq, "q"), (k, wk, "k"), (v, wv, "v")):
            close(o.numpy(), a @ w.astype(np.float64).T, tag)


SWEEPS = [
    ("projections: every entry point", projection_case),
    ("decode: auto / single pass / split-K", ta.test_random_geometry_batches_against_checker),
    ("decode: in-kernel merge, plan-free", tm.test_in_kernel_merge_and_planless_decode_on_random_geometries),
    ("decode: skewed long chunks", tm.test_skewed_chunks_on_random_long_batches),
    ("prefill: paged chunks", tp.test_random_geometry_prefill_against_checker),
]
SWEEPS = [sw for sw in SWEEPS if args.only in sw[0]]
budget = args.minutes * 60 / len(SWEEPS)
bad = 0
for name, fn in SWEEPS:
    t0, seed, fails = time.perf_counter(), args.first, []
    said = t0
    while time.perf_counter() - t0 < budget:
        if time.perf_counter() - said > 30:  # (a run that writes nothing for minutes is taken to be hung)
            print(f"  ... {name}: at seed {seed}", flush=True)
            said = time.perf_counter()
        try:
            fn(seed)
        except Exception as e:  # an assertion, or an error code from the library
            fails.append((seed, (str(e).splitlines() or [type(e).__name__])[0][:200]))
            traceback.print_exc(limit=1)
        seed += 1
    print(f"{name:40s} seeds {args.first}..{seed - 1} ({seed - args.first} cases, {time.perf_counter() - t0:.0f} s): "
          f"{len(fails)} failed {fails if fails else ''}", flush=True)
    bad += len(fails)
sys.exit(1 if bad else 0)
