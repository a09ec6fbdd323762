"""The CPU checker (oracle/nvk_oracle.c) against the reference's own functions (oracle/_ref/libnvk_ref.so, built from
/root/reference where that exists), bit for bit, on seeds beyond the ones tests/test_oracle_vs_ref.py runs: paged and
contiguous attention on random geometry, RoPE, chunked-prefill attention, RMSNorm and matmul on random shapes.
CPU only:    python tools/soak_oracle.py --first 1000 --minutes 2"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as ol  # noqa: E402
import test_oracle_vs_ref as t  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--first", type=int, default=1000)
ap.add_argument("--minutes", type=float, default=2.0)
args = ap.parse_args()
if ol.REF is None:
    sys.exit("oracle/_ref/libnvk_ref.so is not built here (it needs /root/reference)")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def others(seed):
    rng = np.random.default_rng(seed)
    KV = int(rng.choice([1, 2, 3, 4, 6, 8]))
    H = KV * int(rng.choice([1, 2, 3, 4, 8]))
    hd = int(rng.choice([4, 16, 20, 48, 64, 96, 128, 256]))
    q = (rng.random(H * hd, dtype=np.float32) - 0.5) * np.float32(6)
    k = (rng.random(KV * hd, dtype=np.float32) - 0.5) * np.float32(6)
    pos, theta = int(rng.choice([0, 1, 17, 4095, 100_000, int(rng.integers(0, 2_000_000))])), float(rng.choice([10000.0, 500000.0]))
    a = ol.apply_rope(ol.ORACLE.oracle_apply_rope, q, k, pos, hd, H, KV, theta)
    b = ol.apply_rope(ol.REF.ref_apply_rope, q, k, pos, hd, H, KV, theta)
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[1]), bits(b[1])), f"rope seed {seed}"
    start, C_ = int(rng.integers(0, 60)), int(rng.integers(1, 24))
    T = start + C_
    qc = rng.random((C_, H, hd), dtype=np.float32) - np.float32(0.5)
    kc = rng.random((T, KV, hd), dtype=np.float32) - np.float32(0.5)
    vc = rng.random((T, KV, hd), dtype=np.float32) - np.float32(0.5)
    assert np.array_equal(bits(ol.oracle_chunked_attention(qc, kc, vc, start, KV)), bits(ol.ref_chunked_attention(qc, kc, vc, start, KV))), \
        f"chunked attention seed {seed}"
    B, D, N = int(rng.integers(1, 40)), int(rng.integers(1, 600)), int(rng.integers(1, 300))
    x = (rng.random((B, D), dtype=np.float32) - np.float32(0.5)) * np.float32(4)
    g = rng.random(D, dtype=np.float32) + np.float32(0.5)
    w = (rng.random((N, D), dtype=np.float32) - np.float32(0.5)) * np.float32(0.2)
    xn = ol.batch_rms_norm(ol.ORACLE.oracle_batch_rms_norm, x, g)
    for fn in (ol.REF.ref_rms_norm_rows, ol.REF.ref_batch_rms_norm):
        assert np.array_equal(bits(xn), bits(ol.batch_rms_norm(fn, x, g))), f"rms_norm seed {seed}"
    m = ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, xn, w)
    for fn in (ol.REF.ref_matmul_rows, ol.REF.ref_batch_matmul):
        assert np.array_equal(bits(m), bits(ol.batch_matmul(fn, xn, w))), f"matmul seed {seed}"


t0, seed, bad = time.perf_counter(), args.first, 0
while time.perf_counter() - t0 < args.minutes * 60:
    for fn in (t.test_paged_attention_random_geometry, others):
        try:
            fn(seed)
        except AssertionError as e:
            print(f"seed {seed}: {fn.__name__}: {e}", flush=True)
            bad += 1
    seed += 1
print(f"seeds {args.first}..{seed - 1}: {seed - args.first} cases each of attention (paged = contiguous = the reference's) and of RoPE / chunked attention / "
      f"RMSNorm / matmul, bit for bit against the reference: {bad} differences ({time.perf_counter() - t0:.0f} s)")
sys.exit(1 if bad else 0)
