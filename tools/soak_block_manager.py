"""Random operation sequences on the product's BlockManager (nvk_bm_* through the Python mirror) against the reference's
own class (oracle/_ref/libnvk_ref.so, built from /root/reference where that exists) and the CPU checker -- every result
of every operation equal.  CPU only:
    python tools/soak_block_manager.py --first 0 --minutes 2"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as ol  # noqa: E402
from golden_defs import BmCLib, BmProduct, replay  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--first", type=int, default=0)
ap.add_argument("--minutes", type=float, default=2.0)
args = ap.parse_args()


def script(seed):
    rng = np.random.default_rng(seed)
    nb = int(rng.choice([1, 2, 7, 63, 64, 65, 131, 500, 4097]) if rng.random() < 0.5 else rng.integers(1, 3000))
    bs = int(rng.choice([1, 4, 8, 16, 32]))
    ops = []
    for _ in range(int(rng.integers(50, 1500))):
        r = int(rng.integers(0, 100))
        blk = int(rng.integers(-3, nb + 3))
        req = int(rng.integers(0, 6))
        if r < 35:
            ops.append(["alloc"])
        elif r < 60:
            ops.append(["free", blk])
        elif r < 68:
            ops.append(["alloc_seq", int(rng.integers(0, 4 * bs * max(1, nb // 8) + 2))])
        elif r < 72:
            ops.append(["free_seq", [int(b) for b in rng.integers(-1, nb + 1, size=int(rng.integers(0, 6)))]])
        elif r < 78:
            ops.append(["is_free", blk])
        elif r < 82:
            ops.append(["num_free"])
        elif r < 84:
            ops.append(["util"])
        elif r < 90:
            ops.append(["req_alloc", req, int(rng.integers(0, 3 * bs * max(1, nb // 10) + 2))])
        elif r < 93:
            ops.append(["req_alloc_block", req])
        elif r < 96:
            ops.append(["req_free", req])
        elif r < 98:
            ops.append(["req_blocks", req])
        elif r < 99:
            ops.append(["req_count", req])
        else:
            ops.append(["active"])
    return nb, bs, ops


against = [("the checker", ol.ORACLE, "oracle_")] + ([("the reference's class", ol.REF, "ref_")] if ol.REF else [])
t0, seed, n_ops, bad = time.perf_counter(), args.first, 0, 0
while time.perf_counter() - t0 < args.minutes * 60:
    nb, bs, ops = script(seed)
    got = replay(BmProduct(nb, bs), ops)
    for name, lib, prefix in against:
        ref = BmCLib(lib, prefix, nb, bs)
        # (a block freed behind a request's back and drawn again is in its list twice: lists outgrow the pool)
        ref.cap = nb + len(ops)
        ref.buf = (C.c_int * ref.cap)()
        want = replay(ref, ops)
        if got != want:
            i = next(k for k in range(len(ops)) if got[k] != want[k])
            print(f"seed {seed} ({nb} blocks of {bs}): operation {i} {ops[i]}: product {got[i]}, {name} {want[i]}", flush=True)
            bad += 1
    n_ops += len(ops)
    seed += 1
print(f"seeds {args.first}..{seed - 1}: {seed - args.first} sequences, {n_ops} operations, against {' and '.join(a[0] for a in against)}: "
      f"{bad} differences ({time.perf_counter() - t0:.0f} s)")
sys.exit(1 if bad else 0)
