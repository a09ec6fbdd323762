"""The random sweeps of tests/fuzz_cases.py and tests/ from several host threads at once (ctypes drops the GIL inside the
library): every thread owns its pools and arrays, the library's process-wide state (first-use tables, per-stream
workspaces, plan caches, the null stream they all launch on) is shared.  A race shows as a wrong row, an error code or a
crash.    python tools/soak_threads.py --threads 4 --minutes 3"""
import argparse
import os
import sys
import threading
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=4)
ap.add_argument("--minutes", type=float, default=3.0)
ap.add_argument("--first", type=int, default=100)
args = ap.parse_args()

import test_gpu_attention as ta  # noqa: E402
import test_gpu_prefill as tp  # noqa: E402
from fuzz_cases import dropin_case, dropin_contiguous_case, model_case, pool_ops_case  # noqa: E402

# (sweeps whose every device call goes through the library: host arrays in, host arrays out.  The projection and no-alloc
#  sweeps, and the merge sweep, allocate and copy through the HIP runtime directly (tests/hipmem.py), on the null stream; ROCm 7.2 refuses such
#  calls while ANY stream of the process records a graph, which is outside the library's gate -- INTEGRATION.md.)
KINDS = [("decode", ta.test_random_geometry_batches_against_checker),
         ("prefill", tp.test_random_geometry_prefill_against_checker), ("model", model_case),
         ("model, fewer blocks", lambda seed: model_case(seed, scarce=True)),
         ("drop-in, host slabs", dropin_case), ("drop-in, contiguous", dropin_contiguous_case), ("pool", pool_ops_case)]
done, fails, lock = {k: 0 for k, _ in KINDS}, [], threading.Lock()
t_end = time.perf_counter() + args.minutes * 60


def worker(tid):
    seed = args.first + 100000 * tid
    while time.perf_counter() < t_end:
        name, fn = KINDS[seed % len(KINDS)]
        try:
            fn(seed)
            with lock:
                done[name] += 1
        except Exception as e:
            with lock:
                fails.append((tid, name, seed, (str(e).splitlines() or [type(e).__name__])[0][-200:]))
            traceback.print_exc(limit=2)
        seed += 1


threads = [threading.Thread(target=worker, args=(i,)) for i in range(args.threads)]
t0 = time.perf_counter()
for t in threads:
    t.start()
while any(t.is_alive() for t in threads):
    time.sleep(20)
    print(f"  ... {sum(done.values())} cases, {len(fails)} failed", flush=True)
for t in threads:
    t.join()
print(f"{args.threads} threads, {time.perf_counter() - t0:.0f} s: {done}: {len(fails)} failed {fails[:8] if fails else ''}", flush=True)
sys.exit(1 if fails else 0)
