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

from fuzz_cases import cli_case, decode_large_case, dropin_case, dropin_contiguous_case, model_case, noalloc_case, plan_case, pool_ops_case, prefill_long_case, projection_case, shard_one_rank_case  # noqa: E402

SWEEPS = [
    ("model: random checkpoints, batches", model_case),
    ("model group: members on one device", lambda seed: model_case(seed, True)),
    ("model: fewer blocks than needed", lambda seed: model_case(seed, False, True)),
    ("model: real widths", lambda seed: model_case(seed, big=True)),
    ("pool: fill and read entry points", pool_ops_case),
    ("decode: plans over the steps of a batch", plan_case),
    ("shard: one-rank communicator", shard_one_rank_case),
    ("nvk_run: request files", cli_case),
    ("drop-in: host slabs, both policies", dropin_case),
    ("drop-in: contiguous cache", dropin_contiguous_case),
    ("decode: the no-alloc entry point", noalloc_case),
    ("prefill: deep into long prompts", prefill_long_case),
    ("decode: large batches, every modifier", decode_large_case),
    ("projections: every entry point", projection_case),
    ("decode: auto / single pass / split-K", ta.test_random_geometry_batches_against_checker),
    ("decode: in-kernel merge, plan-free", tm.test_in_kernel_merge_and_planless_decode_on_random_geometries),
    ("decode: skewed long chunks", tm.test_skewed_chunks_on_random_long_batches),
    ("prefill: paged chunks", tp.test_random_geometry_prefill_against_checker),
]
SWEEPS_ALL = SWEEPS
SWEEPS = [sw for sw in SWEEPS if args.only in sw[0]]
budget = args.minutes * 60 / max(1, len(SWEEPS))
bad = 0
for name, fn in SWEEPS:
    t0, seed, fails, notes = time.perf_counter(), args.first, [], {}
    said = t0
    while time.perf_counter() - t0 < budget:
        if time.perf_counter() - said > 30:  # (a run that writes nothing for minutes is taken to be hung)
            print(f"  ... {name}: at seed {seed}", flush=True)
            said = time.perf_counter()
        try:
            note = fn(seed)
            if note:
                notes[note] = notes.get(note, 0) + 1
        except Exception as e:  # an assertion, or an error code from the library
            fails.append((seed, (str(e).splitlines() or [type(e).__name__])[0][-160:]))
            traceback.print_exc(limit=1)
        seed += 1
    print(f"{name:40s} seeds {args.first}..{seed - 1} ({seed - args.first} cases, {time.perf_counter() - t0:.0f} s): "
          f"{len(fails)} failed {fails if fails else ''}{' ' + str(notes) if notes else ''}", flush=True)
    bad += len(fails)
if __name__ == "__main__":
    sys.exit(1 if bad else 0)
