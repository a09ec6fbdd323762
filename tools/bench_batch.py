"""The decode batch (SURVEY.md 8f rows f1 + f2) timed on one MI355X: model steps of L layers, each layer =
RoPE + KV append + attention for every running request (nvk_batch_layer), tables kept resident.

    python tools/bench_batch.py [--seqs 512] [--ctx 2048] [--layers 32] [--steps 8]
The pool is pre-filled so that every request already has `ctx` tokens; then `steps` model steps are timed.
Reports ms per model step, tokens/s, and the attention-path HBM rate (K/V read once per layer + the appended rows)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import nano_vllm_amd as nv  # noqa: E402
from hipmem import DeviceArray  # noqa: E402
from nano_vllm_amd import workloads as wl  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seqs", type=int, default=512)
ap.add_argument("--ctx", type=int, default=2048)
ap.add_argument("--layers", type=int, default=4, help="layers held in the pool (a step runs --model-layers layer calls over them)")
ap.add_argument("--model-layers", type=int, default=32)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--graph", action="store_true",
                help="capture the step's layer loop in a hipGraph once and replay it (measurement of the device-side "
                     "time without host launch cost; lengths do not advance between replays)")
args = ap.parse_args()

g = wl.G_LLAMA
bs = g.block_size
max_blocks = (args.ctx + args.steps + bs - 1) // bs + 1
nb = args.seqs * max_blocks
bm = nv.BlockManager(nb, bs)
with nv.PagedKVPool(args.layers, nb, bs, g.n_kv_heads, g.head_dim) as pool, \
        nv.DecodeBatch(pool, bm, args.seqs, max_blocks) as batch:
    for layer in range(args.layers):
        pool.fill_uniform(layer, 1 + layer, 100 + layer)
    ids = list(range(args.seqs))
    for r in ids:
        batch.add_request(r)
    q = DeviceArray(wl.uniform(3, 0, args.seqs * g.n_heads * g.head_dim).reshape(args.seqs, g.n_heads, g.head_dim))
    k = DeviceArray(wl.uniform(4, 0, args.seqs * g.row_len).reshape(args.seqs, g.row_len))
    v = DeviceArray(wl.uniform(5, 0, args.seqs * g.row_len).reshape(args.seqs, g.row_len))
    out = DeviceArray(np.zeros((args.seqs, g.n_heads, g.head_dim), np.float32))

    def model_step(layer_calls):
        batch.begin_step(ids, g.n_heads)
        for i in range(layer_calls):
            batch.layer(i % args.layers, q, k, v, out, 10000.0)
        batch.end_step()

    # bring every request to `ctx` tokens: the bookkeeping advances, the pool content is the synthetic fill
    t0 = time.perf_counter()
    for _ in range(args.ctx):
        batch.begin_step(ids, g.n_heads)
        batch.end_step()
    pool.synchronize()
    warm = time.perf_counter() - t0
    model_step(args.model_layers)  # warm-up (sizes the split-K workspace)
    pool.synchronize()
    if args.graph:
        from hipmem import Stream
        stream = Stream()
        batch.begin_step(ids, g.n_heads, stream=stream)
        stream.synchronize()

        def layers():
            for i in range(args.model_layers):
                batch.layer(i % args.layers, q, k, v, out, 10000.0, stream=stream)

        exe = stream.capture(layers)
        stream.launch(exe)
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            stream.launch(exe)
        stream.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        batch.end_step()
    else:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model_step(args.model_layers)
        pool.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
    st = batch.stats()
ctx_mid = args.ctx + 1 + (args.steps + 1) / 2
kv_bytes = args.model_layers * args.seqs * (2 * ctx_mid * g.row_len * 4 + 2 * g.n_heads * g.head_dim * 4 + 3 * g.row_len * 4)
print(f"decode batch{' (hipGraph replay)' if args.graph else ''}: {args.seqs} requests x ctx {args.ctx}, {args.model_layers} layer calls per step: "
      f"{dt * 1e3:8.3f} ms/step  {args.seqs / dt:10.0f} tokens/s  attention path {kv_bytes / dt / 1e9:7.1f} GB/s "
      f"({100 * kv_bytes / dt / 8e12:4.1f}% of HBM peak)  bookkeeping-only steps {warm / args.ctx * 1e6:6.1f} us each  {st}")
