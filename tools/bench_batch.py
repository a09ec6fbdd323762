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
ap.add_argument("--prefill-chunk", type=int, default=0,
                help="mixed steps: one more request feeds a prompt chunk of this many tokens in every step "
                     "(nvk_batch_begin_step_chunks), beside the --seqs decoding requests")
ap.add_argument("--prefill-ctx", type=int, default=0, help="tokens that request already holds when timing starts")
ap.add_argument("--hidden", type=int, default=0, metavar="DIM",
                help="start every layer call from the hidden state x [rows, DIM]: RMSNorm + q/k/v projections on the "
                     "device (nvk_batch_layer_hidden), one weight set per pool layer (so the Infinity Cache does not "
                     "hold them between calls)")
ap.add_argument("--graph", action="store_true",
                help="capture the step's layer loop in a hipGraph once and replay it (measurement of the device-side "
                     "time without host launch cost; lengths do not advance between replays)")
args = ap.parse_args()

g = wl.G_LLAMA
bs = g.block_size
dec_blocks = (args.ctx + args.steps + bs - 1) // bs + 1
pre_blocks = (args.prefill_ctx + args.prefill_chunk * (args.steps + 2) + bs - 1) // bs + 1 if args.prefill_chunk else 0
max_blocks = max(dec_blocks, pre_blocks)
nb = args.seqs * dec_blocks + pre_blocks
rows = args.seqs + args.prefill_chunk
bm = nv.BlockManager(nb, bs)
with nv.PagedKVPool(args.layers, nb, bs, g.n_kv_heads, g.head_dim) as pool, \
        nv.DecodeBatch(pool, bm, args.seqs + 1, max_blocks) as batch:
    for layer in range(args.layers):
        pool.fill_uniform(layer, 1 + layer, 100 + layer)
    ids = list(range(args.seqs))
    for r in ids:
        batch.add_request(r)
    q = DeviceArray(wl.uniform(3, 0, rows * g.n_heads * g.head_dim).reshape(rows, g.n_heads, g.head_dim))
    k = DeviceArray(wl.uniform(4, 0, rows * g.row_len).reshape(rows, g.row_len))
    v = DeviceArray(wl.uniform(5, 0, rows * g.row_len).reshape(rows, g.row_len))
    out = DeviceArray(np.zeros((rows, g.n_heads, g.head_dim), np.float32))
    PREFILL_ID = args.seqs
    if args.prefill_chunk:
        batch.add_request(PREFILL_ID)
        if args.prefill_ctx:  # bookkeeping only, like the decode requests below
            batch.begin_step_chunks([PREFILL_ID], [args.prefill_ctx], g.n_heads)
            batch.end_step()

    if args.hidden:
        D = args.hidden
        sc = np.float32(1.0 / np.sqrt(D))
        xh = DeviceArray(wl.uniform(6, 0, rows * D).reshape(rows, D))
        Wt = [dict(g=DeviceArray(wl.uniform(7 + 10 * l, 0, D) + np.float32(1.5)),
                   wq=DeviceArray((wl.uniform(8 + 10 * l, 0, g.n_heads * g.head_dim * D) * sc)),
                   wk=DeviceArray((wl.uniform(9 + 10 * l, 0, g.row_len * D) * sc)),
                   wv=DeviceArray((wl.uniform(10 + 10 * l, 0, g.row_len * D) * sc))) for l in range(args.layers)]

    def one_layer(i, stream=None):
        if args.hidden:
            w = Wt[i % args.layers]
            batch.layer_hidden(i % args.layers, xh, args.hidden, w["g"], w["wq"], w["wk"], w["wv"], out, 10000.0,
                               stream=stream)
        else:
            batch.layer(i % args.layers, q, k, v, out, 10000.0, stream=stream)

    def model_step(layer_calls):
        if args.prefill_chunk:
            batch.begin_step_chunks(ids + [PREFILL_ID], [1] * args.seqs + [args.prefill_chunk], g.n_heads)
        else:
            batch.begin_step(ids, g.n_heads)
        for i in range(layer_calls):
            one_layer(i)
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
        assert not args.prefill_chunk, "--graph times pure decode steps"
        from hipmem import Stream
        stream = Stream()
        batch.begin_step(ids, g.n_heads, stream=stream)
        stream.synchronize()

        def layers():
            for i in range(args.model_layers):
                one_layer(i, stream)

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
if args.prefill_chunk:
    pos_mid = args.prefill_ctx + args.prefill_chunk * (1 + args.steps / 2)  # start of the middle timed step's chunk
    flops = args.model_layers * 4 * g.n_heads * g.head_dim * args.prefill_chunk * (pos_mid + (args.prefill_chunk + 1) / 2)
    print(f"mixed steps: + one request feeding {args.prefill_chunk}-token chunks from position {args.prefill_ctx}: "
          f"{flops / 1e12:.2f} TFLOP of prefill attention per step beside the decode rows")
kv_bytes = args.model_layers * args.seqs * (2 * ctx_mid * g.row_len * 4 + 2 * g.n_heads * g.head_dim * 4 + 3 * g.row_len * 4)
if args.hidden:
    print(f"layers start from the hidden state (dim {args.hidden}): RMSNorm + q/k/v projections included in the step "
          f"({2.0 * rows * (g.n_heads * g.head_dim + 2 * g.row_len) * args.hidden * args.model_layers / 1e12:.2f} TFLOP per step)")
print(f"decode batch{' (hipGraph replay)' if args.graph else ''}: {args.seqs} requests x ctx {args.ctx}, {args.model_layers} layer calls per step: "
      f"{dt * 1e3:8.3f} ms/step  {args.seqs / dt:10.0f} tokens/s  attention path {kv_bytes / dt / 1e9:7.1f} GB/s "
      f"({100 * kv_bytes / dt / 8e12:4.1f}% of HBM peak)  bookkeeping-only steps {warm / args.ctx * 1e6:6.1f} us each  {st}")
