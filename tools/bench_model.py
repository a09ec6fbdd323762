"""Engine-level decode throughput of the device-resident model (nvk_model_*) on one MI355X: a synthetic llama2.c
checkpoint of a chosen geometry is written to /tmp, loaded, `--seqs` requests are prefilled with `--ctx` prompt
tokens each (in chunks) and then decoded for `--steps` model steps.

    python tools/bench_model.py [--preset stories110m|tinyllama|llama7b-8l] [--seqs 64] [--ctx 512] [--steps 16]
Reports ms per model step and generated tokens/s (whole forward pass: embedding .. argmax, fp32)."""
import argparse
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import nano_vllm_amd as nv  # noqa: E402

PRESETS = {  # dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab, max_seq_len
    "stories15m": (288, 768, 6, 6, 6, 32000, 256),
    "stories110m": (768, 2048, 12, 12, 12, 32000, 1024),
    "tinyllama": (2048, 5632, 22, 32, 4, 32000, 2048),
    "llama7b-8l": (4096, 11008, 8, 32, 8, 32000, 4096),  # 8 of Llama-2-7B's 32 layers with GQA 32/8 (6.9 GB of fp32)
}
ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="tinyllama", choices=sorted(PRESETS))
ap.add_argument("--seqs", type=int, default=64)
ap.add_argument("--ctx", type=int, default=512)
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--linked", action="store_true", help="carry the norms of 1-16 row steps between the GEMVs (opt-in: a few ulp from the separate kernels)")
ap.add_argument("--no-graphs", action="store_true", help="launch every decode step's kernels directly (no hipGraph replay)")
ap.add_argument("--chunk", type=int, default=2048, help="prompt tokens fed per request and step during prefill")
args = ap.parse_args()
D, Hd, L, H, KV, V, S = PRESETS[args.preset]
hd = D // H
assert args.ctx + args.steps + 4 <= S, "context + steps exceed the preset's max_seq_len"
path = f"/tmp/nvk_bench_{args.preset}.bin"
sizes = [V * D, L * D, L * D * H * hd, L * D * KV * hd, L * D * KV * hd, L * H * hd * D, L * D, L * D * Hd, L * Hd * D,
         L * D * Hd, D]
if not os.path.exists(path) or os.path.getsize(path) != 28 + 4 * sum(sizes):
    t0 = time.perf_counter()
    rng = np.random.default_rng(1)
    with open(path + ".tmp", "wb") as fh:
        fh.write(struct.pack("7i", D, Hd, L, H, KV, V, S))
        for i, n in enumerate(sizes):
            norm = i in (1, 6, 10)
            left = n
            while left:
                m = min(left, 1 << 24)
                w = rng.random(m, dtype=np.float32)
                w = (1.0 + 0.1 * (w - 0.5)) if norm else (w - 0.5) * np.float32(2.0 / np.sqrt(D if i else 1.0) if i else 1.0)
                fh.write(w.astype(np.float32).tobytes())
                left -= m
    os.replace(path + ".tmp", path)
    print(f"wrote {path}: {os.path.getsize(path) / 1e9:.2f} GB in {time.perf_counter() - t0:.1f} s", flush=True)
bs = 16
blocks = args.seqs * ((args.ctx + args.steps + bs) // bs + 1)
t0 = time.perf_counter()
with nv.LlamaModel(path, bs, blocks, max_seqs=args.seqs) as m:
    if args.no_graphs:
        m.use_graphs(False)
    if args.linked:
        m.use_linked_norms(True)
    print(f"{args.preset}: dim {D}, {L} layers, {H}/{KV} heads x {hd}, hidden {Hd}, vocab {V}; loaded in "
          f"{time.perf_counter() - t0:.1f} s; {args.seqs} requests x {args.ctx} prompt tokens", flush=True)
    rng = np.random.default_rng(2)
    ids = list(range(args.seqs))
    for r in ids:
        m.add_request(r)
    t0 = time.perf_counter()
    fed = 0
    while fed < args.ctx:
        c = min(args.chunk, args.ctx - fed)
        nxt, _ = m.step_chunks(ids, [rng.integers(3, V, size=c).tolist() for _ in ids])
        fed += c
    t_prefill = time.perf_counter() - t0
    for _ in range(4):  # warm-up decode steps (the graph is recorded during these)
        nxt, _ = m.step(ids, nxt.tolist())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nxt, _ = m.step(ids, nxt.tolist())
    dt = (time.perf_counter() - t0) / args.steps
weight_bytes = 4 * (sum(sizes) - V * D)  # every layer weight + the classifier are read once per step
kv_bytes = args.seqs * L * 2 * (args.ctx + args.steps / 2) * KV * hd * 4
print(f"prefill {args.seqs * args.ctx / t_prefill:10.0f} tokens/s ({t_prefill * 1e3:.1f} ms)   decode {dt * 1e3:8.3f} ms/step  "
      f"{args.seqs / dt:9.0f} tokens/s   (weights {weight_bytes / 1e9:.2f} GB + KV {kv_bytes / 1e9:.2f} GB per step = "
      f"{(weight_bytes + kv_bytes) / dt / 1e12:.2f} TB/s)")
