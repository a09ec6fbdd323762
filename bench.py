#!/usr/bin/env python3
"""bench.py -- paged-attention decode on MI355X: achieved HBM GB/s and layer-call tokens/s.

    python bench.py --gpus N --steps K --warmup W [--workload config2|config3|config4|5a|5b]
                    [--shard seq|heads] [--mode auto|single|split] [--chunk T] [--no-cpu]

A "step" is one layer-call: one new token for every sequence of the batch in
one layer.  At N=1 the default workload is BASELINE.json
configs[1] (64 seqs x 2K ctx, uniform block tables, fp32 KV, 32q/8kv heads x
128).  For N>1 the path is sequence-sharded: every rank owns a private pool and
its own batch of that shape, there is no data-path collective, and the job's
throughput is the sum over ranks over the slowest rank's time (weak scaling).
--shard heads is the head-group variant: every rank holds n_kv_heads/N kv-heads
of ALL sequences and the out slices are joined by one RCCL all-gather per
layer-call, inside the timed region (strong scaling: the batch is fixed).

Inputs are resident in HBM before the timed region (the pool is filled by a
device kernel; q / block tables / lengths are device tensors).  Rank 0 prints
ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


GEOMS = {"llama": (32, 8, 128, 16), "mha": (32, 32, 128, 16), "tiny": (6, 6, 48, 16), "shard2": (16, 4, 128, 16),
         "shard8": (4, 1, 128, 16), "d64": (32, 8, 64, 16), "d256": (16, 4, 256, 16), "gqa8": (32, 4, 128, 16),
         "kvmul3": (24, 8, 128, 16), "mha64": (64, 64, 128, 16), "tinyllama": (32, 4, 64, 16)}


def make_workload(name, rank=0, geom="llama"):
    from nano_vllm_amd import workloads as wl

    if geom != "llama":  # diagnostic sweeps over head geometry (shardN = one rank's share of a head-sharded G-llama)
        g = wl.Geometry(*GEOMS[geom])
        w = {"config2": lambda: wl.config2(g), "config3": lambda: wl.config3(256, g), "config4": lambda: wl.config4(g),
             "5a": lambda: wl.config5a(512, g, first_seq=512 * rank), "5b": lambda: wl.config5b(512, g)}[name]()
        w.name += f"_{geom}"
        return w
    if name == "config2":
        return wl.config2()
    if name == "config3":
        return wl.config3()
    if name == "config4":
        return wl.config4()
    if name == "config3u":  # diagnostic: config 3's ragged lengths behind contiguous tables
        w = wl.config3()
        need = (w.ctx_lens.astype(np.int64) + w.geom.block_size - 1) // w.geom.block_size
        start = np.concatenate([[0], np.cumsum(need)[:-1]])
        tabs = np.zeros_like(w.block_tables)
        for s in range(w.num_seqs):
            tabs[s, :need[s]] = start[s] + np.arange(need[s])
        return wl.Workload("decode_256xragged_contiguous", w.geom, int(need.sum()), w.ctx_lens, tabs)
    if name == "config2f":  # diagnostic: config 2's uniform lengths behind fragmented tables
        return wl.fragmented_workload("decode_64x2048_fragmented", np.full(64, 2048, np.int32))
    if name.startswith("u") and "x" in name:  # diagnostic: uNxT = N sequences of T tokens behind fragmented tables
        n, t = (int(v) for v in name[1:].split("x"))
        return wl.fragmented_workload(f"decode_{n}x{t}_fragmented", np.full(n, t, np.int32))
    if name == "5a":
        return wl.config5a(512, first_seq=512 * rank)  # rank r's slice of the 4096-sequence batch
    if name == "5b":
        return wl.config5b(512)
    raise SystemExit(f"unknown workload {name}")


def profiled_traffic(work_name, kernel):
    """HBM bytes per launch of `kernel` (the instantiation this run launched, as rocprofv3 names it) from the
    committed PMC profile of this same workload (profiles/r*_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE
    passes, gfx950 corrections applied as MI355X_MICROARCH.md prescribes).  The latest round wins; None when no
    profile holds this (workload, kernel) pair -- never a number measured on another kernel."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):
        try:
            with open(path) as fh:
                prof = json.load(fh)
            if prof.get("workload") == work_name and kernel in prof.get("kernels", {}):
                best = prof["kernels"][kernel].get("hbm_bytes_per_launch_corrected", best)
        except (OSError, ValueError):
            continue
    return best


def cpu_baseline(work, cores):
    """The reference's CPU attention timed on this host (rank 0, N=1 only).

    kind "reference": the reference's own header compiled with its own flags
    (oracle/_ref/libnvk_ref_fast.so); kind "port": our C restatement with the
    same flags when that library did not travel.  The whole workload is run
    when it is small enough, otherwise a bounded sample of its sequences."""
    import oracle_lib as ol
    from concurrent.futures import ThreadPoolExecutor
    from nano_vllm_amd import workloads as wl

    g = work.geom
    # bound the sample: about 64 K context tokens per core-second at these flags
    budget_tokens = 2_500_000
    order = np.arange(work.num_seqs)
    take, tok = [], 0
    for s in order:
        if tok + int(work.ctx_lens[s]) > budget_tokens and take:
            break
        take.append(int(s))
        tok += int(work.ctx_lens[s])
    k_host, v_host, tabs, ctx = wl.host_blocks_for(work, take)
    q = work.q()[take]
    sample_bytes = int(2 * ctx.astype(np.int64).sum() * g.n_kv_heads * g.head_dim * 4
                       + 2 * len(take) * g.n_heads * g.head_dim * 4
                       + ((ctx.astype(np.int64) + g.block_size - 1) // g.block_size).sum() * 4)
    out = np.empty_like(q)
    p = lambda a: C.c_void_p(a.ctypes.data)

    ref = ol.load_fast_ref()
    port = ol.load_fast_oracle()
    if ref is not None:
        kind = "reference"

        def run(lo, hi, scratch):
            ref.ref_paged_attention_batch(p(out), p(q), p(k_host), p(v_host), p(tabs), int(tabs.shape[1]), p(ctx),
                                          lo, hi, p(scratch), g.block_size, g.head_dim, g.n_heads, g.n_kv_heads)
    elif port is not None:
        kind = "port"

        def run(lo, hi, scratch):
            for s in range(lo, hi):
                port.oracle_paged_attention(p(out[s]), p(q[s]), p(k_host), p(v_host), p(tabs[s]), p(scratch),
                                            int(ctx[s]), g.block_size, g.head_dim, g.n_heads, g.n_kv_heads)
    else:
        return None, None

    max_ctx = int(ctx.max())
    n = len(take)

    def timed(n_threads):
        scr = [np.empty(g.n_heads * max_ctx, np.float32) for _ in range(n_threads)]
        bounds = np.linspace(0, n, n_threads + 1).astype(int)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            if n_threads == 1:
                run(0, n, scr[0])
            else:
                with ThreadPoolExecutor(n_threads) as ex:
                    list(ex.map(lambda i: run(int(bounds[i]), int(bounds[i + 1]), scr[i]), range(n_threads)))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    sample = (f"{n} of {work.num_seqs} sequences of {work.name} ({int(ctx.sum())} context tokens), "
              f"best of 3 passes, -O3 -ffast-math -march=x86-64-v3")
    t1 = timed(1)
    one = {"value": round(sample_bytes / t1 / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": kind, "sample": sample,
           "ms_per_layer_call": round(t1 * 1e3 * work.num_seqs / n, 2)}
    many = None
    if cores > 1 and n >= 2:
        nt = min(cores, n)
        tn = timed(nt)
        many = {"value": round(sample_bytes / tn / 1e9, 3), "unit": "GB/s", "cores": nt, "kind": kind,
                "sample": sample, "ms_per_layer_call": round(tn * 1e3 * work.num_seqs / n, 2)}
    return one, many


def extra_workloads(nv, torch, dev, args):
    """The other BASELINE.json configs on one GPU, a few steps each (they are parity-test cases first; these
    lines make their rates visible to whoever runs the default bench): {name: {ms_per_step, GBps, frac, ...}}."""
    res = {}
    for name in ("config3", "config4", "5a", "5b"):
        work = make_workload(name)
        g = work.geom
        uniform_batch = bool((work.ctx_lens == work.ctx_lens[0]).all())
        flags = nv.AUTO | (nv.UNIFORM if uniform_batch else 0)
        with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim, device=dev.index) as pool:
            pool.fill_uniform(0, 1, 2)
            q = torch.from_numpy(work.q()).to(dev)
            tabs = torch.from_numpy(work.block_tables).to(dev)
            ctx = torch.from_numpy(work.ctx_lens).to(dev)
            out = torch.empty_like(q)
            torch.cuda.synchronize()
            kname = nv.decode_kernel_name(pool, work.num_seqs, g.n_heads, work.max_ctx, flags, 0)

            def run(warmup, iters, want_attn):
                return nv.decode_bench(pool, 0, work.num_seqs, g.n_heads, q, tabs, work.max_blocks, ctx, work.max_ctx,
                                       out, flags, 0, warmup, iters, replan_every=1, want_attn=want_attn)

            steps = 10
            est_ms, _ = run(3, 3, False)
            # the same clock pre-warm as the headline's (the GPU idled while the host built this workload's tables)
            run(min(2000, int(args.prewarm_ms / max(est_ms, 1e-3)) + 1), 0, False)
            step_ms, _ = run(0, steps, False)
            _, attn_ms = run(0, steps, True)
            nbytes = work.algorithmic_bytes()
            reused = None
            if not uniform_batch:
                # what an engine does: the plan depends on the lengths only, so it is built once per model step and
                # serves all of its layer-calls (nvk_plan_build + nvk_decode_planned); here once per 32 launches
                r_ms, _ = nv.decode_bench(pool, 0, work.num_seqs, g.n_heads, q, tabs, work.max_blocks, ctx, work.max_ctx,
                                          out, flags, 0, 2, 32, replan_every=32, want_attn=False)
                reused = {"replan_every": 32, "steps": 32, "ms_per_step": round(r_ms, 4),
                          "GBps": round(nbytes / (r_ms * 1e-3) / 1e9, 1),
                          "frac": round(nbytes / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
            res[name] = {"workload": work.name, "steps": steps, "ms_per_step": round(step_ms, 4),
                         "GBps": round(nbytes / (step_ms * 1e-3) / 1e9, 1),
                         "frac": round(nbytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "kernel": kname, "kernel_ms": round(attn_ms, 4),
                         "kernel_frac": round(nbytes / (attn_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "traffic": profiled_traffic(work.name, kname), "algorithmic_bytes": nbytes,
                         "plan": "none: uniform lengths" if uniform_batch else "nvk_plan_kernel every layer-call",
                         "tokens_per_s": round(work.num_seqs / (step_ms * 1e-3), 1)}
            if reused is not None:
                res[name]["plan_reused_over_32_layer_calls"] = reused
        del q, tabs, ctx, out
        torch.cuda.empty_cache()
    return res


class Watchdog:
    """Bounds the N>1 extras: if they are not finished after `seconds` (a rank stuck in a collective, say), rank 0
    prints the headline line with what the extras have so far and every rank's process ends with status 0."""

    def __init__(self, seconds, rank, line, partial):
        import threading

        self._lock = threading.Lock()
        self._state = "running"
        self._done = threading.Event()

        def dog():
            if self._done.wait(seconds):
                return
            with self._lock:
                if self._state != "running":
                    return
                self._state = "timed out"
            if line is not None:
                got = dict(partial)
                got["error"] = f"extras not finished after {seconds:.0f} s on rank {rank}: line printed without the rest"
                line["multi_gpu_workloads"] = got
                print(json.dumps(line), flush=True)
            os._exit(0)

        threading.Thread(target=dog, daemon=True).start()

    def finish(self):
        """True: the extras came back in time and the caller prints; False: the watchdog has taken over."""
        with self._lock:
            if self._state != "running":
                return False
            self._state = "finished"
        self._done.set()
        return True


def multi_gpu_workloads(nv, torch, dist, args, rank, world, local_rank, dev, red_dev, barrier, res):
    """BASELINE.json configs[4] across the ranks of this launch; fills `res` key by key (all ranks; rank 0's is printed).

    5a / 5b: rank r decodes sequences [512 r, 512 (r + 1)) of the 512 x world from a pool of its own: weak scaling, no
    collective, GB/s and tokens/s = all ranks' work over the slowest rank's wall time between two barriers.
    5b_head_sharded: ONE batch of 512 x world sequences, rank r holding kv-heads [r KV / world, (r + 1) KV / world) of
    every sequence (the same bytes per GPU as 5b), out joined by one all-gather per layer-call inside the timed region."""
    from nano_vllm_amd import sharding as sh
    from nano_vllm_amd import workloads as wl

    steps = 10

    def reduce_max(*vals):
        t = torch.tensor(list(vals), dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    def device_inputs(work):
        q = torch.from_numpy(work.q()).to(dev)
        tabs = torch.from_numpy(work.block_tables).to(dev)
        ctx = torch.from_numpy(work.ctx_lens).to(dev)
        return q, tabs, ctx, torch.empty_like(q)

    for name in ("5a", "5b"):
        work = make_workload(name, rank)
        g = work.geom
        uniform_batch = bool((work.ctx_lens == work.ctx_lens[0]).all())
        flags = nv.AUTO | (nv.UNIFORM if uniform_batch else 0)
        with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim, device=local_rank) as pool:
            pool.fill_uniform(0, 1 + 16 * rank, 2 + 16 * rank)
            q, tabs, ctx, out = device_inputs(work)
            torch.cuda.synchronize()

            def run(warmup, iters, want_attn):
                return nv.decode_bench(pool, 0, work.num_seqs, g.n_heads, q, tabs, work.max_blocks, ctx, work.max_ctx,
                                       out, flags, 0, warmup, iters, replan_every=1, want_attn=want_attn)

            est_ms, _ = run(3, 3, False)
            run(min(2000, int(args.prewarm_ms / max(est_ms, 1e-3)) + 1), 0, False)
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            run(0, steps, False)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            barrier()
            _, attn_ms = run(0, steps, True)
            # every rank's batch has its own lengths (5a): the job's bytes are the sum over ranks
            tot = torch.tensor([float(work.algorithmic_bytes()), float(work.num_seqs)], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            wall, attn_ms = reduce_max(wall, attn_ms)
            ms = wall * 1e3 / steps
            res[name] = {"workload": f"{work.name} per GPU x {world} GPUs, sequence-sharded, no collective",
                         "scaling": "weak", "steps": steps, "ms_per_step": round(ms, 4),
                         "GBps": round(float(tot[0]) / (ms * 1e-3) / 1e9, 1),
                         "frac_per_gpu": round(float(tot[0]) / world / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "slowest_kernel_ms": round(attn_ms, 4), "sequences": int(tot[1]),
                         "tokens_per_s": round(float(tot[1]) / (ms * 1e-3), 1)}
            del q, tabs, ctx, out
        torch.cuda.empty_cache()

    full = wl.config5b(512 * world)
    gf = full.geom
    if gf.n_kv_heads % world != 0:
        res["5b_head_sharded"] = {"skipped": f"{world} ranks do not divide {gf.n_kv_heads} kv-heads"}
        return
    hs = sh.HeadShard(rank, world, gf.n_heads, gf.n_kv_heads)
    g = wl.Geometry(hs.local_heads, hs.local_kv_heads, gf.head_dim, gf.block_size)
    work = wl.Workload(full.name + f"_heads{rank}of{world}", g, full.num_blocks, full.ctx_lens, full.block_tables, False)
    flags = nv.AUTO | nv.UNIFORM
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim, device=local_rank) as pool:
        pool.fill_uniform(0, 1 + 16 * rank, 2 + 16 * rank)
        q, tabs, ctx, out = device_inputs(work)
        torch.cuda.synchronize()
        wall, step_ms, attn_ms = head_sharded_leg(nv, torch, dist, hs, pool, work, q, tabs, ctx, out, flags, 0, steps, 3,
                                                  args.backend, rank, world, local_rank, dev, barrier)
        wall, step_ms, attn_ms = reduce_max(wall, step_ms, attn_ms)
        ms = wall * 1e3 / steps
        nbytes = work.algorithmic_bytes()  # this rank's heads of the whole batch
        shard_bytes = work.num_seqs * g.n_heads * g.head_dim * 4
        res["5b_head_sharded"] = {
            "workload": f"{full.name}: ONE batch, {g.n_kv_heads} of {gf.n_kv_heads} kv-heads per GPU x {world} GPUs",
            "scaling": "weak in bytes per GPU (the batch grows with the GPUs, the heads per GPU shrink)",
            "collective": ("ncclAllGather (RCCL) through nvk_shard_*, on its own stream under the next layer-call"
                           if args.backend == "nccl" else "gloo on host copies: rehearsal"),
            "gather_bytes_per_rank": {"sent": shard_bytes, "gathered": shard_bytes * world},
            "steps": steps, "ms_per_step": round(ms, 4), "event_ms_per_step": round(step_ms, 4),
            "GBps": round(nbytes * world / (ms * 1e-3) / 1e9, 1),
            "frac_per_gpu": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "slowest_kernel_ms": round(attn_ms, 4), "sequences": work.num_seqs,
            "tokens_per_s": round(work.num_seqs / (ms * 1e-3), 1)}
        del q, tabs, ctx, out
    torch.cuda.empty_cache()


def head_sharded_leg(nv, torch, dist, hs, pool, work, q, tabs, ctx, out, flags, chunk, steps, warmup, backend, rank, world,
                     local_rank, dev, barrier):
    """K x (attention of this rank's heads, then ONE all-gather of out): (wall s, event ms per step, kernel ms).

    With RCCL the gather goes through the library's own C ABI (nvk_shard_*: ncclAllGather called directly, on a
    communication stream behind an event) and is overlapped: layer-call i's gather runs under layer-call i + 1's
    attention, two (out, gathered) buffer pairs in turn, the consumer of gathered[i] waiting one step behind -- as the
    wo projection of a model's layer would.  With gloo (rehearsal: ranks share a GPU, which RCCL refuses) the
    collective runs on host copies through torch.distributed, not overlapped."""
    from nano_vllm_amd import sharding as sh

    g = work.geom
    side = torch.cuda.Stream(device=dev)
    prev_stream = torch.cuda.current_stream(dev)
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    assert stream != 0  # (the C ABI reads a NULL stream as "the pool's own stream")
    outs = [out, torch.empty_like(out)]
    gath = [torch.empty((world,) + tuple(out.shape), dtype=out.dtype, device=dev) for _ in range(2)]
    shard = None
    try:
        if backend == "nccl":
            ids = [sh.NvkShard.new_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0)
            shard = sh.NvkShard(rank, world, ids[0], local_rank, hs.n_heads, hs.n_kv_heads)

        def step(i):
            b = i % 2
            if shard is not None:
                shard.decode_gather(pool, 0, work.num_seqs, q, tabs, work.max_blocks, ctx, work.max_ctx, outs[b],
                                    gath[b], b, flags, chunk, compute_stream=stream)
                if i > 0:
                    shard.wait_gather(1 - b, stream)  # the consumer of the previous layer-call's rows
            else:
                nv.decode_device(pool, 0, work.num_seqs, g.n_heads, q, tabs, work.max_blocks, ctx, work.max_ctx,
                                 outs[b], flags, chunk, stream=stream)
                gath[b].copy_(hs.all_gather_out(outs[b].cpu()))

        def drain(i):
            if shard is not None and i > 0:
                shard.wait_gather((i - 1) % 2, stream)

        for i in range(warmup):
            step(i)
        drain(warmup)
        torch.cuda.synchronize()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(steps):
            step(i)
        drain(steps)
        ev1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        barrier()
        step_ms = ev0.elapsed_time(ev1) / steps
        # the attention kernel alone in this rank's shape, HIP events on the pool's stream
        _, attn_ms = nv.decode_bench(pool, 0, work.num_seqs, g.n_heads, q, tabs, work.max_blocks, ctx, work.max_ctx, out,
                                     flags, chunk, 2, 16, replan_every=1, want_attn=True, rotate_layers=1)
        torch.cuda.synchronize()
    finally:
        if shard is not None:
            shard.close()
        torch.cuda.set_stream(prev_stream)
    return wall, step_ms, attn_ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--geom", default="llama", choices=sorted(GEOMS), help="diagnostic: head geometry of the workload")
    ap.add_argument("--shard", default="seq", choices=["seq", "heads"],
                    help="seq: a private batch per GPU, no collective (default); heads: kv-head groups per GPU, "
                         "one RCCL all-gather of out per layer-call")
    ap.add_argument("--mode", default="auto", choices=["auto", "single", "split"])
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--replan", type=int, default=1,
                    help="rebuild the decode plan every N layer-calls inside the timed region "
                         "(1 = every call; 32 = as for the 32 layers of one model step)")
    ap.add_argument("--plan", default="auto", choices=["auto", "kernel", "none"],
                    help="auto: a batch whose sequences all have the same length is launched with NVK_DECODE_UNIFORM "
                         "(no plan kernel: its order is the natural one), a ragged batch with the plan kernel; "
                         "kernel: always build the plan; none: never (NVK_DECODE_NO_PLAN for ragged batches)")
    ap.add_argument("--merge", default="auto", choices=["auto", "kernel", "separate"],
                    help="split-K combine: the library's choice, inside the decode kernel, or nvk_merge_kernel")
    ap.add_argument("--strong", action="store_true",
                    help="sequence-sharded strong scaling: ONE batch of the workload's shape is split over the ranks "
                         "by context tokens (sharding.partition_sequences) instead of one batch per rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1: nccl (= RCCL, the real one) or gloo (rehearsal of the "
                         "multi-process path on a box with fewer GPUs than ranks; ranks then share GPUs)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--rotate", type=int, default=4,
                    help="N=1 only: also time the batch over this many different KV slabs in turn (cold rotation, "
                         "SURVEY.md 8d) and report it beside the resident number; 1 = off")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed launches of the same step for about this long before the W warm-up steps, so that the "
                         "clocks have reached their steady state when the timed region starts (0 = off)")
    ap.add_argument("--no-pairs", action="store_true",
                    help="diagnostic: rows of 64 / 192 B one kv-head per wave (NVK_DECODE_NO_PAIRS)")
    ap.add_argument("--free-running", action="store_true",
                    help="diagnostic: the vector kernel without the lockstep barrier and with 8-token steps "
                         "(NVK_DECODE_FREE_RUNNING)")
    ap.add_argument("--no-extra", action="store_true",
                    help="N=1 only: skip the extra BASELINE configs (3, 4, 5a, 5b) timed after the headline workload")
    ap.add_argument("--extra-timeout", type=float, default=150.0,
                    help="N>1 only: seconds the configs[4] extras (5a / 5b sequence-sharded, 5b head-sharded) may take "
                         "before the headline line is printed without them")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="diagnostic: no per-kernel event pairs inside the timed region (roofline from step time)")
    args = ap.parse_args()

    import torch
    import nano_vllm_amd as nv

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if nv.device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()  # rehearsal: more ranks than GPUs
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")  # where the timing reduction lives

    work = make_workload(args.workload, 0 if args.strong else rank, args.geom)
    g = work.geom
    whole_bytes = None
    if args.strong and args.shard == "seq" and world > 1:
        # this rank's share of the one batch: its own pool and tables (block ids are rank-local), nothing exchanged
        from nano_vllm_amd import sharding as sh
        from nano_vllm_amd import workloads as wl

        whole_bytes = work.algorithmic_bytes()
        mine = sh.partition_sequences(work.ctx_lens, world)[rank]
        work = wl.fragmented_workload(work.name + f"_part{rank}of{world}", work.ctx_lens[mine], g) if work.fragmented \
            else wl.Workload(work.name + f"_part{rank}of{world}", g, work.num_blocks, work.ctx_lens[mine],
                             work.block_tables[mine], False)
    flags = {"auto": nv.AUTO, "single": nv.SINGLE_PASS, "split": nv.SPLIT_K}[args.mode]
    flags |= {"auto": 0, "kernel": nv.MERGE_IN_KERNEL, "separate": nv.MERGE_SEPARATE}[args.merge]
    if args.no_pairs:
        flags |= nv.NO_PAIRS
    if args.free_running:
        flags |= nv.FREE_RUNNING
    uniform_batch = bool((work.ctx_lens == work.ctx_lens[0]).all())
    if args.plan == "none" or (args.plan == "auto" and uniform_batch):
        flags |= nv.UNIFORM if uniform_batch else nv.NO_PLAN
    planning = ("none: uniform lengths, natural chunk order" if flags & nv.UNIFORM else
                "none: lengths read by the decode kernel, natural chunk order" if flags & nv.NO_PLAN else
                f"nvk_plan_kernel, rebuilt every {args.replan} layer-call(s)")

    def barrier():
        if world > 1:
            dist.barrier()

    if args.shard == "heads":
        # head-group variant: this rank's pool holds n_kv_heads/world kv-heads of every sequence
        from nano_vllm_amd import sharding as sh
        from nano_vllm_amd import workloads as wl

        hs = sh.HeadShard(rank, world, g.n_heads, g.n_kv_heads)
        g = wl.Geometry(hs.local_heads, hs.local_kv_heads, g.head_dim, g.block_size)
        work = wl.Workload(work.name + f"_heads{rank}of{world}", g, work.num_blocks, work.ctx_lens, work.block_tables,
                           work.fragmented)
    rotate = max(1, args.rotate) if (args.shard == "seq" and world == 1) else 1
    pool = nv.PagedKVPool(rotate, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim, device=local_rank)
    for layer in range(rotate):
        pool.fill_uniform(layer, 1 + 16 * rank + 2 * layer, 2 + 16 * rank + 2 * layer)
    q = torch.from_numpy(work.q()).to(dev)
    tabs = torch.from_numpy(work.block_tables).to(dev)
    ctx = torch.from_numpy(work.ctx_lens).to(dev)
    out = torch.empty_like(q)
    torch.cuda.synchronize()
    kernel_name = nv.decode_kernel_name(pool, work.num_seqs, g.n_heads, work.max_ctx, flags, args.chunk)

    def run(warmup, iters, want_attn, layers=1):
        return nv.decode_bench(pool, 0, work.num_seqs, g.n_heads, q, tabs, work.max_blocks, ctx, work.max_ctx, out,
                               flags, args.chunk, warmup, iters, replan_every=args.replan, want_attn=want_attn,
                               rotate_layers=layers)

    rotated = None
    if args.shard == "seq":
        if args.prewarm_ms > 0:  # (sizes the split-K workspace too)
            t_end = time.perf_counter() + args.prewarm_ms * 1e-3
            while time.perf_counter() < t_end:
                run(20, 0, False)
        run(args.warmup, 0, False)  # W untimed steps
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        step_ms, _ = run(0, args.steps, False)  # EXACTLY K steps, nothing but the path's own launches on the stream
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        barrier()
        # the dominant kernel's own duration: the same K launches again, each between a pair of HIP events on the
        # launch stream (a pair costs the stream ~6 us, which is why it is not inside the region `value` comes from)
        if args.no_kernel_events:
            attn_ms = step_ms
        else:
            _, attn_ms = run(0, args.steps, True)
        if rotate > 1:  # the same batch over `rotate` different slabs in turn: nothing can be served from the Infinity Cache
            run(rotate, 0, False, rotate)
            r_step, _ = run(0, args.steps, False, rotate)
            _, r_attn = run(0, args.steps, True, rotate)
            rotated = (r_step, r_attn)
        torch.cuda.synchronize()
    else:
        wall, step_ms, attn_ms = head_sharded_leg(nv, torch, dist if world > 1 else None, hs, pool, work, q, tabs, ctx, out,
                                                  flags, args.chunk, args.steps, args.warmup, args.backend, rank, world,
                                                  local_rank, dev, barrier)

    if world > 1:
        t = torch.tensor([wall, step_ms, attn_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, step_ms, attn_ms = (float(x) for x in t.tolist())

    line = None
    if rank == 0:
        nbytes = work.algorithmic_bytes()  # per rank, per layer-call
        ms_per_step = wall * 1e3 / args.steps
        job_bytes = whole_bytes if whole_bytes is not None else nbytes * world
        value = job_bytes / (ms_per_step * 1e-3) / 1e9
        achieved = nbytes / (attn_ms * 1e-3) / 1e9
        seqs_all = work.num_seqs * (world if args.shard == "seq" else 1)  # head shards all hold the same batch
        if whole_bytes is not None:
            seqs_all = make_workload(args.workload, 0, args.geom).num_seqs
        line = {
            "metric": "paged_attn_decode_achieved_hbm_GBps",
            "value": round(value, 1),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak" if (args.shard == "seq" and not args.strong) else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": work.name, "num_seqs_per_gpu": work.num_seqs, "total_ctx_tokens_per_gpu":
                       work.total_tokens, "n_heads": g.n_heads, "n_kv_heads": g.n_kv_heads, "head_dim": g.head_dim,
                       "block_size": g.block_size, "fragmented_tables": work.fragmented, "mode": args.mode, "replan_every": args.replan, "plan": planning,
                       "merge": args.merge, "prewarm_ms": args.prewarm_ms,
                       "parallelism": (f"sequence-sharded x{world}, no collective" if args.shard == "seq" else
                                       f"head-sharded x{world}, one all-gather of out per layer-call ("
                                       + ("RCCL via nvk_shard_*, overlapped with the next layer-call" if args.backend == "nccl"
                                          else "gloo on host copies: rehearsal") + ")")},
            "pct_of_hbm_peak": round(100.0 * value / world / HBM_PEAK_GBPS, 2),
            "tokens_per_s": round(seqs_all / (ms_per_step * 1e-3), 1),
            "tokens_per_s_32_layer_model": round(seqs_all / (ms_per_step * 1e-3) / 32.0, 1),
            "event_ms_per_step": round(step_ms, 4),
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": profiled_traffic(work.name, kernel_name) if args.mode == "auto" and args.chunk == 0
                         else None,
                         "kernel_ms": round(attn_ms, 4), "algorithmic_bytes": nbytes,
                         "kernel_events": ("none: step time" if args.no_kernel_events else
                                           f"HIP event pair around each of {args.steps} launches on the launch stream, "
                                           "a second pass right after the timed region")},
        }
        if rotated is not None:
            r_step, r_attn = rotated
            line["rotated"] = {"slabs": rotate, "slab_bytes": int(2 * 4 * work.num_blocks * g.block_size * g.n_kv_heads * g.head_dim),
                               "ms_per_step": round(r_step, 4), "GBps": round(nbytes / (r_step * 1e-3) / 1e9, 1),
                               "kernel_ms": round(r_attn, 4),
                               "kernel_frac": round(nbytes / (r_attn * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        if world == 1 and args.shard == "seq" and not args.no_extra and args.workload == "config2" and args.geom == "llama":
            pool.close()
            pool = None
            line["workloads"] = extra_workloads(nv, torch, dev, args)
        if world == 1 and not args.no_cpu:
            cores = min(16, os.cpu_count() or 1)
            one, many = cpu_baseline(work, cores)
            if one is not None:
                line["cpu_baseline"] = one
            if many is not None:
                line["cpu_baseline_all_cores"] = many

    # N > 1, default launch: BASELINE.json configs[4] while all ranks are here -- 5a and 5b sequence-sharded (rank r's
    # 512 of the 512 x N sequences, no collective) and 5b's batch head-sharded (every rank a kv-head group of ALL
    # 512 x N sequences, one RCCL all-gather of out per layer-call through nvk_shard_*).  Extra keys of the same line;
    # a watchdog prints the headline without them if a rank fails to come back.
    multi = None
    if (world > 1 and args.shard == "seq" and not args.strong and not args.no_extra and args.workload == "config2"
            and args.geom == "llama"):
        pool.close()
        pool = None
        del q, tabs, ctx, out
        torch.cuda.empty_cache()
        multi = {}
        dog = Watchdog(args.extra_timeout, rank, line if rank == 0 else None, multi)
        try:
            multi_gpu_workloads(nv, torch, dist, args, rank, world, local_rank, dev, red_dev, barrier, multi)
        except Exception as exc:  # noqa: BLE001 -- the headline must still be printed
            multi["error"] = f"{type(exc).__name__}: {exc}"[:300]
        if not dog.finish():
            return  # the watchdog is printing (or has printed) the line and ends the process
        if rank == 0:
            line["multi_gpu_workloads"] = multi
    if rank == 0:
        print(json.dumps(line), flush=True)

    if pool is not None:
        pool.close()
    if world > 1:
        sys.stdout.flush()
        if multi is not None and "error" in multi:
            os._exit(0)  # ranks may have left the extras at different points: no further collective is safe
        dog = Watchdog(20.0, rank, None, {})  # (a peer that left that way never answers the shutdown handshake)
        dist.destroy_process_group()
        dog.finish()


if __name__ == "__main__":
    main()
