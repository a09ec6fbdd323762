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
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


GEOMS = {"llama": (32, 8, 128, 16), "mha": (32, 32, 128, 16), "tiny": (6, 6, 48, 16), "shard2": (16, 4, 128, 16),
         "shard8": (4, 1, 128, 16), "d64": (32, 8, 64, 16), "d256": (16, 4, 256, 16), "gqa8": (32, 4, 128, 16),
         "kvmul3": (24, 8, 128, 16), "mha64": (64, 64, 128, 16), "tinyllama": (32, 4, 64, 16),
         "gqa8x8": (64, 8, 128, 16), "shard4": (8, 2, 128, 16)}


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
    """(HBM bytes per launch of `kernel`, the committed profile it was read from): the instantiation this run launched,
    as rocprofv3 names it, looked up in the PMC profile of this same workload (profiles/r*_pmc_traffic.json: separate
    FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections applied as MI355X_MICROARCH.md prescribes).  NOT measured in
    this run -- counters need rocprofv3 around the process -- which is what `traffic_source` in the line says.  The
    latest round wins; (None, None) when no profile holds this (workload, kernel) pair -- never a number measured on
    another kernel."""
    import glob
    best, where = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):
        try:
            with open(path) as fh:
                prof = json.load(fh)
            if prof.get("workload") == work_name and kernel in prof.get("kernels", {}):
                got = prof["kernels"][kernel].get("hbm_bytes_per_launch_corrected")
                if got is not None:
                    best, where = got, "profiles/" + os.path.basename(path)
        except (OSError, ValueError):
            continue
    return best, where


def self_check(work, out_host, seed_k, seed_v, seqs):
    """The checker's leg of the line: rows `seqs` of `out` (what the LAST timed launch left in HBM) against the CPU
    restatement of Attention::paged_attention (oracle/, include/core/attention.hpp:82-159) on the same pool values,
    at the path's tolerance 1e-4 + 1e-5 |ref|.  Test infrastructure used as the checker only; None when the
    checker's library did not travel."""
    try:
        import oracle_lib as ol
    except Exception:  # noqa: BLE001
        return None
    if ol.ORACLE is None:
        return None
    from nano_vllm_amd import workloads as wl

    g = work.geom
    seqs = sorted({int(s) for s in seqs})
    k_host, v_host, tabs, ctx = wl.host_blocks_for(work, seqs, seed_k, seed_v)
    want = ol.oracle_decode_batch(work.q()[seqs], k_host, v_host, tabs, ctx, g.block_size, g.n_kv_heads)
    got = np.asarray(out_host)[seqs]
    err = np.abs(got - want)
    return {"seqs": seqs, "ctx": [int(c) for c in ctx], "max_abs_err": float(err.max()),
            "tolerance": "1e-4 + 1e-5*|ref|", "ok": bool((err <= 1e-4 + 1e-5 * np.abs(want)).all()),
            "against": "oracle/nvk_oracle.c (CPU restatement of attention.hpp:82-159)"}


def check_picks(work):
    """Two sequences of a batch for self_check: the longest and one from the other end of the batch."""
    longest = int(np.argmax(work.ctx_lens))
    other = 0 if longest != 0 else work.num_seqs - 1
    return [longest, other]


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


EXTRA_STEPS = 20    # layer-calls per timed region of an extra workload
EXTRA_REGIONS = 5   # regions per extra workload: ms_per_step = their median, ms_per_step_min = the fastest
EXTRA_WARMUP = 5    # untimed layer-calls in front of every region (the default run's --warmup)


def extra_workloads(nv, torch, dev, args):
    """The other BASELINE.json configs on one GPU (they are parity-test cases first; these lines make their rates
    visible to whoever runs the default bench): {name: {ms_per_step, GBps, frac, ...}}.  EXTRA_REGIONS timed regions
    of EXTRA_STEPS layer-calls each: the median region is the number, the fastest is beside it (box-to-box and
    region-to-region spread is 1-4 % on these)."""
    res = {}
    names = ("config3", "config4", "5a", "5b", "config2_head_shard_rank")
    if os.environ.get("NVK_BENCH_EXTRAS"):  # diagnostic: a subset, in the order given
        names = tuple(n for n in os.environ["NVK_BENCH_EXTRAS"].split(",") if n in names)
    for name in names:
        # (the last one: what each GPU of an 8-way head-sharded config 2 runs -- 4 q-heads / 1 kv-head of every sequence)
        work = make_workload("config2", 0, "shard8") if name == "config2_head_shard_rank" else make_workload(name)
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

            def run(warmup, iters, want_attn, replan=1):
                return nv.decode_bench(pool, 0, work.num_seqs, g.n_heads, q, tabs, work.max_blocks, ctx, work.max_ctx,
                                       out, flags, 0, warmup, iters, replan_every=replan, want_attn=want_attn)

            est_ms, _ = run(3, 3, False)
            # the same clock pre-warm as the headline's (the GPU idled while the host built this workload's tables)
            run(min(2000, int(args.prewarm_ms / max(est_ms, 1e-3)) + 1), 0, False)
            # (every region with EXTRA_WARMUP untimed layer-calls in front of it inside the same call, as the headline's W
            # warmup steps are: between two regions the host synchronises and the GPU idles)
            regions = [run(EXTRA_WARMUP, EXTRA_STEPS, False)[0] for _ in range(EXTRA_REGIONS)]
            step_ms, step_min = statistics.median(regions), min(regions)
            kregions = [run(EXTRA_WARMUP, EXTRA_STEPS, True)[1] for _ in range(EXTRA_REGIONS)]
            attn_ms = statistics.median(kregions)
            out.fill_(float("nan"))
            run(0, 1, False)
            checked = self_check(work, out.cpu().numpy(), 1, 2, check_picks(work))
            nbytes = work.algorithmic_bytes()
            frac = lambda ms: round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            reused = None
            if not uniform_batch:
                # what an engine does: the plan depends on the lengths only, so it is built once per model step and
                # serves all of its layer-calls (nvk_plan_build + nvk_decode_planned); here once per 32 launches
                rr = [run(32, 32, False, 32)[0] for i in range(EXTRA_REGIONS)]
                r_ms = statistics.median(rr)
                reused = {"replan_every": 32, "steps": 32, "regions": EXTRA_REGIONS, "ms_per_step": round(r_ms, 4),
                          "ms_per_step_min": round(min(rr), 4),
                          "GBps": round(nbytes / (r_ms * 1e-3) / 1e9, 1), "frac": frac(r_ms)}
            # the same launches as a model step would replay them: one plan, 32 layer-calls in ONE hipGraph
            g_ms = nv.decode_bench_graph(pool, 0, work.num_seqs, g.n_heads, q, tabs, work.max_blocks, ctx, work.max_ctx,
                                         out, flags, 0, layers_per_graph=32, warmup=2, iters=5)
            graph = {"layers_per_graph": 32, "launches": 5, "ms_per_step": round(g_ms, 4), "frac": frac(g_ms)}
            traffic, traffic_source = profiled_traffic(work.name, kname)
            res[name] = {"workload": work.name, "steps": EXTRA_STEPS, "regions": EXTRA_REGIONS,
                         "warmup_per_region": EXTRA_WARMUP,
                         "ms_per_step": round(step_ms, 4), "ms_per_step_min": round(step_min, 4),
                         "GBps": round(nbytes / (step_ms * 1e-3) / 1e9, 1), "frac": frac(step_ms),
                         "frac_best_region": frac(step_min),
                         "kernel": kname, "kernel_ms": round(attn_ms, 4), "kernel_ms_min": round(min(kregions), 4),
                         "kernel_frac": frac(attn_ms),
                         "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes": nbytes,
                         "plan": "none: uniform lengths" if uniform_batch else "nvk_plan_kernel every layer-call",
                         "tokens_per_s": round(work.num_seqs / (step_ms * 1e-3), 1), "checked": checked,
                         "graph_replay_of_32_layer_calls": graph}
            if reused is not None:
                res[name]["plan_reused_over_32_layer_calls"] = reused
        del q, tabs, ctx, out
        torch.cuda.empty_cache()
        settle(torch)
    return res


def settle(torch, seconds=0.6):
    """Between two workloads: the pool just freed was gigabytes, and for a few hundred milliseconds after a release of
    that size the next workload's launches measure 3-4 % slower (config 4 right after config 3 or 5b: 0.859 of the peak
    for the first ~150 ms of launches, 0.892 after that and when it runs first -- profiles/r03_extras_order.log).
    The wait keeps one workload's clean-up out of the next one's number."""
    torch.cuda.synchronize()
    time.sleep(seconds)


MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz)


class HipEvents:
    """hipEvent timing on a named stream through ctypes (torch.cuda.Event only sees torch's current stream; the
    library launches on the pool's own stream or on the device's null stream)."""

    def __init__(self):
        hip = C.CDLL("libamdhip64.so")
        hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        hip.hipEventSynchronize.argtypes = [C.c_void_p]
        hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        hip.hipEventDestroy.argtypes = [C.c_void_p]
        self.hip = hip
        self.e0, self.e1 = C.c_void_p(), C.c_void_p()
        assert hip.hipEventCreate(C.byref(self.e0)) == 0 and hip.hipEventCreate(C.byref(self.e1)) == 0

    def ms_per_call(self, stream, fn, iters):
        """iters back-to-back calls of fn() between one event pair on `stream` (an address, 0 = the null stream)."""
        s = C.c_void_p(stream or None)
        assert self.hip.hipEventRecord(self.e0, s) == 0
        for _ in range(iters):
            fn()
        assert self.hip.hipEventRecord(self.e1, s) == 0
        assert self.hip.hipEventSynchronize(self.e1) == 0
        ms = C.c_float(0.0)
        assert self.hip.hipEventElapsedTime(C.byref(ms), self.e0, self.e1) == 0
        return float(ms.value) / iters

    def regions(self, stream, fn, iters, n):
        got = [self.ms_per_call(stream, fn, iters) for _ in range(n)]
        return statistics.median(got), min(got)

    def close(self):
        self.hip.hipEventDestroy(self.e0)
        self.hip.hipEventDestroy(self.e1)


def _tol_ok(got, want):
    err = np.abs(got - want)
    return float(err.max()), bool((err <= 1e-4 + 1e-5 * np.abs(want)).all())


def prefill_row(nv, torch, dev, ev, args):
    """SURVEY.md 8f row f3 on the record: paged chunked prefill, 2 prompts x 8192 tokens from position 0, causal,
    G-llama (ref: experimental/chunked_prefill/model_chunked.hpp:167-224).  Roofline: fp32 MFMA; algorithmic flops
    4 x keys seen x n_heads x head_dim per query token (QK^T + PV)."""
    from nano_vllm_amd import workloads as wl

    g = wl.G_LLAMA
    seqs, chunk = 2, 8192
    work = wl.uniform_workload("prefill_2x8192", seqs, chunk, g)
    total = seqs * chunk
    starts, lens = np.zeros(seqs, np.int32), np.full(seqs, chunk, np.int32)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim, device=dev.index) as pool:
        pool.fill_uniform(0, 1, 2)
        gen = torch.Generator(device=dev).manual_seed(9)
        q = torch.rand((total, g.n_heads, g.head_dim), generator=gen, device=dev) * 2 - 1
        out = torch.full_like(q, float("nan"))
        tabs = torch.from_numpy(work.block_tables).to(dev)
        torch.cuda.synchronize()
        run = lambda: nv.prefill_device(pool, 0, g.n_heads, q, tabs, seqs, work.max_blocks, starts, lens, out,
                                        nv.PREFILL_TILED)
        run()
        run()
        pool.synchronize()
        ms, ms_min = ev.regions(pool.stream, run, 4, 3)
        pool.synchronize()
        checked = None
        try:
            import oracle_lib as ol
            if ol.ORACLE is not None:
                rows, worst, ok = [(1, chunk - 1), (0, 77)], 0.0, True
                for s, b in rows:
                    k_host, v_host, tab, _ = wl.host_blocks_for(work, [s])
                    want = ol.oracle_paged_attention(q[s * chunk + b].cpu().numpy(), k_host, v_host, tab[0], b + 1,
                                                     g.block_size, g.head_dim, g.n_heads, g.n_kv_heads)
                    e, good = _tol_ok(out[s * chunk + b].cpu().numpy(), want)
                    worst, ok = max(worst, e), ok and good
                checked = {"rows": [f"seq {s} query {b}" for s, b in rows], "max_abs_err": worst, "ok": ok,
                           "tolerance": "1e-4 + 1e-5*|ref|",
                           "against": "oracle/nvk_oracle.c paged_attention with num_tokens = query + 1"}
        except ImportError:
            pass
    keys = seqs * chunk * (chunk + 1) // 2
    flops = 4.0 * keys * g.n_heads * g.head_dim
    del q, out, tabs
    torch.cuda.empty_cache()
    return {"workload": "paged chunked prefill, 2 prompts x 8192 tokens, causal, 32/8/128, TILED form",
            "ref": "experimental/chunked_prefill/model_chunked.hpp:167-224", "bound": "mfma",
            "kernel": "nvk::nvk_prefill_kernel<128>", "calls_per_region": 4, "regions": 3,
            "kernel_ms": round(ms, 4), "kernel_ms_min": round(ms_min, 4), "algorithmic_flops": flops,
            "TFLOPs": round(flops / (ms * 1e-3) / 1e12, 2), "peak": MFMA_F32_PEAK_TFLOPS,
            "frac": round(flops / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
            "query_tokens_per_s": round(total / (ms * 1e-3), 1), "checked": checked}


def projection_rows(nv, torch, dev, ev, args):
    """SURVEY.md 8f row f1b on the record: the q | k | v projections of a layer's head (ref: include/core/model.hpp:
    164-166 = Ops::matmul, include/ops/linear.hpp:12-22; BatchOps::batch_matmul, batch_ops.hpp:8-23) at dim 4096 ->
    4096 + 2 x 1024, ONE launch for the three, for 1 / 16 / 512 activation rows.  Few rows stream the weights once:
    HBM roofline, algorithmic bytes = the weights; 512 rows: fp32 MFMA roofline, 2 x rows x out x in flops."""
    D, QD, KD = 4096, 4096, 1024
    gen = torch.Generator(device=dev).manual_seed(5)
    sc = 1.0 / float(np.sqrt(D))
    wq, wk, wv = ((torch.rand((n, D), generator=gen, device=dev) * 2 - 1) * sc for n in (QD, KD, KD))
    wbytes = (QD + 2 * KD) * D * 4
    w_host = None
    res = {}
    for rows in (1, 16, 512):
        x = torch.rand((rows, D), generator=gen, device=dev) * 2 - 1
        q, k, v = (torch.full((rows, n), float("nan"), device=dev) for n in (QD, KD, KD))
        torch.cuda.synchronize()
        run = lambda: nv.qkv_project_device(q, k, v, x, None, None, wq, wk, wv, rows, D, QD, KD, device=dev.index)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        ms, ms_min = ev.regions(0, run, 30, 5)
        torch.cuda.synchronize()
        checked = None
        try:
            import oracle_lib as ol
            if ol.ORACLE is not None:
                if w_host is None:
                    w_host = [w.cpu().numpy() for w in (wq, wk, wv)]
                pick = sorted({0, rows - 1})
                xh = x[pick].cpu().numpy()
                worst, ok = 0.0, True
                for got, w in zip((q, k, v), w_host):
                    e, good = _tol_ok(got[pick].cpu().numpy(), ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, xh, w))
                    worst, ok = max(worst, e), ok and good
                checked = {"rows": pick, "max_abs_err": worst, "ok": ok, "tolerance": "1e-4 + 1e-5*|ref|",
                           "against": "oracle_batch_matmul (pinned bit-exact to Ops::matmul, linear.hpp:12-22)"}
        except ImportError:
            pass
        flops = 2.0 * rows * (QD + 2 * KD) * D
        row = {"workload": f"q|k|v projection, {rows} row(s), 4096 -> 4096 + 2 x 1024, one launch",
               "ref": "include/core/model.hpp:164-166, include/ops/linear.hpp:12-22", "calls_per_region": 30,
               "regions": 5, "ms_per_call": round(ms, 5), "ms_per_call_min": round(ms_min, 5), "checked": checked}
        if rows <= 32:
            row.update({"bound": "hbm", "kernel": ("nvk::nvk_gemv_kernel<1, 4, 2, false, false, false, false>" if rows <= 2 else
                                                   "nvk::nvk_gemm_skinny16_kernel<2, 0>"),
                        "algorithmic_bytes": wbytes, "GBps": round(wbytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                        "frac": round(wbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)})
        else:
            row.update({"bound": "mfma", "kernel": "nvk::nvk_gemm_nt_kernel<64, 64, 32, 0, 2>", "algorithmic_flops": flops,
                        "TFLOPs": round(flops / (ms * 1e-3) / 1e12, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                        "frac": round(flops / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)})
        res[f"qkv_{rows}row" if rows == 1 else f"qkv_{rows}rows"] = row
        del x, q, k, v
    del wq, wk, wv
    torch.cuda.empty_cache()
    return res


def model_step_rows(nv, torch, dev, args):
    """SURVEY.md 8f row f4 / the caller of the whole path on the record: LlamaModel::forward (ref: include/core/
    model.hpp:150-247) as device-resident batched decode steps, TinyLlama geometry (dim 2048, hidden 5632, 22 layers,
    32/4 heads x 64, vocab 32000; 3.9 GB of fp32 weights, synthetic values in the llama2.c file format), 1 and 64
    requests at ~1024 context tokens.  A step = embedding .. greedy argmax for every request, wall clock (the call is
    synchronous: token ids in, next tokens out).  HBM roofline: every layer weight + the classifier once per step,
    plus the requests' K/V."""
    import struct

    D, Hd, L, H, KV, V, S = 2048, 5632, 22, 32, 4, 32000, 2048
    hd = D // H
    sizes = [V * D, L * D, L * D * H * hd, L * D * KV * hd, L * D * KV * hd, L * H * hd * D, L * D, L * D * Hd,
             L * Hd * D, L * D * Hd, D]  # model.hpp:322-347's order; no separate classifier: shared with the embedding
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = os.path.join(tmpdir, f"nvk_bench_tinyllama_{os.getpid()}.bin")
    rng = np.random.default_rng(1)
    tile = rng.random(1 << 22, dtype=np.float32) - np.float32(0.5)
    try:
        with open(path, "wb") as fh:
            fh.write(struct.pack("7i", D, Hd, L, H, KV, V, S))
            for i, n in enumerate(sizes):
                norm = i in (1, 6, 10)
                w = (1.0 + 0.1 * tile) if norm else tile * np.float32(2.0 / np.sqrt(D) if i else 2.0)
                buf = w.astype(np.float32).tobytes()
                left = n * 4
                while left:
                    m = min(left, len(buf))
                    fh.write(buf[:m])
                    left -= m
        ctx, steps, bs = 1024, 16, 16
        blocks = 64 * ((ctx + 2 * steps + bs) // bs + 1)
        res = {}
        with nv.LlamaModel(path, bs, blocks, max_seqs=64, device=dev.index) as m:
            os.unlink(path)  # resident in HBM now
            for n_req, linked in ((1, False), (1, True), (64, False)):
                m.use_linked_norms(linked)
                ids = list(range(n_req))
                for r in ids:
                    m.add_request(r)
                fed = 0
                while fed < ctx:
                    c = min(512, ctx - fed)
                    nxt, _ = m.step_chunks(ids, [rng.integers(3, V, size=c).tolist() for _ in ids])
                    fed += c
                for _ in range(3):
                    nxt, _ = m.step(ids, nxt.tolist())
                times = []
                for _ in range(steps):
                    t0 = time.perf_counter()
                    nxt, _ = m.step(ids, nxt.tolist())
                    times.append(time.perf_counter() - t0)
                for r in ids:
                    m.finish_request(r)
                dt = statistics.median(times)
                weight_bytes = 4 * (sum(sizes) - V * D) + 4 * V * D  # layers + final norm, and the classifier (= embedding rows)
                kv_bytes = int(n_req * L * 2 * (ctx + 3 + steps / 2) * KV * hd * 4)
                res[f"model_step_tinyllama_b{n_req}" + ("_linked_norms" if linked else "")] = {
                    "workload": f"LlamaModel::forward as one batched decode step, TinyLlama geometry, {n_req} request(s) "
                                f"at ~{ctx} context tokens, greedy"
                                + ("; nvk_model_use_linked_norms(1): the norms carried between the GEMVs, 6 launches per "
                                   "layer instead of 8 (opt-in: a few ulp from the separate norm kernels)" if linked else ""),
                    "ref": "include/core/model.hpp:150-247", "bound": "hbm", "steps": steps,
                    "ms_per_step": round(dt * 1e3, 4), "ms_per_step_min": round(min(times) * 1e3, 4),
                    "tokens_per_s": round(n_req / dt, 1),
                    "algorithmic_bytes": weight_bytes + kv_bytes, "weight_bytes": weight_bytes, "kv_bytes": kv_bytes,
                    "GBps": round((weight_bytes + kv_bytes) / dt / 1e9, 1), "peak": HBM_PEAK_GBPS,
                    "frac": round((weight_bytes + kv_bytes) / dt / 1e9 / HBM_PEAK_GBPS, 4),
                    "checked": None, "parity": "tests/test_gpu_model_stream.py (reference token streams, fixture F4)"}
        return res
    finally:
        if os.path.exists(path):
            os.unlink(path)


def f_row_workloads(nv, torch, dev, args):
    """SURVEY.md 8f's rows on the driver's record (N = 1 extras, headline untouched): prefill, the projections, the model
    step.  A row that fails is reported as {"error": ...} and the line's top-level `extras_failed` says so."""
    ev = HipEvents()
    res = {}
    for name, fn in (("prefill_2x8192", lambda: {"prefill_2x8192": prefill_row(nv, torch, dev, ev, args)}),
                     ("qkv", lambda: projection_rows(nv, torch, dev, ev, args)),
                     ("model_step_tinyllama", lambda: model_step_rows(nv, torch, dev, args))):
        try:
            res.update(fn())
        except Exception as exc:  # noqa: BLE001 -- reported in the line, and the exit status says so
            res[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    ev.close()
    return res


class Watchdog:
    """Bounds the shutdown handshake of an N > 1 run (dist.destroy_process_group can wait for a peer for ever).  By the
    time it is armed every number has been printed and checked and this process has no GPU work outstanding; if the
    handshake does not come back within `seconds` the process says so on stderr and leaves with `status`."""

    def __init__(self, seconds, what, status=0):
        import threading

        self._done = threading.Event()

        def dog():
            if self._done.wait(seconds):
                return
            print(f"bench.py: {what} not finished after {seconds:.0f} s; leaving with status {status}", file=sys.stderr,
                  flush=True)
            os._exit(status)

        threading.Thread(target=dog, daemon=True).start()

    def finish(self):
        self._done.set()


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start the ranks ourselves, as the child process
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` (one process per GPU),
    BEFORE this process has imported torch or touched the GPU; its output is ours and so is its exit status."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def extras_in_children(dist, args, rank, world, local_rank):
    """The N > 1 extras (configs[4]) run in a FRESH process per rank, not in the process that holds the headline: a rank
    stuck in a collective or a hung kernel is then a child this rank can end (by its pid) while staying healthy itself.
    Rank 0 picks a free rendezvous port for the children; every rank starts `bench.py --extras-child` with its own
    RANK / LOCAL_RANK / WORLD_SIZE and waits for it at most --extra-timeout seconds.  Returns (results or None,
    error text or None) on rank 0, (None, error) elsewhere."""
    box = [free_port() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    out_path = os.path.join(tempfile.gettempdir(), f"nvk_bench_extras_{box[0]}_{rank}.json")
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}  # (the agent's store is not theirs)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(box[0])})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), "--extras-child", out_path, "--gpus", str(world), "--backend",
           args.backend, "--prewarm-ms", str(args.prewarm_ms)]
    error = None
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL)
    try:
        rc = proc.wait(timeout=args.extra_timeout)
        if rc != 0:
            error = f"extras child of rank {rank} ended with status {rc}"
    except subprocess.TimeoutExpired:
        proc.kill()  # this rank's own child, by pid
        proc.wait()
        error = f"extras not finished after {args.extra_timeout:.0f} s on rank {rank}: child ended, partial results kept"
    got = None
    if rank == 0:
        try:
            with open(out_path) as fh:
                got = json.load(fh)
        except (OSError, ValueError):
            got = None
    try:
        os.unlink(out_path)
    except OSError:
        pass
    # one verdict for the job: any rank's failure is the job's
    errs = [None] * world
    dist.all_gather_object(errs, error)
    errs = [e for e in errs if e]
    return got, ("; ".join(errs)[:600] if errs else None)


def extras_child_main(args):
    """`bench.py --extras-child PATH`: one rank of the N > 1 extras in its own process group.  Rank 0 rewrites PATH
    (atomically) after every finished workload, so whatever was done before a time-out is kept."""
    nv, torch, dist, rank, local_rank, world, dev, red_dev, barrier = init_ranks(args)
    res = {}

    def publish():
        if rank == 0:
            with open(args.extras_child + ".tmp", "w") as fh:
                json.dump(res, fh)
            os.replace(args.extras_child + ".tmp", args.extras_child)

    try:
        multi_gpu_workloads(nv, torch, dist, args, rank, world, local_rank, dev, red_dev, barrier, res, publish)
    except Exception as exc:  # noqa: BLE001 -- recorded, and this child's status says so
        res["error"] = f"{type(exc).__name__}: {exc}"[:300]
        publish()
        os._exit(3)  # (peers may be waiting in a collective: no handshake)
    publish()
    dog = Watchdog(20.0, "extras child: destroy_process_group")
    dist.destroy_process_group()
    dog.finish()


def init_ranks(args):
    """torch + the library + this process's place in the job: RANK / LOCAL_RANK / WORLD_SIZE from the environment
    (torch.distributed.run's), one process per GPU, RCCL ("nccl") or gloo (rehearsal: ranks share GPUs)."""
    import torch
    import nano_vllm_amd as nv

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if nv.device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()  # rehearsal: more ranks than GPUs
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")  # where the timing reduction lives

    def barrier():
        if world > 1:
            dist.barrier()

    return nv, torch, dist, rank, local_rank, world, dev, red_dev, barrier


def multi_gpu_workloads(nv, torch, dist, args, rank, world, local_rank, dev, red_dev, barrier, res, publish):
    """BASELINE.json configs[4] across the ranks of this launch; fills `res` key by key (all ranks; rank 0's is printed).

    5a / 5b: rank r decodes sequences [512 r, 512 (r + 1)) of the 512 x world from a pool of its own: weak scaling, no
    collective, GB/s and tokens/s = all ranks' work over the slowest rank's wall time between two barriers.
    5b_head_sharded: ONE batch of 512 x world sequences, rank r holding kv-heads [r KV / world, (r + 1) KV / world) of
    every sequence (the same bytes per GPU as 5b), out joined by one all-gather per layer-call inside the timed region."""
    from nano_vllm_amd import sharding as sh
    from nano_vllm_amd import workloads as wl

    steps = EXTRA_STEPS

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
            out.fill_(float("nan"))
            run(0, 1, False)
            checked = self_check(work, out.cpu().numpy(), 1 + 16 * rank, 2 + 16 * rank, check_picks(work))  # rank 0's is printed
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
                         "tokens_per_s": round(float(tot[1]) / (ms * 1e-3), 1), "checked_on_rank0": checked}
            publish()
            del q, tabs, ctx, out
        torch.cuda.empty_cache()
        settle(torch)

    full = wl.config5b(512 * world)
    gf = full.geom
    if gf.n_kv_heads % world != 0:
        res["5b_head_sharded"] = {"skipped": f"{world} ranks do not divide {gf.n_kv_heads} kv-heads"}
        publish()
        return
    hs = sh.HeadShard(rank, world, gf.n_heads, gf.n_kv_heads)
    g = wl.Geometry(hs.local_heads, hs.local_kv_heads, gf.head_dim, gf.block_size)
    work = wl.Workload(full.name + f"_heads{rank}of{world}", g, full.num_blocks, full.ctx_lens, full.block_tables, False)
    flags = nv.AUTO | nv.UNIFORM
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim, device=local_rank) as pool:
        pool.fill_uniform(0, 1 + 16 * rank, 2 + 16 * rank)
        q, tabs, ctx, out = device_inputs(work)
        torch.cuda.synchronize()
        # both places the collective can run (nvk_shard_set_gather_mode): hidden on the shard's own stream under the
        # next layer-call (the default with more than one rank), and in line behind the attention on the compute stream
        for key, mode in (("5b_head_sharded", "overlapped"), ("5b_head_sharded_inline", "inline")):
            if mode == "inline" and args.backend != "nccl":
                continue  # (the gloo rehearsal has one form only)
            info = {}
            wall, step_ms, attn_ms = head_sharded_leg(nv, torch, dist, hs, pool, work, q, tabs, ctx, out, flags, 0, steps, 3,
                                                      args.backend, rank, world, local_rank, dev, barrier, mode, info)
            wall, step_ms, attn_ms = reduce_max(wall, step_ms, attn_ms)
            checked = self_check(work, out.cpu().numpy(), 1 + 16 * rank, 2 + 16 * rank, check_picks(work))  # this rank's heads
            ms = wall * 1e3 / steps
            nbytes = work.algorithmic_bytes()  # this rank's heads of the whole batch
            shard_bytes = work.num_seqs * g.n_heads * g.head_dim * 4
            res[key] = {
                "workload": f"{full.name}: ONE batch, {g.n_kv_heads} of {gf.n_kv_heads} kv-heads per GPU x {world} GPUs",
                "scaling": "weak in bytes per GPU (the batch grows with the GPUs, the heads per GPU shrink)",
                "collective": (f"ncclAllGather (RCCL) through nvk_shard_*, {info.get('gather_mode')}: "
                               + ("on its own stream under the next layer-call" if info.get("gather_mode") == "overlapped"
                                  else "on the compute stream behind the attention")
                               if args.backend == "nccl" else "gloo on host copies: rehearsal"),
                "gather_bytes_per_rank": {"sent": shard_bytes, "gathered": shard_bytes * world},
                "steps": steps, "ms_per_step": round(ms, 4), "event_ms_per_step": round(step_ms, 4),
                "GBps": round(nbytes * world / (ms * 1e-3) / 1e9, 1),
                "frac_per_gpu": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "slowest_kernel_ms": round(attn_ms, 4), "sequences": work.num_seqs,
                "tokens_per_s": round(work.num_seqs / (ms * 1e-3), 1), "checked_on_rank0": checked}
            publish()
        del q, tabs, ctx, out
    torch.cuda.empty_cache()


def head_sharded_leg(nv, torch, dist, hs, pool, work, q, tabs, ctx, out, flags, chunk, steps, warmup, backend, rank, world,
                     local_rank, dev, barrier, gather_mode="auto", info=None):
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
    gath = [torch.empty((world,) + tuple(out.shape), dtype=out.dtype, device=dev) for _ in range(2)]
    # with RCCL the attention writes its slice where the gather wants it, gathered[b][rank] (an IN-PLACE all-gather:
    # no copy of the rank's own rows; with one rank nothing moves at all); the gloo rehearsal keeps separate buffers
    outs = [gath[0][rank], gath[1][rank]] if backend == "nccl" else [out, torch.empty_like(out)]
    shard = None
    try:
        if backend == "nccl":
            ids = [sh.NvkShard.new_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0)
            shard = sh.NvkShard(rank, world, ids[0], local_rank, hs.n_heads, hs.n_kv_heads)
            if gather_mode != "auto":
                shard.set_gather_mode({"overlapped": shard.GATHER_OVERLAPPED, "inline": shard.GATHER_INLINE}[gather_mode])
            if info is not None:
                info["gather_mode"] = shard.gather_mode

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
    ap.add_argument("--gather", default="auto", choices=["auto", "overlapped", "inline"],
                    help="--shard heads: where the all-gather runs (nvk_shard_set_gather_mode): the shard's own stream, hidden "
                         "under the next layer-call, or the compute stream; auto = overlapped with more than one rank")
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
    ap.add_argument("--row-pairs", action="store_true",
                    help="diagnostic: whole 128-float rows as kv-head pairs in 4-wave workgroups, two to a CU "
                         "(NVK_DECODE_ROW_PAIRS)")
    ap.add_argument("--no-extra", action="store_true",
                    help="N=1 only: skip the extra BASELINE configs (3, 4, 5a, 5b) timed after the headline workload")
    ap.add_argument("--extra-timeout", type=float, default=150.0,
                    help="N>1 only: seconds the configs[4] extras (5a / 5b sequence-sharded, 5b head-sharded) may take "
                         "before the headline line is printed without them")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="diagnostic: no per-kernel event pairs inside the timed region (roofline from step time)")
    ap.add_argument("--no-frows", action="store_true",
                    help="N=1 only: skip SURVEY.md 8f's rows (prefill, projections, model step) timed after the extras")
    ap.add_argument("--extras-child", default=None, metavar="PATH",
                    help="internal: run one rank of the N>1 extras in this (fresh) process and leave rank 0's results "
                         "at PATH; started by the rank processes themselves, see extras_in_children()")
    args = ap.parse_args()

    if args.extras_child is not None:
        return extras_child_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python bench.py --gpus N` as the driver's N = 1 command is shaped: start the ranks ourselves (a child process,
        # before torch is imported or the GPU touched) and pass its line and status on
        return self_launch(args.gpus, sys.argv[1:])

    nv, torch, dist, rank, local_rank, world, dev, red_dev, barrier = init_ranks(args)

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
    if args.row_pairs:
        flags |= nv.ROW_PAIRS
    uniform_batch = bool((work.ctx_lens == work.ctx_lens[0]).all())
    if args.plan == "none" or (args.plan == "auto" and uniform_batch):
        flags |= nv.UNIFORM if uniform_batch else nv.NO_PLAN
    planning = ("none: uniform lengths, natural chunk order" if flags & nv.UNIFORM else
                "none: lengths read by the decode kernel, natural chunk order" if flags & nv.NO_PLAN else
                f"nvk_plan_kernel, rebuilt every {args.replan} layer-call(s)")

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
        out_host = out.cpu().numpy()  # what the last of those launches left in HBM (layer 0's slab)
        if rotate > 1:  # the same batch over `rotate` different slabs in turn: nothing can be served from the Infinity Cache
            run(rotate, 0, False, rotate)
            r_step, _ = run(0, args.steps, False, rotate)
            _, r_attn = run(0, args.steps, True, rotate)
            rotated = (r_step, r_attn)
        torch.cuda.synchronize()
    else:
        shard_info = {}
        wall, step_ms, attn_ms = head_sharded_leg(nv, torch, dist if world > 1 else None, hs, pool, work, q, tabs, ctx, out,
                                                  flags, args.chunk, args.steps, args.warmup, args.backend, rank, world,
                                                  local_rank, dev, barrier, args.gather, shard_info)
        out_host = out.cpu().numpy()

    if world > 1:
        t = torch.tensor([wall, step_ms, attn_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, step_ms, attn_ms = (float(x) for x in t.tolist())

    line = None
    status = 0
    if rank == 0:
        nbytes = work.algorithmic_bytes()  # per rank, per layer-call
        ms_per_step = wall * 1e3 / args.steps
        job_bytes = whole_bytes if whole_bytes is not None else nbytes * world
        value = job_bytes / (ms_per_step * 1e-3) / 1e9
        achieved = nbytes / (attn_ms * 1e-3) / 1e9
        seqs_all = work.num_seqs * (world if args.shard == "seq" else 1)  # head shards all hold the same batch
        if whole_bytes is not None:
            seqs_all = make_workload(args.workload, 0, args.geom).num_seqs
        traffic, traffic_source = (profiled_traffic(work.name, kernel_name) if args.mode == "auto" and args.chunk == 0
                                   else (None, None))
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
                                       + (f"RCCL via nvk_shard_*, {shard_info.get('gather_mode')}" if args.backend == "nccl"
                                          else "gloo on host copies: rehearsal") + ")")},
            "pct_of_hbm_peak": round(100.0 * value / world / HBM_PEAK_GBPS, 2),
            "tokens_per_s": round(seqs_all / (ms_per_step * 1e-3), 1),
            "tokens_per_s_32_layer_model": round(seqs_all / (ms_per_step * 1e-3) / 32.0, 1),
            "event_ms_per_step": round(step_ms, 4),
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic,
                         "traffic_source": (f"{traffic_source}: rocprofv3 --pmc passes over this workload and kernel, "
                                            "committed; not measured in this run") if traffic_source else None,
                         "kernel_ms": round(attn_ms, 4), "algorithmic_bytes": nbytes,
                         "kernel_events": ("none: step time" if args.no_kernel_events else
                                           f"HIP event pair around each of {args.steps} launches on the launch stream, "
                                           "a second pass right after the timed region")},
            # rows of `out` as the last timed launch left them, against the CPU checker (rank 0's batch)
            "checked": self_check(work, out_host, 1 + 16 * rank, 2 + 16 * rank, check_picks(work)),
        }
        if rotated is not None:
            r_step, r_attn = rotated
            line["rotated"] = {"slabs": rotate, "slab_bytes": int(2 * 4 * work.num_blocks * g.block_size * g.n_kv_heads * g.head_dim),
                               "ms_per_step": round(r_step, 4), "GBps": round(nbytes / (r_step * 1e-3) / 1e9, 1),
                               "kernel_ms": round(r_attn, 4),
                               "kernel_frac": round(nbytes / (r_attn * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        default_run = args.shard == "seq" and args.workload == "config2" and args.geom == "llama" and not args.strong
        if world == 1 and default_run and not (args.no_extra and args.no_frows):
            pool.close()
            pool = None
            del q, tabs, ctx, out
            torch.cuda.empty_cache()
            settle(torch)
            if not args.no_extra:
                line["workloads"] = extra_workloads(nv, torch, dev, args)
            if not args.no_frows:
                line["f_rows"] = f_row_workloads(nv, torch, dev, args)
        if world == 1 and not args.no_cpu:
            cores = min(16, os.cpu_count() or 1)
            one, many = cpu_baseline(work, cores)
            if one is not None:
                line["cpu_baseline"] = one
            if many is not None:
                line["cpu_baseline_all_cores"] = many

    # N > 1, default launch: BASELINE.json configs[4] while all ranks are here -- 5a and 5b sequence-sharded (rank r's
    # 512 of the 512 x N sequences, no collective) and 5b's batch head-sharded (every rank a kv-head group of ALL
    # 512 x N sequences, one RCCL all-gather of out per layer-call through nvk_shard_*).  Extra keys of the same line,
    # measured by a fresh child process per rank (extras_in_children): the headline's processes stay healthy whatever
    # happens there, and say so in `extras_failed` / `multi_gpu_workloads.error`.
    if (world > 1 and args.shard == "seq" and not args.strong and not args.no_extra and args.workload == "config2"
            and args.geom == "llama"):
        pool.close()
        pool = None
        del q, tabs, ctx, out
        torch.cuda.empty_cache()
        multi, error = extras_in_children(dist, args, rank, world, local_rank)
        if rank == 0:
            multi = multi if isinstance(multi, dict) else {}
            if error and "error" not in multi:
                multi["error"] = error
            line["multi_gpu_workloads"] = multi
            line["extras_failed"] = bool(error) or "error" in multi

    if rank == 0:
        def bad(node):  # any {"ok": False} check or {"error": ...} row below `node`
            if isinstance(node, dict):
                if node.get("ok") is False:
                    return "check"
                if "error" in node:
                    return "error"
                for v in node.values():
                    got = bad(v)
                    if got:
                        return got
            return None

        line.setdefault("extras_failed", False)
        for key in ("workloads", "f_rows"):
            if bad(line.get(key)) == "error":
                line["extras_failed"] = True
        failed_check = (line["checked"] or {}).get("ok") is False or any(
            bad(line.get(key)) == "check" for key in ("workloads", "f_rows", "multi_gpu_workloads"))
        line["check_failed"] = failed_check
        print(json.dumps(line), flush=True)
        if line["extras_failed"]:
            print("bench.py: an extra workload failed (see extras_failed and the error keys of the line); the headline "
                  "number stands", file=sys.stderr, flush=True)
        if failed_check:
            print("bench.py: a timed launch's output differs from the CPU checker: the line is NOT a valid measurement",
                  file=sys.stderr, flush=True)
            status = 4

    if pool is not None:
        pool.close()
    if world > 1:
        sys.stdout.flush()
        dog = Watchdog(20.0, "destroy_process_group", status)
        dist.destroy_process_group()
        dog.finish()
    return status


if __name__ == "__main__":
    sys.exit(main())
