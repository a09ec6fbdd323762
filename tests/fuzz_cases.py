"""Random cases for the differential sweeps: tests/test_gpu_fuzz.py runs a few seeds of each in the suite,
tools/soak_parity.py runs them for minutes at a time (profiles/r03_soak_parity.log).  Each function takes a seed,
builds its case from it alone, and raises AssertionError (with the case in the message) when the device and the checker
disagree.  GPU only."""
import os

import numpy as np

import nano_vllm_amd as nv
from hipmem import DeviceArray


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


_CLI_DIR = None


def cli_case(seed):
    """nvk_run (the reference's command line over the library) on a random request file -- prompts of random words and
    bytes, token limits, JSON fields in any order with fields it must ignore -- at a random batch size, on one model or
    sequence-sharded over two or three standing on GPU 0: every request prints exactly the text the Python mirror
    (LlamaModel + Tokenizer over the same C ABI) generates for it alone."""
    import json
    import subprocess
    import tempfile

    import test_cli as tc
    from golden_defs import write_checkpoint, write_tokenizer

    global _CLI_DIR
    if _CLI_DIR is None:
        _CLI_DIR = tempfile.mkdtemp(prefix="nvk_cli_fuzz_")
        write_checkpoint(os.path.join(_CLI_DIR, "model.bin"), tc.SPEC)
        write_tokenizer(os.path.join(_CLI_DIR, "tokenizer.bin"), tc.SPEC["vocab_size"])
    rng = np.random.default_rng(seed)
    words = ["once", "upon", "a", "time", "the", "theory", "of", "thing", "hello", "inn", "ant", "hen", "and", "in", "an", "é", "x.", "?!"]
    reqs = []
    for _ in range(int(rng.integers(1, 7))):
        prompt = " ".join(rng.choice(words, size=int(rng.integers(1, 9))).tolist())
        fields = [("prompt", prompt), ("temperature", 0.0), ("max_tokens", int(rng.integers(1, 40)))]
        if rng.random() < 0.5:
            fields.append(("top_p", 0.9))
        if rng.random() < 0.3:
            fields.append(("extra", {"ignored": [1, {"deep": None}], "s": "\"quoted\" \\ text"}))
        rng.shuffle(fields)
        reqs.append(dict((k, v) for k, v in fields))
    path = os.path.join(_CLI_DIR, f"reqs{seed}.json")
    with open(path, "w") as fh:
        json.dump({"name": f"fuzz{seed}", "requests": reqs}, fh, indent=int(rng.integers(0, 3)) or None)
    batch = str(int(rng.integers(1, 6)))
    sharding = [(), ("--devices", "0,0"), ("--devices", "0,0,0")][int(rng.integers(0, 3))]
    what = f"seed {seed}: {len(reqs)} requests, -b {batch} {' '.join(sharding)}"
    r = subprocess.run([tc.CLI, _CLI_DIR, "--input-json", path, "-b", batch, *sharding], capture_output=True, timeout=120)
    os.remove(path)
    assert r.returncode == 0, f"{what}: exit status {r.returncode}: {r.stderr[-300:]!r}"
    parts = (b"\n" + r.stdout).split(b"\n[request ")[1:]
    assert parts and parts[-1].endswith(b"\n"), f"{what}: output {r.stdout[:200]!r}"
    parts[-1] = parts[-1][:-1]
    got = {int(part.split(b"] ", 1)[0]): part.split(b"] ", 1)[1] for part in parts}
    for i, rq in enumerate(reqs):
        want = rq["prompt"].encode() + tc.expected_text(nv, _CLI_DIR, rq["prompt"], rq["max_tokens"], True)
        assert got.get(i) == want, f"{what}: request {i} {rq['prompt']!r}: printed {got.get(i)!r}, expected {want!r}"


def shard_one_rank_case(seed):
    """nvk_shard_* on a communicator of one rank (all a one-GPU box can form; the head slicing at world 2 and 4 is covered
    on CPU over gloo): random geometry, batch and mode, both places the collective can run, the two-buffer protocol over
    2-5 layer-calls -- gathered[0] must be what nvk_decode writes, bit for bit."""
    from hipmem import Stream
    from nano_vllm_amd import sharding as sh
    from nano_vllm_amd import workloads as wl

    rng = np.random.default_rng(seed)
    KV = int(rng.choice([1, 2, 4, 8]))
    H = KV * int(rng.choice([1, 2, 4, 8]))
    hd = int(rng.choice([16, 48, 64, 128]))
    bs = int(rng.choice([8, 16, 32]))
    n = int(rng.integers(1, 10))
    top = int(rng.choice([50, 600, 3000]))
    ctx = rng.integers(0, top + 1, size=n).astype(np.int32)
    ctx[int(rng.integers(0, n))] = top
    g = wl.Geometry(H, KV, hd, bs)
    work = wl.fragmented_workload(f"shard{seed}", ctx, g, slack=0.2, perm_seed=seed)
    L = int(rng.integers(2, 6))
    flags, chunk = [(nv.AUTO, 0), (nv.SPLIT_K, int(rng.choice([32, 128, 512]))), (nv.SINGLE_PASS, 0)][int(rng.integers(0, 3))]
    what = f"seed {seed}: H{H} KV{KV} hd{hd} bs{bs} ctx{ctx.tolist()} L{L} flags={flags:#x} chunk={chunk}"
    with nv.PagedKVPool(L, work.num_blocks, bs, KV, hd) as pool:
        for layer in range(L):
            pool.fill_uniform(layer, 10 + layer, 20 + layer)
        shard = sh.NvkShard(0, 1, sh.NvkShard.new_unique_id(), 0, H, KV)
        st = Stream()
        try:
            shard.set_gather_mode([shard.GATHER_AUTO, shard.GATHER_OVERLAPPED, shard.GATHER_INLINE][int(rng.integers(0, 3))])
            q = work.q()
            qd, td, cd = DeviceArray(q), DeviceArray(work.block_tables), DeviceArray(ctx)
            outs = [DeviceArray(np.full_like(q, np.nan)) for _ in range(2)]
            gath = [DeviceArray(np.full((1,) + q.shape, np.nan, np.float32)) for _ in range(2)]
            got = []
            for layer in range(L):
                b = layer % 2
                shard.decode_gather(pool, layer, n, qd, td, work.max_blocks, cd, work.max_ctx, outs[b], gath[b], b, flags, chunk,
                                    compute_stream=st)
                if layer > 0:
                    shard.wait_gather((layer - 1) % 2, st)
                    shard.wait_gather((layer - 1) % 2)
                    got.append(gath[(layer - 1) % 2].numpy().copy())
            shard.wait_gather((L - 1) % 2)
            got.append(gath[(L - 1) % 2].numpy().copy())
            st.synchronize()
            for layer in range(L):
                want = nv.decode(pool, layer, q, work.block_tables, ctx, flags, chunk)
                assert np.isfinite(want).all() and np.array_equal(got[layer][0].view(np.uint32), want.view(np.uint32)), \
                    f"{what}: layer-call {layer}"
        finally:
            shard.close()
            st.close()


def plan_case(seed):
    """A model step's use of decode plans (nvk_plan_build once per step, nvk_decode_planned per layer), over several
    steps of a batch whose sequences grow by a token, finish and join: each step the plan is rebuilt -- or, when the
    lengths did not change, reused --, run on every layer of a pool whose layers hold different rows, and compared bit
    for bit with nvk_decode_device of the same mode and, on one layer, with the checker.  Plans of different sizes follow
    each other in one object (its work list and counters grow and are reused)."""
    import oracle_lib as ol
    from nano_vllm_amd import workloads as wl

    rng = np.random.default_rng(seed)
    KV = int(rng.choice([1, 2, 4, 8]))
    H = KV * int(rng.choice([1, 2, 4, 8, 16]))
    hd = int(rng.choice([16, 48, 64, 128]))
    bs = int(rng.choice([8, 16, 32]))
    L = int(rng.integers(1, 4))
    n0 = int(rng.integers(1, 12))
    top = int(rng.choice([60, 400, 2500]))
    g = wl.Geometry(H, KV, hd, bs)
    room = int(rng.integers(3, 12))  # tokens a sequence may grow by
    ctx = rng.integers(0, top + 1, size=n0).astype(np.int32)
    ctx[int(rng.integers(0, n0))] = top
    work = wl.fragmented_workload(f"plan{seed}", ctx + room, g, slack=0.2, perm_seed=seed)
    k_host, v_host, tabs, _ = wl.host_blocks_for(work, list(range(n0)))
    flags, chunk = [(nv.AUTO, 0), (nv.SPLIT_K, int(rng.choice([16, 64, 256]))), (nv.SPLIT_K | nv.MERGE_SEPARATE, int(rng.choice([32, 128]))),
                    (nv.SINGLE_PASS, 0)][int(rng.integers(0, 4))]
    what = f"seed {seed}: H{H} KV{KV} hd{hd} bs{bs} L{L} flags={flags:#x} chunk={chunk}"
    with nv.PagedKVPool(L, work.num_blocks, bs, KV, hd) as pool:
        for l in range(L):
            pool.fill_uniform(l, wl.SEED_K + l, wl.SEED_V + l)
        plan = nv.DecodePlan(pool)
        try:
            alive = list(range(n0))
            lens = ctx.copy()
            for step in range(int(rng.integers(2, 7))):
                if step and rng.random() < 0.7:
                    lens[alive] = np.minimum(lens[alive] + 1, ctx[alive] + room)  # (a token each; otherwise the last plan still fits)
                if len(alive) > 1 and rng.random() < 0.3:
                    alive.pop(int(rng.integers(0, len(alive))))
                n = len(alive)
                q = rng.standard_normal((n, H, hd), dtype=np.float32)
                qd, td = DeviceArray(q), DeviceArray(np.ascontiguousarray(work.block_tables[alive]))
                cd = DeviceArray(np.ascontiguousarray(lens[alive]))
                max_ctx = int(lens[alive].max()) if n else 0
                plan.build(n, H, cd, max_ctx, flags, chunk)
                for l in range(L):
                    o1, o2 = DeviceArray(np.full_like(q, np.nan)), DeviceArray(np.full_like(q, np.nan))
                    plan.run(l, qd, td, work.max_blocks, o1)
                    nv.decode_device(pool, l, n, H, qd, td, work.max_blocks, cd, max_ctx, o2, flags, chunk)
                    pool.synchronize()
                    a, b = o1.numpy(), o2.numpy()
                    assert np.isfinite(a).all() and np.array_equal(a.view(np.uint32), b.view(np.uint32)), \
                        f"{what}: step {step} layer {l}: planned and direct launches differ"
                    if l == 0:
                        kh = wl.host_blocks_for(work, alive)  # (layer 0's fill is the workload's own)
                        want = ol.oracle_decode_batch(q, kh[0], kh[1], kh[2], lens[alive], bs, KV, n_threads=8)
                        err = np.abs(a - want)
                        assert (err <= 1e-4 + 1e-5 * np.abs(want)).all(), f"{what}: step {step}: max err {err.max():.3e}"
        finally:
            plan.close()


def pool_ops_case(seed):
    """The pool's filling and reading entry points in random order on a random geometry (odd head widths included): rows
    appended one by one and in batches, RoPE + append (Ops::apply_rope, positional.hpp:13-39, then model.hpp:190-207),
    whole blocks uploaded; against a host copy of the pool -- every block downloaded bit for bit (rotated K rows within
    4e-7 of the checker's rotation), every kv-head gathered through a shuffled table, and one decode over it all."""
    import oracle_lib as ol

    rng = np.random.default_rng(seed)
    KV = int(rng.choice([1, 2, 3, 4, 6, 8]))
    H = KV * int(rng.choice([1, 2, 3, 4, 8]))
    hd = int(rng.choice([4, 16, 20, 48, 64, 80, 128, 132, 256]))
    bs = int(rng.choice([1, 4, 8, 16, 32]))
    L, nb = int(rng.integers(1, 4)), int(rng.integers(3, 40))
    theta = float(rng.choice([10000.0, 500000.0]))
    hk = np.zeros((L, nb, bs, KV, hd), np.float32)
    hv = np.zeros_like(hk)
    loose = np.zeros((L, nb, bs), bool)  # slots holding a rotated K row (compared within the rotation's bar)
    what = f"seed {seed}: {L} layers x {nb} blocks of {bs}, KV{KV} H{H} hd{hd}"
    with nv.PagedKVPool(L, nb, bs, KV, hd) as pool:
        for op in range(int(rng.integers(5, 40))):
            l, kind = int(rng.integers(0, L)), int(rng.integers(0, 4))
            if kind == 0:
                pb, off = int(rng.integers(0, nb)), int(rng.integers(0, bs))
                kr, vr = rng.standard_normal((KV, hd), dtype=np.float32), rng.standard_normal((KV, hd), dtype=np.float32)
                pool.append(l, pb, off, kr, vr)
                hk[l, pb, off], hv[l, pb, off], loose[l, pb, off] = kr, vr, False
            elif kind == 3:
                pb = int(rng.integers(0, nb))
                kb, vb = rng.standard_normal((bs, KV, hd), dtype=np.float32), rng.standard_normal((bs, KV, hd), dtype=np.float32)
                pool.upload_block(l, pb, kb, vb)
                hk[l, pb], hv[l, pb], loose[l, pb] = kb, vb, False
            else:  # a batch: n sequences over disjoint blocks, one new row each
                n = int(rng.integers(1, min(nb, 9) + 1))
                max_blocks = max(1, nb // n)
                tabs = rng.permutation(nb)[: n * max_blocks].reshape(n, max_blocks).astype(np.int32)
                pos = rng.integers(0, max_blocks * bs, size=n).astype(np.int32)
                kr, vr = rng.standard_normal((n, KV, hd), dtype=np.float32), rng.standard_normal((n, KV, hd), dtype=np.float32)
                if kind == 1:
                    pool.append_batch(l, kr, vr, tabs, pos)
                    want_k = kr
                else:
                    q = rng.standard_normal((n, H, hd), dtype=np.float32)
                    q_rot = pool.rope_append_batch(l, q, kr, vr, tabs, pos, theta)
                    want_k = np.empty_like(kr)
                    for s_ in range(n):
                        wq, want_k[s_] = ol.apply_rope(ol.ORACLE.oracle_apply_rope, q[s_], kr[s_], int(pos[s_]), hd, H, KV, theta)
                        assert np.abs(q_rot[s_] - wq).max() <= 4e-7 * max(1.0, float(np.abs(wq).max())), f"{what}: op {op}: rotated q"
                for s_ in range(n):
                    pb, off = int(tabs[s_, pos[s_] // bs]), int(pos[s_] % bs)
                    hk[l, pb, off], hv[l, pb, off], loose[l, pb, off] = want_k[s_], vr[s_], kind == 2
        for l in range(L):
            for pb in range(nb):
                kb, vb = pool.download_block(l, pb)
                assert np.array_equal(vb.view(np.uint32), hv[l, pb].view(np.uint32)), f"{what}: V of layer {l} block {pb}"
                exact = ~loose[l, pb]
                assert np.array_equal(kb[exact].view(np.uint32), hk[l, pb][exact].view(np.uint32)), f"{what}: K of layer {l} block {pb}"
                if loose[l, pb].any():
                    d = np.abs(kb[loose[l, pb]] - hk[l, pb][loose[l, pb]])
                    assert d.max() <= 4e-7 * max(1.0, float(np.abs(hk[l, pb]).max())), f"{what}: rotated K of layer {l} block {pb}"
                    hk[l, pb][loose[l, pb]] = kb[loose[l, pb]]  # (what the pool holds is what the reads below see)
        l = int(rng.integers(0, L))
        table = rng.permutation(nb).astype(np.int32)
        T = int(rng.integers(1, nb * bs + 1))
        for h in range(KV):
            gk, gv = pool.gather(l, table, T, h)
            lin_k = hk[l][table].reshape(nb * bs, KV, hd)[:T, h]
            lin_v = hv[l][table].reshape(nb * bs, KV, hd)[:T, h]
            assert np.array_equal(gk.view(np.uint32), lin_k.view(np.uint32)) and np.array_equal(gv.view(np.uint32), lin_v.view(np.uint32)), \
                f"{what}: gather of kv-head {h}, {T} tokens"
        q = rng.standard_normal((1, H, hd), dtype=np.float32)
        got = nv.decode(pool, l, q, table[None, :], np.array([T], np.int32))
        want = ol.oracle_paged_attention(q[0], hk[l], hv[l], table, T, bs, hd, H, KV)
        err = np.abs(got[0] - want)
        assert np.isfinite(got).all() and (err <= 1e-4 + 1e-5 * np.abs(want)).all(), f"{what}: decode over {T} tokens: max err {err.max():.3e}"


def dropin_case(seed):
    """An engine's calls to the drop-in entry point with HOST slabs (Attention::paged_attention's eleven arguments,
    include/core/attention.hpp:82-92), under both staging policies: two layer slabs, several requests decoding in turn,
    requests finishing and new ones taking their blocks (lowest id first, as the reference's BlockManager hands them out),
    prompts written token by token or many rows at once before the first decode call -- half of the new requests arrive
    holding exactly one token more than the call before, on blocks just freed.  Every call against the checker on the
    slabs as they are at that moment."""
    import oracle_lib as ol

    rng = np.random.default_rng(seed)
    KV = int(rng.choice([1, 2, 4]))
    H = KV * int(rng.choice([1, 2, 4]))
    hd = int(rng.choice([16, 64]))
    bs = int(rng.choice([4, 8, 16]))
    nb, L = int(rng.integers(6, 30)), 2
    k = [np.zeros((nb, bs, KV, hd), np.float32) for _ in range(L)]
    v = [np.zeros((nb, bs, KV, hd), np.float32) for _ in range(L)]
    free, live, next_id, last_tokens = list(range(nb)), {}, 0, 0
    append_only = bool(rng.integers(0, 3))
    what = f"seed {seed}: H{H} KV{KV} hd{hd} bs{bs}, {nb} blocks, append-only mirror {append_only}"

    def write_row(req, pos):
        while pos // bs >= len(req["table"]):
            req["table"].append(free.pop(0))
        pb, off = req["table"][pos // bs], pos % bs
        for l in range(L):
            k[l][pb, off] = rng.standard_normal((KV, hd), dtype=np.float32)
            v[l][pb, off] = rng.standard_normal((KV, hd), dtype=np.float32)

    nv.set_host_cache_policy(append_only)
    try:
        for step in range(int(rng.integers(30, 120))):
            r = rng.random()
            if live and r < 0.08:  # a request finishes: its blocks go back, lowest id first out again
                done = live.pop(list(live)[int(rng.integers(0, len(live)))])
                free.extend(done["table"])
                free.sort()
            if (not live or r > 0.85) and len(free) >= 2 and len(live) < 4:  # a new one, its prompt written in one go
                room = len(free) * bs - 1
                prompt = last_tokens if (rng.random() < 0.5 and 0 < last_tokens <= room) else int(rng.integers(0, min(room, 3 * bs) + 1))
                live[next_id] = {"table": [], "len": 0}
                for pos in range(prompt):
                    write_row(live[next_id], pos)
                live[next_id]["len"] = prompt
                next_id += 1
            if not live:
                continue
            req = live[list(live)[int(rng.integers(0, len(live)))]]
            if req["len"] // bs >= len(req["table"]) and not free:
                continue  # (no block for its next token)
            write_row(req, req["len"])
            req["len"] += 1
            table = np.array(req["table"], np.int32)
            for l in range(L):
                q = rng.standard_normal((H, hd), dtype=np.float32)
                out = np.full(H * hd, np.nan, np.float32)
                nv.paged_attention(out, q, k[l], v[l], table, None, req["len"], bs, hd, H, KV)
                want = ol.oracle_paged_attention(q, k[l], v[l], table, req["len"], bs, hd, H, KV).reshape(-1)
                err = np.abs(out - want)
                assert np.isfinite(out).all() and (err <= 1e-4 + 1e-5 * np.abs(want)).all(), \
                    f"{what}: step {step} layer {l}, {req['len']} tokens over blocks {req['table']}: max err {err.max():.3e}"
            last_tokens = req["len"]
    finally:
        nv.set_host_cache_policy(False)


def dropin_contiguous_case(seed):
    """The same for the contiguous-cache twin (Attention::standard_attention, attention.hpp:16-76; one sequence per
    slab, as LlamaModel::attention calls it, model.hpp:419-430): tokens appended one by one, now and then a new sequence
    from row 0 -- its prompt written in one go, half of the time exactly as long as the last call's context -- under both
    staging policies, two layer slabs, every call against the checker."""
    import oracle_lib as ol

    rng = np.random.default_rng(seed)
    KV = int(rng.choice([1, 2, 4]))
    H = KV * int(rng.choice([1, 2, 4]))
    hd = int(rng.choice([16, 48, 64]))
    S, L = int(rng.integers(8, 120)), 2
    k = [np.zeros((S, KV, hd), np.float32) for _ in range(L)]
    v = [np.zeros((S, KV, hd), np.float32) for _ in range(L)]
    append_only = bool(rng.integers(0, 3))
    what = f"seed {seed}: H{H} KV{KV} hd{hd}, max_seq_len {S}, append-only mirror {append_only}"
    n = 0  # rows of the current sequence
    nv.set_host_cache_policy(append_only)
    try:
        for step in range(int(rng.integers(20, 100))):
            if n >= S or rng.random() < 0.07:  # a new sequence over the same slabs
                last = n
                n = last if (rng.random() < 0.5 and 1 <= last < S) else int(rng.integers(0, max(1, S // 2)))
                for l in range(L):
                    k[l][:n] = rng.standard_normal((n, KV, hd), dtype=np.float32)
                    v[l][:n] = rng.standard_normal((n, KV, hd), dtype=np.float32)
            for l in range(L):
                k[l][n] = rng.standard_normal((KV, hd), dtype=np.float32)
                v[l][n] = rng.standard_normal((KV, hd), dtype=np.float32)
            n += 1
            for l in range(L):
                q = rng.standard_normal((H, hd), dtype=np.float32)
                out = np.full(H * hd, np.nan, np.float32)
                nv.standard_attention(out, q, k[l], v[l], None, n - 1, hd, H, KV, S)
                want = ol.standard_attention(ol.ORACLE.oracle_standard_attention, q, k[l], v[l], n - 1, hd, H, KV, S).reshape(-1)
                err = np.abs(out - want)
                assert np.isfinite(out).all() and (err <= 1e-4 + 1e-5 * np.abs(want)).all(), \
                    f"{what}: step {step} layer {l}, pos {n - 1}: max err {err.max():.3e}"
    finally:
        nv.set_host_cache_policy(False)


def decode_large_case(seed):
    """Decode at the sizes the launch shapes were tuned on: the BASELINE geometries and their neighbours, 10-600
    sequences, contexts to 20 000 tokens (ragged, some empty, some uniform), fragmented tables; every mode and modifier
    (plan kernel with thousands of items, plan-free and uniform launches with skewed chunks, in-kernel and separate merge,
    the vector kernel where the MFMA one would run, free-running waves, whole-row pairs) against the checker."""
    import oracle_lib as ol
    from nano_vllm_amd import workloads as wl

    rng = np.random.default_rng(seed)
    H, KV, hd = [(32, 8, 128), (32, 4, 128), (32, 4, 64), (64, 8, 128), (32, 32, 128), (16, 4, 128), (4, 1, 128), (8, 8, 64),
                 (40, 8, 128), (6, 6, 48), (128, 8, 128), (16, 1, 128), (12, 2, 64), (32, 8, 64)][int(rng.integers(0, 14))]
    bs = int(rng.choice([16, 16, 16, 8, 32]))
    shape = int(rng.integers(0, 4))
    if shape == 0:    # many short sequences (5a)
        n, top = int(rng.integers(100, 600)), int(rng.choice([64, 256, 600]))
    elif shape == 1:  # a batch of medium ones (configs 2, 3)
        n, top = int(rng.integers(10, 130)), int(rng.choice([1024, 2048, 4096]))
    elif shape == 2:  # few long ones (config 4)
        n, top = int(rng.integers(1, 9)), int(rng.choice([8192, 20000]))
    else:             # mixed
        n, top = int(rng.integers(10, 200)), int(rng.choice([300, 3000]))
    # (keep the checker's work bounded: ~ tokens x q-heads x head_dim)
    while n * top * H * hd > 6e9 and n > 1:
        n = max(1, n // 2)
    uniform = rng.random() < 0.25
    ctx = np.full(n, top, np.int32) if uniform else rng.integers(0, top + 1, size=n).astype(np.int32)
    ctx[int(rng.integers(0, n))] = top
    g = wl.Geometry(H, KV, hd, bs)
    work = wl.fragmented_workload(f"large{seed}", ctx, g, slack=0.1, perm_seed=seed)
    q = work.q() * np.float32(rng.choice([1, 3]))
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(n)))
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, bs, KV, n_threads=16)
    chunk = int(rng.choice([128, 256, 512, 1024, 2048]))
    modes = [(nv.AUTO, 0), (nv.SPLIT_K, chunk), (nv.SPLIT_K | nv.NO_PLAN, chunk), (nv.SPLIT_K | nv.MERGE_SEPARATE, chunk),
             (nv.AUTO | nv.VECTOR_ONLY, 0), (nv.AUTO | nv.FREE_RUNNING, 0), (nv.AUTO | nv.NO_PAIRS, 0), (nv.SINGLE_PASS, 0)]
    if uniform:
        modes += [(nv.AUTO | nv.UNIFORM, 0), (nv.SPLIT_K | nv.UNIFORM, chunk)]
    if KV == 8 and hd == 128:
        modes.append((nv.AUTO | nv.ROW_PAIRS, 0))
    what = f"seed {seed}: H{H} KV{KV} hd{hd} bs{bs}, {n} sequences to {top} tokens{' (uniform)' if uniform else ''}"
    with nv.PagedKVPool(1, work.num_blocks, bs, KV, hd) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for i in rng.permutation(len(modes))[:4]:
            flags, ch = modes[int(i)]
            got = nv.decode(pool, 0, q, work.block_tables, work.ctx_lens, flags, ch)
            err = np.abs(got - want)
            assert np.isfinite(got).all() and (err <= 1e-4 + 1e-5 * np.abs(want)).all(), \
                f"{what} flags={flags:#x} chunk={ch}: max err {err.max():.3e} in row {np.unravel_index(err.argmax(), err.shape)}"


def prefill_long_case(seed):
    """Paged chunked prefill deep into long prompts: 1-3 sequences, chunks of up to 400 queries behind up to 4000 cached
    tokens (split keys, tile skipping under the causal line, ragged last tiles), every form, against the checker (query b
    is the decode attention over start + b + 1 tokens; the checker's rows spread over host threads)."""
    from concurrent.futures import ThreadPoolExecutor

    import oracle_lib as ol
    from nano_vllm_amd import workloads as wl

    rng = np.random.default_rng(seed)
    KV = int(rng.choice([1, 2, 4, 8]))
    H = KV * int(rng.choice([1, 2, 4, 8]))
    hd = int(rng.choice([48, 64, 128]))
    bs = int(rng.choice([8, 16, 32]))
    n = int(rng.integers(1, 4))
    starts = rng.integers(0, int(rng.choice([300, 1500, 4000])), size=n).astype(np.int32)
    lens = rng.integers(1, int(rng.choice([40, 150, 400])), size=n).astype(np.int32)
    g = wl.Geometry(H, KV, hd, bs)
    work = wl.fragmented_workload(f"plong{seed}", starts + lens, g, slack=0.2, perm_seed=seed)
    k_host, v_host, tabs, _ = wl.host_blocks_for(work, list(range(n)))
    total = int(lens.sum())
    q = (rng.random((total, H, hd), dtype=np.float32) - np.float32(0.5)) * np.float32(rng.choice([1, 2, 4]))
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(int)
    pieces = [(s, a, min(a + 32, int(offs[s + 1]))) for s in range(n) for a in range(int(offs[s]), int(offs[s + 1]), 32)]

    def rows(piece):
        s, a, b = piece
        return ol.oracle_paged_prefill(q[a:b], k_host, v_host, tabs[s], int(starts[s]) + a - int(offs[s]), bs, KV)

    with ThreadPoolExecutor(16) as ex:
        want = np.concatenate(list(ex.map(rows, pieces)))
    what = f"seed {seed}: H{H} KV{KV} hd{hd} bs{bs} starts {starts.tolist()} lens {lens.tolist()}"
    with nv.PagedKVPool(1, work.num_blocks, bs, KV, hd) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for flags in (nv.PREFILL_ROWWISE, nv.PREFILL_TILED, nv.PREFILL_AUTO):
            got = nv.prefill(pool, 0, q, work.block_tables, starts, lens, flags)
            err = np.abs(got - want)
            assert np.isfinite(got).all() and (err <= 1e-4 + 1e-5 * np.abs(want)).all(), f"{what} flags={flags}: max err {err.max():.3e}"


def noalloc_case(seed):
    """nvk_decode_device_noalloc (the entry point that allocates nothing) against nvk_decode_device, bit for bit, on a
    random geometry / batch / mode; sizes past its stated maximum must be refused with an error, nothing else."""
    from nano_vllm_amd import workloads as wl

    rng = np.random.default_rng(seed)
    KV = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 16]))
    H = KV * int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 16]))
    hd = int(rng.choice([4, 16, 20, 48, 64, 80, 96, 128, 132, 192, 256]))
    bs = int(rng.choice([1, 4, 8, 12, 16, 32]))
    n = int(rng.integers(1, 12))
    top = int(rng.choice([40, 300, 1500, 5000]))
    ctx = rng.integers(0, top + 1, size=n).astype(np.int32)
    ctx[int(rng.integers(0, n))] = top  # (a pool needs a block)
    g = wl.Geometry(H, KV, hd, bs)
    work = wl.fragmented_workload(f"na{seed}", ctx, g, slack=0.3, perm_seed=seed)
    q = work.q()
    flags, chunk = [(nv.AUTO, 0), (nv.SPLIT_K, int(rng.choice([8, 24, 64, 200, 512]))),
                    (nv.SPLIT_K | nv.MERGE_SEPARATE, int(rng.choice([24, 64, 256]))),
                    (nv.SPLIT_K | nv.NO_PLAN, int(rng.choice([64, 128, 512]))), (nv.SINGLE_PASS, 0)][int(rng.integers(0, 5))]
    what = f"seed {seed}: H{H} KV{KV} hd{hd} bs{bs} ctx{ctx.tolist()} flags={flags} chunk={chunk}"
    with nv.PagedKVPool(1, work.num_blocks, bs, KV, hd) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        qd, td, cd = DeviceArray(q), DeviceArray(work.block_tables), DeviceArray(work.ctx_lens)
        o1, o2 = DeviceArray(np.full_like(q, np.nan)), DeviceArray(np.full_like(q, np.nan))
        try:
            nv.decode_device_noalloc(pool, 0, n, H, qd, td, work.max_blocks, cd, work.max_ctx, o1, flags, chunk)
        except nv.NvkError as e:
            assert "the static array holds" in str(e), f"{what}: {e}"
            return "refused (more scratch than the static array holds)"
        nv.decode_device(pool, 0, n, H, qd, td, work.max_blocks, cd, work.max_ctx, o2, flags, chunk)
        pool.synchronize()
        a, b = o1.numpy(), o2.numpy()
        assert np.isfinite(b).all() and np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{what}: bits differ"


def model_case(seed, group=False, scarce=False, big=False):
    """(group: the same through nvk_model_group_* -- two or three models on device 0 standing in for as many GPUs, each
    request placed on one of them at admission.)
    A random llama2.c checkpoint (head geometry, layers, widths, vocabulary, block size, shared or own classifier)
    through the device-resident model: 1-6 requests with different prompts, joining at different steps, prompts fed
    in chunks, greedy decode.  Every logits row against the checker's LlamaModel::forward of that request alone, fed
    the tokens the device chose (2e-3 + 1e-4 |ref| + 1e-4 max |ref|: the model-level bar of tests/test_gpu_model_stream.py
    plus a term in the row's scale), and the
    argmax wherever the checker's first two logits are further apart than that bar."""
    import ctypes as C
    import tempfile

    import oracle_lib as ol
    from golden_defs import write_checkpoint

    rng = np.random.default_rng(seed)
    while True:
        hd = int(rng.choice([64, 128]) if big else rng.choice([4, 8, 16, 20, 32, 48, 64, 128]))
        KV = int(rng.choice([1, 2, 3, 4, 6, 8]))
        H = KV * int(rng.choice([1, 2, 3, 4, 8]))
        if (H * hd in (1024, 2048, 4096)) if big else (H * hd <= 768):
            break
    bs = int(rng.choice([4, 8, 16, 32]))
    n_req = int(rng.choice([1, 2, 3, 8, 16, 17, 24, 33, 40]) if big else rng.integers(1, 7))
    max_seq = int(rng.integers(24, 40) if big else rng.integers(48, 129))
    spec = dict(dim=H * hd, hidden_dim=int(rng.choice([2816, 5632, 11008])) if big else 4 * int(rng.integers(8, 300)),
                n_layers=int(rng.integers(1, 3) if big else rng.integers(1, 5)), n_heads=H,
                n_kv_heads=KV, vocab_size=int(rng.choice([8000, 32000]) if big else rng.integers(40, 600)), max_seq_len=max_seq, block_size=bs,
                num_blocks=n_req * ((max_seq + bs - 1) // bs) + int(rng.integers(0, 4)), weight_seed=1000 + seed,
                weight_gain=0.0, shared=bool(rng.integers(0, 2)))
    # (the fixtures' conditioning: 0.35 at dim 96, 0.30 at dim 288; wider models with such weights saturate the softmax
    #  and the logits' differences stop being rounding)
    spec["weight_gain"] = float(min(0.35, 5.0 / np.sqrt(spec["dim"])))
    if scarce:
        spec["num_blocks"] = max(2, int(spec["num_blocks"] * rng.uniform(0.25, 0.8)))
    gen = int(rng.integers(2, 5) if big else rng.integers(4, 14))
    prompts = [[1] + rng.integers(3, spec["vocab_size"], size=int(rng.integers(0, 8 if big else max_seq - gen - 2))).tolist()
               for _ in range(n_req)]
    join = [0] + [int(rng.integers(0, 5)) for _ in range(n_req - 1)]
    chunk = int(rng.choice([1, 3, 7, 16, 200]))  # prompt tokens fed per step
    what = f"seed {seed}: {spec} prompts {[len(q) for q in prompts]} join {join} chunk {chunk}"
    V = spec["vocab_size"]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.bin")
        write_checkpoint(path, spec)
        # (the checker's model, like the reference's, takes every layer's blocks from one manager)
        oracles = [ol.ORACLE.oracle_model_create(path.encode(), bs, spec["n_layers"] * ((max_seq + bs - 1) // bs + 1))
                   for _ in range(n_req)]
        assert all(oracles), what
        o_pos = [0] * n_req
        def oracle_feed(r, tokens):  # returns the logits behind the last token
            want = np.empty(V, np.float32)
            for i, t in enumerate(tokens):
                last = i == len(tokens) - 1
                rc = ol.ORACLE.oracle_model_forward(oracles[r], int(t), o_pos[r], want.ctypes.data_as(C.c_void_p) if last else None)
                assert rc == 0, f"{what}: checker refused token {o_pos[r]} of request {r}"
                o_pos[r] += 1
            return want

        try:
            members = int(rng.integers(2, 4))
            with (nv.LlamaModelGroup(path, [0] * members, bs, spec["num_blocks"], max_seqs_per_member=n_req) if group
                  else nv.LlamaModel(path, bs, spec["num_blocks"], max_seqs=n_req)) as m:
                graphs, linked = bool(rng.integers(0, 4)), bool(rng.integers(0, 3) == 0)
                m.use_graphs(graphs)  # (on by default)
                if not group:
                    m.use_linked_norms(linked)  # (off by default: a few ulp from the separate norm kernels)
                what += f" graphs {graphs} linked norms {linked and not group}"
                fed = [0] * n_req          # prompt tokens already given
                out = [[] for _ in range(n_req)]
                joined = [False] * n_req
                dropped = 0
                for step in range(400):
                    ids, chunks = [], []
                    for r in range(n_req):
                        if join[r] > step or len(out[r]) >= gen:
                            continue
                        if not joined[r]:
                            if group:
                                m.add_request(10 + r, len(prompts[r]) + gen)
                            else:
                                m.add_request(10 + r)
                            joined[r] = True
                        piece = prompts[r][fed[r]:fed[r] + chunk] if fed[r] < len(prompts[r]) else [out[r][-1]]
                        ids.append(r), chunks.append(piece)
                    if not ids:
                        if all(len(o) >= gen for o in out):
                            break
                        continue
                    try:
                        nxt, logits = m.step_chunks([10 + r for r in ids], chunks, True)
                    except nv.NvkError as e:
                        assert scarce and "no free blocks" in str(e), f"{what}: step {step}: {e}"
                        # nothing advanced: finish the request that holds the most, repeat the step without it
                        victim = max((r for r in range(n_req) if joined[r] and len(out[r]) < gen), key=lambda r: o_pos[r])
                        m.finish_request(10 + victim)
                        out[victim] = [0] * gen  # (counts as done)
                        dropped += 1
                        continue
                    if big:  # the checker's forward passes of this step, a request to a thread (ctypes drops the GIL)
                        from concurrent.futures import ThreadPoolExecutor
                        with ThreadPoolExecutor(16) as ex:
                            refs = list(ex.map(lambda a: oracle_feed(*a).copy(), zip(ids, chunks)))
                    for row, (r, piece) in enumerate(zip(ids, chunks)):
                        if fed[r] < len(prompts[r]):
                            fed[r] += len(piece)
                        ref = refs[row] if big else oracle_feed(r, piece)
                        got = logits[row]
                        err = np.abs(got - ref)
                        # (the fixtures' logits are O(5); these reach 50 at dim 768, so the bar gets a term in the row's scale)
                        bar = 2e-3 + 1e-4 * np.abs(ref) + 1e-4 * float(np.abs(ref).max())
                        assert np.isfinite(got).all() and (err <= bar).all(), \
                            f"{what}: step {step} request {r} at {o_pos[r]} tokens: max err {err.max():.3e}, largest logit {np.abs(ref).max():.1f}"
                        top = np.sort(ref)[-2:]
                        if fed[r] >= len(prompts[r]):  # the token behind a finished prompt, or behind a decode token
                            if top[1] - top[0] > 2 * float(bar.max()):
                                assert int(nxt[row]) == int(np.argmax(ref)), f"{what}: step {step} request {r}: argmax"
                            out[r].append(int(nxt[row]))
                            if len(out[r]) >= gen:
                                m.finish_request(10 + r)
                assert all(len(o) >= gen for o in out), what
                assert group or m.free_blocks == spec["num_blocks"], what
                if scarce:
                    return f"{min(dropped, 3)}{'+' if dropped > 3 else ''} request(s) dropped for blocks"
        finally:
            for o in oracles:
                ol.ORACLE.oracle_model_destroy(o)
