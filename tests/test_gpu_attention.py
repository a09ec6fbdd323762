"""Parity of the HIP path, through the C ABI, against the golden vectors and the CPU checker.

Tolerance (BASELINE.json north_star): fp32 outputs within 1e-4 max-abs / 1e-5
relative of the reference, i.e. |got - want| <= 1e-4 + 1e-5 * |want| elementwise.
The KV gather and the block allocator are integer/byte work and must be bit-exact.
"""
import os

import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from golden_defs import ATTENTION_CASES, attention_case_inputs
from nano_vllm_amd import workloads as wl

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-4, 1e-5


def assert_close(got, want, what=""):
    got = np.asarray(got, np.float32)
    want = np.asarray(want, np.float32)
    assert got.shape == want.shape, what
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: NaN pattern differs"
    ok = np.abs(got - want) <= ATOL + RTOL * np.abs(want)
    ok |= np.isnan(want)
    assert ok.all(), f"{what}: max abs err {np.nanmax(np.abs(got - want)):.3e}"


def pool_from_host(k_pool, v_pool, n_layers=1, layer=0):
    nb, bs, kv, hd = k_pool.shape
    pool = nv.PagedKVPool(n_layers, nb, bs, kv, hd)
    for b in range(nb):
        pool.upload_block(layer, b, k_pool[b], v_pool[b])
    return pool


@pytest.mark.parametrize("case", ATTENTION_CASES, ids=[c["name"] for c in ATTENTION_CASES])
def test_golden_cases_all_modes(case, golden_dir):
    """Every golden case through decode(): auto, single-pass and two split-K chunkings."""
    gold = np.load(os.path.join(golden_dir, "attention_cases.npz"))[case["name"] + "/out"]
    q, k_pool, v_pool, table = attention_case_inputs(case)
    T = case["num_tokens"]
    with pool_from_host(k_pool, v_pool) as pool:
        tabs = table[None, :]
        ctx = np.array([T], np.int32)
        for flags, chunk in ((nv.AUTO, 0), (nv.SINGLE_PASS, 0), (nv.SPLIT_K, 8), (nv.SPLIT_K, 64)):
            out = nv.decode(pool, 0, q[None], tabs, ctx, flags, chunk)[0]
            assert_close(out, gold, f"{case['name']} flags={flags} chunk={chunk}")


@pytest.mark.parametrize("case", ATTENTION_CASES[:12], ids=[c["name"] for c in ATTENTION_CASES[:12]])
def test_drop_in_signature(case, golden_dir):
    """nvk_paged_attention with the reference's eleven arguments: host slabs (unbound) and bound pool."""
    gold = np.load(os.path.join(golden_dir, "attention_cases.npz"))[case["name"] + "/out"]
    q, k_pool, v_pool, table = attention_case_inputs(case)
    H, KV, hd, bs, T = (case[k] for k in ("n_heads", "n_kv_heads", "head_dim", "block_size", "num_tokens"))
    scratch = np.empty(H * max(T, 1), np.float32)

    nv.bind_pool(None)
    out = np.full((H, hd), np.nan, np.float32)
    nv.paged_attention(out, q, k_pool, v_pool, table, scratch, T, bs, hd, H, KV)
    assert_close(out, gold, "unbound")

    with pool_from_host(k_pool, v_pool, n_layers=3, layer=2) as pool:
        nv.bind_pool(pool, 2)
        try:
            out = np.full((H, hd), np.nan, np.float32)
            nv.paged_attention(out, q, None, None, table, None, T, bs, hd, H, KV)
            assert_close(out, gold, "bound")
            with pytest.raises(nv.NvkError):  # geometry mismatch is an error, not UB
                nv.paged_attention(out, q, None, None, table, None, T, bs + 1, hd, H, KV)
        finally:
            nv.bind_pool(None)


@pytest.mark.parametrize("case", [c for c in ATTENTION_CASES if c["num_tokens"] > 0][:16],
                         ids=[c["name"] for c in ATTENTION_CASES if c["num_tokens"] > 0][:16])
def test_standard_attention_drop_in(case, golden_dir):
    """nvk_standard_attention with the reference's ten arguments over a contiguous cache (golden F3:
    the reference's standard_attention on the same rows, which equals its paged_attention bit for bit)."""
    gold = np.load(os.path.join(golden_dir, "attention_cases.npz"))[case["name"] + "/std"]
    q, k_pool, v_pool, table = attention_case_inputs(case)
    H, KV, hd, bs, T = (case[k] for k in ("n_heads", "n_kv_heads", "head_dim", "block_size", "num_tokens"))
    max_seq = T + 5
    k_lin = np.full((max_seq, KV, hd), np.nan, np.float32)  # rows past pos must never be read
    v_lin = np.full((max_seq, KV, hd), np.nan, np.float32)
    for t in range(T):
        k_lin[t] = k_pool[table[t // bs], t % bs]
        v_lin[t] = v_pool[table[t // bs], t % bs]
    out = np.full((H, hd), np.nan, np.float32)
    nv.standard_attention(out, q, k_lin, v_lin, None, T - 1, hd, H, KV, max_seq)
    assert_close(out, gold, "standard")
    out2 = np.full((H, hd), np.nan, np.float32)
    nv.standard_attention(out2, q, k_lin, v_lin, None, -1, hd, H, KV, max_seq)
    assert not out2.any()
    with pytest.raises(nv.NvkError):
        nv.standard_attention(out2, q, k_lin, v_lin, None, max_seq, hd, H, KV, max_seq)


def test_gather_is_bit_exact():
    """Block-table indexing and KV gather: identical bytes to the closed form of attention.hpp:109-115."""
    case = next(c for c in ATTENTION_CASES if c["name"].startswith("h12_kv3_d64_bs12"))  # non power-of-two block
    for case in (case, ATTENTION_CASES[4], ATTENTION_CASES[8]):
        q, k_pool, v_pool, table = attention_case_inputs(case)
        T, bs = case["num_tokens"], case["block_size"]
        with pool_from_host(k_pool, v_pool) as pool:
            for kvh in range(case["n_kv_heads"]):
                k, v = pool.gather(0, table, T, kvh)
                want_k = np.stack([k_pool[table[t // bs], t % bs, kvh] for t in range(T)])
                want_v = np.stack([v_pool[table[t // bs], t % bs, kvh] for t in range(T)])
                assert np.array_equal(k.view(np.uint32), want_k.view(np.uint32))
                assert np.array_equal(v.view(np.uint32), want_v.view(np.uint32))


def test_device_fill_matches_generator():
    g = wl.Geometry(8, 2, 64, 16)
    with nv.PagedKVPool(2, 37, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(1, 11, 22)
        for pb in (0, 5, 36):
            k, v = pool.download_block(1, pb)
            assert np.array_equal(k.reshape(-1), wl.uniform_block(11, pb, g.block_floats))
            assert np.array_equal(v.reshape(-1), wl.uniform_block(22, pb, g.block_floats))
            assert np.array_equal(k.reshape(-1), ol.oracle_fill_uniform(11, pb * g.block_floats, g.block_floats))
        k0, _ = pool.download_block(0, 0)
        assert not k0.any()  # other layers stay zero-initialised (model.hpp:267-268)


def test_append_then_attend_like_the_model_loop():
    """KV append + attention in the order of LlamaModel::forward (model.hpp:178-221): token pos attends to itself."""
    H, KV, hd, bs, L = 8, 2, 64, 16, 2
    nb = 12
    rng = np.random.default_rng(5)
    bm = nv.BlockManager(nb, bs)
    host_k = np.zeros((L, nb, bs, KV, hd), np.float32)
    host_v = np.zeros_like(host_k)
    tables = [[] for _ in range(L)]
    with nv.PagedKVPool(L, nb, bs, KV, hd) as pool:
        for pos in range(40):
            for layer in range(L):
                if pos % bs == 0:
                    tables[layer].append(bm.allocate_block())
                k = rng.standard_normal((KV, hd), dtype=np.float32)
                v = rng.standard_normal((KV, hd), dtype=np.float32)
                q = rng.standard_normal((H, hd), dtype=np.float32)
                pb, off = tables[layer][pos // bs], pos % bs
                pool.append(layer, pb, off, k, v)
                host_k[layer, pb, off], host_v[layer, pb, off] = k, v
                tab = np.asarray(tables[layer], np.int32)
                got = nv.decode(pool, layer, q[None], tab[None], np.array([pos + 1], np.int32))[0]
                want = ol.oracle_paged_attention(q, host_k[layer], host_v[layer], tab, pos + 1, bs, hd, H, KV)
                assert_close(got, want, f"pos {pos} layer {layer}")
        assert tables[0] == [0, 2, 4] and tables[1] == [1, 3, 5]  # shared free list interleaves layers


def test_append_batch_matches_per_row_append():
    H, KV, hd, bs = 4, 4, 32, 8
    n, nb = 9, 40
    rng = np.random.default_rng(6)
    tabs = rng.permutation(nb)[: n * 4].reshape(n, 4).astype(np.int32)
    pos = rng.integers(0, 4 * bs, n).astype(np.int32)
    k = rng.standard_normal((n, KV * hd), dtype=np.float32)
    v = rng.standard_normal((n, KV * hd), dtype=np.float32)
    with nv.PagedKVPool(1, nb, bs, KV, hd) as a, nv.PagedKVPool(1, nb, bs, KV, hd) as b:
        a.append_batch(0, k, v, tabs, pos)
        for s in range(n):
            b.append(0, int(tabs[s, pos[s] // bs]), int(pos[s] % bs), k[s], v[s])
        for pb in range(nb):
            ka, va = a.download_block(0, pb)
            kb, vb = b.download_block(0, pb)
            assert np.array_equal(ka, kb) and np.array_equal(va, vb)
        bad = tabs.copy()
        bad[3, pos[3] // bs] = nb
        with pytest.raises(nv.NvkError) as e:
            a.append_batch(0, k, v, bad, pos)
        assert e.value.code == nv._capi.ERR_BAD_BLOCK


def test_ragged_batch_against_checker():
    """A small ragged, fragmented batch (config-3 shape in miniature), every sequence checked."""
    g = wl.Geometry(32, 8, 128, 16)
    ctx = np.array([1, 7, 16, 17, 100, 255, 256, 257, 700, 1, 33, 513, 0, 64], np.int32)
    work = wl.fragmented_workload("mini_ragged", ctx, g)
    q = work.q()
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(work.num_seqs)))
        want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads)
        for flags, chunk in ((nv.AUTO, 0), (nv.SINGLE_PASS, 0), (nv.SPLIT_K, 128), (nv.SPLIT_K, 40)):
            got = nv.decode(pool, 0, q, work.block_tables, work.ctx_lens, flags, chunk)
            assert_close(got, want, f"flags={flags} chunk={chunk}")
        assert not got[12].any()  # ctx 0: the reference zeroes out and does nothing else (attention.hpp:98)


def test_bad_block_table_is_rejected_not_dereferenced():
    with nv.PagedKVPool(1, 8, 16, 2, 64) as pool:
        q = np.zeros((1, 4, 64), np.float32)
        with pytest.raises(nv.NvkError) as e:
            nv.decode(pool, 0, q, np.array([[0, 8]], np.int32), np.array([20], np.int32))
        assert e.value.code == nv._capi.ERR_BAD_BLOCK
        with pytest.raises(nv.NvkError):
            nv.decode(pool, 0, q, np.array([[0, -1]], np.int32), np.array([20], np.int32))
        with pytest.raises(nv.NvkError):  # table row too short for the context
            nv.decode(pool, 0, q, np.array([[0, 1]], np.int32), np.array([40], np.int32))
        with pytest.raises(nv.NvkError):  # n_heads not a multiple of n_kv_heads
            nv.decode(pool, 0, np.zeros((1, 3, 64), np.float32), np.array([[0, 1]], np.int32),
                      np.array([20], np.int32))


def test_scores_below_reference_floor_give_reference_nans():
    """max_val starts at -1e10 (attention.hpp:126): if every score is below it the reference divides 0 by 0."""
    H, KV, hd, bs, T = 2, 1, 64, 16, 5
    k_pool = np.full((1, bs, KV, hd), -1.0e6, np.float32)
    v_pool = np.ones((1, bs, KV, hd), np.float32)
    q = np.full((H, hd), 1.0e6, np.float32)
    q[1] = 1.0  # head 1 stays in the normal range
    table = np.array([0], np.int32)
    want = ol.oracle_paged_attention(q, k_pool, v_pool, table, T, bs, hd, H, KV)
    assert np.isnan(want[0]).all() and not np.isnan(want[1]).any()
    with pool_from_host(k_pool, v_pool) as pool:
        for flags, chunk in ((nv.SINGLE_PASS, 0), (nv.SPLIT_K, 8)):
            got = nv.decode(pool, 0, q[None], table[None], np.array([T], np.int32), flags, chunk)[0]
            assert_close(got, want, f"flags={flags}")


def test_online_softmax_rescale_is_forced():
    """A late, dominant key makes the running max jump mid-stream and across chunk boundaries."""
    H, KV, hd, bs, T = 4, 1, 128, 16, 400
    rng = np.random.default_rng(9)
    nb = T // bs
    k_pool = (0.05 * rng.standard_normal((nb, bs, KV, hd))).astype(np.float32)
    v_pool = rng.standard_normal((nb, bs, KV, hd)).astype(np.float32)
    q = rng.standard_normal((H, hd)).astype(np.float32)
    for spike in (3, 130, 399):  # first step, a middle chunk, the very last token
        k2 = k_pool.copy()
        k2[spike // bs, spike % bs, 0] = 4.0 * q[0]  # score ~ +45 for head 0 only
        table = np.arange(nb, dtype=np.int32)
        want = ol.oracle_paged_attention(q, k2, v_pool, table, T, bs, hd, H, KV)
        with pool_from_host(k2, v_pool) as pool:
            for flags, chunk in ((nv.SINGLE_PASS, 0), (nv.SPLIT_K, 64), (nv.SPLIT_K, 8)):
                got = nv.decode(pool, 0, q[None], table[None], np.array([T], np.int32), flags, chunk)[0]
                assert_close(got, want, f"spike {spike} flags={flags} chunk={chunk}")


def _full_size_check(work, sample, modes):
    g = work.geom
    q = work.q()
    k_host, v_host, tabs, c = wl.host_blocks_for(work, sample)
    want = ol.oracle_decode_batch(q[sample], k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=8)
    outs = []
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for flags, chunk in modes:
            got = nv.decode(pool, 0, q, work.block_tables, work.ctx_lens, flags, chunk)
            assert_close(got[sample], want, f"{work.name} flags={flags} chunk={chunk}")
            assert np.isfinite(got).all()
            outs.append(got)
        # size-independent property at full size: a softmax-weighted mean of V stays inside V's range
        assert np.abs(outs[0]).max() <= 1.0
        # and the split-K and single-pass answers agree everywhere, not just on the sampled sequences
        for o in outs[1:]:
            assert_close(o, outs[0], f"{work.name} mode agreement")
        # permutation invariance: the same data behind a different physical placement gives the same bits
        return outs[0]


def test_config2_full_size():
    """BASELINE.json configs[1]: 64 seqs x 2K ctx, uniform tables, 1 GiB of KV."""
    work = wl.config2()
    assert work.algorithmic_bytes() == 1_075_871_744
    _full_size_check(work, [0, 1, 31, 62, 63], [(nv.AUTO, 0), (nv.SINGLE_PASS, 0), (nv.SPLIT_K, 256)])


def test_config3_ragged_fragmented_full_size():
    """configs[2]: 256 seqs, ctx 128..8192, fragmented tables."""
    work = wl.config3()
    order = np.argsort(work.ctx_lens)
    sample = sorted({int(order[0]), int(order[1]), int(order[len(order) // 2]), int(order[-1]), 0, 255})
    _full_size_check(work, sample, [(nv.AUTO, 0), (nv.SPLIT_K, 512)])


def test_config4_long_context_split_k():
    """configs[3]: 16 seqs x 32K ctx; split-K against single-pass and the checker."""
    work = wl.config4()
    _full_size_check(work, [0, 15], [(nv.AUTO, 0), (nv.SINGLE_PASS, 0), (nv.SPLIT_K, 1024)])


def test_config5a_many_short_sequences():
    """configs[4] (one GPU's share of 5a): 512 short sequences, ctx 64..192."""
    work = wl.config5a(512)
    _full_size_check(work, list(range(0, 512, 37)), [(nv.AUTO, 0), (nv.SINGLE_PASS, 0)])


def test_config5b_one_gpu_share_full_size():
    """configs[4], the bandwidth-scaling point (5b): one GPU's 512 sequences x 2K context, 8 GiB of KV."""
    work = wl.config5b(512)
    assert work.total_tokens == 512 * 2048
    _full_size_check(work, [0, 200, 511], [(nv.AUTO, 0), (nv.SPLIT_K, 512)])


def test_block_placement_does_not_change_bits():
    """Paged == contiguous: moving the same rows to other physical blocks leaves every output bit unchanged."""
    case = ATTENTION_CASES[8]  # 32/8/128, 2048 tokens
    q, k_pool, v_pool, table = attention_case_inputs(case)
    T, bs = case["num_tokens"], case["block_size"]
    nb = k_pool.shape[0]
    ident = np.arange((T + bs - 1) // bs, dtype=np.int32)
    k_lin = np.zeros_like(k_pool)
    v_lin = np.zeros_like(v_pool)
    k_lin[: ident.size] = k_pool[table[: ident.size]]
    v_lin[: ident.size] = v_pool[table[: ident.size]]
    ctx = np.array([T], np.int32)
    with pool_from_host(k_pool, v_pool) as a, pool_from_host(k_lin, v_lin) as b:
        for flags, chunk in ((nv.SINGLE_PASS, 0), (nv.SPLIT_K, 256)):
            oa = nv.decode(a, 0, q[None], table[None], ctx, flags, chunk)
            ob = nv.decode(b, 0, q[None], ident[None], ctx, flags, chunk)
            assert np.array_equal(oa.view(np.uint32), ob.view(np.uint32))
    assert nb > ident.size


def test_plan_reused_across_layers_matches_per_call_planning():
    """One nvk_plan built from the batch's lengths serves every layer (device-pointer API)."""
    from hipmem import DeviceArray

    g = wl.Geometry(8, 2, 64, 16)
    ctx = np.array([5, 300, 0, 64, 1000, 17, 129], np.int32)
    work = wl.fragmented_workload("plan_reuse", ctx, g)
    L = 3
    with nv.PagedKVPool(L, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        for layer in range(L):
            pool.fill_uniform(layer, 10 + layer, 20 + layer)
        q = work.q()
        qd, tabs, ctxd = DeviceArray(q), DeviceArray(work.block_tables), DeviceArray(ctx)
        plan = nv.DecodePlan(pool).build(work.num_seqs, g.n_heads, ctxd, work.max_ctx, nv.SPLIT_K, 128)
        assert plan.info()["chunk_tokens"] == 128 and plan.info()["max_chunks"] == 8
        for layer in range(L):
            out = DeviceArray(np.full_like(q, np.nan))
            plan.run(layer, qd, tabs, work.max_blocks, out)
            pool.synchronize()
            got = out.numpy()
            want = nv.decode(pool, layer, q, work.block_tables, ctx, nv.SPLIT_K, 128)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))  # same plan, same bits
            single = nv.decode(pool, layer, q, work.block_tables, ctx, nv.SINGLE_PASS)
            assert_close(got, single, f"layer {layer}")
        plan.close()


def test_many_sequences_cross_plan_tiles():
    """More sequences than the plan kernel has threads (several scan tiles), ragged, with empty ones."""
    g = wl.Geometry(4, 2, 32, 8)
    n = 2500
    ctx = (wl.splitmix64(31, np.arange(n, dtype=np.uint64)) % np.uint64(70)).astype(np.int32)
    ctx[::97] = 0
    work = wl.fragmented_workload("many", ctx, g, slack=0.1)
    q = work.q()
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(n)))
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=8)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for flags, chunk in ((nv.AUTO, 0), (nv.SPLIT_K, 16), (nv.SPLIT_K, 24), (nv.SINGLE_PASS, 0)):
            got = nv.decode(pool, 0, q, work.block_tables, work.ctx_lens, flags, chunk)
            assert_close(got, want, f"flags={flags} chunk={chunk}")
            assert not got[ctx == 0].any()


def test_head_sharded_pools_reproduce_the_full_result():
    """SURVEY.md 8e, head-sharded variant, emulated on one GPU: G pools of n_kv_heads/G kv-heads each, the rank
    slices of q, and the gathered layout [G, n, H/G, hd] assembled back, against the one-pool result and the
    checker (the all-gather itself is covered over gloo in tests/test_sharding_gloo.py).  A shard with fewer
    than 8 kv-heads splits each chunk's tokens over its spare waves, so sums are associated differently from
    the one-pool run: equal to a few ulp, not bit for bit."""
    from nano_vllm_amd import sharding as sh

    g = wl.Geometry(16, 8, 128, 16)
    ctx = np.array([700, 1, 64, 2049, 0, 130], np.int32)
    work = wl.fragmented_workload("head_shards", ctx, g)
    q = work.q()
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(work.num_seqs)))
    nb = k_host.shape[0]
    with pool_from_host(k_host, v_host) as full_pool:
        full = nv.decode(full_pool, 0, q, tabs, c, nv.SPLIT_K, 512)
    for world in (2, 8):
        parts = []
        for rank in range(world):
            hs = sh.HeadShard(rank, world, g.n_heads, g.n_kv_heads)
            with nv.PagedKVPool(1, nb, g.block_size, hs.local_kv_heads, g.head_dim) as pool:
                for b in range(nb):
                    pool.upload_block(0, b, hs.local_kv_block(k_host[b]), hs.local_kv_block(v_host[b]))
                parts.append(nv.decode(pool, 0, hs.local_q(q), tabs, c, nv.SPLIT_K, 512))
        got = sh.assemble_heads(np.stack(parts))
        assert_close(got, full, f"world {world}")
        assert np.abs(got - full).max() <= 2e-6
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads)
    assert_close(got, want, "vs checker")


@pytest.mark.parametrize("geom", [(6, 6, 48, 16), (32, 8, 128, 16), (8, 2, 64, 32), (4, 4, 20, 8)],
                         ids=["g_tiny", "g_llama", "h8kv2d64", "d20"])
def test_rope_append_batch_matches_reference_rope_then_append(geom):
    """Ops::apply_rope (positional.hpp:13-39) + the append (model.hpp:190-207) in one kernel.
    Rotated q and the K/V rows that land in the pool are compared with the checker; then attention over them."""
    H, KV, hd, bs = geom
    n, max_blocks = 13, 6
    nb = n * max_blocks + 5
    tabs = wl.shuffled_permutation(nb, 17)[: n * max_blocks].reshape(n, max_blocks).astype(np.int32)
    pos = (wl.splitmix64(5, np.arange(n, dtype=np.uint64)) % np.uint64(max_blocks * bs)).astype(np.int32)
    pos[0], pos[1] = 0, max_blocks * bs - 1
    q = (wl.uniform(31, 0, n * H * hd) * np.float32(2)).reshape(n, H, hd)
    k = (wl.uniform(32, 0, n * KV * hd) * np.float32(2)).reshape(n, KV, hd)
    v = wl.uniform(33, 0, n * KV * hd).reshape(n, KV, hd)
    for theta in (10000.0, 500000.0):
        with nv.PagedKVPool(2, nb, bs, KV, hd) as pool:
            q_rot = pool.rope_append_batch(1, q, k, v, tabs, pos, theta)
            exact = total = 0
            for s in range(n):
                want_q, want_k = ol.apply_rope(ol.ORACLE.oracle_apply_rope, q[s], k[s], int(pos[s]), hd, H, KV, theta)
                kb, vb = pool.download_block(1, int(tabs[s, pos[s] // bs]))
                got_k, got_v = kb[pos[s] % bs], vb[pos[s] % bs]
                assert np.array_equal(got_v.view(np.uint32), v[s].view(np.uint32))  # V is a plain copy: bit-exact
                assert_close(q_rot[s], want_q, f"q seq {s} pos {pos[s]}")
                assert_close(got_k, want_k, f"k seq {s} pos {pos[s]}")
                # far tighter than the bar in practice: cos/sin differ from libm's by at most one rounding
                assert np.abs(q_rot[s] - want_q).max() <= 4e-7 * max(1.0, float(np.abs(want_q).max()))
                exact += int((q_rot[s].view(np.uint32) == want_q.view(np.uint32)).sum())
                total += want_q.size
            assert exact >= 0.98 * total, (exact, total)
            k0, _ = pool.download_block(0, int(tabs[0, 0]))
            assert not k0.any()  # the other layer is untouched


def test_rope_append_then_decode_like_a_batched_model_step():
    """positions advance token by token for a small batch: rope+append, then decode, against the checker's
    own rope -> append -> paged_attention on a host pool (the order of LlamaModel::forward, model.hpp:169-221)."""
    H, KV, hd, bs = 8, 2, 64, 16
    n, steps = 5, 37
    max_blocks = (steps + bs - 1) // bs
    nb = n * max_blocks
    tabs = wl.shuffled_permutation(nb, 3).reshape(n, max_blocks).astype(np.int32)
    host_k = np.zeros((nb, bs, KV, hd), np.float32)
    host_v = np.zeros_like(host_k)
    rng = np.random.default_rng(12)
    with nv.PagedKVPool(1, nb, bs, KV, hd) as pool:
        for t in range(steps):
            q = rng.standard_normal((n, H, hd), dtype=np.float32)
            k = rng.standard_normal((n, KV, hd), dtype=np.float32)
            v = rng.standard_normal((n, KV, hd), dtype=np.float32)
            pos = np.full(n, t, np.int32)
            q_rot = pool.rope_append_batch(0, q, k, v, tabs, pos)
            got = nv.decode(pool, 0, q_rot, tabs, pos + 1)
            for s in range(n):
                wq, wk = ol.apply_rope(ol.ORACLE.oracle_apply_rope, q[s], k[s], t, hd, H, KV)
                pb = int(tabs[s, t // bs])
                host_k[pb, t % bs], host_v[pb, t % bs] = wk, v[s]
                want = ol.oracle_paged_attention(wq, host_k, host_v, tabs[s], t + 1, bs, hd, H, KV)
                assert_close(got[s], want, f"step {t} seq {s}")


@pytest.mark.parametrize("seed", range(36))
def test_random_geometry_batches_against_checker(seed):
    """Differential sweep: random head geometry (every VPL / QPK / token-split / kv-group combination), block size,
    ragged lengths and fragmented tables, each in auto, single-pass and a random split-K chunking."""
    d = wl.splitmix64(4200 + seed, np.arange(12, dtype=np.uint64)).tolist()
    KV = [1, 2, 3, 4, 5, 6, 8, 12, 16][d[0] % 9]
    H = KV * [1, 2, 3, 4, 5, 6, 7, 8, 16][(d[1] + seed) % 9]  # every q-heads-per-wave count, and two q sub-groups
    hd = [4, 16, 20, 48, 64, 80, 96, 128, 132, 192, 256][d[2] % 11]
    bs = [1, 4, 8, 12, 16, 32][d[3] % 6]
    n = 1 + d[4] % 9
    top = [40, 300, 1500][d[5] % 3]
    ctx = (wl.splitmix64(seed, np.arange(n, dtype=np.uint64)) % np.uint64(top)).astype(np.int32)
    ctx[d[6] % n] = top  # at least one sequence at the top of the range
    g = wl.Geometry(H, KV, hd, bs)
    work = wl.fragmented_workload(f"fuzz{seed}", ctx, g, slack=0.3, perm_seed=99 + seed)
    q = work.q() * np.float32(1 + d[7] % 5)
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(n)))
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, bs, KV, n_threads=4)
    chunk = [8, 24, 64, 200, 512][d[8] % 5]
    with nv.PagedKVPool(1, work.num_blocks, bs, KV, hd) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for flags, ch in ((nv.AUTO, 0), (nv.SINGLE_PASS, 0), (nv.SPLIT_K, chunk)):
            got = nv.decode(pool, 0, q, work.block_tables, work.ctx_lens, flags, ch)
            assert_close(got, want, f"H{H} KV{KV} hd{hd} bs{bs} ctx{ctx.tolist()} flags={flags} chunk={ch}")


def test_planned_decode_is_capturable_in_a_hip_graph():
    """The per-layer entry point allocates nothing and synchronises nothing, so the layer loop of a model step
    can be captured once and replayed: plan build + L x (rope+append, attention) as one hipGraph."""
    from hipmem import DeviceArray, Stream, _hip

    g = wl.Geometry(8, 2, 64, 16)
    L = 3
    ctx = np.array([300, 17, 1000, 64, 129], np.int32)
    work = wl.fragmented_workload("graph", ctx + 1, g)  # room for the token the step appends
    n = work.num_seqs
    rng = np.random.default_rng(8)
    q = rng.standard_normal((n, g.n_heads, g.head_dim), dtype=np.float32)
    k = rng.standard_normal((n, g.row_len), dtype=np.float32)
    v = rng.standard_normal((n, g.row_len), dtype=np.float32)

    def run(use_graph):
        with nv.PagedKVPool(L, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
            for layer in range(L):
                pool.fill_uniform(layer, 10 + layer, 20 + layer)
            st = Stream()
            qd = [DeviceArray(q) for _ in range(L)]  # rope rotates q in place: one buffer per layer
            kd, vd = DeviceArray(k), DeviceArray(v)
            tabs, pos, lens = DeviceArray(work.block_tables), DeviceArray(ctx), DeviceArray(ctx + 1)
            outs = [DeviceArray(np.full_like(q, np.nan)) for _ in range(L)]
            plan = nv.DecodePlan(pool)
            pool.reserve_workspace(n, g.n_heads, work.max_ctx)
            plan.build(n, g.n_heads, lens, work.max_ctx, stream=st)  # sizes the plan: the captured rebuild allocates nothing
            st.synchronize()

            def step():
                plan.build(n, g.n_heads, lens, work.max_ctx, stream=st)
                for layer in range(L):
                    pool.rope_append_batch_device(layer, n, g.n_heads, qd[layer], kd, vd, tabs, work.max_blocks, pos,
                                                  stream=st)
                    plan.run(layer, qd[layer], tabs, work.max_blocks, outs[layer], stream=st)

            if use_graph:
                exe = st.capture(step)
                assert all(np.isnan(o.numpy()).all() for o in outs)  # capture records, it does not run
                st.launch(exe)
                st.synchronize()
                _hip.hipGraphExecDestroy(exe)
            else:
                step()
                st.synchronize()
            res = [o.numpy() for o in outs]
            plan.close()
            st.close()
            return res

    eager, graphed = run(False), run(True)
    for layer in range(L):
        assert np.isfinite(eager[layer]).all()
        assert np.array_equal(eager[layer].view(np.uint32), graphed[layer].view(np.uint32)), layer


def test_pool_beyond_int32_offsets():
    """The reference computes pool offsets in int (attention.hpp:114-115, model.hpp:196-198) and is only defined
    below 2^31 floats per slab; the device pool is addressed in 64 bits.  Two layers of 9.2 GB each per K and V,
    sequences placed in blocks whose float offsets exceed 2^31 and, in layer 1, 2^32."""
    g = wl.G_LLAMA
    nb = 140_000  # 140000 * 16 * 1024 floats = 2.29e9 floats per slab > 2^31
    top = [139_999, 131_073, 135_000, 131_072, 139_000, 132_000, 137_777, 138_888]
    ctx = np.array([50, 64], np.int32)
    tabs = np.array([top[:4], top[4:]], np.int32)
    assert int(tabs.min()) * g.block_floats >= 2 ** 31  # every block used lies beyond what an int offset reaches
    q = wl.uniform(3, 0, 2 * g.n_heads * g.head_dim).reshape(2, g.n_heads, g.head_dim)
    with nv.PagedKVPool(2, nb, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(1, 5, 6)
        work = wl.Workload("big", g, nb, ctx, tabs)
        k_host = np.stack([wl.uniform_block(5, int(pb), g.block_floats).reshape(g.block_size, g.n_kv_heads, g.head_dim)
                           for pb in tabs.reshape(-1)])
        v_host = np.stack([wl.uniform_block(6, int(pb), g.block_floats).reshape(g.block_size, g.n_kv_heads, g.head_dim)
                           for pb in tabs.reshape(-1)])
        local = np.arange(8, dtype=np.int32).reshape(2, 4)
        want = ol.oracle_decode_batch(q, k_host, v_host, local, ctx, g.block_size, g.n_kv_heads)
        for flags, chunk in ((nv.AUTO, 0), (nv.SPLIT_K, 16)):
            got = nv.decode(pool, 1, q, work.block_tables, ctx, flags, chunk)
            assert_close(got, want, f"flags={flags}")
        kb, _ = pool.download_block(1, 139_999)
        assert np.array_equal(kb.reshape(-1), wl.uniform_block(5, 139_999, g.block_floats))
        k0, _ = pool.download_block(0, 139_999)
        assert not k0.any()  # layer 0 untouched: the layer offset is 64-bit too
        # an appended row lands where the 64-bit arithmetic says
        krow = np.full(g.row_len, 7.0, np.float32)
        pool.append(1, 139_999, 15, krow, -krow)
        kb, vb = pool.download_block(1, 139_999)
        assert np.array_equal(kb[15].reshape(-1), krow) and np.array_equal(vb[15].reshape(-1), -krow)


def test_two_pools_driven_from_two_host_threads():
    """Distinct pools may be driven concurrently from different host threads (include/nvk_paged_attn.h, threading)."""
    import threading

    g = wl.Geometry(8, 2, 64, 16)
    ctx = np.array([300, 17, 1000, 64, 129, 0, 511], np.int32)
    results, errors = {}, []

    def worker(tid):
        try:
            work = wl.fragmented_workload(f"thread{tid}", ctx + tid, g, perm_seed=50 + tid)
            q = work.q()
            k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(work.num_seqs)))
            with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
                pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
                for _ in range(20):
                    got = nv.decode(pool, 0, q, work.block_tables, work.ctx_lens, nv.SPLIT_K, 64)
            want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads)
            results[tid] = (got, want)
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for tid, (got, want) in results.items():
        assert_close(got, want, f"thread {tid}")


def test_layers_and_slabs_of_a_pool_do_not_overlap():
    """The pool is one allocation, [layer][K slab | V slab] (the reference keeps two arrays of [layer][slab],
    include/core/model.hpp:95,262-265; where the slabs lie is the library's business).  Every (layer, K/V, block)
    is its own storage: distinct contents written to all of them come back unchanged, the per-layer addresses are
    disjoint ranges of one slab each, and attention on one layer sees that layer's rows only."""
    L, nb, bs, KV, hd, H = 3, 5, 4, 2, 16, 4
    rng = np.random.default_rng(12)
    blocks = rng.standard_normal((L, 2, nb, bs, KV, hd)).astype(np.float32)
    slab = nb * bs * KV * hd * 4
    with nv.PagedKVPool(L, nb, bs, KV, hd) as pool:
        for layer in range(L):
            for b in range(nb):
                pool.upload_block(layer, b, blocks[layer, 0, b], blocks[layer, 1, b])
        spans = []
        for layer in range(L):
            k_ptr, v_ptr = pool.layer_ptrs(layer)
            spans += [(k_ptr, k_ptr + slab), (v_ptr, v_ptr + slab)]
            for b in range(nb):
                k, v = pool.download_block(layer, b)
                assert np.array_equal(k, blocks[layer, 0, b]) and np.array_equal(v, blocks[layer, 1, b])
        spans.sort()
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), spans
        q = rng.standard_normal((1, H, hd)).astype(np.float32)
        table = np.arange(nb, dtype=np.int32)[None]
        T = nb * bs - 1
        for layer in range(L):
            want = ol.oracle_paged_attention(q[0], blocks[layer, 0], blocks[layer, 1], table[0], T, bs, hd, H, KV)
            got = nv.decode(pool, layer, q, table, np.array([T], np.int32), nv.AUTO)[0]
            assert_close(got, want, f"layer {layer}")
