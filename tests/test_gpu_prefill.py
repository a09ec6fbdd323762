"""paged chunked-prefill attention and the chunk append, through the C ABI.

Oracle: golden F6 (the reference's LlamaModelChunked::chunked_attention on a contiguous cache, recorded by
oracle/make_golden.py) with the same rows placed behind a shuffled block table, and the checker's
per-query paged attention for multi-sequence batches.  Tolerance as for decode: 1e-4 abs + 1e-5 rel."""
import os

import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from golden_defs import PREFILL_CASES, paged_from_contiguous, prefill_case_inputs
from nano_vllm_amd import workloads as wl

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-4, 1e-5
MODES = [nv.PREFILL_ROWWISE, nv.PREFILL_TILED, nv.PREFILL_AUTO]


def assert_close(got, want, what=""):
    assert got.shape == want.shape, what
    ok = np.abs(got - want) <= ATOL + RTOL * np.abs(want)
    assert ok.all(), f"{what}: max abs err {np.abs(got - want).max():.3e} at {np.argwhere(~ok)[:3].tolist()}"


def pool_from_host(k_pool, v_pool):
    nb, bs, kv, hd = k_pool.shape
    pool = nv.PagedKVPool(1, nb, bs, kv, hd)
    for b in range(nb):
        pool.upload_block(0, b, k_pool[b], v_pool[b])
    return pool


@pytest.mark.parametrize("case", PREFILL_CASES, ids=[c["name"] for c in PREFILL_CASES])
def test_golden_chunked_prefill_cases(case, golden_dir):
    gold = np.load(os.path.join(golden_dir, "prefill_cases.npz"))[case["name"] + "/out"]
    q, k_lin, v_lin = prefill_case_inputs(case)
    for bs in (16, 8):
        k_pool, v_pool, table = paged_from_contiguous(k_lin, v_lin, bs, 8100 + case["index"])
        with pool_from_host(k_pool, v_pool) as pool:
            for flags in MODES:
                out = nv.prefill(pool, 0, q, table[None, :], [case["start_pos"]], [case["chunk_size"]], flags)
                assert_close(out, gold, f"{case['name']} bs={bs} flags={flags}")


def test_ragged_multi_sequence_chunks():
    """Several sequences with different starts and chunk lengths (incl. empty chunks and a 1-token chunk,
    which is exactly a decode step) in one call, fragmented tables."""
    g = wl.Geometry(8, 2, 64, 16)
    starts = np.array([0, 100, 37, 0, 511, 64, 5], np.int32)
    lens = np.array([70, 33, 0, 1, 1, 128, 40], np.int32)
    work = wl.fragmented_workload("prefill_ragged", starts + lens, g)
    k_host, v_host, tabs, _ = wl.host_blocks_for(work, list(range(work.num_seqs)))
    total = int(lens.sum())
    q = wl.uniform(77, 0, total * g.n_heads * g.head_dim).reshape(total, g.n_heads, g.head_dim)
    want = np.empty_like(q)
    r = 0
    for s in range(starts.size):
        if lens[s]:
            want[r:r + lens[s]] = ol.oracle_paged_prefill(q[r:r + lens[s]], k_host, v_host, tabs[s], int(starts[s]),
                                                          g.block_size, g.n_kv_heads)
        r += int(lens[s])
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for flags in MODES:
            got = nv.prefill(pool, 0, q, work.block_tables, starts, lens, flags)
            assert_close(got, want, f"flags={flags}")
        with pytest.raises(nv.NvkError):  # chunk runs past its table row
            nv.prefill(pool, 0, q, work.block_tables, starts + work.max_blocks * 16, lens, nv.PREFILL_ROWWISE)
        bad = work.block_tables.copy()
        bad[0, 1] = work.num_blocks
        with pytest.raises(nv.NvkError) as e:
            nv.prefill(pool, 0, q, bad, starts, lens)
        assert e.value.code == nv._capi.ERR_BAD_BLOCK


def test_chunk_append_then_prefill_equals_token_by_token_decode():
    """forward_chunk's order (model_chunked.hpp:79-96): rope + append the whole chunk, then chunked attention --
    against appending and decoding the same tokens one at a time (the reference's own prefill, runner.hpp:39-43)."""
    H, KV, hd, bs = 8, 2, 64, 16
    chunks = [(0, 48), (48, 31), (79, 1), (80, 50)]  # one sequence, prompt fed in four chunks
    n_tok = 130
    nb = 12
    table = wl.shuffled_permutation(nb, 9)[: (n_tok + bs - 1) // bs].astype(np.int32)
    rng = np.random.default_rng(4)
    q = rng.standard_normal((n_tok, H, hd), dtype=np.float32)
    k = rng.standard_normal((n_tok, KV, hd), dtype=np.float32)
    v = rng.standard_normal((n_tok, KV, hd), dtype=np.float32)
    with nv.PagedKVPool(1, nb, bs, KV, hd) as a, nv.PagedKVPool(1, nb, bs, KV, hd) as b:
        step = np.empty_like(q)
        for t in range(n_tok):  # token at a time
            qr = a.rope_append_batch(0, q[t:t + 1], k[t:t + 1], v[t:t + 1], table[None, :], [t])
            step[t] = nv.decode(a, 0, qr, table[None, :], [t + 1])[0]
        for flags in MODES:
            chunked = np.empty_like(q)
            for start, n in chunks:
                qr = nv.rope_append_chunks(b, 0, q[start:start + n], k[start:start + n], v[start:start + n],
                                           table[None, :], [start], [n])
                chunked[start:start + n] = nv.prefill(b, 0, qr, table[None, :], [start], [n], flags)
            assert_close(chunked, step, f"flags={flags}")
        for pb in range(nb):  # the two pools hold the same bytes: chunk append == per-token append
            ka, va = a.download_block(0, pb)
            kb, vb = b.download_block(0, pb)
            assert np.array_equal(ka.view(np.uint32), kb.view(np.uint32)) and np.array_equal(va, vb)


@pytest.mark.parametrize("seed", range(12))
def test_random_geometry_prefill_against_checker(seed):
    """Differential sweep of the prefill forms over head geometry, block size, starts and chunk lengths."""
    d = wl.splitmix64(5200 + seed, np.arange(12, dtype=np.uint64)).tolist()
    KV = [1, 2, 3, 4, 6, 8][d[0] % 6]
    H = KV * [1, 2, 3, 4, 8][d[1] % 5]
    hd = [16, 20, 48, 64, 80, 128][d[2] % 6]
    bs = [4, 8, 16, 32][d[3] % 4]
    n = 1 + d[4] % 4
    starts = (wl.splitmix64(seed, np.arange(n, dtype=np.uint64)) % np.uint64(200)).astype(np.int32)
    lens = (wl.splitmix64(70 + seed, np.arange(n, dtype=np.uint64)) % np.uint64(150)).astype(np.int32)
    lens[d[5] % n] = 1 + d[6] % 140
    g = wl.Geometry(H, KV, hd, bs)
    work = wl.fragmented_workload(f"pfuzz{seed}", starts + lens, g, slack=0.3, perm_seed=7 + seed)
    k_host, v_host, tabs, _ = wl.host_blocks_for(work, list(range(n)))
    total = int(lens.sum())
    q = wl.uniform(900 + seed, 0, total * H * hd).reshape(total, H, hd) * np.float32(1 + d[7] % 4)
    want = np.empty_like(q)
    r = 0
    for s in range(n):
        if lens[s]:
            want[r:r + lens[s]] = ol.oracle_paged_prefill(q[r:r + lens[s]], k_host, v_host, tabs[s], int(starts[s]), bs, KV)
        r += int(lens[s])
    with nv.PagedKVPool(1, work.num_blocks, bs, KV, hd) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for flags in MODES:
            got = nv.prefill(pool, 0, q, work.block_tables, starts, lens, flags)
            assert_close(got, want, f"H{H} KV{KV} hd{hd} bs{bs} starts{starts.tolist()} lens{lens.tolist()} flags={flags}")


def test_long_context_small_batch_splits_keys():
    """One sequence deep into a long prompt: too few tiles to fill the chip, so key ranges are split over
    workgroups and merged (split-K prefill); checked against the checker and the row-wise form."""
    g = wl.Geometry(8, 2, 128, 16)
    for start, n in ((3000, 100), (0, 700), (5000, 1)):
        work = wl.fragmented_workload("prefill_long", np.array([start + n], np.int32), g)
        k_host, v_host, tabs, _ = wl.host_blocks_for(work, [0])
        q = wl.uniform(55, 0, n * g.n_heads * g.head_dim).reshape(n, g.n_heads, g.head_dim)
        want = ol.oracle_paged_prefill(q, k_host, v_host, tabs[0], start, g.block_size, g.n_kv_heads)
        with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
            pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
            tiled = nv.prefill(pool, 0, q, work.block_tables, [start], [n], nv.PREFILL_TILED)
            assert_close(tiled, want, f"tiled start={start} n={n}")
            rowwise = nv.prefill(pool, 0, q, work.block_tables, [start], [n], nv.PREFILL_ROWWISE)
            assert_close(rowwise, want, f"rowwise start={start} n={n}")
