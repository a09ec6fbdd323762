"""The split-K combine inside the decode kernel, and decoding without a plan.

Where a sequence is cut into chunks the reference's softmax (include/core/attention.hpp:126-140) is
re-assembled from the chunks' (max, sum, output) partials.  Two ways to do that must give the same bits:
``nvk_merge_kernel`` (a kernel of its own, one thread per output float4, chunks in order) and the
workgroup of ``nvk_decode_kernel`` that completes a sequence (same arithmetic, handed the other
workgroups' partials through an agent-scope arrival counter).  A wrong hand-off shows as stale or
missing partials, so every comparison here is bit for bit, every word, on launches whose inputs
change from one call to the next (a stale line would hold the previous call's values).

Without a plan (NVK_DECODE_UNIFORM / NVK_DECODE_NO_PLAN) the kernel works its chunk out of its
workgroup index; only the ORDER of the chunks differs from a planned launch, never a result.
"""
import os

import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from golden_defs import ATTENTION_CASES, attention_case_inputs
from hipmem import DeviceArray
from nano_vllm_amd import workloads as wl
from test_gpu_attention import assert_close, pool_from_host

pytestmark = pytest.mark.gpu

CASES = [c for c in ATTENTION_CASES if c["num_tokens"] > 0]


def same_bits(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def decode_dev(pool, work_or_none, q, tabs, ctx, flags, chunk=0, max_ctx=None, ctx_dev="lens"):
    """One layer-call through the device-pointer entry point; returns the host copy of out."""
    n, H, _ = q.shape
    qd, td = DeviceArray(q), DeviceArray(tabs)
    cd = DeviceArray(np.asarray(ctx, np.int32)) if ctx_dev == "lens" else None
    out = DeviceArray(np.full_like(q, np.nan))
    nv.decode_device(pool, 0, n, H, qd, td, tabs.shape[1], cd, int(max_ctx if max_ctx is not None else max(ctx)), out,
                     flags, chunk)
    pool.synchronize()
    return out.numpy()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_in_kernel_merge_equals_merge_kernel_on_golden_cases(case, golden_dir):
    """Golden cases x {8, 64}-token chunks (up to 256 chunks per sequence): in-kernel combine == nvk_merge_kernel,
    bit for bit, and both within the path's tolerance of the reference's output."""
    gold = np.load(os.path.join(golden_dir, "attention_cases.npz"))[case["name"] + "/out"]
    q, k_pool, v_pool, table = attention_case_inputs(case)
    ctx = np.array([case["num_tokens"]], np.int32)
    with pool_from_host(k_pool, v_pool) as pool:
        for chunk in (8, 64):
            fused = nv.decode(pool, 0, q[None], table[None], ctx, nv.SPLIT_K | nv.MERGE_IN_KERNEL, chunk)[0]
            apart = nv.decode(pool, 0, q[None], table[None], ctx, nv.SPLIT_K | nv.MERGE_SEPARATE, chunk)[0]
            assert same_bits(fused, apart), f"{case['name']} chunk={chunk}"
            assert_close(fused, gold, f"{case['name']} chunk={chunk}")


@pytest.mark.parametrize("seed", range(36))
def test_in_kernel_merge_and_planless_decode_on_random_geometries(seed):
    """The geometries of test_random_geometry_batches_against_checker (every VPL / QPK / token-split / kv-group
    combination, ragged lengths with empty sequences, fragmented tables): the combine inside the kernel, the merge
    kernel, and the same launch without a plan all give the same bits; and they match the checker."""
    d = wl.splitmix64(4200 + seed, np.arange(12, dtype=np.uint64)).tolist()
    KV = [1, 2, 3, 4, 5, 6, 8, 12, 16][d[0] % 9]
    H = KV * [1, 2, 3, 4, 5, 6, 7, 8, 16][(d[1] + seed) % 9]
    hd = [4, 16, 20, 48, 64, 80, 96, 128, 132, 192, 256][d[2] % 11]
    bs = [1, 4, 8, 12, 16, 32][d[3] % 6]
    n = 1 + d[4] % 9
    top = [40, 300, 1500][d[5] % 3]
    ctx = (wl.splitmix64(seed, np.arange(n, dtype=np.uint64)) % np.uint64(top)).astype(np.int32)
    ctx[d[6] % n] = top
    if n > 2:
        ctx[(d[6] + 1) % n] = 0  # an empty sequence among the others
    g = wl.Geometry(H, KV, hd, bs)
    work = wl.fragmented_workload(f"fuzzm{seed}", ctx, g, slack=0.3, perm_seed=99 + seed)
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(n)))
    chunk = [8, 24, 64, 200, 512][d[8] % 5]
    with nv.PagedKVPool(1, work.num_blocks, bs, KV, hd) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for rnd in range(2):  # two rounds with different queries: a stale partial would carry round 0's values
            q = work.q() * np.float32(1 + (d[7] + rnd) % 5)
            want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, bs, KV, n_threads=4)
            what = f"H{H} KV{KV} hd{hd} bs{bs} ctx{ctx.tolist()} chunk={chunk} round {rnd}"
            fused = decode_dev(pool, work, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_IN_KERNEL, chunk)
            apart = decode_dev(pool, work, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_SEPARATE, chunk)
            assert same_bits(fused, apart), what
            assert_close(fused, want, what)
            for merge in (nv.MERGE_IN_KERNEL, nv.MERGE_SEPARATE, 0):
                direct = decode_dev(pool, work, q, work.block_tables, ctx, nv.SPLIT_K | nv.NO_PLAN | merge, chunk)
                if merge:
                    assert same_bits(direct, fused), what + f" no plan, merge 0x{merge:x}"
                else:  # the library's own choice may be the chunk-parallel merge kernel (another order of sums)
                    assert_close(direct, want, what + " no plan")
            auto = decode_dev(pool, work, q, work.block_tables, ctx, nv.AUTO | nv.NO_PLAN)
            assert_close(auto, want, what + " auto, no plan")


def test_many_chunks_per_sequence_in_kernel():
    """>= 128 chunks per sequence combined by one workgroup each, against the checker; sequences of different
    lengths so the arrival counts differ."""
    g = wl.G_LLAMA
    ctx = np.array([2048, 1500, 1029, 3], np.int32)
    work = wl.fragmented_workload("many_chunks", ctx, g)
    q = work.q()
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(work.num_seqs)))
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=4)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        fused = nv.decode(pool, 0, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_IN_KERNEL, 8)  # 256 / 188 / 129 / 1
        apart = nv.decode(pool, 0, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_SEPARATE, 8)
        assert_close(fused, want, "256 chunks")
        assert same_bits(fused, apart)


def test_reference_floor_nans_survive_the_in_kernel_merge():
    """max_val starts at -1e10 (attention.hpp:126): all scores below it -> the reference's 0/0, chunked or not."""
    H, KV, hd, bs, T = 2, 1, 64, 16, 37
    nb = (T + bs - 1) // bs
    k_pool = np.full((nb, bs, KV, hd), -1.0e6, np.float32)
    v_pool = np.ones((nb, bs, KV, hd), np.float32)
    q = np.full((H, hd), 1.0e6, np.float32)
    q[1] = 1.0
    table = np.arange(nb, dtype=np.int32)
    want = ol.oracle_paged_attention(q, k_pool, v_pool, table, T, bs, hd, H, KV)
    assert np.isnan(want[0]).all() and not np.isnan(want[1]).any()
    with pool_from_host(k_pool, v_pool) as pool:
        for merge in (nv.MERGE_IN_KERNEL, nv.MERGE_SEPARATE):
            got = nv.decode(pool, 0, q[None], table[None], np.array([T], np.int32), nv.SPLIT_K | merge, 8)[0]
            assert_close(got, want, f"merge 0x{merge:x}")


def test_uniform_batch_needs_no_plan_and_no_lengths():
    """NVK_DECODE_UNIFORM: every sequence holds max_ctx tokens, ctx_lens_dev is NULL, no plan kernel runs; the bits
    are those of the planned launch (the plan of a uniform batch IS the natural order)."""
    g = wl.Geometry(32, 8, 128, 16)
    for n, T in ((5, 700), (64, 257), (3, 16)):
        ctx = np.full(n, T, np.int32)
        work = wl.fragmented_workload(f"uniform{n}x{T}", ctx, g)
        q = work.q()
        with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
            pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
            for mode, chunk in ((nv.AUTO, 0), (nv.SPLIT_K, 64), (nv.SINGLE_PASS, 0)):
                planned = decode_dev(pool, work, q, work.block_tables, ctx, mode, chunk)
                uniform = decode_dev(pool, work, q, work.block_tables, ctx, mode | nv.UNIFORM, chunk, max_ctx=T,
                                     ctx_dev=None)
                assert same_bits(planned, uniform), f"{n}x{T} mode {mode}"
            k_host, v_host, tabs, c = wl.host_blocks_for(work, [0, n - 1])
            want = ol.oracle_decode_batch(q[[0, n - 1]], k_host, v_host, tabs, c, g.block_size, g.n_kv_heads)
            assert_close(uniform[[0, n - 1]], want, f"{n}x{T}")


@pytest.mark.parametrize("name", ["config2", "config3", "config4"])
def test_in_kernel_merge_at_full_size_under_load(name):
    """BASELINE configs at full size: every CU streaming (config 2 / 4) and unevenly loaded (config 3: ragged,
    longest first), the combining workgroups' L1s warm with the stream.  Default launch (in-kernel combine)
    == merge kernel on every word, twice with different queries."""
    work = {"config2": wl.config2, "config3": wl.config3, "config4": wl.config4}[name]()
    g = work.geom
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for rnd in range(2):
            q = work.q() * np.float32(1 + rnd)
            fused = decode_dev(pool, work, q, work.block_tables, work.ctx_lens, nv.AUTO | nv.MERGE_IN_KERNEL)
            apart = decode_dev(pool, work, q, work.block_tables, work.ctx_lens, nv.AUTO | nv.MERGE_SEPARATE)
            assert np.isfinite(fused).all()
            assert same_bits(fused, apart), f"{name} round {rnd}"
            auto = decode_dev(pool, work, q, work.block_tables, work.ctx_lens, nv.AUTO)
            assert same_bits(auto, fused), f"{name}: the default launch is the in-kernel combine here"


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 15, 17, 63, 65, 130, 257])
def test_uniform_batches_of_any_size_cover_every_item(n):
    """Uniform batches deal each XCD a contiguous eighth of the items (xcd_range_item: workgroup w takes item
    (w % 8) * n / 8 + w / 8, the first n % 8 ranges one longer).  Whatever n, every (sequence, chunk) must be taken
    exactly once: rows pre-filled with NaN come back complete and bit-equal to the planned launch."""
    g = wl.G_LLAMA
    T = 200
    ctx = np.full(n, T, np.int32)
    work = wl.fragmented_workload(f"ranges{n}", ctx, g, perm_seed=n)
    q = work.q()
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for mode, chunk in ((nv.SINGLE_PASS, 0), (nv.SPLIT_K, 48), (nv.SPLIT_K | nv.MERGE_SEPARATE, 64)):
            planned = decode_dev(pool, work, q, work.block_tables, ctx, mode, chunk)
            uniform = decode_dev(pool, work, q, work.block_tables, ctx, mode | nv.UNIFORM, chunk, max_ctx=T, ctx_dev=None)
            assert np.isfinite(uniform).all(), f"n={n} mode 0x{mode:x}: an item was never computed"
            assert same_bits(planned, uniform), f"n={n} mode 0x{mode:x}"


@pytest.mark.parametrize("bs,n,T,chunk,H", [(16, 4, 4096, 1024, 32), (24, 8, 2048, 1024, 32), (16, 2, 8192, 2048, 32),
                                             (16, 4, 4096, 1024, 64), (24, 4, 4096, 2048, 64)])  # 64/8/128: the MFMA kernel
def test_long_chunks_skewed_between_even_and_odd_xcds(bs, n, T, chunk, H):
    """Sequences cut into long chunks (>= 1024 tokens): every full chunk that runs on an even XCD hands 4 % of its
    tokens to the chunk behind it on the odd XCD beside it (LaunchShape::chunk_skew: the even XCDs stream slower).
    With a plan the work list carries the bounds (nvk::skewed_chunk, by the item's place in the list); without one
    (uniform batch, an even number of full chunks, whole rounds of the XCDs) the kernel works them out of its id.
    Chunk bounds that no longer sit on multiples of the chunk size -- nor, at block size 24, on block boundaries:
    both combines give the same bits, every row matches the checker, and -- the same bounds either way here -- the
    planned and the plan-free launch give the same bits.  One token short of full chunks only the plan skews."""
    g = wl.Geometry(H, 8, 128, bs)
    ctx = np.full(n, T, np.int32)
    work = wl.fragmented_workload(f"skew{bs}x{n}x{T}", ctx, g, perm_seed=bs + n)
    q = work.q()
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(n)))
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, bs, g.n_kv_heads, n_threads=4)
    with nv.PagedKVPool(1, work.num_blocks, bs, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        planned = decode_dev(pool, work, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_IN_KERNEL, chunk)
        fused = decode_dev(pool, work, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_IN_KERNEL | nv.UNIFORM, chunk,
                           max_ctx=T, ctx_dev=None)
        apart = decode_dev(pool, work, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_SEPARATE | nv.UNIFORM, chunk,
                           max_ctx=T, ctx_dev=None)
        host = nv.decode(pool, 0, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_IN_KERNEL, chunk)
        assert np.isfinite(fused).all()
        assert same_bits(fused, apart)
        assert same_bits(fused, planned)
        assert same_bits(fused, host)
        assert_close(fused, want, f"skewed chunks bs={bs} {n}x{T}")
        # ragged: the plan skews the pairs of full chunks it finds, the last (partial) chunks stay as they are
        ragged = np.array([T - 1 - 1000 * (i % 3) for i in range(n)], np.int32)
        k2, v2, t2, c2 = wl.host_blocks_for(work, list(range(n)))
        want2 = ol.oracle_decode_batch(q, k2, v2, t2, ragged, bs, g.n_kv_heads, n_threads=4)
        p2 = decode_dev(pool, work, q, work.block_tables, ragged, nv.SPLIT_K | nv.MERGE_IN_KERNEL, chunk)
        s2 = decode_dev(pool, work, q, work.block_tables, ragged, nv.SPLIT_K | nv.MERGE_SEPARATE, chunk)
        d2 = decode_dev(pool, work, q, work.block_tables, ragged, nv.SPLIT_K | nv.NO_PLAN, chunk)
        h2 = nv.decode(pool, 0, q, work.block_tables, ragged, nv.SPLIT_K | nv.MERGE_IN_KERNEL, chunk)
        assert same_bits(p2, s2) and same_bits(p2, h2)
        assert_close(p2, want2, f"ragged, planned bs={bs}")
        assert_close(d2, want2, f"ragged, no plan bs={bs}")


@pytest.mark.parametrize("seed", range(6))
def test_skewed_chunks_on_random_long_batches(seed):
    """Random batches cut into 1024- or 2048-token chunks (the sizes that are skewed): sequences with 0..5 full chunks
    at even and odd places of the work list, ragged last chunks, an empty sequence, block sizes 16 and 24.  Planned
    launch with either combine (same bits), the host-pointer entry point (same bits: it builds the same list), the
    plan-free launch (tolerance: other bounds) -- all against the checker."""
    d = wl.splitmix64(7700 + seed, np.arange(8, dtype=np.uint64)).tolist()
    bs = (16, 24)[d[0] % 2]
    chunk = (1024, 2048)[d[1] % 2]
    n = 3 + d[2] % 6
    g = wl.Geometry(32, 8, 128, bs)
    ctx = (wl.splitmix64(seed + 50, np.arange(n, dtype=np.uint64)) % np.uint64(5 * chunk + 700)).astype(np.int32)
    ctx[d[3] % n] = 0
    ctx[(d[3] + 1) % n] = 3 * chunk  # an odd number of full chunks, no remainder: the next sequence starts at an odd place
    work = wl.fragmented_workload(f"skewfuzz{seed}", ctx, g, slack=0.2, perm_seed=seed)
    q = work.q()
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(n)))
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, bs, g.n_kv_heads, n_threads=4)
    with nv.PagedKVPool(1, work.num_blocks, bs, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        what = f"bs{bs} chunk{chunk} ctx{ctx.tolist()}"
        fused = decode_dev(pool, work, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_IN_KERNEL, chunk)
        apart = decode_dev(pool, work, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_SEPARATE, chunk)
        host = nv.decode(pool, 0, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_IN_KERNEL, chunk)
        free = decode_dev(pool, work, q, work.block_tables, ctx, nv.SPLIT_K | nv.NO_PLAN, chunk)
        auto = decode_dev(pool, work, q, work.block_tables, ctx, nv.AUTO)
        assert same_bits(fused, apart), what
        assert same_bits(fused, host), what
        assert_close(fused, want, what)
        assert_close(free, want, what + " no plan")
        assert_close(auto, want, what + " auto")
