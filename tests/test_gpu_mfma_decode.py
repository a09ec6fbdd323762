"""Decode on fp32 MFMA tiles (nvk_decode_mfma_kernel): GQA groups of 5..16 q-heads over head_dim 64 / 128.

The matrix-core path computes the same attention (include/core/attention.hpp:82-159) as the vector kernel with
v_mfma_f32_16x16x4_f32 products -- exact fp32 fmaf chains in another order -- so it is held to the same bar against
the CPU checker, and to the vector kernel (NVK_DECODE_VECTOR_ONLY) within that bar.
"""
import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from nano_vllm_amd import workloads as wl
from test_gpu_attention import assert_close, pool_from_host

pytestmark = pytest.mark.gpu

# (n_heads, n_kv_heads, head_dim, block_size): token-split parts 8 / 4 / 2 / 1, two kv-head groups, odd group sizes;
# the last three are the head-sharded ranks of a 128-head / 8-kv-head model: 16 q-heads per kv-head x 128 floats, where
# the token-split records are 58 KiB (7 of them), 50 KiB and 33 KiB of the workgroup's LDS
GEOMS = [(8, 1, 128, 16), (16, 2, 64, 16), (32, 4, 128, 16), (32, 4, 64, 16), (64, 8, 128, 8), (96, 12, 64, 32),
         (5, 1, 64, 4), (14, 2, 128, 16), (48, 3, 128, 16), (80, 16, 64, 16), (16, 1, 128, 16), (32, 2, 128, 16),
         (64, 4, 128, 16)]


@pytest.mark.parametrize("geom", GEOMS, ids=[f"H{g[0]}_KV{g[1]}_hd{g[2]}_bs{g[3]}" for g in GEOMS])
def test_mfma_decode_against_checker_and_vector_kernel(geom):
    g = wl.Geometry(*geom)
    ctx = np.array([1, 15, 16, 17, 300, 1027, 0, 64, 2048, 33], np.int32)
    work = wl.fragmented_workload("mfma", ctx, g, perm_seed=3)
    q = work.q() * np.float32(3.0)
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(work.num_seqs)))
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=8)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for flags, chunk in ((nv.AUTO, 0), (nv.SINGLE_PASS, 0), (nv.SPLIT_K, 48), (nv.SPLIT_K | nv.MERGE_SEPARATE, 16),
                             (nv.SPLIT_K | nv.MERGE_IN_KERNEL, 512)):
            got = nv.decode(pool, 0, q, work.block_tables, ctx, flags, chunk)
            assert_close(got, want, f"{geom} flags=0x{flags:x} chunk={chunk}")
            vec = nv.decode(pool, 0, q, work.block_tables, ctx, flags | nv.VECTOR_ONLY, chunk)
            assert_close(vec, want, f"{geom} vector kernel flags=0x{flags:x} chunk={chunk}")
        assert not got[6].any()  # ctx 0 -> zero row (attention.hpp:98)


def test_mfma_token_split_records_and_a_long_table_slice_share_the_lds():
    """16 q-heads x 128 floats x 7 token-split records leave 6 KiB of the launch's 64 KiB for the block-table slice:
    a one-chunk request over 30 000 tokens is cut to what fits (and combined) instead of overrunning the LDS."""
    g = wl.Geometry(16, 1, 128, 16)
    ctx = np.array([30000, 5], np.int32)
    work = wl.fragmented_workload("mfma_lds", ctx, g, perm_seed=5)
    q = work.q() * np.float32(2.0)
    k_host, v_host, tabs, c = wl.host_blocks_for(work, [0, 1])
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=2)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for flags, chunk in ((nv.SINGLE_PASS, 0), (nv.AUTO, 0), (nv.SPLIT_K, 28000)):
            assert_close(nv.decode(pool, 0, q, work.block_tables, ctx, flags, chunk), want, f"flags={flags} chunk={chunk}")


def test_mfma_decode_rescale_and_reference_floor():
    """A late dominant key forces the running max to jump inside and across tiles; scores below the reference's
    -1e10 floor give its 0/0 (attention.hpp:126-140) on the matrix-core path too."""
    H, KV, hd, bs, T = 8, 1, 128, 16, 400
    rng = np.random.default_rng(19)
    nb = T // bs
    k_pool = (0.05 * rng.standard_normal((nb, bs, KV, hd))).astype(np.float32)
    v_pool = rng.standard_normal((nb, bs, KV, hd)).astype(np.float32)
    q = rng.standard_normal((H, hd)).astype(np.float32)
    table = np.arange(nb, dtype=np.int32)
    for spike in (3, 130, 399):
        k2 = k_pool.copy()
        k2[spike // bs, spike % bs, 0] = 4.0 * q[5]  # score ~ +45 for head 5 only
        want = ol.oracle_paged_attention(q, k2, v_pool, table, T, bs, hd, H, KV)
        with pool_from_host(k2, v_pool) as pool:
            for flags, chunk in ((nv.SINGLE_PASS, 0), (nv.SPLIT_K, 64), (nv.SPLIT_K, 16)):
                got = nv.decode(pool, 0, q[None], table[None], np.array([T], np.int32), flags, chunk)[0]
                assert_close(got, want, f"spike {spike} flags={flags} chunk={chunk}")
    kf = np.full((3, bs, KV, hd), -1.0e6, np.float32)
    vf = np.ones((3, bs, KV, hd), np.float32)
    qf = np.full((H, hd), 1.0e6, np.float32)
    qf[1::2] = 1.0  # odd heads stay in the normal range
    want = ol.oracle_paged_attention(qf, kf, vf, np.arange(3, dtype=np.int32), 37, bs, hd, H, KV)
    assert np.isnan(want[0]).all() and not np.isnan(want[1]).any()
    with pool_from_host(kf, vf) as pool:
        for flags, chunk in ((nv.SINGLE_PASS, 0), (nv.SPLIT_K, 16)):
            got = nv.decode(pool, 0, qf[None], np.arange(3, dtype=np.int32)[None], np.array([37], np.int32), flags, chunk)[0]
            assert_close(got, want, f"floor flags={flags}")


def test_mfma_decode_full_size_gqa8():
    """One GPU's 512 x 2048 share (5b) at 32 q-heads / 4 kv-heads: sampled sequences against the checker, outputs
    inside V's range, single-pass == split-K within the bar."""
    g = wl.Geometry(32, 4, 128, 16)
    work = wl.config5b(512, g)
    q = work.q()
    sample = [0, 255, 511]
    k_host, v_host, tabs, c = wl.host_blocks_for(work, sample)
    want = ol.oracle_decode_batch(q[sample], k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=8)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        a = nv.decode(pool, 0, q, work.block_tables, work.ctx_lens, nv.AUTO)
        b = nv.decode(pool, 0, q, work.block_tables, work.ctx_lens, nv.SPLIT_K, 256)
        assert_close(a[sample], want, "5b gqa8")
        assert np.abs(a).max() <= 1.0 and np.isfinite(a).all()
        assert_close(b, a, "split-K vs single pass")
