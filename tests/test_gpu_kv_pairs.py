"""Rows of 64 / 192 B (head_dim 16 / 48) streamed a PAIR of adjacent kv-heads per wave (nvk_decode_pairs_kernel<QPK>).

The reference's row is head_dim floats at (token, kv_head) (include/core/attention.hpp:114-115): at head_dim 48 --
its own stories15M geometry -- a row is one and a half 128-B lines.  The paired launch reads the same values as the
one-head-per-wave launch (NVK_DECODE_NO_PAIRS) and is held to the same bar against the CPU checker.
"""
import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from nano_vllm_amd import workloads as wl
from test_gpu_attention import assert_close
from test_gpu_fused_merge import same_bits

pytestmark = pytest.mark.gpu

# (n_heads, n_kv_heads, head_dim, block_size): token-split parts 8 / 4 / 2 / 1, two kv-head groups with idle waves in
# the second (20 heads = 10 pairs over 8 waves), GQA groups of 2..4 q-heads, a 64-B row
GEOMS = [(6, 6, 48, 16), (2, 2, 48, 16), (4, 4, 48, 4), (16, 16, 48, 8), (20, 20, 48, 16), (12, 6, 48, 16),
         (24, 6, 48, 32), (18, 6, 48, 1), (8, 8, 16, 16), (8, 2, 16, 12)]


@pytest.mark.parametrize("geom", GEOMS, ids=[f"H{g[0]}_KV{g[1]}_hd{g[2]}_bs{g[3]}" for g in GEOMS])
def test_paired_heads_against_checker_and_single_heads(geom):
    g = wl.Geometry(*geom)
    ctx = np.array([1, 15, 16, 17, 31, 32, 33, 300, 1027, 0, 64, 2048], np.int32)
    work = wl.fragmented_workload("pairs", ctx, g, perm_seed=5)
    q = work.q() * np.float32(2.0)
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(work.num_seqs)))
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=8)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        name = nv.decode_kernel_name(pool, work.num_seqs, g.n_heads, 2048)
        assert name.startswith("nvk::nvk_decode_pairs_kernel<"), name
        single = nv.decode_kernel_name(pool, work.num_seqs, g.n_heads, 2048, nv.NO_PAIRS)
        # (more kv-heads than a workgroup has waves: the free-running kernel, see nvk_decode_free_kernel)
        assert single.startswith("nvk::nvk_decode_free_kernel<1, " if g.n_kv_heads > 8 else "nvk::nvk_decode_kernel<1, "), single
        for flags, chunk in ((nv.AUTO, 0), (nv.SINGLE_PASS, 0), (nv.SPLIT_K, 48), (nv.SPLIT_K | nv.NO_PLAN, 16)):
            got = nv.decode(pool, 0, q, work.block_tables, ctx, flags, chunk)
            assert_close(got, want, f"{geom} flags=0x{flags:x} chunk={chunk}")
            one = nv.decode(pool, 0, q, work.block_tables, ctx, flags | nv.NO_PAIRS, chunk)
            assert_close(one, want, f"{geom} one head per wave, flags=0x{flags:x} chunk={chunk}")
        fused = nv.decode(pool, 0, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_IN_KERNEL, 64)
        apart = nv.decode(pool, 0, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_SEPARATE, 64)
        assert same_bits(fused, apart), f"{geom}: in-kernel combine vs merge kernel"
        assert not got[9].any()  # ctx 0 -> zero row (attention.hpp:98)


def test_odd_head_counts_stay_on_single_heads():
    """Token rows of an odd number of 192-B pieces do not start on a line: no pairing."""
    with nv.PagedKVPool(1, 64, 16, 3, 48) as pool:
        assert nv.decode_kernel_name(pool, 4, 3, 512) == "nvk::nvk_decode_kernel<1, 1, false>"
    with nv.PagedKVPool(1, 64, 16, 6, 64) as pool:
        assert nv.decode_kernel_name(pool, 4, 6, 512) == "nvk::nvk_decode_kernel<1, 1, true>"


def test_paired_heads_full_size_tiny_geometry():
    """G-tiny (6 / 6 / 48, the reference's stories15M) at one GPU's 512 x 2048 share: sampled sequences against the
    checker, every output inside V's range, paired == single heads within the bar everywhere."""
    g = wl.G_TINY
    work = wl.config5b(512, g)
    q = work.q()
    sample = [0, 255, 511]
    k_host, v_host, tabs, c = wl.host_blocks_for(work, sample)
    want = ol.oracle_decode_batch(q[sample], k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=8)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        a = nv.decode(pool, 0, q, work.block_tables, work.ctx_lens, nv.AUTO)
        b = nv.decode(pool, 0, q, work.block_tables, work.ctx_lens, nv.AUTO | nv.NO_PAIRS)
        assert_close(a[sample], want, "5b tiny, pairs")
        assert np.abs(a).max() <= 1.0 and np.isfinite(a).all()
        assert_close(a, b, "pairs vs single heads")


# NVK_DECODE_ROW_PAIRS: whole 128-float rows, a pair of kv-heads per wave, 4-wave workgroups (two to a CU)
ROW_PAIR_GEOMS = [(32, 8, 128, 16), (8, 8, 128, 16), (12, 4, 128, 8), (4, 2, 128, 16), (24, 8, 128, 32), (16, 8, 128, 1)]


@pytest.mark.parametrize("geom", ROW_PAIR_GEOMS, ids=[f"H{g[0]}_KV{g[1]}_hd{g[2]}_bs{g[3]}" for g in ROW_PAIR_GEOMS])
def test_row_pairs_against_checker_and_the_default_kernel(geom):
    """nvk_decode_rowpairs_kernel<QPK> is held to the bar of the kernel it replaces: the CPU checker on ragged
    fragmented batches in every mode, token-split shapes (2 / 4 kv-heads: the spare waves split the tokens) included,
    and in-kernel combine == merge kernel bit for bit."""
    g = wl.Geometry(*geom)
    ctx = np.array([1, 3, 4, 5, 15, 16, 17, 31, 33, 300, 1027, 0, 64, 2048, 129], np.int32)
    work = wl.fragmented_workload("rowpairs", ctx, g, perm_seed=7)
    q = work.q() * np.float32(2.0)
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(work.num_seqs)))
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=8)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        name = nv.decode_kernel_name(pool, work.num_seqs, g.n_heads, 2048, nv.ROW_PAIRS)
        assert name.startswith("nvk::nvk_decode_rowpairs_kernel<"), name
        for flags, chunk in ((nv.AUTO, 0), (nv.SINGLE_PASS, 0), (nv.SPLIT_K, 48), (nv.SPLIT_K | nv.NO_PLAN, 16),
                             (nv.SPLIT_K, 1024)):
            got = nv.decode(pool, 0, q, work.block_tables, ctx, flags | nv.ROW_PAIRS, chunk)
            assert_close(got, want, f"{geom} row pairs flags=0x{flags:x} chunk={chunk}")
        fused = nv.decode(pool, 0, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_IN_KERNEL | nv.ROW_PAIRS, 64)
        apart = nv.decode(pool, 0, q, work.block_tables, ctx, nv.SPLIT_K | nv.MERGE_SEPARATE | nv.ROW_PAIRS, 64)
        assert same_bits(fused, apart), f"{geom}: in-kernel combine vs merge kernel"
        assert not got[11].any()  # ctx 0 -> zero row (attention.hpp:98)


def test_row_pairs_is_declined_where_it_has_no_kernel():
    """Odd kv-head counts, other row widths, GQA groups past 4 q-heads per wave: the modifier is accepted and the
    default kernel runs."""
    for geom in ((9, 3, 128, 16), (32, 8, 64, 16), (64, 8, 128, 16), (32, 32, 128, 16)):
        g = wl.Geometry(*geom)
        with nv.PagedKVPool(1, 64, g.block_size, g.n_kv_heads, g.head_dim) as pool:
            assert "rowpairs" not in nv.decode_kernel_name(pool, 4, g.n_heads, 512, nv.ROW_PAIRS), geom
