"""Lockstep (a barrier per double-step keeps a workgroup's waves on the same token rows) is about WHEN rows are asked
for, not about what is computed: nvk_decode_kernel must give the bits of the free-running kernel
(NVK_DECODE_FREE_RUNNING -> nvk_decode_free_kernel) on every shape, token parts and split-K included, and
nvk_decode_wide_kernel (16-token steps) the same values within the bar of include/core/attention.hpp's checker.
"""
import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from nano_vllm_amd import workloads as wl
from test_gpu_attention import assert_close

pytestmark = pytest.mark.gpu

# (n_heads, n_kv_heads, head_dim, block_size): whole rows; 2 / 4 / 8 token parts; 64-float rows; partial rows
GEOMS = [(32, 8, 128, 16), (24, 8, 128, 16), (8, 8, 128, 8), (16, 4, 128, 16), (8, 2, 128, 16), (4, 1, 128, 16),
         (32, 8, 64, 16), (4, 4, 64, 16), (16, 8, 96, 16), (3, 1, 40, 4)]


@pytest.mark.parametrize("geom", GEOMS, ids=[f"H{g[0]}_KV{g[1]}_hd{g[2]}_bs{g[3]}" for g in GEOMS])
def test_lockstep_kernel_gives_the_bits_of_the_free_running_kernel(geom):
    g = wl.Geometry(*geom)
    ctx = np.array([1, 8, 9, 16, 17, 100, 511, 512, 513, 777, 1000, 0, 33], np.int32)
    work = wl.fragmented_workload("lock", ctx, g, perm_seed=29)
    q = work.q() * np.float32(2.0)
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(work.num_seqs)))
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=8)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        base = nv.VECTOR_ONLY | nv.NO_PAIRS
        locked = nv.decode_kernel_name(pool, work.num_seqs, g.n_heads, int(ctx.max()), nv.SPLIT_K | base, 256)
        free = nv.decode_kernel_name(pool, work.num_seqs, g.n_heads, int(ctx.max()), nv.SPLIT_K | base | nv.FREE_RUNNING, 256)
        assert locked.startswith("nvk::nvk_decode_kernel<") and free.startswith("nvk::nvk_decode_free_kernel<"), (locked, free)
        for flags, chunk in ((nv.SINGLE_PASS, 0), (nv.SPLIT_K, 64), (nv.SPLIT_K, 256), (nv.SPLIT_K | nv.NO_PLAN, 128),
                             (nv.SPLIT_K | nv.MERGE_SEPARATE, 96)):
            a = nv.decode(pool, 0, q, work.block_tables, ctx, flags | base, chunk)
            b = nv.decode(pool, 0, q, work.block_tables, ctx, flags | base | nv.FREE_RUNNING, chunk)
            assert_close(a, want, f"{geom} flags=0x{flags:x} chunk={chunk}")
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{geom} flags=0x{flags:x} chunk={chunk}: lockstep != free-running"


def test_wide_steps_against_the_free_running_kernel_at_full_size():
    """config 3 (256 ragged sequences behind fragmented tables): AUTO runs nvk_decode_wide_kernel; the free-running
    8-token-step kernel on the same batch agrees within the bar everywhere, and sampled sequences meet the checker."""
    work = wl.config3()
    g = work.geom
    q = work.q()
    sample = [0, 17, 101, 255]
    k_host, v_host, tabs, c = wl.host_blocks_for(work, sample)
    want = ol.oracle_decode_batch(q[sample], k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=8)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        assert nv.decode_kernel_name(pool, work.num_seqs, g.n_heads, work.max_ctx, nv.AUTO) == "nvk::nvk_decode_wide_kernel<4>"
        assert nv.decode_kernel_name(pool, work.num_seqs, g.n_heads, work.max_ctx, nv.AUTO | nv.FREE_RUNNING) == \
            "nvk::nvk_decode_free_kernel<2, 4, true>"
        a = nv.decode(pool, 0, q, work.block_tables, work.ctx_lens, nv.AUTO)
        b = nv.decode(pool, 0, q, work.block_tables, work.ctx_lens, nv.AUTO | nv.FREE_RUNNING)
        assert_close(a[sample], want, "config 3, 16-token steps")
        assert_close(a, b, "16-token steps in lockstep vs 8-token steps free-running")
