"""16-token steps (nvk_decode_wide_kernel): chunks of 1024 tokens and more over whole 128-float rows, one kv-head
group, no token parts.  Same attention (include/core/attention.hpp:82-159), same work items and combine as
nvk_decode_kernel -- held to the checker, and to the 8-token-step kernel (shorter chunks) within the bar.
"""
import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from nano_vllm_amd import workloads as wl
from test_gpu_attention import assert_close

pytestmark = pytest.mark.gpu

# (n_heads, n_kv_heads, head_dim, block_size): 4 / 3 / 2 / 1 q-heads per wave; block sizes around the 16-token step
GEOMS = [(32, 8, 128, 16), (24, 8, 128, 16), (16, 8, 128, 8), (8, 8, 128, 32), (32, 8, 128, 5)]


@pytest.mark.parametrize("geom", GEOMS, ids=[f"H{g[0]}_KV{g[1]}_hd{g[2]}_bs{g[3]}" for g in GEOMS])
def test_wide_steps_against_checker_and_narrow_steps(geom):
    g = wl.Geometry(*geom)
    # lengths around the step (16), the chunk bounds (1024 / 2048) and ragged tails; 0 and 1 included
    ctx = np.array([2048, 1024, 1025, 1039, 1040, 1041, 2047, 3000, 4096, 4111, 1, 0, 15, 16, 17, 1500], np.int32)
    work = wl.fragmented_workload("wide", ctx, g, perm_seed=17)
    q = work.q() * np.float32(2.0)
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(work.num_seqs)))
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=8)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        qpk = g.n_heads // g.n_kv_heads
        assert nv.decode_kernel_name(pool, work.num_seqs, g.n_heads, int(ctx.max()), nv.SPLIT_K, 1024) == \
            f"nvk::nvk_decode_wide_kernel<{qpk}>"
        assert nv.decode_kernel_name(pool, work.num_seqs, g.n_heads, int(ctx.max()), nv.SPLIT_K, 512) == \
            f"nvk::nvk_decode_kernel<2, {qpk}, true>"
        narrow = nv.decode(pool, 0, q, work.block_tables, ctx, nv.SPLIT_K, 512)
        assert_close(narrow, want, f"{geom} 8-token steps")
        for flags, chunk in ((nv.SINGLE_PASS, 0), (nv.SPLIT_K, 1024), (nv.SPLIT_K, 2048), (nv.SPLIT_K | nv.NO_PLAN, 1040),
                             (nv.SPLIT_K | nv.MERGE_SEPARATE, 1024)):
            got = nv.decode(pool, 0, q, work.block_tables, ctx, flags, chunk)
            assert_close(got, want, f"{geom} wide flags=0x{flags:x} chunk={chunk}")
            assert_close(got, narrow, f"{geom} wide vs narrow flags=0x{flags:x} chunk={chunk}")
        assert not got[11].any()  # ctx 0 -> zero row (attention.hpp:98)
