"""Token-split workgroups of the vector decode kernel: a pool with fewer kv-heads than a workgroup has waves (a
head-sharded rank, a small model) lets the spare waves split the chunk's tokens; the parts' (m, l, o) meet in LDS and
the merge is dealt over all threads of the workgroup, each storing its float4 of `out` or of the chunk's partial.

Held to the checker (include/core/attention.hpp:82-159) over every row width (head_dim <= 64 / <= 128 / <= 256,
whole and partial), 2 / 4 / 8 parts, pairs of kv-heads per wave, one chunk (the merge writes `out`) and several
(the merge writes the split-K partial; combined in the kernel or by nvk_merge_kernel, bit-equal).
"""
import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from nano_vllm_amd import workloads as wl
from test_gpu_attention import assert_close, pool_from_host

pytestmark = pytest.mark.gpu

# (n_heads, n_kv_heads, head_dim, block_size)
GEOMS = [(4, 1, 128, 16),   # the head-shard rank of G-llama at G = 8: 8 parts
         (8, 2, 128, 16),   # G = 4: 4 parts
         (16, 4, 128, 16),  # G = 2: 2 parts
         (4, 1, 96, 16), (2, 2, 80, 8),    # 128-float lanes, partial rows
         (1, 1, 64, 16), (4, 2, 64, 4), (3, 1, 40, 16),  # 64-float rows, whole and partial
         (6, 6, 48, 16), (4, 2, 16, 16),   # a pair of kv-heads per wave (192-B / 64-B rows) with token parts
         (4, 1, 256, 16), (2, 2, 192, 16), (8, 2, 256, 8)]  # 256-float rows: 4-wave workgroups, 4 / 2 parts


@pytest.mark.parametrize("geom", GEOMS, ids=[f"H{g[0]}_KV{g[1]}_hd{g[2]}_bs{g[3]}" for g in GEOMS])
def test_token_split_against_checker(geom):
    g = wl.Geometry(*geom)
    ctx = np.array([1, 7, 8, 9, 31, 64, 65, 300, 1027, 0, 2048, 33, 129], np.int32)
    work = wl.fragmented_workload("parts", ctx, g, perm_seed=11)
    q = work.q() * np.float32(2.5)
    k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(work.num_seqs)))
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=8)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        for flags, chunk in ((nv.AUTO, 0), (nv.SINGLE_PASS, 0), (nv.SPLIT_K, 32), (nv.SPLIT_K, 256),
                             (nv.AUTO | nv.NO_PLAN, 0), (nv.SPLIT_K | nv.NO_PLAN, 128)):
            fused = nv.decode(pool, 0, q, work.block_tables, ctx, flags | nv.MERGE_IN_KERNEL | nv.VECTOR_ONLY, chunk)
            assert_close(fused, want, f"{geom} flags=0x{flags:x} chunk={chunk}")
            apart = nv.decode(pool, 0, q, work.block_tables, ctx, flags | nv.MERGE_SEPARATE | nv.VECTOR_ONLY, chunk)
            assert np.array_equal(fused.view(np.uint32), apart.view(np.uint32)), \
                f"{geom} flags=0x{flags:x} chunk={chunk}: in-kernel combine != nvk_merge_kernel"
        assert not fused[9].any()  # ctx 0 -> zero row (attention.hpp:98)


def test_token_split_rescale_and_reference_floor():
    """A dominant key in ONE token part makes the other parts' sums vanish in the LDS merge; scores below the
    reference's -1e10 floor give its 0/0 (attention.hpp:126-140) through the merge as well."""
    H, KV, hd, bs, T = 4, 1, 128, 16, 512
    rng = np.random.default_rng(23)
    nb = T // bs
    k_pool = (0.05 * rng.standard_normal((nb, bs, KV, hd))).astype(np.float32)
    v_pool = rng.standard_normal((nb, bs, KV, hd)).astype(np.float32)
    q = rng.standard_normal((H, hd)).astype(np.float32)
    table = np.arange(nb, dtype=np.int32)
    for spike in (0, 9, 70, 511):  # steps are dealt round-robin over the parts: tokens of different parts
        k2 = k_pool.copy()
        k2[spike // bs, spike % bs, 0] = 4.0 * q[2]
        want = ol.oracle_paged_attention(q, k2, v_pool, table, T, bs, hd, H, KV)
        with pool_from_host(k2, v_pool) as pool:
            for flags, chunk in ((nv.SINGLE_PASS, 0), (nv.SPLIT_K, 128)):
                got = nv.decode(pool, 0, q[None], table[None], np.array([T], np.int32), flags, chunk)[0]
                assert_close(got, want, f"spike {spike} flags={flags} chunk={chunk}")
    kf = np.full((4, bs, KV, hd), -1.0e6, np.float32)
    vf = np.ones((4, bs, KV, hd), np.float32)
    qf = np.full((H, hd), 1.0e6, np.float32)
    qf[1::2] = 1.0  # odd heads stay in the normal range
    want = ol.oracle_paged_attention(qf, kf, vf, np.arange(4, dtype=np.int32), 61, bs, hd, H, KV)
    assert np.isnan(want[0]).all() and not np.isnan(want[1]).any()
    with pool_from_host(kf, vf) as pool:
        for flags, chunk in ((nv.SINGLE_PASS, 0), (nv.SPLIT_K, 16)):
            got = nv.decode(pool, 0, qf[None], np.arange(4, dtype=np.int32)[None], np.array([61], np.int32), flags,
                            chunk)[0]
            assert_close(got, want, f"floor flags={flags}")
