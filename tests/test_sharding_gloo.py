"""The N>1 path on CPU: world_size 2 over gloo.  Both partitions must reproduce the
one-rank result bit for bit, since sequences and heads are independent."""
import os
import socket
import tempfile

import numpy as np
import pytest

import oracle_lib as ol
from nano_vllm_amd import sharding as sh

pytestmark = pytest.mark.skipif(ol.ORACLE is None, reason="oracle/libnvk_oracle.so not built")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_over_gloo():
    import torch.multiprocessing as mp

    import shard_worker

    with tempfile.TemporaryDirectory() as tmp:
        result = os.path.join(tmp, "result.txt")
        mp.spawn(shard_worker.run_rank, args=(2, _free_port(), result), nprocs=2, join=True)
        text = open(result).read()
    assert text.startswith("[1, 1, 1]"), text


def test_four_ranks_over_gloo_with_the_c_abi_head_ranges():
    """world_size 4: both partitions reproduce the one-rank result bit for bit; the head ranges come from the
    C ABI (nvk_shard_head_range) and must be HeadShard's."""
    import torch.multiprocessing as mp

    import shard_worker

    with tempfile.TemporaryDirectory() as tmp:
        result = os.path.join(tmp, "result.txt")
        mp.spawn(shard_worker.run_rank, args=(4, _free_port(), result, True), nprocs=4, join=True)
        text = open(result).read()
    assert text.startswith("[1, 1, 1]"), text


def test_c_abi_head_ranges_match_the_mirror():
    for world, H, KV in ((1, 32, 8), (2, 32, 8), (4, 32, 8), (8, 32, 8), (4, 8, 4), (3, 24, 6), (2, 6, 6)):
        for rank in range(world):
            hs = sh.HeadShard(rank, world, H, KV)
            assert sh.c_head_range(rank, world, H, KV) == (hs.local_heads, hs.local_kv_heads, hs.q_range[0], hs.kv_range[0])
    for bad in ((0, 3, 32, 8), (2, 2, 32, 8), (0, 0, 32, 8), (0, 2, 30, 8)):
        with pytest.raises(ValueError):
            sh.c_head_range(*bad)
    from nano_vllm_amd import _capi
    assert _capi.lib.nvk_shard_gather_bytes(4096, 32, 128, 8) == 8 * 2 ** 20  # 8 MiB per rank
    assert _capi.lib.nvk_shard_gather_bytes(64, 32, 128, 8) == 128 * 2 ** 10


def test_partition_sequences_balances_tokens_and_is_a_partition():
    from nano_vllm_amd import workloads as wl

    ctx = wl.config3().ctx_lens
    for world in (1, 2, 4, 8):
        owners = sh.partition_sequences(ctx, world)
        flat = np.sort(np.concatenate(owners))
        assert np.array_equal(flat, np.arange(ctx.size))
        loads = np.array([ctx[o].sum() for o in owners], np.int64)
        assert loads.max() - loads.min() <= ctx.max()  # LPT bound
        assert all(np.all(np.diff(o) > 0) for o in owners if o.size > 1)
    assert sh.contiguous_ranges(4096, 8) == [(512 * r, 512 * (r + 1)) for r in range(8)]
    assert sh.contiguous_ranges(10, 4) == [(0, 2), (2, 5), (5, 7), (7, 10)]


def test_head_shard_ranges_and_validation():
    hs = sh.HeadShard(3, 8, 32, 8)
    assert (hs.kv_range, hs.q_range, hs.local_heads, hs.local_kv_heads, hs.kv_mul) == ((3, 4), (12, 16), 4, 1, 4)
    with pytest.raises(ValueError):
        sh.HeadShard(0, 3, 32, 8)  # 3 does not divide 8 kv-heads
    with pytest.raises(ValueError):
        sh.HeadShard(2, 2, 32, 8)
    q = np.arange(2 * 32 * 4, dtype=np.float32).reshape(2, 32, 4)
    assert np.array_equal(hs.local_q(q), q[:, 12:16])
    assert sh.all_gather_bytes(4096, 32, 128, 8) == (8 * 2 ** 20, 64 * 2 ** 20)  # 8 MiB / 64 MiB
    g = np.arange(2 * 3 * 2 * 4, dtype=np.float32).reshape(2, 3, 2, 4)  # [G, n, H/G, hd]
    full = sh.assemble_heads(g)
    assert full.shape == (3, 4, 4) and np.array_equal(full[1, 2], g[1, 1, 0])
