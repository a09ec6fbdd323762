"""The head-sharded split behind the C ABI (nvk_shard_*): RCCL called directly, the gather on its own stream.

A one-GPU box can only form a communicator of one rank (RCCL refuses two ranks on one device), so what runs here
is the whole code path -- unique id, ncclCommInitRank, ncclAllGather on the communication stream behind the
compute stream's event, the two-buffer protocol -- with G = 1, where gathered[0] must be out, bit for bit.
World sizes 2 and 4 are covered on CPU over gloo (tests/test_sharding_gloo.py) with the same head ranges."""
import numpy as np
import pytest

import nano_vllm_amd as nv
from hipmem import DeviceArray, Stream
from nano_vllm_amd import sharding as sh
from nano_vllm_amd import workloads as wl

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["auto", "overlapped", "inline"])
def test_one_rank_communicator_gathers_what_decode_device_writes(mode):
    """Both places the collective can run (nvk_shard_set_gather_mode): on the shard's own stream behind an event
    (what more than one rank gets), or in line on the compute stream (what one rank gets by default)."""
    g = wl.Geometry(32, 8, 128, 16)
    ctx = np.array([700, 33, 2048, 1, 513, 0, 640], np.int32)
    work = wl.fragmented_workload("shard1", ctx, g)
    L = 3
    with nv.PagedKVPool(L, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        for layer in range(L):
            pool.fill_uniform(layer, 10 + layer, 20 + layer)
        shard = sh.NvkShard(0, 1, sh.NvkShard.new_unique_id(), 0, g.n_heads, g.n_kv_heads)
        assert shard.info() == {"rank": 0, "world": 1, "local_heads": 32, "local_kv_heads": 8, "first_head": 0,
                                "first_kv_head": 0}
        assert shard.gather_mode == "inline"  # AUTO with one rank
        if mode != "auto":
            shard.set_gather_mode({"overlapped": shard.GATHER_OVERLAPPED, "inline": shard.GATHER_INLINE}[mode])
            assert shard.gather_mode == mode
        q = work.q()
        n = work.num_seqs
        qd, td, cd = DeviceArray(q), DeviceArray(work.block_tables), DeviceArray(ctx)
        outs = [DeviceArray(np.full_like(q, np.nan)) for _ in range(2)]
        gath = [DeviceArray(np.full((1,) + q.shape, np.nan, np.float32)) for _ in range(2)]
        st = Stream()
        got = []
        # layer-call l uses buffer l % 2; its gather runs under layer-call l + 1's attention
        for layer in range(L):
            b = layer % 2
            shard.decode_gather(pool, layer, n, qd, td, work.max_blocks, cd, work.max_ctx, outs[b], gath[b], b,
                                compute_stream=st)
            if layer > 0:  # the consumer of the previous layer-call's gathered rows
                shard.wait_gather((layer - 1) % 2, st)  # (on the stream: a second wait for the same gather is skipped)
                shard.wait_gather((layer - 1) % 2, st)
                shard.wait_gather((layer - 1) % 2)
                got.append(gath[(layer - 1) % 2].numpy().copy())
        shard.wait_gather((L - 1) % 2)
        got.append(gath[(L - 1) % 2].numpy().copy())
        st.synchronize()
        for layer in range(L):
            want = nv.decode(pool, layer, q, work.block_tables, ctx)
            assert got[layer].shape == (1,) + q.shape
            assert np.array_equal(got[layer][0].view(np.uint32), want.view(np.uint32)), layer
        # the collective alone, on a caller stream
        alone = DeviceArray(np.zeros((1,) + q.shape, np.float32))
        shard.all_gather_out(outs[(L - 1) % 2], alone, n, g.head_dim, stream=st)
        st.synchronize()
        assert np.array_equal(alone.numpy()[0].view(np.uint32), got[L - 1][0].view(np.uint32))
        shard.close()
        st.close()


def test_shard_rejects_a_pool_of_the_wrong_width():
    g = wl.Geometry(32, 8, 128, 16)
    with nv.PagedKVPool(1, 8, g.block_size, 4, g.head_dim) as pool:  # 4 kv-heads, the shard expects 8
        shard = sh.NvkShard(0, 1, sh.NvkShard.new_unique_id(), 0, g.n_heads, g.n_kv_heads)
        z = DeviceArray(np.zeros(16, np.float32))
        with pytest.raises(nv.NvkError):
            shard.decode_gather(pool, 0, 1, z, z, 1, z, 1, z, z, 0)
        with pytest.raises(nv.NvkError):
            shard.decode_gather(pool, 0, 1, z, z, 1, z, 1, z, z, 2)  # buffer must be 0 or 1
        shard.close()


def test_in_place_gather_writes_the_ranks_own_slot_directly():
    """out_local = gathered + rank * slice (ncclAllGather's in-place form): the attention writes its rows where the
    gather wants them; with one rank the result is simply there."""
    g = wl.Geometry(8, 2, 64, 16)
    ctx = np.array([90, 300, 17], np.int32)
    work = wl.fragmented_workload("shard_inplace", ctx, g, perm_seed=9)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, 5, 6)
        shard = sh.NvkShard(0, 1, sh.NvkShard.new_unique_id(), 0, g.n_heads, g.n_kv_heads)
        q = work.q()
        qd, td, cd = DeviceArray(q), DeviceArray(work.block_tables), DeviceArray(ctx)
        gath = DeviceArray(np.full((1,) + q.shape, np.nan, np.float32))
        for mode in (shard.GATHER_INLINE, shard.GATHER_OVERLAPPED):
            shard.set_gather_mode(mode)
            shard.decode_gather(pool, 0, work.num_seqs, qd, td, work.max_blocks, cd, work.max_ctx, gath, gath, 0)
            shard.wait_gather(0)
            want = nv.decode(pool, 0, q, work.block_tables, ctx)
            assert np.array_equal(gath.numpy()[0].view(np.uint32), want.view(np.uint32))
        shard.close()
