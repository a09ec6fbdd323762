"""nvk_decode_device_noalloc: the decode entry point that allocates nothing (static device scratch array).

Through the C ABI, against nvk_decode_device on the same inputs: same kernels, same chunking, so the same bits
(and the checker's tolerance, as everywhere).  Two calls in a row, the second with a smaller input, so whatever
the first call left in the shared array -- partials, counters, work list -- must not leak into the second.
Collected last (see the file name): the round's end runs pytest -x.
"""
import ctypes as C

import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from hipmem import DeviceArray, _hip
from nano_vllm_amd import workloads as wl

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-4, 1e-5
_hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]


def free_device_bytes():
    free, total = C.c_size_t(), C.c_size_t()
    assert _hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return int(free.value)


def run_both(pool, work, q, flags, chunk):
    """(noalloc, reference entry point) outputs for one batch; also checks that the no-alloc call left the
    device's free memory where it was."""
    g = work.geom
    qd, td, cd = DeviceArray(q), DeviceArray(work.block_tables), DeviceArray(work.ctx_lens)
    o1, o2 = DeviceArray(np.full_like(q, np.nan)), DeviceArray(np.full_like(q, np.nan))
    scrap = DeviceArray(np.full_like(q, np.nan))
    nv.decode_device_noalloc(pool, 0, work.num_seqs, g.n_heads, qd, td, work.max_blocks, cd, work.max_ctx, scrap, flags,
                             chunk)  # (the runtime's own first-launch set-up of a kernel is not the library's doing)
    pool.synchronize()
    before = free_device_bytes()
    nv.decode_device_noalloc(pool, 0, work.num_seqs, g.n_heads, qd, td, work.max_blocks, cd, work.max_ctx, o1, flags,
                             chunk)
    pool.synchronize()
    assert free_device_bytes() == before, "the no-alloc entry point changed the device's free memory"
    nv.decode_device(pool, 0, work.num_seqs, g.n_heads, qd, td, work.max_blocks, cd, work.max_ctx, o2, flags, chunk)
    pool.synchronize()
    return o1.numpy(), o2.numpy()


@pytest.mark.parametrize("flags,chunk", [(nv.SPLIT_K, 64), (nv.SPLIT_K | nv.MERGE_SEPARATE, 24), (nv.AUTO, 0),
                                          (nv.SPLIT_K | nv.NO_PLAN, 128)],
                         ids=["splitk64_in_kernel_merge", "splitk24_merge_kernel", "auto", "splitk128_no_plan"])
def test_noalloc_matches_decode_device_on_two_calls_in_a_row(flags, chunk):
    g = wl.G_LLAMA
    big = wl.fragmented_workload("noalloc_big", np.array([700, 1, 2048, 77, 513, 0, 1999, 640, 33], np.int32), g)
    small = wl.fragmented_workload("noalloc_small", np.array([130, 65, 5], np.int32), g, perm_seed=5)
    for work in (big, small):  # second call: fewer rows, fewer chunks, the array still holds the first call's state
        with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
            pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
            q = work.q()
            got, ref = run_both(pool, work, q, flags, chunk)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"{work.name}: bits differ"
            k_host, v_host, tabs, c = wl.host_blocks_for(work, list(range(work.num_seqs)))
            want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=4)
            assert (np.abs(got - want) <= ATOL + RTOL * np.abs(want)).all(), work.name
            assert not got[work.ctx_lens == 0].any()


def test_noalloc_same_pool_big_then_small_then_big():
    """One pool, three calls back to back on its stream (no wait in between): sizes change, nothing stale."""
    g = wl.Geometry(8, 2, 64, 16)
    ctxs = [np.array([900, 400, 1500, 7], np.int32), np.array([70, 300], np.int32), np.array([900, 400, 1500, 7], np.int32)]
    works = [wl.fragmented_workload(f"seq{i}", c, g, perm_seed=11) for i, c in enumerate(ctxs)]
    nb = max(w.num_blocks for w in works)
    with nv.PagedKVPool(1, nb, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        outs, keep = [], []
        for w in works:
            q = w.q()
            qd, td, cd = DeviceArray(q), DeviceArray(w.block_tables), DeviceArray(w.ctx_lens)
            o = DeviceArray(np.full_like(q, np.nan))
            nv.decode_device_noalloc(pool, 0, w.num_seqs, g.n_heads, qd, td, w.max_blocks, cd, w.max_ctx, o,
                                     nv.SPLIT_K, 32)
            outs.append(o)
            keep.append((qd, td, cd))
        pool.synchronize()
        for w, o in zip(works, outs):
            want = nv.decode(pool, 0, w.q(), w.block_tables, w.ctx_lens, nv.SPLIT_K, 32)
            assert np.array_equal(o.numpy().view(np.uint32), want.view(np.uint32)), w.name


def test_noalloc_refuses_what_does_not_fit():
    g = wl.G_LLAMA
    assert nv.noalloc_scratch_bytes() == 24 << 20
    work = wl.uniform_workload("too_big", 64, 2048, g)
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        q = work.q()
        qd, td, cd = DeviceArray(q), DeviceArray(work.block_tables), DeviceArray(work.ctx_lens)
        out = DeviceArray(np.zeros_like(q))
        with pytest.raises(nv.NvkError) as e:  # 64 rows x 256 chunks x 16 KiB = 256 MiB of partials
            nv.decode_device_noalloc(pool, 0, 64, g.n_heads, qd, td, work.max_blocks, cd, 2048, out, nv.SPLIT_K, 8)
        assert e.value.code == nv._capi.ERR_INVALID_ARG and "scratch" in e.value.message
        # the same batch in chunks that fit is served
        nv.decode_device_noalloc(pool, 0, 64, g.n_heads, qd, td, work.max_blocks, cd, 2048, out, nv.AUTO, 0)
        pool.synchronize()
