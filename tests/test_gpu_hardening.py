"""The launch path's behaviour where a caller breaks an assumption: growth under graph capture, a host slab that is
not the one the append-only mirror staged, device memory held by a thread that ended."""
import ctypes as C
import threading

import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from hipmem import DeviceArray, Stream, _hip
from nano_vllm_amd import workloads as wl

pytestmark = pytest.mark.gpu

_hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]


def free_device_bytes():
    free, total = C.c_size_t(), C.c_size_t()
    assert _hip.hipDeviceSynchronize() == 0
    assert _hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return int(free.value)


def test_workspace_growth_under_capture_is_refused_before_anything_is_touched():
    """nvk_decode_device on a stream being captured, with a shape the pool has no workspace for yet: a status and a
    message that names the remedy, no hipFree / hipMalloc / device-wide wait inside the capture -- which therefore
    still ends cleanly; the same call outside the capture (or after nvk_pool_reserve_workspace) is served."""
    g = wl.Geometry(8, 2, 64, 16)
    ctx = np.array([900, 400, 1500, 7], np.int32)
    work = wl.fragmented_workload("capture", ctx, g)
    q = work.q()
    with nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim) as pool:
        pool.fill_uniform(0, wl.SEED_K, wl.SEED_V)
        qd, td, cd = DeviceArray(q), DeviceArray(work.block_tables), DeviceArray(ctx)
        out = DeviceArray(np.full_like(q, np.nan))
        st = Stream()
        seen = {}

        def attempt():
            try:
                nv.decode_device(pool, 0, work.num_seqs, g.n_heads, qd, td, work.max_blocks, cd, work.max_ctx, out,
                                 nv.SPLIT_K, 64, stream=st)
            except nv.NvkError as e:
                seen["err"] = e

        exe = st.capture(attempt)  # hipStreamEndCapture succeeds: nothing illegal was attempted under the capture
        _hip.hipGraphExecDestroy(exe)
        assert seen["err"].code == nv._capi.ERR_INVALID_ARG and "captured" in seen["err"].message
        assert "nvk_pool_reserve_workspace" in seen["err"].message
        # sized outside the capture, the same call is capturable and replays to the eager result
        pool.reserve_workspace(work.num_seqs, g.n_heads, work.max_ctx, 64)
        nv.decode_device(pool, 0, work.num_seqs, g.n_heads, qd, td, work.max_blocks, cd, work.max_ctx, out, nv.SPLIT_K, 64,
                         stream=st)
        st.synchronize()
        eager = out.numpy().copy()
        out2 = DeviceArray(np.full_like(q, np.nan))
        seen.clear()

        def again():
            nv.decode_device(pool, 0, work.num_seqs, g.n_heads, qd, td, work.max_blocks, cd, work.max_ctx, out2,
                             nv.SPLIT_K, 64, stream=st)

        exe = st.capture(again)
        st.launch(exe)
        st.synchronize()
        _hip.hipGraphExecDestroy(exe)
        assert np.array_equal(out2.numpy().view(np.uint32), eager.view(np.uint32))
        st.close()


def test_append_only_mirror_notices_a_slab_that_changed_under_it():
    """NVK_HOST_CACHE_APPEND_ONLY keys its device mirror on the slab's address and trusts "the previous call plus one
    token".  A slab freed and created again at the same address (here: rewritten in place) whose next call happens
    to be last + 1 breaks the promise; the row staged last no longer matches, so everything is staged again."""
    H, KV, hd, bs = 8, 2, 64, 16
    nb, T = 6, 40
    rng = np.random.default_rng(5)
    k_slab = rng.standard_normal((nb, bs, KV, hd)).astype(np.float32)
    v_slab = rng.standard_normal((nb, bs, KV, hd)).astype(np.float32)
    table = np.array([4, 1, 3, 0, 5, 2], np.int32)
    q = rng.standard_normal((H, hd)).astype(np.float32)
    out = np.empty(H * hd, np.float32)
    nv.set_host_cache_policy(True)
    try:
        nv.host_cache_staged_bytes(reset=True)
        nv.paged_attention(out, q, k_slab, v_slab, table, None, T, bs, hd, H, KV)
        want = ol.oracle_paged_attention(q, k_slab, v_slab, table, T, bs, hd, H, KV)
        assert np.allclose(out.reshape(H, hd), want, atol=1e-4, rtol=1e-5)
        first = nv.host_cache_staged_bytes(reset=True)
        assert first == 2 * T * KV * hd * 4
        # the promise kept: one more token, only its row crosses PCIe
        nv.paged_attention(out, q, k_slab, v_slab, table, None, T + 1, bs, hd, H, KV)
        assert nv.host_cache_staged_bytes(reset=True) == 2 * KV * hd * 4
        # the promise broken: "another request's slab" at the same address, one token further
        k_slab[...] = rng.standard_normal(k_slab.shape).astype(np.float32)
        v_slab[...] = rng.standard_normal(v_slab.shape).astype(np.float32)
        nv.paged_attention(out, q, k_slab, v_slab, table, None, T + 2, bs, hd, H, KV)
        want = ol.oracle_paged_attention(q, k_slab, v_slab, table, T + 2, bs, hd, H, KV)
        assert np.allclose(out.reshape(H, hd), want, atol=1e-4, rtol=1e-5), "stale rows of the old slab were used"
        assert nv.host_cache_staged_bytes(reset=True) == 2 * (T + 2) * KV * hd * 4  # everything staged again
        # the contiguous twin (nvk_standard_attention) keeps the same watch
        kc = rng.standard_normal((64, KV, hd)).astype(np.float32)
        vc = rng.standard_normal((64, KV, hd)).astype(np.float32)
        nv.standard_attention(out, q, kc, vc, None, 20, hd, H, KV, 64)
        kc[...] = rng.standard_normal(kc.shape).astype(np.float32)
        nv.standard_attention(out, q, kc, vc, None, 21, hd, H, KV, 64)
        want = ol.oracle_paged_attention(q, kc.reshape(4, 16, KV, hd), vc.reshape(4, 16, KV, hd), np.arange(4, dtype=np.int32),
                                         22, 16, hd, H, KV)  # tokens 0..pos behind an identity table
        assert np.allclose(out.reshape(H, hd), want, atol=1e-4, rtol=1e-5), "contiguous mirror used stale rows"
    finally:
        nv.set_host_cache_policy(False)


def test_a_finished_thread_gives_its_device_memory_back():
    """The host-slab forms of the drop-in keep a staging pool (and, append-only, mirrors) per host thread; they are
    released when the thread ends."""
    H, KV, hd, bs, T = 8, 2, 128, 16, 300
    nb = (T + bs - 1) // bs
    rng = np.random.default_rng(6)
    k_slab = rng.standard_normal((nb, bs, KV, hd)).astype(np.float32)
    v_slab = rng.standard_normal((nb, bs, KV, hd)).astype(np.float32)
    table = np.arange(nb, dtype=np.int32)
    q = rng.standard_normal((H, hd)).astype(np.float32)

    def body(append_only, res):
        out = np.empty(H * hd, np.float32)
        nv.set_host_cache_policy(append_only)
        nv.paged_attention(out, q, k_slab, v_slab, table, None, T, bs, hd, H, KV)
        res.append(out.copy())
        res.append(free_device_bytes())

    warm = []
    for append_only in (False, True):  # the runtime's own first-use allocations (and what it keeps of them) happen here
        t = threading.Thread(target=body, args=(append_only, warm))
        t.start()
        t.join()
    # The HIP runtime hands memory back in chunks of its own (a freed 2 MiB buffer may stay in its cache), so the
    # measure is that threads coming and going do not ADD UP: eight more threads, each holding ~4 MiB while it runs,
    # leave free memory where the first two left it, to within one runtime chunk.
    before = free_device_bytes()
    after = []
    for i in range(8):
        res = []
        t = threading.Thread(target=body, args=(bool(i & 1), res))
        t.start()
        t.join()
        after.append(free_device_bytes())
        assert res[1] < after[-1], "the thread held device memory while it ran"
        assert np.array_equal(res[0], warm[0])
    slack = 4 << 20
    assert before - min(after) <= slack, f"device memory not returned at thread exit: {before} -> {after}"
    assert after[-1] >= after[1] - (2 << 20), f"free device memory keeps shrinking: {after}"
