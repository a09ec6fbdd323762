"""The product BlockManager (C ABI, bitmap tree) against the reference's recorded behaviour.

Integer work: every id, count and error must match exactly (golden traces F1 were
recorded from the reference's include/scheduler/block_manager.hpp).
"""
import json
import os
import threading

import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from golden_defs import BmCLib, BmProduct, replay
from nano_vllm_amd.workloads import splitmix64


def test_golden_traces(golden_dir):
    with open(os.path.join(golden_dir, "block_manager_traces.json")) as fh:
        traces = json.load(fh)["traces"]
    for t in traces:
        got = replay(BmProduct(t["num_blocks"], t["block_size"]), t["ops"])
        assert got == t["results"], t["name"]


DROPIN_PATH = os.path.join(ol.ROOT, "oracle", "_ref", "libnvk_ref_dropin.so")


@pytest.mark.skipif(not os.path.exists(DROPIN_PATH), reason="oracle/_ref/libnvk_ref_dropin.so not built")
def test_golden_traces_through_the_cpp_drop_in_class(golden_dir):
    """include/nvk_dropin/scheduler/block_manager.hpp: the reference's class name and methods over nvk_bm_*,
    driven through the same shim that recorded the traces from the reference's own class."""
    lib = ol._load_ref(DROPIN_PATH)
    with open(os.path.join(golden_dir, "block_manager_traces.json")) as fh:
        traces = json.load(fh)["traces"]
    for t in traces:
        got = replay(BmCLib(lib, "ref_", t["num_blocks"], t["block_size"]), t["ops"])
        assert got == t["results"], t["name"]


def test_reference_error_behaviour():
    bm = nv.BlockManager(4, 16)
    assert [bm.allocate_block() for _ in range(5)] == [0, 1, 2, 3, -1]  # -1 when full (ref :30-33)
    with pytest.raises(RuntimeError, match="Invalid block_id"):  # ref :50-53
        bm.free_block(4)
    with pytest.raises(RuntimeError, match="Invalid block_id"):
        bm.free_block(-1)
    bm.free_block(2)
    bm.free_block(2)  # double free: warning + no-op (ref :55-58)
    assert bm.get_num_free_blocks() == 1
    with pytest.raises(RuntimeError, match="Out of memory"):  # ref :71-74
        bm.allocate_sequence(17)
    assert bm.allocate_sequence(16) == [2]
    assert bm.get_utilization() == 1.0
    assert not bm.is_free(99) and not bm.is_free(-5)
    bm.free_block_safe(1234)  # the locked variant ignores bad ids (ref :249-251)
    assert bm.allocate_for_request(1, 1) == []  # non-throwing variant (ref :263-265)
    assert bm.get_num_active_requests() == 0


def test_lowest_free_first_on_a_large_pool_matches_the_linear_scan():
    """540K blocks (the size config 3 needs): same ids as the reference's O(n) scan, via the checker's restatement."""
    nb = 540_001
    ours = BmProduct(nb, 16)
    theirs = BmCLib(ol.ORACLE, "oracle_", nb, 16)
    draws = splitmix64(77, np.arange(3000, dtype=np.uint64)).tolist()
    held = []
    for d in draws:
        r = d % 10
        if r < 6 or not held:
            op = ["alloc"]
        elif r < 9:
            op = ["free", held.pop(int((d >> 8) % len(held)))]
        else:
            op = ["alloc_seq", int((d >> 8) % 3000)]
        a, b = ours.do(op), theirs.do(op)
        assert a == b, op
        if op[0] == "alloc" and a >= 0:
            held.append(a)
        elif op[0] == "alloc_seq" and a != "throw":
            held.extend(a)
    # free a scattering of ids across bitmap words and levels, then re-allocate: ascending order
    for i in (0, 63, 64, 4095, 4096, 262143, 262144, nb - 1):
        for side in (ours, theirs):
            if not side.do(["is_free", i]):
                side.do(["free", i])
    for _ in range(12):
        assert ours.do(["alloc"]) == theirs.do(["alloc"])
    ours.close()
    theirs.close()


def test_exhaust_and_refill_everything():
    nb = 5000
    bm = nv.BlockManager(nb, 8)
    assert bm.allocate_sequence(nb * 8) == list(range(nb))
    assert bm.allocate_block() == -1 and bm.get_num_free_blocks() == 0
    bm.free_sequence(list(range(0, nb, 2)))
    assert bm.get_num_free_blocks() == nb // 2
    assert bm.allocate_sequence(8 * 10) == list(range(0, 20, 2))
    assert abs(bm.get_utilization() - (1.0 - (nb // 2 - 10) / nb)) < 1e-6


def test_locked_api_is_thread_safe():
    nb = 4096
    bm = nv.BlockManager(nb, 16)
    got = [[] for _ in range(8)]

    def worker(i):
        for _ in range(nb // 8):
            got[i].append(bm.allocate_block_for_request(i))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    flat = sorted(x for g in got for x in g)
    assert flat == list(range(nb))  # every block handed out exactly once
    assert bm.get_num_active_requests() == 8
    for i in range(8):
        assert sorted(bm.get_request_blocks(i)) == sorted(got[i])
        bm.free_request(i)
    assert bm.get_num_free_blocks() == nb and bm.get_num_active_requests() == 0


def test_model_style_shared_free_list():
    """All layers share one free list (model.hpp:182,256): layer l of an L-layer model owns blocks l, L+l, 2L+l, ..."""
    L, bs = 6, 16
    bm = nv.BlockManager(256, bs)
    tables = [[] for _ in range(L)]
    for pos in range(256):
        for layer in range(L):
            if pos % bs == 0:
                tables[layer].append(bm.allocate_block())
    for layer in range(L):
        assert tables[layer] == [layer + L * i for i in range(16)]
    assert bm.get_num_free_blocks() == 256 - 96


def test_release_block_for_request_undoes_a_partial_draw():
    """What nvk_batch's roll-back needs when another thread took blocks between its check and its draw: a block
    given back must also leave its request's list, or free_request would later free it under its next owner
    (two sequences on one KV block).  The reference's per-request API has no such call (block_manager.hpp:157-171
    frees whole requests only)."""
    import nano_vllm_amd as nv

    bm = nv.BlockManager(6, 16)
    a = [bm.allocate_block_for_request(7) for _ in range(3)]  # request 7 draws 0, 1, 2
    assert a == [0, 1, 2]
    bm.release_block_for_request(7, 2)                         # roll the last one back
    assert bm.get_request_blocks(7) == [0, 1] and bm.is_free(2) and bm.get_num_free_blocks() == 4
    assert bm.allocate_block_for_request(9) == 2               # the id goes to another request
    bm.free_request(7)                                         # must free 0 and 1 only
    assert not bm.is_free(2) and bm.get_request_blocks(9) == [2]
    assert bm.is_free(0) and bm.is_free(1) and bm.get_num_active_requests() == 1
    with pytest.raises(nv.NvkError) as e:
        bm.release_block_for_request(9, 5)                     # not held by that request
    assert e.value.code == nv._capi.ERR_BAD_BLOCK
    with pytest.raises(nv.NvkError):
        bm.release_block_for_request(7, 0)                     # request 7 holds nothing any more
    bm.release_block_for_request(9, 2)
    assert bm.get_num_active_requests() == 0 and bm.get_num_free_blocks() == 6


def test_small_guards_of_the_c_abi():
    import ctypes as C

    import nano_vllm_amd as nv
    lib = nv._capi.lib
    lib.nvk_bm_utilization.restype = C.c_float
    assert lib.nvk_bm_utilization(None) == 0.0
    bm = nv.BlockManager(4, 16)
    assert bm.allocate_for_request(1, 2 ** 31 - 1) == []       # 134 M blocks wanted, computed without overflow
    ids = (C.c_int * 1)()
    assert lib.nvk_bm_allocate_for_request(bm._h, 3, 40, ids, 1) == 0  # 3 blocks needed, room for 1: nothing drawn
    assert b"ids holds 1" in lib.nvk_last_error() and bm.get_num_free_blocks() == 4
