"""the decode batch (nvk_batch_*) under a continuous-batching schedule.

Requests join and finish at different steps; every step runs RoPE + append + attention for all running
requests, layer by layer.  Checked per request against the CPU checker's own rope -> append ->
paged_attention on a host pool (parity for a batch is defined per sequence, SURVEY.md TL;DR), and the
block ids against the checker's restatement of the reference's per-request BlockManager API."""
import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from golden_defs import BmCLib
from hipmem import DeviceArray

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-4, 1e-5


def close(got, want):
    return bool((np.abs(got - want) <= ATOL + RTOL * np.abs(want)).all())


def test_continuous_batching_schedule():
    L, H, KV, hd, bs = 2, 8, 2, 64, 8
    nb, max_seqs, max_blocks = 48, 6, 8
    theta = 10000.0
    rng = np.random.default_rng(21)
    # request id -> (step it joins, tokens it will take)
    plan = {10: (0, 30), 11: (0, 9), 12: (3, 17), 13: (5, 40), 14: (12, 8), 15: (12, 25), 16: (20, 12)}
    host_k = np.zeros((L, nb, bs, KV, hd), np.float32)
    host_v = np.zeros_like(host_k)
    ref_bm = BmCLib(ol.ORACLE, "oracle_", nb, bs)
    lens, running = {}, []
    bm = nv.BlockManager(nb, bs)
    with nv.PagedKVPool(L, nb, bs, KV, hd) as pool, nv.DecodeBatch(pool, bm, max_seqs, max_blocks) as batch:
        step = 0
        while step < 70 and (running or any(j >= step for j, _ in plan.values())):
            for rid, (join, _) in plan.items():
                if join == step:
                    batch.add_request(rid)
                    running.append(rid)
                    lens[rid] = 0
            ids = list(running)  # ScheduledBatch.decode_requests: the running requests, in admission order
            if ids:
                batch.begin_step(ids, H)
                # the same allocations, in the same order, through the reference-equivalent allocator
                for rid in ids:
                    if lens[rid] % bs == 0:
                        ref_bm.do(["req_alloc_block", rid])
                for rid in ids:
                    assert batch.request_blocks(rid) == ref_bm.do(["req_blocks", rid]), (step, rid)
                n = len(ids)
                for layer in range(L):
                    q = rng.standard_normal((n, H, hd), dtype=np.float32)
                    k = rng.standard_normal((n, KV, hd), dtype=np.float32)
                    v = rng.standard_normal((n, KV, hd), dtype=np.float32)
                    qd, kd, vd = DeviceArray(q), DeviceArray(k), DeviceArray(v)
                    outd = DeviceArray(np.full_like(q, np.nan))
                    batch.layer(layer, qd, kd, vd, outd, theta)
                    pool.synchronize()
                    got, q_rot = outd.numpy(), qd.numpy()
                    for i, rid in enumerate(ids):
                        pos = lens[rid]
                        table = np.asarray(batch.request_blocks(rid), np.int32)
                        wq, wk = ol.apply_rope(ol.ORACLE.oracle_apply_rope, q[i], k[i], pos, hd, H, KV, theta)
                        host_k[layer, table[pos // bs], pos % bs] = wk
                        host_v[layer, table[pos // bs], pos % bs] = v[i]
                        want = ol.oracle_paged_attention(wq, host_k[layer], host_v[layer], table, pos + 1, bs, hd, H, KV)
                        assert close(q_rot[i], wq), (step, layer, rid)
                        assert close(got[i], want), (step, layer, rid, float(np.abs(got[i] - want).max()))
                batch.end_step()
                for rid in ids:
                    lens[rid] += 1
                    assert batch.request_len(rid) == lens[rid]
            for rid in list(running):
                if lens[rid] >= plan[rid][1]:
                    batch.finish_request(rid)
                    ref_bm.do(["req_free", rid])
                    running.remove(rid)
            assert bm.get_num_free_blocks() == ref_bm.do(["num_free"])
            step += 1
        assert not running and batch.num_requests() == 0
        assert bm.get_num_free_blocks() == nb  # every block came back
        st = batch.stats()
        assert st["patch_steps"] > 20 and 5 <= st["full_uploads"] <= 16, st  # tables re-sent only when the list changes
    ref_bm.close()


def test_out_of_blocks_leaves_the_batch_unchanged():
    L, H, KV, hd, bs = 1, 4, 2, 32, 4
    bm = nv.BlockManager(3, bs)
    with nv.PagedKVPool(L, 3, bs, KV, hd) as pool, nv.DecodeBatch(pool, bm, 4, 4) as batch:
        for rid in (1, 2, 3, 4):
            batch.add_request(rid)
        with pytest.raises(nv.NvkError) as e:  # four requests at pos 0 need four blocks, three exist
            batch.begin_step([1, 2, 3, 4], H)
        assert e.value.code == nv._capi.ERR_OOM and "Out of memory: no free blocks" in e.value.message
        assert bm.get_num_free_blocks() == 3 and all(batch.request_blocks(r) == [] for r in (1, 2, 3, 4))
        batch.begin_step([1, 2, 3], H)
        with pytest.raises(nv.NvkError):  # a step is in flight
            batch.begin_step([1], H)
        with pytest.raises(nv.NvkError):
            batch.finish_request(1)
        batch.end_step()
        assert [batch.request_blocks(r) for r in (1, 2, 3)] == [[0], [1], [2]]
        with pytest.raises(nv.NvkError):
            batch.begin_step([1, 1], H)  # listed twice
        with pytest.raises(nv.NvkError):
            batch.begin_step([9], H)  # not running
        with pytest.raises(nv.NvkError):
            batch.add_request(2)  # already running
        batch.finish_request(2)
        batch.begin_step([4], H)  # request 4 gets request 2's block back: lowest free id first
        batch.end_step()
        assert batch.request_blocks(4) == [1]


def test_plain_append_mode_matches_append_batch():
    """rope_theta = 0: nvk_batch_layer appends K as it is and leaves q alone."""
    L, H, KV, hd, bs = 1, 4, 2, 32, 4
    rng = np.random.default_rng(3)
    bm = nv.BlockManager(16, bs)
    with nv.PagedKVPool(L, 16, bs, KV, hd) as pool, nv.DecodeBatch(pool, bm, 3, 4) as batch:
        for rid in (5, 6, 7):
            batch.add_request(rid)
        hk = np.zeros((16, bs, KV, hd), np.float32)
        hv = np.zeros_like(hk)
        for t in range(10):
            batch.begin_step([5, 6, 7], H)
            q = rng.standard_normal((3, H, hd), dtype=np.float32)
            k = rng.standard_normal((3, KV, hd), dtype=np.float32)
            v = rng.standard_normal((3, KV, hd), dtype=np.float32)
            qd, outd = DeviceArray(q), DeviceArray(np.empty_like(q))
            batch.layer(0, qd, DeviceArray(k), DeviceArray(v), outd, 0.0)
            pool.synchronize()
            assert np.array_equal(qd.numpy(), q)
            got = outd.numpy()
            for i, rid in enumerate((5, 6, 7)):
                table = np.asarray(batch.request_blocks(rid), np.int32)
                hk[table[t // bs], t % bs], hv[table[t // bs], t % bs] = k[i], v[i]
                want = ol.oracle_paged_attention(q[i], hk, hv, table, t + 1, bs, hd, H, KV)
                assert close(got[i], want), (t, rid)
            batch.end_step()


def test_chunked_prefill_beside_decode_in_one_batch():
    """ScheduledBatch's two lists in one step (scheduler.hpp:24-43): running requests decode one token while newly
    admitted ones feed their prompts in chunks of at most 24 tokens; per layer rope + append + attention for all.
    Every output row is checked against the checker's rope -> append -> paged_attention for that token."""
    L, H, KV, hd, bs = 2, 8, 2, 64, 8
    nb, max_seqs, max_blocks, chunk_cap = 96, 6, 12, 24
    theta = 10000.0
    rng = np.random.default_rng(33)
    # request id -> (step it joins, prompt tokens, tokens to generate after the prompt)
    plan = {1: (0, 50, 12), 2: (0, 7, 20), 3: (2, 61, 5), 4: (4, 24, 9), 5: (9, 1, 6)}
    host_k = np.zeros((L, nb, bs, KV, hd), np.float32)
    host_v = np.zeros_like(host_k)
    fed, running = {}, []
    bm = nv.BlockManager(nb, bs)
    steps_with_both = 0
    with nv.PagedKVPool(L, nb, bs, KV, hd) as pool, nv.DecodeBatch(pool, bm, max_seqs, max_blocks) as batch:
        step = 0
        while step < 80 and (running or any(j >= step for j, _, _ in plan.values())):
            for rid, (join, _, _) in plan.items():
                if join == step:
                    batch.add_request(rid)
                    running.append(rid)
                    fed[rid] = 0
            decode_ids = [r for r in running if fed[r] >= plan[r][1]]
            prefill_ids = [r for r in running if fed[r] < plan[r][1]]
            ids = decode_ids + prefill_ids  # decode requests first
            chunks = [1] * len(decode_ids) + [min(chunk_cap, plan[r][1] - fed[r]) for r in prefill_ids]
            if ids:
                steps_with_both += bool(decode_ids) and any(c > 1 for c in chunks)
                rows = batch.begin_step_chunks(ids, chunks, H)
                assert rows == sum(chunks)
                for layer in range(L):
                    q = rng.standard_normal((rows, H, hd), dtype=np.float32)
                    k = rng.standard_normal((rows, KV, hd), dtype=np.float32)
                    v = rng.standard_normal((rows, KV, hd), dtype=np.float32)
                    qd, kd, vd = DeviceArray(q), DeviceArray(k), DeviceArray(v)
                    outd = DeviceArray(np.full_like(q, np.nan))
                    batch.layer(layer, qd, kd, vd, outd, theta)
                    pool.synchronize()
                    got = outd.numpy()
                    r0 = 0
                    for rid, c in zip(ids, chunks):
                        table = np.asarray(batch.request_blocks(rid), np.int32)
                        for b in range(c):  # append the whole chunk first (forward_chunk's order), then attend
                            pos = fed[rid] + b
                            _, wk = ol.apply_rope(ol.ORACLE.oracle_apply_rope, q[r0 + b], k[r0 + b], pos, hd, H, KV, theta)
                            host_k[layer, table[pos // bs], pos % bs] = wk
                            host_v[layer, table[pos // bs], pos % bs] = v[r0 + b]
                        for b in range(c):
                            pos = fed[rid] + b
                            wq, _ = ol.apply_rope(ol.ORACLE.oracle_apply_rope, q[r0 + b], k[r0 + b], pos, hd, H, KV, theta)
                            want = ol.oracle_paged_attention(wq, host_k[layer], host_v[layer], table, pos + 1, bs, hd, H, KV)
                            assert close(got[r0 + b], want), (step, layer, rid, b, float(np.abs(got[r0 + b] - want).max()))
                        r0 += c
                batch.end_step()
                for rid, c in zip(ids, chunks):
                    fed[rid] += c
                    assert batch.request_len(rid) == fed[rid]
                    assert len(batch.request_blocks(rid)) == (fed[rid] + bs - 1) // bs
            for rid in list(running):
                if fed[rid] >= plan[rid][1] + plan[rid][2]:
                    batch.finish_request(rid)
                    running.remove(rid)
            step += 1
        assert not running and bm.get_num_free_blocks() == nb
        assert steps_with_both >= 3  # the schedule really mixes decode rows and prefill chunks in one step


def test_layer_from_the_hidden_state():
    """nvk_batch_layer_hidden = model.hpp:161-221 for every row of a mixed step: rms_norm, q/k/v projections, RoPE,
    append, attention -- against the checker doing the same per token on the host."""
    L, H, KV, hd, bs, D = 2, 8, 2, 64, 8, 192
    nb, max_blocks = 64, 8
    rng = np.random.default_rng(91)
    sc = np.float32(1.0 / np.sqrt(D))
    W = [dict(g=rng.random(D, dtype=np.float32) + np.float32(0.5),
              wq=rng.standard_normal((H * hd, D), dtype=np.float32) * sc,
              wk=rng.standard_normal((KV * hd, D), dtype=np.float32) * sc,
              wv=rng.standard_normal((KV * hd, D), dtype=np.float32) * sc) for _ in range(L)]
    host_k = np.zeros((L, nb, bs, KV, hd), np.float32)
    host_v = np.zeros_like(host_k)
    bm = nv.BlockManager(nb, bs)
    fed = {1: 0, 2: 0, 3: 0}
    schedule = [([1, 2], [19, 5]), ([2, 1, 3], [1, 1, 21]), ([1, 2, 3], [1, 1, 1]), ([1, 2, 3], [1, 1, 1])]
    with nv.PagedKVPool(L, nb, bs, KV, hd) as pool, nv.DecodeBatch(pool, bm, 4, max_blocks) as batch:
        dW = [{k: DeviceArray(v) for k, v in w.items()} for w in W]
        for rid in fed:
            batch.add_request(rid)
        for ids, chunks in schedule:
            rows = batch.begin_step_chunks(ids, chunks, H)
            for layer in range(L):
                x = rng.standard_normal((rows, D), dtype=np.float32)
                outd = DeviceArray(np.full((rows, H, hd), np.nan, np.float32))
                w = dW[layer]
                batch.layer_hidden(layer, DeviceArray(x), D, w["g"], w["wq"], w["wk"], w["wv"], outd, 10000.0)
                pool.synchronize()
                got = outd.numpy()
                xb = ol.batch_rms_norm(ol.ORACLE.oracle_batch_rms_norm, x, W[layer]["g"])
                q, k, v = (ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, xb, W[layer][n]) for n in ("wq", "wk", "wv"))
                r0 = 0
                for rid, c in zip(ids, chunks):
                    table = np.asarray(batch.request_blocks(rid), np.int32)
                    rq = []
                    for b in range(c):
                        pos = fed[rid] + b
                        wq_, wk_ = ol.apply_rope(ol.ORACLE.oracle_apply_rope, q[r0 + b], k[r0 + b], pos, hd, H, KV, 10000.0)
                        host_k[layer, table[pos // bs], pos % bs] = wk_.reshape(KV, hd)
                        host_v[layer, table[pos // bs], pos % bs] = v[r0 + b].reshape(KV, hd)
                        rq.append(wq_.reshape(H, hd))
                    for b in range(c):
                        pos = fed[rid] + b
                        want = ol.oracle_paged_attention(rq[b], host_k[layer], host_v[layer], table, pos + 1, bs, hd, H, KV)
                        assert close(got[r0 + b], want), (ids, layer, rid, b, float(np.abs(got[r0 + b] - want).max()))
                    r0 += c
            batch.end_step()
            for rid, c in zip(ids, chunks):
                fed[rid] += c


@pytest.mark.parametrize("n_req", [1, 2, 5, 16, 27, 40, 70])
def test_steps_of_few_decoding_rows_fuse_rope_and_append_into_the_projection(n_req):
    """With up to 16 decoding rows (96 where every row tile of the product has a workgroup of its own: 16 x 16 tiles at
    27 and 40 rows here, 32 x 32 at 70) nvk_batch_layer_hidden runs the q/k/v projections, RoPE and the KV append as ONE
    launch (the epilogue of the GEMV, or of the 16-row MFMA kernel, writes where nvk_rope_append_kernel would).  Same arithmetic: the attention output
    and every K/V block must equal, bit for bit, the piecewise path (nvk_qkv_project_device + nvk_batch_layer) run on
    a second pool with the same history."""
    L, H, KV, hd, bs, D = 2, 8, 2, 64, 8, 192
    nb, max_blocks = 320, 8
    rng = np.random.default_rng(92)
    sc = np.float32(1.0 / np.sqrt(D))
    W = [dict(g=rng.random(D, dtype=np.float32) + np.float32(0.5),
              wq=rng.standard_normal((H * hd, D), dtype=np.float32) * sc,
              wk=rng.standard_normal((KV * hd, D), dtype=np.float32) * sc,
              wv=rng.standard_normal((KV * hd, D), dtype=np.float32) * sc) for _ in range(L)]
    ids = list(range(1, n_req + 1))
    prompt = ([13, 6, 9, 3, 11, 1, 20, 7, 2, 15, 4, 8, 10, 5, 12, 14] * 5)[:n_req]
    xs = [[rng.standard_normal((sum(prompt) if s == 0 else n_req, D), dtype=np.float32) for _ in range(L)] for s in range(12)]
    outs, caches = [], []
    for fused in (True, False):
        bm = nv.BlockManager(nb, bs)
        with nv.PagedKVPool(L, nb, bs, KV, hd) as pool, nv.DecodeBatch(pool, bm, max(2, n_req), max_blocks) as batch:
            dW = [{k: DeviceArray(v) for k, v in w.items()} for w in W]
            for rid in ids:
                batch.add_request(rid)
            got = []
            for s in range(12):
                chunks = prompt if s == 0 else [1] * n_req
                rows = batch.begin_step_chunks(ids, chunks, H)
                for layer in range(L):
                    x, w = DeviceArray(xs[s][layer]), dW[layer]
                    outd = DeviceArray(np.full((rows, H, hd), np.nan, np.float32))
                    if fused or s == 0:
                        batch.layer_hidden(layer, x, D, w["g"], w["wq"], w["wk"], w["wv"], outd, 10000.0)
                    else:
                        q = DeviceArray(np.zeros((rows, H * hd), np.float32))
                        k = DeviceArray(np.zeros((rows, KV * hd), np.float32))
                        v = DeviceArray(np.zeros((rows, KV * hd), np.float32))
                        xb = DeviceArray(np.zeros((rows, D), np.float32))
                        nv.qkv_project_device(q, k, v, x, xb, w["g"], w["wq"], w["wk"], w["wv"], rows, D, H * hd, KV * hd,
                                              stream=pool.stream)
                        batch.layer(layer, q, k, v, outd, 10000.0)
                    pool.synchronize()
                    got.append(outd.numpy().copy())
                batch.end_step()
            outs.append(got)
            blocks = sorted(b for rid in ids for b in batch.request_blocks(rid))
            caches.append([pool.download_block(layer, b) for layer in range(L) for b in blocks])
    for a, b in zip(outs[0], outs[1]):
        assert np.isfinite(a).all() and np.array_equal(a, b)
    for (ka, va), (kb, vb) in zip(caches[0], caches[1]):
        assert np.array_equal(ka, kb) and np.array_equal(va, vb)
