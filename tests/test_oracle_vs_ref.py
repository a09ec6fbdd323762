"""Pins oracle/nvk_oracle.c to the real reference, bit for bit, on many seeded inputs.

Runs where oracle/_ref/libnvk_ref.so exists (built from /root/reference by `make ref`;
it travels to the GPU box as a built file).  Elsewhere it is skipped and the committed
fixtures (test_oracle_golden.py) carry the pin.
"""
import numpy as np
import pytest

import oracle_lib as ol
from golden_defs import BmCLib, bm_trace_script, replay
from nano_vllm_amd.workloads import shuffled_permutation, splitmix64, uniform

pytestmark = pytest.mark.skipif(ol.REF is None or ol.ORACLE is None,
                                reason="oracle/_ref/libnvk_ref.so not built here")


@pytest.mark.parametrize("seed", range(40))
def test_paged_attention_random_geometry(seed):
    d = splitmix64(900 + seed, np.arange(8, dtype=np.uint64)).tolist()
    KV = [1, 2, 3, 4, 6, 8][d[0] % 6]
    H = KV * [1, 2, 3, 4, 8][d[1] % 5]
    hd = [4, 16, 20, 48, 64, 96, 128][d[2] % 7]
    bs = [1, 4, 8, 12, 16, 32][d[3] % 6]
    T = int(d[4] % 300)
    nb = (T + bs - 1) // bs + 2
    table = shuffled_permutation(nb, seed)[: max((T + bs - 1) // bs, 1)].astype(np.int32)
    n = nb * bs * KV * hd
    k = uniform(10 + seed, 0, n).reshape(nb, bs, KV, hd)
    v = uniform(50 + seed, 0, n).reshape(nb, bs, KV, hd)
    q = (uniform(90 + seed, 0, H * hd) * np.float32(1 + d[5] % 7)).reshape(H, hd)
    a = ol.oracle_paged_attention(q, k, v, table, T, bs, hd, H, KV)
    b = ol.ref_paged_attention(q, k, v, table, T, bs, hd, H, KV)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    if T > 0:
        k_lin = np.stack([k[table[t // bs], t % bs] for t in range(T)])
        v_lin = np.stack([v[table[t // bs], t % bs] for t in range(T)])
        c = ol.standard_attention(ol.ORACLE.oracle_standard_attention, q, k_lin, v_lin, T - 1, hd, H, KV, T)
        e = ol.standard_attention(ol.REF.ref_standard_attention, q, k_lin, v_lin, T - 1, hd, H, KV, T)
        assert np.array_equal(c.view(np.uint32), e.view(np.uint32))
        assert np.array_equal(a.view(np.uint32), e.view(np.uint32))  # paged == contiguous


def test_block_manager_scripts_and_random_ops():
    for script in bm_trace_script():
        a = replay(BmCLib(ol.ORACLE, "oracle_", script["num_blocks"], script["block_size"]), script["ops"])
        b = replay(BmCLib(ol.REF, "ref_", script["num_blocks"], script["block_size"]), script["ops"])
        assert a == b, script["name"]
    for seed in range(5):
        nb = int(3 + splitmix64(seed, np.array([0], np.uint64))[0] % 200)
        draws = splitmix64(4000 + seed, np.arange(600, dtype=np.uint64)).tolist()
        ops = []
        for d in draws:
            r = d % 12
            arg = int((d >> 8) % (nb + 4)) - 2
            ops.append([["alloc"], ["alloc"], ["alloc"], ["free", arg], ["free", arg], ["alloc_seq", int((d >> 8) % 64)],
                        ["is_free", arg], ["num_free"], ["req_alloc", int(d >> 40) % 4, int((d >> 8) % 50)],
                        ["req_free", int(d >> 40) % 4], ["req_alloc_block", int(d >> 40) % 4],
                        ["req_blocks", int(d >> 40) % 4]][r])
        a = replay(BmCLib(ol.ORACLE, "oracle_", nb, 4), ops)
        b = replay(BmCLib(ol.REF, "ref_", nb, 4), ops)
        assert a == b, seed


def test_apply_rope_bit_exact():
    for seed, (H, KV, hd) in enumerate([(6, 6, 48), (32, 8, 128), (8, 2, 64), (4, 4, 20), (2, 1, 256)]):
        q = uniform(300 + seed, 0, H * hd) * np.float32(3)
        k = uniform(400 + seed, 0, KV * hd) * np.float32(3)
        for pos in (0, 1, 15, 255, 2047, 32767, 1_000_003):
            for theta in (10000.0, 500000.0):
                a = ol.apply_rope(ol.ORACLE.oracle_apply_rope, q, k, pos, hd, H, KV, theta)
                b = ol.apply_rope(ol.REF.ref_apply_rope, q, k, pos, hd, H, KV, theta)
                assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
                assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def test_chunked_prefill_attention_bit_exact():
    """oracle_chunked_attention vs the reference's LlamaModelChunked::chunked_attention, and both vs the
    decode attention applied per query (the identity the paged prefill kernels are checked through)."""
    for seed, (H, KV, hd, start, C_) in enumerate([(6, 6, 48, 0, 20), (8, 2, 64, 13, 9), (32, 8, 128, 50, 17), (4, 4, 20, 3, 5)]):
        T = start + C_
        q = uniform(500 + seed, 0, C_ * H * hd).reshape(C_, H, hd)
        k = uniform(510 + seed, 0, T * KV * hd).reshape(T, KV, hd)
        v = uniform(520 + seed, 0, T * KV * hd).reshape(T, KV, hd)
        a = ol.oracle_chunked_attention(q, k, v, start, KV)
        b = ol.ref_chunked_attention(q, k, v, start, KV)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for bq in (0, C_ - 1):
            one = ol.standard_attention(ol.REF.ref_standard_attention, q[bq], k, v, start + bq, hd, H, KV, T)
            assert np.array_equal(one.view(np.uint32), b[bq].view(np.uint32))


def test_rms_norm_and_matmul_bit_exact():
    """oracle_batch_rms_norm / oracle_batch_matmul vs Ops::rms_norm / Ops::matmul per row and vs the BatchOps forms."""
    for seed, (B, D, N) in enumerate([(1, 288, 288), (3, 64, 40), (5, 4096, 96), (33, 100, 7), (2, 8, 1)]):
        x = (uniform(600 + seed, 0, B * D) * np.float32(2)).reshape(B, D)
        g = (uniform(610 + seed, 0, D) + np.float32(1.5))
        w = (uniform(620 + seed, 0, N * D) * np.float32(0.1)).reshape(N, D)
        a = ol.batch_rms_norm(ol.ORACLE.oracle_batch_rms_norm, x, g)
        for fn in (ol.REF.ref_rms_norm_rows, ol.REF.ref_batch_rms_norm):
            assert np.array_equal(a.view(np.uint32), ol.batch_rms_norm(fn, x, g).view(np.uint32))
        m = ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, a, w)
        for fn in (ol.REF.ref_matmul_rows, ol.REF.ref_batch_matmul):
            assert np.array_equal(m.view(np.uint32), ol.batch_matmul(fn, a, w).view(np.uint32))
