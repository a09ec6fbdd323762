"""The CPU checker (oracle/nvk_oracle.c) against the committed golden vectors.

The vectors were recorded from the reference's own headers (oracle/make_golden.py);
the oracle is built with the same flags (-O2 -ffp-contract=off) and keeps the
reference's loop order, so every comparison here is bit-exact.
"""
import ctypes as C
import json
import os
import tempfile

import numpy as np
import pytest

import oracle_lib as ol
from golden_defs import (ATTENTION_CASES, PREFILL_CASES, QKV_CASES, TINY_MODELS, qkv_case_inputs, BmCLib, attention_case_inputs,
                         paged_from_contiguous, prefill_case_inputs, replay, write_checkpoint)

pytestmark = pytest.mark.skipif(ol.ORACLE is None, reason="oracle/libnvk_oracle.so not built (make oracle)")


def _p(a):
    return C.c_void_p(a.ctypes.data)


def test_block_manager_traces(golden_dir):
    with open(os.path.join(golden_dir, "block_manager_traces.json")) as fh:
        traces = json.load(fh)["traces"]
    assert len(traces) >= 4
    for t in traces:
        got = replay(BmCLib(ol.ORACLE, "oracle_", t["num_blocks"], t["block_size"]), t["ops"])
        assert got == t["results"], t["name"]


@pytest.mark.parametrize("case", ATTENTION_CASES, ids=[c["name"] for c in ATTENTION_CASES])
def test_attention_case_bit_exact(case, golden_dir):
    gold = np.load(os.path.join(golden_dir, "attention_cases.npz"))
    q, k_pool, v_pool, table = attention_case_inputs(case)
    out = ol.oracle_paged_attention(q, k_pool, v_pool, table, case["num_tokens"], case["block_size"],
                                    case["head_dim"], case["n_heads"], case["n_kv_heads"])
    assert np.array_equal(out, gold[case["name"] + "/out"])
    if case["num_tokens"] > 0:
        # paged == contiguous on the same rows (the reference's structural identity)
        T, bs = case["num_tokens"], case["block_size"]
        k_lin = np.stack([k_pool[table[t // bs], t % bs] for t in range(T)])
        v_lin = np.stack([v_pool[table[t // bs], t % bs] for t in range(T)])
        std = ol.standard_attention(ol.ORACLE.oracle_standard_attention, q, k_lin, v_lin, T - 1, case["head_dim"],
                                    case["n_heads"], case["n_kv_heads"], T)
        assert np.array_equal(std, gold[case["name"] + "/std"])
        assert np.array_equal(std, out)
    else:
        assert not out.any()


@pytest.mark.parametrize("case", PREFILL_CASES, ids=[c["name"] for c in PREFILL_CASES])
def test_chunked_prefill_case_bit_exact(case, golden_dir):
    """F6: the reference's chunked_attention (contiguous cache), and the paged form on the same rows."""
    gold = np.load(os.path.join(golden_dir, "prefill_cases.npz"))[case["name"] + "/out"]
    q, k_lin, v_lin = prefill_case_inputs(case)
    out = ol.oracle_chunked_attention(q, k_lin, v_lin, case["start_pos"], case["n_kv_heads"])
    assert np.array_equal(out.view(np.uint32), gold.view(np.uint32))
    k_pool, v_pool, table = paged_from_contiguous(k_lin, v_lin, 16, 8000 + case["index"])
    paged = ol.oracle_paged_prefill(q, k_pool, v_pool, table, case["start_pos"], 16, case["n_kv_heads"])
    assert np.array_equal(paged.view(np.uint32), gold.view(np.uint32))  # paged == contiguous, per query


@pytest.mark.parametrize("case", QKV_CASES, ids=[c["name"] for c in QKV_CASES])
def test_qkv_projection_case_bit_exact(case, golden_dir):
    """F7: Ops::rms_norm and the three Ops::matmul of model.hpp:161-166, recorded from the reference."""
    gold = np.load(os.path.join(golden_dir, "qkv_cases.npz"))
    x, g, wq, wk, wv = qkv_case_inputs(case)
    xb = ol.batch_rms_norm(ol.ORACLE.oracle_batch_rms_norm, x, g)
    assert np.array_equal(xb.view(np.uint32), gold[case["name"] + "/xb"].view(np.uint32))
    for key, w in (("q", wq), ("k", wk), ("v", wv)):
        out = ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, xb, w)
        assert np.array_equal(out.view(np.uint32), gold[case["name"] + "/" + key].view(np.uint32)), key


@pytest.mark.parametrize("spec", TINY_MODELS, ids=[s["name"] for s in TINY_MODELS])
def test_tiny_llama_stream(spec, golden_dir):
    gold = np.load(os.path.join(golden_dir, "tiny_llama_stream.npz"))
    n = spec["name"]
    vocab = spec["vocab_size"]
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "model.bin")
        write_checkpoint(path, spec)
        m = ol.ORACLE.oracle_model_create(path.encode(), spec["block_size"], spec["num_blocks"])
        assert m
        logits = np.empty(vocab, np.float32)
        attn = np.empty(spec["dim"], np.float32)
        for pos, tok in enumerate(spec["forced_tokens"]):
            assert ol.ORACLE.oracle_model_forward(m, int(tok), pos, _p(logits)) == 0
            ol.ORACLE.oracle_model_last_attn_out(m, _p(attn), spec["dim"])
            assert np.array_equal(logits, gold[n + "/tf_logits"][pos]), pos
            assert np.array_equal(attn, gold[n + "/tf_attn"][pos]), pos
            assert ol.ORACLE.oracle_greedy_sample(_p(logits), vocab) == gold[n + "/tf_argmax"][pos]
        # F5: all layers share one free list, so layer l owns blocks l, L+l, 2L+l, ...
        buf = (C.c_int * spec["num_blocks"])()
        L = spec["n_layers"]
        for layer in range(L):
            cnt = ol.ORACLE.oracle_model_block_table(m, layer, buf, spec["num_blocks"])
            tab = list(buf[:cnt])
            assert tab == gold[f"{n}/table_{layer}"].tolist()
            assert tab == [layer + L * i for i in range(len(tab))]
        ol.ORACLE.oracle_model_destroy(m)

        m = ol.ORACLE.oracle_model_create(path.encode(), spec["block_size"], spec["num_blocks"])
        prompt = spec["prompt_tokens"]
        pos = 0
        for tok in prompt[:-1]:
            assert ol.ORACLE.oracle_model_forward(m, int(tok), pos, None) == 0
            pos += 1
        tok, stream = int(prompt[-1]), []
        for _ in range(spec["gen_steps"]):
            assert ol.ORACLE.oracle_model_forward(m, tok, pos, _p(logits)) == 0
            pos += 1
            tok = ol.ORACLE.oracle_greedy_sample(_p(logits), vocab)
            stream.append(tok)
        ol.ORACLE.oracle_model_destroy(m)
        assert stream == gold[n + "/greedy_stream"].tolist()
        assert len(set(stream)) > 10  # the fixture pins something: not a collapsed constant stream


def test_model_out_of_blocks_is_reported():
    """The reference throws "Out of memory: no free blocks" (model.hpp:183-185) when the shared list runs dry."""
    spec = dict(TINY_MODELS[0])
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "model.bin")
        write_checkpoint(path, spec)
        # 3 layers x 16-token blocks: 4 blocks cover layer 0..2 at pos 0 and only layer 0 at pos 16
        m = ol.ORACLE.oracle_model_create(path.encode(), 16, 4)
        for pos in range(16):
            assert ol.ORACLE.oracle_model_forward(m, 1, pos, None) == 0
        assert ol.ORACLE.oracle_model_forward(m, 1, 16, None) == -1
        ol.ORACLE.oracle_model_destroy(m)
