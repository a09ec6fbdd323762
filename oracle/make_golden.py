#!/usr/bin/env python3
"""oracle/make_golden.py -- TEST INFRASTRUCTURE: generates tests/golden/*.

Runs only in the build container, where /root/reference exists and
oracle/_ref/libnvk_ref.so (the reference's own headers behind oracle/ref_shim.cpp,
built by `make ref` at -O2 -ffp-contract=off) can be loaded.  It calls the
*reference* functions on seeded inputs and stores inputs-by-seed plus expected
outputs; the fixtures are data only (no reference source in any form).

  F1  tests/golden/block_manager_traces.json   BlockManager op traces -> results
  F2  tests/golden/attention_cases.npz         single-sequence paged_attention outputs
  F3  (same file) standard_attention outputs on the same rows (paged == contiguous)
  F6  tests/golden/prefill_cases.npz           chunked_attention outputs (experimental/chunked_prefill)
  F8  tests/golden/tokenizer_cases.json        Tokenizer::encode / decode over a synthetic tokenizer.bin
  F7  tests/golden/qkv_cases.npz               Ops::rms_norm + Ops::matmul (q/k/v projections), = the BatchOps forms
  F4  tests/golden/tiny_llama_stream.npz       synthetic llama2.c checkpoint: token stream,
                                               logits, attention outputs, F5 block tables

Usage: python oracle/make_golden.py
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ol  # noqa: E402
from golden_defs import (ATTENTION_CASES, PREFILL_CASES, QKV_CASES, TINY_MODELS, TOKENIZER_TEXTS, qkv_case_inputs, write_tokenizer, BmCLib, attention_case_inputs,  # noqa: E402
                         bm_trace_script, prefill_case_inputs, replay, write_checkpoint)

GOLD = os.path.join(ROOT, "tests", "golden")


def p(a):
    return C.c_void_p(a.ctypes.data)


def main():
    if ol.REF is None:
        sys.exit("oracle/_ref/libnvk_ref.so missing: run `make ref` where /root/reference exists")
    os.makedirs(GOLD, exist_ok=True)

    # ---- F1 ------------------------------------------------------------------
    traces = []
    for script in bm_trace_script():
        res = replay(BmCLib(ol.REF, "ref_", script["num_blocks"], script["block_size"]), script["ops"])
        traces.append({"name": script["name"], "num_blocks": script["num_blocks"],
                       "block_size": script["block_size"], "ops": script["ops"], "results": res})
    with open(os.path.join(GOLD, "block_manager_traces.json"), "w") as fh:
        json.dump({"generator": "oracle/make_golden.py (reference BlockManager via oracle/_ref)", "traces": traces},
                  fh, indent=0)
    print(f"F1: {len(traces)} traces, {sum(len(t['ops']) for t in traces)} ops")

    # ---- F2 / F3 ---------------------------------------------------------------
    store = {}
    for case in ATTENTION_CASES:
        name, H, KV, hd, bs, T = case["name"], case["n_heads"], case["n_kv_heads"], case["head_dim"], \
            case["block_size"], case["num_tokens"]
        q, k_pool, v_pool, table = attention_case_inputs(case)
        out = ol.ref_paged_attention(q, k_pool, v_pool, table, T, bs, hd, H, KV)
        store[name + "/out"] = out
        # F3: the same rows laid out contiguously, through the reference's standard_attention
        if T > 0:
            k_lin = np.empty((T, KV, hd), np.float32)
            v_lin = np.empty((T, KV, hd), np.float32)
            for t in range(T):
                k_lin[t] = k_pool[table[t // bs], t % bs]
                v_lin[t] = v_pool[table[t // bs], t % bs]
            std = ol.standard_attention(ol.REF.ref_standard_attention, q, k_lin, v_lin, T - 1, hd, H, KV, T)
            store[name + "/std"] = std
            assert np.array_equal(std, out), name  # the reference's own structural identity
    np.savez_compressed(os.path.join(GOLD, "attention_cases.npz"), **store)
    print(f"F2/F3: {len(ATTENTION_CASES)} attention cases")

    # ---- F6: chunked-prefill attention (experimental/chunked_prefill) -------------
    store = {}
    for case in PREFILL_CASES:
        q, k_lin, v_lin = prefill_case_inputs(case)
        store[case["name"] + "/out"] = ol.ref_chunked_attention(q, k_lin, v_lin, case["start_pos"], case["n_kv_heads"])
    np.savez_compressed(os.path.join(GOLD, "prefill_cases.npz"), **store)
    print(f"F6: {len(PREFILL_CASES)} chunked-prefill cases")

    # ---- F7: rms_norm + q/k/v projections (model.hpp:161-166) ----------------------
    store = {}
    for case in QKV_CASES:
        x, g, wq, wk, wv = qkv_case_inputs(case)
        xb = ol.batch_rms_norm(ol.REF.ref_rms_norm_rows, x, g)
        assert np.array_equal(xb, ol.batch_rms_norm(ol.REF.ref_batch_rms_norm, x, g)), case["name"]
        store[case["name"] + "/xb"] = xb
        for key, w in (("q", wq), ("k", wk), ("v", wv)):
            out = ol.batch_matmul(ol.REF.ref_matmul_rows, xb, w)
            assert np.array_equal(out, ol.batch_matmul(ol.REF.ref_batch_matmul, xb, w)), case["name"]
            store[case["name"] + "/" + key] = out
    np.savez_compressed(os.path.join(GOLD, "qkv_cases.npz"), **store)
    print(f"F7: {len(QKV_CASES)} rms_norm + qkv projection cases")

    # ---- F8: tokenizer.bin + Tokenizer::encode / decode (tokenizer.hpp) ----------------
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "tokenizer.bin")
        write_tokenizer(path, 512)
        tok = ol.REF.ref_tokenizer_create(path.encode(), 512)
        assert tok
        f8 = {"vocab_size": 512,
              "encode": [{"text_hex": t.hex(), "bos": bos, "eos": eos, "ids": ol.ref_tokenizer_encode(tok, t, bos, eos)}
                         for t in TOKENIZER_TEXTS for bos, eos in ((True, False), (True, True))
                         ] + [{"text_hex": t.hex(), "bos": False, "eos": False, "ids": ol.ref_tokenizer_encode(tok, t, False, False)}
                              for t in TOKENIZER_TEXTS if len(t) > 0],
              "decode_hex": [ol.ref_tokenizer_decode(tok, i).hex() for i in range(-1, 513)]}
        ol.REF.ref_tokenizer_destroy(tok)
    with open(os.path.join(GOLD, "tokenizer_cases.json"), "w") as fh:
        json.dump(f8, fh)
    print(f"F8: {len(f8['encode'])} encodings, {len(f8['decode_hex'])} decoded pieces")

    # ---- F4 / F5 ---------------------------------------------------------------
    store = {}
    for spec in TINY_MODELS:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "model.bin")
            write_checkpoint(path, spec)
            m = ol.REF.ref_model_create(path.encode(), 1, spec["block_size"], spec["num_blocks"])
            assert m, spec["name"]
            vocab = spec["vocab_size"]
            logits = np.empty(vocab, np.float32)
            attn = np.empty(spec["dim"], np.float32)
            # teacher-forced pass: fixed inputs, record argmax + full logits + last-layer attention output
            forced = spec["forced_tokens"]
            tf_argmax, tf_logits, tf_attn = [], [], []
            for pos, tok in enumerate(forced):
                rc = ol.REF.ref_model_forward(m, int(tok), pos, p(logits))
                assert rc == 0
                ol.REF.ref_model_last_attn_out(m, p(attn), spec["dim"])
                tf_argmax.append(int(ol.REF.ref_greedy_sample(p(logits), vocab)))
                tf_logits.append(logits.copy())
                tf_attn.append(attn.copy())
            tabs = []
            buf = (C.c_int * spec["num_blocks"])()
            for layer in range(spec["n_layers"]):
                n = ol.REF.ref_model_block_table(m, layer, buf, spec["num_blocks"])
                tabs.append(list(buf[:n]))
            ol.REF.ref_model_destroy(m)

            # free-running greedy pass (run_single_prompt, runner.hpp:39-57): prompt then argmax feedback
            m = ol.REF.ref_model_create(path.encode(), 1, spec["block_size"], spec["num_blocks"])
            prompt = spec["prompt_tokens"]
            pos = 0
            for tok in prompt[:-1]:
                assert ol.REF.ref_model_forward(m, int(tok), pos, None) == 0
                pos += 1
            tok = int(prompt[-1])
            stream = []
            for _ in range(spec["gen_steps"]):
                assert ol.REF.ref_model_forward(m, tok, pos, p(logits)) == 0
                pos += 1
                tok = int(ol.REF.ref_greedy_sample(p(logits), vocab))
                stream.append(tok)
            ol.REF.ref_model_destroy(m)

        n = spec["name"]
        store[n + "/tf_argmax"] = np.asarray(tf_argmax, np.int32)
        store[n + "/tf_logits"] = np.asarray(tf_logits, np.float32)
        store[n + "/tf_attn"] = np.asarray(tf_attn, np.float32)
        store[n + "/greedy_stream"] = np.asarray(stream, np.int32)
        for layer, t in enumerate(tabs):
            store[f"{n}/table_{layer}"] = np.asarray(t, np.int32)
        print(f"F4: {n}: {len(forced)} forced steps, greedy stream {stream[:12]}... "
              f"({len(set(stream))} distinct tokens)")
    np.savez_compressed(os.path.join(GOLD, "tiny_llama_stream.npz"), **store)


if __name__ == "__main__":
    main()
