"""BASELINE.json configs[0] on the GPU: the reference ENGINE, unmodified, with the drop-in in place.

oracle/_ref/libnvk_ref_dropin.so is the reference's own LlamaModel::forward
(include/core/model.hpp:150-247) compiled in the build container with
include/nvk_dropin/ in front of its include path, so that its one attention
call site (model.hpp:405-415 / :421-430) and its BlockManager resolve to
libnvk_paged_attn.so.  Nothing else of the engine changes: it still owns the
host KV slabs and memcpys rows into them (model.hpp:190-207), and the drop-in
stages the blocks a call names to the device (host-slab form of
nvk_paged_attention).  The library only exists where it was built from
/root/reference; it travels to the GPU box as a built file.

Pass = teacher-forced logits within tolerance of the recorded reference run,
identical argmax per step, identical greedy token stream, identical per-layer
block tables (golden F4/F5, tests/golden/tiny_llama_stream.npz).

Tolerances: the bars in this file (5e-4 + 1e-4 |ref| on the last layer's attention output, 2e-3 + 1e-4 |ref| on the
logits) are NOT north_star's per-call bar.  They are a stated deviation for quantities that sit behind L layers of
fp32 GEMVs whose inputs already differ in the last bits; what they guard is exact: the argmax at every position, the
greedy stream and the block tables.  The authoritative bar for the path itself, 1e-4 + 1e-5 |ref| per attention call,
is held where a single call can be compared: tests/test_gpu_attention.py.

The library is a built file that has to travel with the tree: with NVK_EXPECT_REF=1 (tests/conftest.py sets it
whenever a GPU is visible) its absence FAILS these tests instead of skipping them, so BASELINE.json configs[0] cannot
go untested without anyone noticing.  NVK_EXPECT_REF=0 restores the skip (a checkout built without /root/reference).
"""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

import oracle_lib as ol
from golden_defs import TINY_MODELS, write_checkpoint

pytestmark = pytest.mark.gpu

DROPIN_PATH = os.path.join(ol.ROOT, "oracle", "_ref", "libnvk_ref_dropin.so")


def _p(a):
    return C.c_void_p(a.ctypes.data)


@pytest.fixture(scope="module")
def engine():
    if not os.path.exists(DROPIN_PATH):
        why = "oracle/_ref/libnvk_ref_dropin.so was not built (needs /root/reference at build time: make ref)"
        if os.environ.get("NVK_EXPECT_REF", "0") == "1":
            pytest.fail(why + "; NVK_EXPECT_REF=1 says it should have travelled with the tree")
        pytest.skip(why)
    return ol._load_ref(DROPIN_PATH)


@pytest.mark.parametrize("paged", [1, 0], ids=["paged", "contiguous"])
@pytest.mark.parametrize("spec", TINY_MODELS, ids=[s["name"] for s in TINY_MODELS])
def test_reference_engine_with_drop_in(engine, spec, paged, golden_dir):
    gold = np.load(os.path.join(golden_dir, "tiny_llama_stream.npz"))
    n, vocab = spec["name"], spec["vocab_size"]
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "model.bin")
        write_checkpoint(path, spec)
        m = engine.ref_model_create(path.encode(), paged, spec["block_size"], spec["num_blocks"])
        assert m
        logits = np.empty(vocab, np.float32)
        attn = np.empty(spec["dim"], np.float32)
        worst = 0.0
        for pos, tok in enumerate(spec["forced_tokens"]):
            assert engine.ref_model_forward(m, int(tok), pos, _p(logits)) == 0
            engine.ref_model_last_attn_out(m, _p(attn), spec["dim"])
            want_attn = gold[n + "/tf_attn"][pos]
            # the LAST layer's attention output: its q/k/v come through L-1 earlier layers whose attention
            # outputs already differ from the recorded run in the last bits (the single-call 1e-4/1e-5 bar is
            # held in test_gpu_attention.py); values here reach |4|, so this is ~1e-4 relative
            assert np.allclose(attn, want_attn, atol=5e-4, rtol=1e-4), (pos, float(np.abs(attn - want_attn).max()))
            want = gold[n + "/tf_logits"][pos]
            worst = max(worst, float(np.abs(logits - want).max()))
            # logits sit behind L layers of fp32 GEMVs fed by attention outputs that differ in the last bits
            assert np.allclose(logits, want, atol=2e-3, rtol=1e-4), (pos, worst)
            assert engine.ref_greedy_sample(_p(logits), vocab) == gold[n + "/tf_argmax"][pos], pos
        if paged:
            buf = (C.c_int * spec["num_blocks"])()
            for layer in range(spec["n_layers"]):
                cnt = engine.ref_model_block_table(m, layer, buf, spec["num_blocks"])
                assert list(buf[:cnt]) == gold[f"{n}/table_{layer}"].tolist()  # F5, via the drop-in BlockManager
        engine.ref_model_destroy(m)

        # free-running greedy decode, the loop of run_single_prompt (runner.hpp:39-57)
        m = engine.ref_model_create(path.encode(), paged, spec["block_size"], spec["num_blocks"])
        prompt, pos = spec["prompt_tokens"], 0
        for tok in prompt[:-1]:
            assert engine.ref_model_forward(m, int(tok), pos, None) == 0
            pos += 1
        tok, stream = int(prompt[-1]), []
        for _ in range(spec["gen_steps"]):
            assert engine.ref_model_forward(m, tok, pos, _p(logits)) == 0
            pos += 1
            tok = int(engine.ref_greedy_sample(_p(logits), vocab))
            stream.append(tok)
        engine.ref_model_destroy(m)
        assert stream == gold[n + "/greedy_stream"].tolist()


@pytest.mark.parametrize("paged", [1, 0], ids=["paged", "contiguous"])
def test_append_only_host_cache_policy_same_stream_far_less_pcie(engine, paged, golden_dir):
    """NVK_HOST_CACHE_APPEND_ONLY: the engine's host slabs are mirrored on the device and only new rows are
    copied.  Same token stream as the recorded reference run; K/V bytes over PCIe drop from O(T^2) to O(T);
    a second request on the same model (pos back to 0) is noticed without being announced."""
    import nano_vllm_amd as nv

    spec = TINY_MODELS[1]
    gold = np.load(os.path.join(golden_dir, "tiny_llama_stream.npz"))
    n, vocab = spec["name"], spec["vocab_size"]
    logits = np.empty(vocab, np.float32)

    def greedy(m):
        prompt, pos = spec["prompt_tokens"], 0
        for tok in prompt[:-1]:
            assert engine.ref_model_forward(m, int(tok), pos, None) == 0
            pos += 1
        tok, stream = int(prompt[-1]), []
        for _ in range(spec["gen_steps"]):
            assert engine.ref_model_forward(m, tok, pos, _p(logits)) == 0
            pos += 1
            tok = int(engine.ref_greedy_sample(_p(logits), vocab))
            stream.append(tok)
        return stream, pos

    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "model.bin")
        write_checkpoint(path, spec)
        staged = {}
        for append_only in (False, True):
            nv.set_host_cache_policy(append_only)
            try:
                nv.host_cache_staged_bytes(reset=True)
                m = engine.ref_model_create(path.encode(), paged, spec["block_size"], spec["num_blocks"])
                stream, n_tok = greedy(m)
                assert stream == gold[n + "/greedy_stream"].tolist(), append_only
                staged[append_only] = nv.host_cache_staged_bytes(reset=True)
                engine.ref_model_destroy(m)
                if append_only:  # a fresh model very likely reuses the freed slabs' addresses: must not see stale rows
                    m = engine.ref_model_create(path.encode(), paged, spec["block_size"], spec["num_blocks"])
                    stream2, _ = greedy(m)
                    assert stream2 == gold[n + "/greedy_stream"].tolist()
                    engine.ref_model_destroy(m)
            finally:
                nv.set_host_cache_policy(False)
        row_bytes = spec["n_kv_heads"] * (spec["dim"] // spec["n_heads"]) * 4
        L = spec["n_layers"]
        assert staged[True] == 2 * row_bytes * n_tok * L  # every K and V row crossed PCIe exactly once
        assert staged[False] == 2 * row_bytes * L * n_tok * (n_tok + 1) // 2  # the whole context on every call
