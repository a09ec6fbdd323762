"""The reference's greedy token streams (fixture F4, recorded from LlamaModel::forward + the greedy sampler) reproduced
with every matmul, RMSNorm, RoPE, KV append and attention of the forward pass on the GPU through the C ABI
(include/core/model.hpp:150-247): per layer nvk_batch_layer_hidden (rms_norm + q/k/v + RoPE + append + paged
attention), nvk_batch_matmul_device for wo / w_gate / w_up / w_down / lm_head, nvk_batch_rms_norm_device for the
FFN and final norms.  Only the residual adds and SwiGLU (elementwise, model.hpp:227-239) run in numpy here.

Bar: identical greedy streams and teacher-forced argmax (BASELINE.json north_star: "identical greedy-decoded token
streams"); logits within 2e-3 + 1e-4 relative of the reference's.  That logit bar is a stated DEVIATION from
north_star's 1e-4 + 1e-5 |ref|, which is a per-call bar: logits sit behind every layer's GEMVs and attention calls, each
inside its own bar, and the differences compound.  What this file holds exactly is what the engine's user sees -- the
argmax at every position and the greedy streams; the per-call bar stays authoritative and is held in
tests/test_gpu_attention.py (attention) and tests/test_gpu_linear.py (projections)."""
import os

import numpy as np
import pytest

from golden_defs import TINY_MODELS, checkpoint_tensors

pytestmark = pytest.mark.gpu


class DeviceModel:
    def __init__(self, nv, DA, spec):
        self.nv, self.DA, self.spec = nv, DA, spec
        D, L, H, KV = spec["dim"], spec["n_layers"], spec["n_heads"], spec["n_kv_heads"]
        self.hd = D // H
        t = dict(checkpoint_tensors(spec))
        self.tok_emb = t["tok_emb"].reshape(spec["vocab_size"], D)
        per = lambda name, rows, cols: [DA(t[name].reshape(L, rows, cols)[l]) for l in range(L)]
        self.rms_att = [DA(t["rms_att"].reshape(L, D)[l]) for l in range(L)]
        self.rms_ffn = [DA(t["rms_ffn"].reshape(L, D)[l]) for l in range(L)]
        self.wq, self.wk, self.wv = per("wq", H * self.hd, D), per("wk", KV * self.hd, D), per("wv", KV * self.hd, D)
        self.wo = per("wo", D, H * self.hd)
        self.w_gate, self.w_up = per("w_gate", spec["hidden_dim"], D), per("w_up", spec["hidden_dim"], D)
        self.w_down = per("w_down", D, spec["hidden_dim"])
        self.rms_final = DA(t["rms_final"])
        self.lm_head = DA(t["tok_emb"] if spec["shared"] else t["lm_head"])  # shared-classifier rule, model.hpp:349-356
        self.bm = nv.BlockManager(spec["num_blocks"], spec["block_size"])
        self.pool = nv.PagedKVPool(L, spec["num_blocks"], spec["block_size"], KV, self.hd)
        self.batch = nv.DecodeBatch(self.pool, self.bm, 1, (spec["max_seq_len"] + spec["block_size"] - 1) // spec["block_size"])
        self.batch.add_request(0)

    def close(self):
        self.batch.close()
        self.pool.close()

    def matmul(self, x, w_dev, out_dim):
        out = self.DA(np.empty((1, out_dim), np.float32))
        self.nv.batch_matmul_device(out, self.DA(x.reshape(1, -1)), w_dev, 1, x.size, out_dim, stream=self.pool.stream)
        self.pool.synchronize()
        return out.numpy().reshape(-1)

    def rms_norm(self, x, w_dev):
        out = self.DA(np.empty((1, x.size), np.float32))
        self.nv.batch_rms_norm_device(out, self.DA(x.reshape(1, -1)), w_dev, 1, x.size, stream=self.pool.stream)
        self.pool.synchronize()
        return out.numpy().reshape(-1)

    def forward(self, token):
        s, nv = self.spec, self.nv
        D, H, Hd = s["dim"], s["n_heads"], s["hidden_dim"]
        x = self.tok_emb[token].copy()
        self.batch.begin_step([0], H)
        for l in range(s["n_layers"]):
            att = self.DA(np.empty((1, H, self.hd), np.float32))
            self.batch.layer_hidden(l, self.DA(x.reshape(1, D)), D, self.rms_att[l], self.wq[l], self.wk[l], self.wv[l], att)
            self.pool.synchronize()
            x = x + self.matmul(att.numpy().reshape(-1), self.wo[l], D)               # model.hpp:224-228
            xb = self.rms_norm(x, self.rms_ffn[l])                                    # :231
            gate, up = self.matmul(xb, self.w_gate[l], Hd), self.matmul(xb, self.w_up[l], Hd)
            hb = (gate / (np.float32(1) + np.exp(-gate, dtype=np.float32))) * up      # Ops::swiglu, activation.hpp:34-40
            x = x + self.matmul(hb.astype(np.float32), self.w_down[l], D)             # :235-239
        self.batch.end_step()
        return self.matmul(self.rms_norm(x, self.rms_final), self.lm_head, s["vocab_size"])  # :243-246


@pytest.mark.parametrize("spec", TINY_MODELS, ids=[s["name"] for s in TINY_MODELS])
def test_greedy_streams_with_the_forward_pass_on_the_gpu(spec, golden_dir):
    import nano_vllm_amd as nv
    from hipmem import DeviceArray

    gold = np.load(os.path.join(golden_dir, "tiny_llama_stream.npz"))
    n = spec["name"]
    m = DeviceModel(nv, DeviceArray, spec)
    try:
        for pos, tok in enumerate(spec["forced_tokens"]):
            logits = m.forward(int(tok))
            want = gold[n + "/tf_logits"][pos]
            assert np.all(np.abs(logits - want) <= 2e-3 + 1e-4 * np.abs(want)), (pos, float(np.abs(logits - want).max()))
            assert int(np.argmax(logits)) == int(gold[n + "/tf_argmax"][pos]), pos
    finally:
        m.close()
    m = DeviceModel(nv, DeviceArray, spec)
    try:
        prompt = spec["prompt_tokens"]
        for tok in prompt[:-1]:
            m.forward(int(tok))
        tok, stream = int(prompt[-1]), []
        for _ in range(spec["gen_steps"]):
            tok = int(np.argmax(m.forward(tok)))  # sampler.hpp:27-29: first maximum
            stream.append(tok)
        assert stream == gold[n + "/greedy_stream"].tolist()
    finally:
        m.close()


# ---- the same through the device-resident model (nvk_model_*): nothing but token ids and logits cross PCIe ----
@pytest.fixture(scope="module")
def checkpoints(tmp_path_factory):
    from golden_defs import write_checkpoint
    d = tmp_path_factory.mktemp("ckpt")
    out = {}
    for spec in TINY_MODELS:
        path = str(d / (spec["name"] + ".bin"))
        write_checkpoint(path, spec)
        out[spec["name"]] = path
    return out


@pytest.mark.parametrize("spec", TINY_MODELS, ids=[s["name"] for s in TINY_MODELS])
def test_device_model_forward_matches_the_reference_stream(spec, checkpoints, golden_dir):
    import nano_vllm_amd as nv

    gold = np.load(os.path.join(golden_dir, "tiny_llama_stream.npz"))
    n = spec["name"]
    with nv.LlamaModel(checkpoints[n], spec["block_size"], spec["num_blocks"]) as m:
        assert (m.dim, m.hidden_dim, m.n_layers, m.n_heads, m.n_kv_heads, m.vocab_size, m.max_seq_len) == \
            tuple(spec[k] for k in ("dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "max_seq_len"))
        assert m.weights_shared == spec["shared"]  # the shared-classifier rule, model.hpp:349-364
        for pos, tok in enumerate(spec["forced_tokens"]):
            logits = m.forward(int(tok), pos)
            want = gold[n + "/tf_logits"][pos]
            assert np.all(np.abs(logits - want) <= 2e-3 + 1e-4 * np.abs(want)), (pos, float(np.abs(logits - want).max()))
            assert int(np.argmax(logits)) == int(gold[n + "/tf_argmax"][pos]), pos
        with pytest.raises(nv.NvkError):
            m.forward(0, 3)  # pos must be the sequence's length
    with nv.LlamaModel(checkpoints[n], spec["block_size"], spec["num_blocks"]) as m:
        prompt = spec["prompt_tokens"]
        m.add_request(7)
        nxt, _ = m.step_chunks([7], [prompt])  # the whole prompt as one prefill chunk
        stream = [int(nxt[0])]
        for _ in range(spec["gen_steps"] - 1):
            nxt, _ = m.step([7], [stream[-1]])
            stream.append(int(nxt[0]))
        assert stream == gold[n + "/greedy_stream"].tolist()
        m.finish_request(7)
        assert m.free_blocks == spec["num_blocks"]


def test_device_model_batched_requests_equal_their_single_runs(checkpoints):
    """Continuous batching at the model level: requests with different prompts join at different steps (prompt as one
    chunk beside the others' decode tokens); every request's greedy stream equals the stream it produces alone."""
    import nano_vllm_amd as nv

    spec = TINY_MODELS[0]
    rng = np.random.default_rng(3)
    prompts = {r: [1] + rng.integers(3, spec["vocab_size"], size=k).tolist() for r, k in ((1, 4), (2, 11), (3, 1), (4, 23))}
    join = {1: 0, 2: 0, 3: 3, 4: 5}
    gen = 20
    alone = {}
    for r, prompt in prompts.items():
        with nv.LlamaModel(checkpoints[spec["name"]], spec["block_size"], spec["num_blocks"]) as m:
            m.add_request(0)
            nxt, _ = m.step_chunks([0], [prompt])
            s = [int(nxt[0])]
            while len(s) < gen:
                nxt, _ = m.step([0], [s[-1]])
                s.append(int(nxt[0]))
            alone[r] = s
    with nv.LlamaModel(checkpoints[spec["name"]], spec["block_size"], spec["num_blocks"], max_seqs=4) as m:
        streams, running = {}, []
        step = 0
        while step < 60 and (running or any(j >= step for j in join.values())):
            decode = [r for r in running if len(streams[r]) < gen]
            fresh = [r for r, j in join.items() if j == step]
            for r in fresh:
                m.add_request(r)
                streams[r] = []
            ids = decode + fresh
            if ids:
                nxt, _ = m.step_chunks(ids, [[streams[r][-1]] for r in decode] + [prompts[r] for r in fresh])
                for r, t in zip(ids, nxt.tolist()):
                    streams[r].append(int(t))
            running = [r for r in decode + fresh if len(streams[r]) < gen]
            for r in decode + fresh:
                if len(streams[r]) >= gen:
                    m.finish_request(r)
            step += 1
        assert streams == alone
        assert m.free_blocks == spec["num_blocks"]


@pytest.mark.parametrize("n_requests", [1, 3])
@pytest.mark.parametrize("spec", TINY_MODELS, ids=[s["name"] for s in TINY_MODELS])
def test_graph_replay_gives_the_bits_of_direct_launches(spec, n_requests, checkpoints):
    """Pure decode steps replay a hipGraph (recorded the second time a launch shape is seen, recorded again when the
    plan's shape changes as the contexts grow): every step's logits must equal, bit for bit, those of launching the
    same kernels one by one."""
    import nano_vllm_amd as nv

    rng = np.random.default_rng(11)
    prompts = [[1] + rng.integers(3, spec["vocab_size"], size=k).tolist() for k in (5, 9, 2)][:n_requests]
    ids = list(range(10, 10 + n_requests))
    steps = min(40, spec["max_seq_len"] - 12)
    runs = {}
    for graphs in (False, True):
        with nv.LlamaModel(checkpoints[spec["name"]], spec["block_size"], spec["num_blocks"], max_seqs=3) as m:
            m.use_graphs(graphs)
            for r in ids:
                m.add_request(r)
            nxt, logits = m.step_chunks(ids, prompts, True)
            got = [logits.copy()]
            for _ in range(steps):
                nxt, logits = m.step(ids, nxt.tolist(), True)
                got.append(logits.copy())
            runs[graphs] = np.stack(got)
    assert np.isfinite(runs[False]).all()
    assert np.array_equal(runs[True], runs[False])


@pytest.mark.parametrize("n_req", [1, 2, 3, 7])
@pytest.mark.parametrize("spec", TINY_MODELS, ids=[s["name"] for s in TINY_MODELS])
def test_norms_carried_between_the_gemvs(spec, n_req, checkpoints):
    """A step of up to 16 decoding rows carries its RMSNorms between the GEMVs (nvk_model_use_linked_norms: the
    producer leaves x * gain and partial sums of squares, the consumer multiplies its product by 1 / rms).  Against
    the norms as kernels the logits agree 10x tighter than the bar the whole model is held to against the reference
    (teacher forced, step by step), with and without the graph; the graph does not change a bit."""
    import nano_vllm_amd as nv

    rng = np.random.default_rng(12)
    prompts = [[1] + rng.integers(3, spec["vocab_size"], size=k).tolist() for k in (6, 3, 1, 4, 2, 5, 3)][:n_req]
    ids = [5, 9, 2, 11, 3, 8, 6][:n_req]
    steps = min(30, spec["max_seq_len"] - 10)
    runs, fed = {}, None
    for linked, graphs in ((False, False), (True, False), (True, True)):
        with nv.LlamaModel(checkpoints[spec["name"]], spec["block_size"], spec["num_blocks"], max_seqs=max(2, n_req)) as m:
            m.use_linked_norms(linked)
            m.use_graphs(graphs)
            for r in ids:
                m.add_request(r)
            nxt, logits = m.step_chunks(ids, prompts, True)
            got, toks = [logits.copy()], [nxt.tolist()]
            for i in range(steps):
                nxt, logits = m.step(ids, fed[i] if fed else nxt.tolist(), True)
                got.append(logits.copy())
                toks.append(nxt.tolist())
            if fed is None:
                fed = toks
            runs[(linked, graphs)] = np.stack(got)
    base = runs[(False, False)]
    assert np.isfinite(base).all()
    assert np.array_equal(runs[(True, True)], runs[(True, False)])
    err = np.abs(runs[(True, False)] - base)
    assert np.all(err <= 2e-4 + 1e-5 * np.abs(base)), float(err.max())
    assert (runs[(True, False)].argmax(-1) == base.argmax(-1)).all()
