"""nvk_run (tools/nvk_run.cpp), the reference's command line over the library.
CPU part: argument / path / JSON handling and the loud failure without a GPU.  GPU part: generated text equals
what the Python mirror (LlamaModel + Tokenizer over the same C ABI) produces token for token, single prompt and
--input-json with real batching."""
import json
import os
import subprocess

import numpy as np
import pytest

from golden_defs import TINY_MODELS, write_checkpoint, write_tokenizer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "nano_vllm_amd", "nvk_run")
SPEC = TINY_MODELS[1]  # g_tiny: vocab 512, the synthetic tokenizer's size

pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="nano_vllm_amd/nvk_run not built (make cli)")


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("model")
    write_checkpoint(str(d / "model.bin"), SPEC)
    write_tokenizer(str(d / "tokenizer.bin"), SPEC["vocab_size"])
    return str(d)


def run(*args):
    return subprocess.run([CLI, *args], capture_output=True, timeout=120)


def test_argument_and_path_errors(model_dir, tmp_path):
    assert run().returncode == 1
    assert run(model_dir).returncode == 1  # neither --prompt nor --input-json (main.cpp:47-51)
    r = run(model_dir, "-i", "x", "--input-json", "y.json")
    assert r.returncode == 1 and b"not both" in r.stderr
    r = run(str(tmp_path / "nowhere"), "-i", "x")
    assert r.returncode == 1 and b"does not exist" in r.stderr
    empty = tmp_path / "empty_dir"
    empty.mkdir()
    r = run(str(empty), "-i", "x")
    assert r.returncode == 1 and b"model.bin not found" in r.stderr  # path.hpp:31-34
    bad = tmp_path / "bad.json"
    bad.write_text('{"requests": [{"prompt": "", "max_tokens": 3}]}')
    r = run(model_dir, "--input-json", str(bad))
    assert r.returncode == 1 and b"has empty prompt" in r.stderr  # json_parser.hpp:318-320
    bad.write_text('{"requests": [{"prompt": "a" "max_tokens": 3}]}')
    assert run(model_dir, "--input-json", str(bad)).returncode == 1
    # the reference samples by default (-t 1.0, src/main.cpp:22); this build only decodes greedily and says so
    # instead of producing another kind of output silently
    r = run(model_dir, "-i", "hello", "-n", "3")
    assert r.returncode == 2 and b"no sampler" in r.stderr and b"--greedy" in r.stderr


def test_fails_loudly_without_a_gpu(model_dir):
    import nano_vllm_amd as nv
    if nv.device_count() > 0:
        pytest.skip("a GPU is visible")
    r = run(model_dir, "-i", "hello", "-n", "3", "-t", "0")
    assert r.returncode == 1 and b"no HIP device visible" in r.stderr  # no CPU path


def expected_text(nv, model_dir, prompt, max_tokens, stop_at_eos):
    tok = nv.Tokenizer(os.path.join(model_dir, "tokenizer.bin"), SPEC["vocab_size"])
    with nv.LlamaModel(os.path.join(model_dir, "model.bin"), 16, 256) as m:
        m.add_request(0)
        nxt, _ = m.step_chunks([0], [tok.encode(prompt)])
        out = [int(nxt[0])]
        while len(out) < max_tokens and m.request_len(0) < m.max_seq_len and not (stop_at_eos and out[-1] == 2):
            nxt, _ = m.step([0], [out[-1]])
            out.append(int(nxt[0]))
    return b"".join(tok.decode_bytes(t) for t in out)


@pytest.mark.gpu
def test_single_prompt_text(model_dir):
    import nano_vllm_amd as nv
    r = run(model_dir, "-i", "hello world", "-n", "24", "-t", "0")
    assert r.returncode == 0, r.stderr
    assert r.stdout == b"\nhello world" + expected_text(nv, model_dir, "hello world", 24, False) + b"\n"
    assert b"tokens/s" in r.stderr
    # a model.bin path finds tokenizer.bin next to it (path.hpp:43-52)
    r2 = run(os.path.join(model_dir, "model.bin"), "--prompt", "hello world", "--steps", "24", "-t", "0")
    assert r2.returncode == 0 and r2.stdout == r.stdout


@pytest.mark.gpu
def test_input_json_batched(model_dir, tmp_path):
    import nano_vllm_amd as nv
    reqs = [{"prompt": "once upon a time", "temperature": 0.0, "top_p": 0.9, "max_tokens": 12},
            {"prompt": "the theory of the thing", "temperature": 0.0, "max_tokens": 30},
            {"prompt": "hello", "max_tokens": 5, "temperature": 0.0},
            {"prompt": "in the inn, an ant and a hen", "temperature": 0.0, "max_tokens": 17, "extra": {"ignored": [1, 2]}}]
    path = tmp_path / "reqs.json"
    path.write_text(json.dumps({"name": "test", "requests": reqs}))
    want = {i: r["prompt"].encode() + expected_text(nv, model_dir, r["prompt"], r["max_tokens"], True)
            for i, r in enumerate(reqs)}
    for batch, sharding in (("1", ()), ("3", ()), ("1", ("--devices", "0,0")), ("2", ("--devices", "0,0,0"))):
        # --devices a,b,...: the engine sequence-sharded, a model per entry (here all on GPU 0, standing in for several
        # GPUs): each request still prints exactly the text it prints alone
        r = run(model_dir, "--input-json", str(path), "-b", batch, *sharding)
        assert r.returncode == 0, r.stderr
        if sharding:
            assert b"sequence-sharded over" in r.stderr
        parts = (b"\n" + r.stdout).split(b"\n[request ")[1:]  # every request prints "[request i] prompt+text\n"
        assert parts and parts[-1].endswith(b"\n")
        parts[-1] = parts[-1][:-1]
        got = {int(part.split(b"] ", 1)[0]): part.split(b"] ", 1)[1] for part in parts}
        assert got == want, batch
