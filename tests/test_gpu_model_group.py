"""The engine sequence-sharded over several devices (nvk_model_group_*, SURVEY.md 8e at engine level): a model per
device with its own host thread, a request placed on one member at admission (fewest promised tokens first), nothing
exchanged.  On a one-GPU box the members all live on device 0 -- two models standing in for two GPUs -- which exercises
everything but the second device: placement, per-member threads running their shares of a step at once, results in
list order.  Pass = every request's greedy stream (and logits) is exactly what a single model produces for it alone
(ref: sequences are independent, include/core/attention.hpp:82-159; the running list being split is
include/scheduler/scheduler.hpp:66-103)."""
import os

import numpy as np
import pytest

import nano_vllm_amd as nv
from golden_defs import TINY_MODELS, write_checkpoint

pytestmark = pytest.mark.gpu

SPEC = TINY_MODELS[1]


@pytest.fixture(scope="module")
def checkpoint(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("group") / "model.bin")
    write_checkpoint(path, SPEC)
    return path


def alone(path, prompt, n_new):
    """Stream and per-step logits of one request on a single model."""
    with nv.LlamaModel(path, 16, 64) as m:
        m.add_request(0)
        nxt, lg = m.step_chunks([0], [prompt], want_logits=True)
        toks, logits = [int(nxt[0])], [lg[0].copy()]
        for _ in range(n_new - 1):
            nxt, lg = m.step([0], [toks[-1]], want_logits=True)
            toks.append(int(nxt[0]))
            logits.append(lg[0].copy())
    return toks, np.stack(logits)


def test_two_members_on_one_device_reproduce_every_requests_own_stream(checkpoint):
    rng = np.random.default_rng(5)
    V = SPEC["vocab_size"]
    prompts = [rng.integers(3, V, size=n).tolist() for n in (9, 1, 23, 4, 17)]
    n_new = [12, 20, 6, 15, 9]
    want = [alone(checkpoint, p, n) for p, n in zip(prompts, n_new)]
    with nv.LlamaModelGroup(checkpoint, [0, 0], 16, 64, max_seqs_per_member=3) as g:
        assert g.size == 2
        placed = [g.add_request(i, len(prompts[i]) + n_new[i]) for i in range(5)]
        # promised tokens 21, 21, 29, 19, 26: member 0 takes the first, member 1 the second, then the lighter one each time
        assert placed == [0, 1, 0, 1, 1] and [g.member_of(i) for i in range(5)] == placed
        with pytest.raises(nv.NvkError):
            g.add_request(0, 5)  # already running
        got = {i: ([], []) for i in range(5)}
        live = list(range(5))
        nxt, lg = g.step_chunks(live, [prompts[i] for i in live], want_logits=True)  # all prompts in one step, both members busy
        last = {}
        for k, i in enumerate(live):
            got[i][0].append(int(nxt[k]))
            got[i][1].append(lg[k].copy())
            last[i] = int(nxt[k])
        while live:
            live = [i for i in live if len(got[i][0]) < n_new[i]]
            for i in list(last):
                if i not in live:
                    g.finish_request(i)
                    del last[i]
            if not live:
                break
            nxt, lg = g.step(live, [last[i] for i in live], want_logits=True)
            for k, i in enumerate(live):
                got[i][0].append(int(nxt[k]))
                got[i][1].append(lg[k].copy())
                last[i] = int(nxt[k])
        assert g.member_of(0) == -1
        for i in range(5):
            assert got[i][0] == want[i][0], i  # the stream, token for token
            # the same kernels on the same rows in another batch composition: batch-size-dependent kernel choices
            # (GEMV / skinny MFMA) reassociate the sums, so logits are held to the model-level bar, argmax exactly
            assert np.allclose(np.stack(got[i][1]), want[i][1], atol=2e-3, rtol=1e-4), i
        # a freed slot is taken again, by the lighter member
        assert g.add_request(7, 3) in (0, 1)
        g.finish_request(7)


def test_group_is_full_when_every_member_is(checkpoint):
    with nv.LlamaModelGroup(checkpoint, [0, 0], 16, 64, max_seqs_per_member=1) as g:
        assert g.add_request(0, 10) == 0 and g.add_request(1, 10) == 1
        with pytest.raises(nv.NvkError) as e:
            g.add_request(2, 10)
        assert "every member" in str(e.value)
        with pytest.raises(nv.NvkError):
            g.step_chunks([5], [[3]])  # unknown request
        with pytest.raises(nv.NvkError):
            g.finish_request(9)


def test_group_load_errors(tmp_path):
    with pytest.raises(nv.NvkError) as e:
        nv.LlamaModelGroup(str(tmp_path / "missing.bin"), [0, 0])
    assert "member" in str(e.value)
    with pytest.raises(nv.NvkError):
        nv.LlamaModelGroup(str(tmp_path / "missing.bin"), [])
