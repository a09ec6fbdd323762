"""Property test of the decode path: ANY geometry the library accepts, ANY ragged batch behind ANY fragmented table, in
ANY launch mode, equals the CPU checker's restatement of Attention::paged_attention (include/core/attention.hpp:82-159)
within the path's bar, 1e-4 + 1e-5 |ref|.  hypothesis draws the cases (derandomized: the same cases in every run), the
library decides which of its kernels each one runs -- vector lanes, kv-head pairs, MFMA tiles, token-split workgroups,
in-kernel or separate combine -- and the modifiers force the alternatives."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import nano_vllm_amd as nv
import oracle_lib as ol
from nano_vllm_amd import workloads as wl
from test_gpu_attention import assert_close

pytestmark = pytest.mark.gpu

MODS = [0, nv.NO_PLAN, nv.MERGE_IN_KERNEL, nv.MERGE_SEPARATE, nv.VECTOR_ONLY, nv.NO_PAIRS, nv.FREE_RUNNING, nv.ROW_PAIRS,
        nv.NO_PLAN | nv.MERGE_IN_KERNEL, nv.VECTOR_ONLY | nv.FREE_RUNNING, nv.ROW_PAIRS | nv.MERGE_SEPARATE]


@st.composite
def cases(draw):
    head_dim = draw(st.sampled_from([4, 16, 32, 48, 64, 80, 96, 128, 192, 256]))
    n_kv = draw(st.integers(1, 12))
    kv_mul = draw(st.sampled_from([1, 1, 2, 3, 4, 4, 5, 7, 8, 12, 16]))
    if head_dim > 128:
        kv_mul = min(kv_mul, 6)  # (keeps the checker's share of the run small: 256-float rows x many heads)
    block_size = draw(st.sampled_from([1, 3, 4, 8, 16, 16, 32]))
    n_seqs = draw(st.integers(1, 9))
    ctx = draw(st.lists(st.one_of(st.integers(0, 40), st.integers(0, 700)), min_size=n_seqs, max_size=n_seqs))
    mode = draw(st.sampled_from(["auto", "single", "split"]))
    chunk = draw(st.sampled_from([0, 8, 16, 24, 48, 64, 100, 256, 1000])) if mode == "split" else 0
    mods = draw(st.sampled_from(MODS))
    scale = draw(st.sampled_from([0.5, 1.0, 3.0]))
    seed = draw(st.integers(1, 1000))
    return head_dim, n_kv, kv_mul, block_size, ctx, mode, chunk, mods, scale, seed


@settings(max_examples=400, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(cases())
def test_any_batch_in_any_mode_equals_the_checker(case):
    head_dim, n_kv, kv_mul, block_size, ctx, mode, chunk, mods, scale, seed = case
    g = wl.Geometry(n_kv * kv_mul, n_kv, head_dim, block_size)
    ctx = np.asarray(ctx, np.int32)
    work = wl.fragmented_workload("prop", ctx, g, perm_seed=seed)
    q = work.q() * np.float32(scale)
    seqs = list(range(work.num_seqs))
    k_host, v_host, tabs, c = wl.host_blocks_for(work, seqs, seed, seed + 1)
    want = ol.oracle_decode_batch(q, k_host, v_host, tabs, c, g.block_size, g.n_kv_heads, n_threads=4)
    flags = {"auto": nv.AUTO, "single": nv.SINGLE_PASS, "split": nv.SPLIT_K}[mode] | mods
    with nv.PagedKVPool(1, max(work.num_blocks, 1), g.block_size, g.n_kv_heads, g.head_dim) as pool:  # (all-empty batches)
        pool.fill_uniform(0, seed, seed + 1)
        got = nv.decode(pool, 0, q, work.block_tables, ctx, flags, chunk)
    assert_close(got, want, f"{case}")
    assert not got[ctx == 0].any()  # an empty sequence is a zero row (attention.hpp:98)
