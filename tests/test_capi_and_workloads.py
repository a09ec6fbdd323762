"""CPU-side checks: the C-ABI library loads and exports everything include/*.h declares,
fails loudly without a GPU (no CPU fallback), and the synthetic workloads are what BASELINE.md says."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import nano_vllm_amd as nv
import oracle_lib as ol
from nano_vllm_amd import _capi
from nano_vllm_amd import workloads as wl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nvk_paged_attn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nvk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 40
    for name in names:
        assert hasattr(_capi.lib, name), f"{name} declared in include/nvk_paged_attn.h but not exported"
    assert sorted(_capi.SIGNATURES) == names  # the ctypes table and the header list the same entry points


def test_version_and_error_string():
    assert "gfx950" in nv.version()
    assert isinstance(_capi.last_error(), str)


def test_library_links_the_hip_runtime_and_not_the_oracle():
    with open(_capi.LIB_PATH, "rb") as fh:
        blob = fh.read()
    assert b"libamdhip64" in blob
    assert b"oracle_paged_attention" not in blob and b"libnvk_oracle" not in blob and b"libnvk_ref" not in blob
    assert b"nvk_decode_kernel" in blob  # the gfx950 code object is embedded


def test_product_sources_never_touch_the_checker():
    bad = re.compile(r"oracle|libnvk_ref|_ref/")
    pkg = os.path.join(ROOT, "nano_vllm_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(base, f), errors="replace").read()
                hits = [ln for ln in text.splitlines() if bad.search(ln) and "oracle_fill_uniform" not in ln]
                assert not hits, f"{f}: {hits[:2]}"


@pytest.mark.skipif(nv.device_count() > 0, reason="only meaningful on a machine without a GPU")
def test_no_gpu_means_loud_failure_not_fallback():
    with pytest.raises(nv.NvkError) as e:
        nv.PagedKVPool(1, 8, 16, 2, 64)
    assert e.value.code == _capi.ERR_NO_DEVICE
    out = np.zeros((4, 64), np.float32)
    k = np.zeros((2, 16, 2, 64), np.float32)
    with pytest.raises(nv.NvkError):  # the drop-in cannot stage to a device that is not there
        nv.paged_attention(out, out.copy(), k, k, np.array([0], np.int32), None, 5, 16, 64, 4, 2)


def test_argument_validation_without_a_device():
    assert _capi.lib.nvk_pool_create(0, 8, 16, 2, 64, 0, C.byref(C.c_void_p())) == _capi.ERR_INVALID_ARG
    assert _capi.lib.nvk_pool_create(1, 8, 16, 2, 66, 0, C.byref(C.c_void_p())) == _capi.ERR_INVALID_ARG
    assert "head_dim" in _capi.last_error()
    assert _capi.lib.nvk_pool_create(1, 8, 16, 2, 64, 0, None) == _capi.ERR_INVALID_ARG
    assert _capi.lib.nvk_decode(None, 0, 1, 4, None, None, 1, None, None, 0, 0) == _capi.ERR_INVALID_ARG
    assert _capi.lib.nvk_bm_create(8, 0, C.byref(C.c_void_p())) == _capi.ERR_INVALID_ARG
    assert _capi.lib.nvk_bind_pool(None, 0) == _capi.OK


# ------------------------------------------------------------------ workloads
def test_generator_matches_the_checker_bit_for_bit():
    for seed, first, n in ((1, 0, 1000), (2, 123456789, 4097), (3, 2 ** 33 + 5, 64)):
        assert np.array_equal(wl.uniform(seed, first, n), ol.oracle_fill_uniform(seed, first, n))
    v = wl.uniform(1, 0, 200000)
    assert -1.0 <= v.min() < -0.99 and 0.99 < v.max() < 1.0 and abs(float(v.mean())) < 0.01
    assert int(ol.ORACLE.oracle_splitmix64(1234, 7)) == int(wl.splitmix64(1234, np.array([7], np.uint64))[0])


def test_config2_is_baseline_md():
    w = wl.config2()
    assert (w.num_seqs, w.max_ctx, w.num_blocks, w.max_blocks) == (64, 2048, 8192, 128)
    assert w.algorithmic_bytes() == 1_075_871_744
    assert w.block_tables[5].tolist() == list(range(5 * 128, 6 * 128))
    assert 2 * w.num_blocks * w.geom.block_floats * 4 == 2 * 512 * 2 ** 20  # 512 MiB K + 512 MiB V


def test_config3_tables_are_a_fragmented_partition():
    w = wl.config3()
    assert w.num_seqs == 256 and w.ctx_lens.min() >= 128 and w.ctx_lens.max() <= 8192
    need = (w.ctx_lens.astype(np.int64) + 15) // 16
    used = np.concatenate([w.blocks_of(s) for s in range(w.num_seqs)])
    assert used.size == need.sum() == np.unique(used).size  # no block owned twice
    assert used.min() >= 0 and used.max() < w.num_blocks
    assert w.num_blocks == need.sum() + int(need.sum() * 0.25)
    adjacent = sum(int((np.diff(w.blocks_of(s)) == 1).sum()) for s in range(w.num_seqs))
    assert adjacent < 0.001 * used.size  # consecutive logical blocks are (almost) never physically adjacent
    assert abs(w.algorithmic_bytes() / 1e9 - 8.7) < 0.6  # about 8.7 GB


def test_config4_and_5_shapes():
    w4 = wl.config4()
    assert (w4.num_seqs, w4.max_ctx, w4.num_blocks) == (16, 32768, 32768)
    assert abs(w4.algorithmic_bytes() / 1e9 - 4.295) < 0.01
    w5a = wl.config5a(4096)
    assert w5a.ctx_lens.min() >= 64 and w5a.ctx_lens.max() <= 192
    shard = wl.config5a(512, first_seq=1024)
    assert np.array_equal(shard.ctx_lens, w5a.ctx_lens[1024:1536])  # a rank's share is a slice of the whole
    w5b = wl.config5b(512)
    assert abs(8 * w5b.algorithmic_bytes() / 1e9 - 68.7) < 0.2


def test_host_blocks_for_reproduces_the_device_fill():
    w = wl.fragmented_workload("t", np.array([40, 3, 100], np.int32), wl.Geometry(8, 2, 64, 16))
    k, v, tabs, ctx = wl.host_blocks_for(w, [2, 0])
    assert ctx.tolist() == [100, 40]
    g = w.geom
    for row, s in enumerate([2, 0]):
        for j, pb in enumerate(w.blocks_of(s)):
            assert np.array_equal(k[tabs[row, j]].reshape(-1), wl.uniform_block(wl.SEED_K, int(pb), g.block_floats))
            assert np.array_equal(v[tabs[row, j]].reshape(-1), wl.uniform_block(wl.SEED_V, int(pb), g.block_floats))
