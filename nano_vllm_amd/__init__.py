"""nano_vllm_amd: MI355X-native paged-attention decode behind nano-vllm's attention interface.

The package is a thin host layer over ``libnvk_paged_attn.so`` (hand-written
gfx950 HIP kernels + a C ABI, ``include/nvk_paged_attn.h``).  Importing it
without the built library raises ImportError: there is no CPU fallback.
(The directory is ``nano_vllm_amd`` rather than ``nano-vllm_amd`` only because a
hyphen is not importable in Python.)
"""
from . import _capi
from ._capi import NvkError, device_count, version
from .attention import AUTO, FREE_RUNNING, ROW_PAIRS, MERGE_IN_KERNEL, MERGE_SEPARATE, NO_PAIRS, NO_PLAN, SINGLE_PASS, SPLIT_K, UNIFORM, VECTOR_ONLY, DecodePlan, bind_pool, decode, decode_bench, decode_bench_graph, decode_device, decode_kernel_name, decode_device_noalloc, host_cache_staged_bytes, noalloc_scratch_bytes, paged_attention, set_host_cache_policy, standard_attention
from .batch import DecodeBatch
from .block_manager import BlockManager
from .kv_pool import PagedKVPool
from .model import LlamaModel, LlamaModelGroup
from .tokenizer import Tokenizer
from .linear import batch_add_rms_norm_device, batch_matmul2_device, batch_swiglu_ffn_device, batch_matmul_device, batch_rms_norm_device, qkv_project_device
from .prefill import PREFILL_AUTO, PREFILL_ROWWISE, PREFILL_TILED, prefill, prefill_device, rope_append_chunks

__all__ = [
    "AUTO", "FREE_RUNNING", "ROW_PAIRS", "MERGE_IN_KERNEL", "MERGE_SEPARATE", "NO_PAIRS", "NO_PLAN", "UNIFORM", "VECTOR_ONLY", "PREFILL_AUTO", "PREFILL_ROWWISE", "PREFILL_TILED", "SINGLE_PASS", "SPLIT_K", "BlockManager", "DecodeBatch", "DecodePlan", "LlamaModel", "LlamaModelGroup", "NvkError", "PagedKVPool", "Tokenizer", "batch_add_rms_norm_device", "batch_matmul2_device", "batch_matmul_device", "batch_swiglu_ffn_device", "batch_rms_norm_device", "bind_pool", "decode", "decode_kernel_name", "decode_device_noalloc", "noalloc_scratch_bytes",
    "decode_bench", "decode_bench_graph", "decode_device", "device_count", "host_cache_staged_bytes", "paged_attention", "prefill", "prefill_device", "qkv_project_device", "rope_append_chunks", "set_host_cache_policy", "standard_attention", "version",
]
