"""ctypes binding of libnvk_paged_attn.so (include/nvk_paged_attn.h).

The library is the product: if it is missing this module raises at import
time, loudly, and nothing in the package falls back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NVK_LIB_PATH selects another build of the same library (kernel-tuning experiments: tools/build_variants.sh)
LIB_PATH = os.environ.get("NVK_LIB_PATH") or os.path.join(_HERE, "libnvk_paged_attn.so")

OK = 0
ERR_INVALID_ARG = -1
ERR_OOM = -2
ERR_HIP = -3
ERR_BAD_BLOCK = -4
ERR_NO_DEVICE = -5
ERR_UNBOUND = -6

DECODE_AUTO = 0
DECODE_SINGLE_PASS = 1
DECODE_SPLIT_K = 2
DECODE_MERGE_IN_KERNEL = 0x100
DECODE_MERGE_SEPARATE = 0x200
DECODE_UNIFORM = 0x400
DECODE_NO_PLAN = 0x800
DECODE_VECTOR_ONLY = 0x1000
DECODE_NO_PAIRS = 0x2000
DECODE_FREE_RUNNING = 0x4000
DECODE_ROW_PAIRS = 0x8000

PREFILL_AUTO = 0
PREFILL_ROWWISE = 1
PREFILL_TILED = 2

HOST_CACHE_RESTAGE = 0
HOST_CACHE_APPEND_ONLY = 1

PTR_HOST = 0
PTR_DEVICE = 1


class NvkError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[nvk {code}] {message}")
        self.code = code
        self.message = message


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make lib` (hipcc --offload-arch=gfx950). "
        "nano_vllm_amd has no CPU fallback."
    )

lib = C.CDLL(LIB_PATH)

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/nvk_paged_attn.h declares
SIGNATURES = {
    "nvk_version": (C.c_char_p, []),
    "nvk_last_error": (C.c_char_p, []),
    "nvk_device_count": (C.c_int, []),
    "nvk_pool_create": (C.c_int, [C.c_int] * 6 + [C.POINTER(_vp)]),
    "nvk_pool_destroy": (C.c_int, [_vp]),
    "nvk_pool_geometry": (C.c_int, [_vp, _i]),
    "nvk_pool_layer_ptrs": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp)]),
    "nvk_pool_stream": (C.c_int, [_vp, C.POINTER(_vp)]),
    "nvk_pool_synchronize": (C.c_int, [_vp]),
    "nvk_pool_append": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "nvk_pool_append_batch": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp, C.c_int]),
    "nvk_pool_rope_append_batch": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp,
                                             C.c_float, C.c_int, _vp]),
    "nvk_pool_set_rope_theta": (C.c_int, [_vp, C.c_float]),
    "nvk_pool_rope_append_chunks": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp,
                                              C.c_float, C.c_int, _vp]),
    "nvk_prefill": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int]),
    "nvk_batch_rms_norm_device": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_int, _vp]),
    "nvk_batch_add_rms_norm_device": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_int, _vp]),
    "nvk_batch_matmul_device": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "nvk_batch_matmul2_device": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "nvk_batch_swiglu_ffn_device": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "nvk_qkv_project_device": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_float, C.c_int, _vp]),
    "nvk_prefill_device": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp]),
    "nvk_pool_upload_block": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "nvk_pool_download_block": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "nvk_pool_gather": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
    "nvk_pool_fill_uniform": (C.c_int, [_vp, C.c_int, C.c_uint64, C.c_uint64]),
    "nvk_pool_reserve_workspace": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "nvk_decode": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int]),
    "nvk_decode_device": (
        C.c_int,
        [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp],
    ),
    "nvk_decode_device_noalloc": (
        C.c_int,
        [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp],
    ),
    "nvk_noalloc_scratch_bytes": (C.c_size_t, []),
    "nvk_decode_bench": (
        C.c_int,
        [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
         C.c_int, _f, _f],
    ),
    "nvk_decode_bench_layers": (
        C.c_int,
        [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
         C.c_int, _f, _f],
    ),
    "nvk_decode_bench_graph": (
        C.c_int,
        [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
         C.c_int, _f],
    ),
    "nvk_decode_kernel_name": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "nvk_shard_head_range": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _i]),
    "nvk_shard_gather_bytes": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "nvk_shard_unique_id": (C.c_int, [_vp]),
    "nvk_shard_create": (C.c_int, [C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "nvk_shard_destroy": (C.c_int, [_vp]),
    "nvk_shard_info": (C.c_int, [_vp, _i]),
    "nvk_shard_all_gather_out": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "nvk_shard_decode_gather": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, C.c_int,
                                          C.c_int, C.c_int, _vp]),
    "nvk_shard_wait_gather": (C.c_int, [_vp, C.c_int, _vp]),
    "nvk_shard_set_gather_mode": (C.c_int, [_vp, C.c_int]),
    "nvk_shard_gather_mode": (C.c_int, [_vp]),
    "nvk_model_group_load": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _i, C.POINTER(_vp)]),
    "nvk_model_group_destroy": (C.c_int, [_vp]),
    "nvk_model_group_size": (C.c_int, [_vp]),
    "nvk_model_group_config": (C.c_int, [_vp, _i]),
    "nvk_model_group_member_device": (C.c_int, [_vp, C.c_int]),
    "nvk_model_group_member_of": (C.c_int, [_vp, C.c_int]),
    "nvk_model_group_add_request": (C.c_int, [_vp, C.c_int, C.c_int]),
    "nvk_model_group_finish_request": (C.c_int, [_vp, C.c_int]),
    "nvk_model_group_request_len": (C.c_int, [_vp, C.c_int]),
    "nvk_model_group_step_chunks": (C.c_int, [_vp, _i, _i, _i, C.c_int, _i, _f]),
    "nvk_model_group_use_graphs": (C.c_int, [_vp, C.c_int]),
    "nvk_plan_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "nvk_plan_destroy": (C.c_int, [_vp]),
    "nvk_plan_build": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "nvk_plan_info": (C.c_int, [_vp, _i]),
    "nvk_decode_planned": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp]),
    "nvk_paged_attention": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "nvk_standard_attention": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "nvk_set_host_cache_policy": (C.c_int, [C.c_int]),
    "nvk_host_cache_staged_bytes": (C.c_longlong, [C.c_int]),
    "nvk_bind_pool": (C.c_int, [_vp, C.c_int]),
    "nvk_batch_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "nvk_batch_destroy": (C.c_int, [_vp]),
    "nvk_batch_add_request": (C.c_int, [_vp, C.c_int]),
    "nvk_batch_finish_request": (C.c_int, [_vp, C.c_int]),
    "nvk_batch_num_requests": (C.c_int, [_vp]),
    "nvk_batch_request_len": (C.c_int, [_vp, C.c_int]),
    "nvk_batch_request_blocks": (C.c_int, [_vp, C.c_int, _i, C.c_int]),
    "nvk_batch_begin_step": (C.c_int, [_vp, _i, C.c_int, C.c_int, _vp]),
    "nvk_batch_begin_step_chunks": (C.c_int, [_vp, _i, _i, C.c_int, C.c_int, _vp]),
    "nvk_batch_step_rows": (C.c_int, [_vp]),
    "nvk_batch_abort_step": (C.c_int, [_vp]),
    "nvk_batch_graph_signature": (C.c_int, [_vp, C.POINTER(C.c_longlong)]),
    "nvk_pool_workspace_generation": (C.c_longlong, [_vp]),
    "nvk_tokenizer_load": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "nvk_tokenizer_destroy": (None, [_vp]),
    "nvk_tokenizer_vocab_size": (C.c_int, [_vp]),
    "nvk_tokenizer_max_token_length": (C.c_int, [_vp]),
    "nvk_tokenizer_decode": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int]),
    "nvk_tokenizer_encode": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, C.c_int, _i, C.c_int]),
    "nvk_model_load": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "nvk_model_destroy": (C.c_int, [_vp]),
    "nvk_model_config": (C.c_int, [_vp, _i]),
    "nvk_model_add_request": (C.c_int, [_vp, C.c_int]),
    "nvk_model_finish_request": (C.c_int, [_vp, C.c_int]),
    "nvk_model_request_len": (C.c_int, [_vp, C.c_int]),
    "nvk_model_free_blocks": (C.c_int, [_vp]),
    "nvk_model_use_graphs": (C.c_int, [_vp, C.c_int]),
    "nvk_model_use_linked_norms": (C.c_int, [_vp, C.c_int]),
    "nvk_model_step_chunks": (C.c_int, [_vp, _i, _i, _i, C.c_int, _i, _f]),
    "nvk_model_step": (C.c_int, [_vp, _i, _i, C.c_int, _i, _f]),
    "nvk_model_forward": (C.c_int, [_vp, C.c_int, C.c_int, _f]),
    "nvk_batch_layer_hidden": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_float, _vp, C.c_float, _vp]),
    "nvk_batch_layer": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_float, _vp]),
    "nvk_batch_end_step": (C.c_int, [_vp]),
    "nvk_batch_stats": (C.c_int, [_vp, C.POINTER(C.c_longlong)]),
    "nvk_bm_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_vp)]),
    "nvk_bm_destroy": (None, [_vp]),
    "nvk_bm_allocate_block": (C.c_int, [_vp]),
    "nvk_bm_free_block": (C.c_int, [_vp, C.c_int]),
    "nvk_bm_allocate_sequence": (C.c_int, [_vp, C.c_int, _i, C.c_int]),
    "nvk_bm_free_sequence": (C.c_int, [_vp, _i, C.c_int]),
    "nvk_bm_num_free_blocks": (C.c_int, [_vp]),
    "nvk_bm_num_blocks": (C.c_int, [_vp]),
    "nvk_bm_block_size": (C.c_int, [_vp]),
    "nvk_bm_is_free": (C.c_int, [_vp, C.c_int]),
    "nvk_bm_utilization": (C.c_float, [_vp]),
    "nvk_bm_allocate_for_request": (C.c_int, [_vp, C.c_int, C.c_int, _i, C.c_int]),
    "nvk_bm_allocate_block_for_request": (C.c_int, [_vp, C.c_int]),
    "nvk_bm_free_request": (None, [_vp, C.c_int]),
    "nvk_bm_release_block_for_request": (C.c_int, [_vp, C.c_int, C.c_int]),
    "nvk_bm_get_request_blocks": (C.c_int, [_vp, C.c_int, _i, C.c_int]),
    "nvk_bm_get_request_block_count": (C.c_int, [_vp, C.c_int]),
    "nvk_bm_num_active_requests": (C.c_int, [_vp]),
    "nvk_bm_allocate_block_safe": (C.c_int, [_vp]),
    "nvk_bm_free_block_safe": (None, [_vp, C.c_int]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library disagree
    _fn.restype = _res
    _fn.argtypes = _args


def last_error() -> str:
    return lib.nvk_last_error().decode("utf-8", "replace")


def check(code: int) -> int:
    if code < 0:
        raise NvkError(code, last_error())
    return code


def version() -> str:
    return lib.nvk_version().decode()


def device_count() -> int:
    return int(lib.nvk_device_count())
