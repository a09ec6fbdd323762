"""ctypes access to the CPU checker (oracle/) -- test infrastructure only.

``ORACLE`` is our C restatement (oracle/libnvk_oracle.so, always present after
``make oracle``).  ``REF`` is the reference's own headers behind oracle/ref_shim.cpp
(oracle/_ref/libnvk_ref.so); it exists only where it was built from
/root/reference and is ``None`` elsewhere.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "libnvk_oracle.so")
ORACLE_FAST_PATH = os.path.join(ROOT, "oracle", "libnvk_oracle_fast.so")
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libnvk_ref.so")
REF_FAST_PATH = os.path.join(ROOT, "oracle", "_ref", "libnvk_ref_fast.so")

_vp = C.c_void_p
_i = C.c_int


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _load_oracle(path):
    lib = C.CDLL(path)
    lib.oracle_splitmix64.restype = C.c_uint64
    lib.oracle_splitmix64.argtypes = [C.c_uint64, C.c_uint64]
    lib.oracle_fill_uniform.restype = None
    lib.oracle_fill_uniform.argtypes = [_vp, C.c_uint64, C.c_uint64, C.c_uint64]
    lib.oracle_paged_attention.restype = None
    lib.oracle_paged_attention.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]
    lib.oracle_standard_attention.restype = None
    lib.oracle_standard_attention.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]
    lib.oracle_chunked_attention.restype = None
    lib.oracle_chunked_attention.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]
    lib.oracle_paged_prefill_attention.restype = None
    lib.oracle_paged_prefill_attention.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]
    lib.oracle_decode_batch.restype = _i
    lib.oracle_decode_batch.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i]
    lib.oracle_apply_rope.restype = None
    lib.oracle_apply_rope.argtypes = [_vp, _vp, _i, _i, _i, _i, C.c_float]
    lib.oracle_batch_rms_norm.restype = None
    lib.oracle_batch_rms_norm.argtypes = [_vp, _vp, _vp, _i, _i]
    lib.oracle_batch_matmul.restype = None
    lib.oracle_batch_matmul.argtypes = [_vp, _vp, _vp, _i, _i, _i]
    lib.oracle_kv_append.restype = None
    lib.oracle_kv_append.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]
    lib.oracle_bm_create.restype = _vp
    lib.oracle_bm_create.argtypes = [_i, _i]
    lib.oracle_bm_destroy.restype = None
    lib.oracle_bm_destroy.argtypes = [_vp]
    for name, args in {
        "oracle_bm_allocate_block": [_vp],
        "oracle_bm_free_block": [_vp, _i],
        "oracle_bm_allocate_sequence": [_vp, _i, _vp, _i],
        "oracle_bm_free_sequence": [_vp, _vp, _i],
        "oracle_bm_num_free": [_vp],
        "oracle_bm_num_blocks": [_vp],
        "oracle_bm_block_size": [_vp],
        "oracle_bm_is_free": [_vp, _i],
        "oracle_bm_allocate_for_request": [_vp, _i, _i, _vp, _i],
        "oracle_bm_allocate_block_for_request": [_vp, _i],
        "oracle_bm_get_request_blocks": [_vp, _i, _vp, _i],
        "oracle_bm_get_request_block_count": [_vp, _i],
        "oracle_bm_num_active_requests": [_vp],
    }.items():
        getattr(lib, name).restype = _i
        getattr(lib, name).argtypes = args
    lib.oracle_bm_free_request.restype = None
    lib.oracle_bm_free_request.argtypes = [_vp, _i]
    lib.oracle_bm_utilization.restype = C.c_float
    lib.oracle_bm_utilization.argtypes = [_vp]
    lib.oracle_model_create.restype = _vp
    lib.oracle_model_create.argtypes = [C.c_char_p, _i, _i]
    lib.oracle_model_destroy.restype = None
    lib.oracle_model_destroy.argtypes = [_vp]
    lib.oracle_model_forward.restype = _i
    lib.oracle_model_forward.argtypes = [_vp, _i, _i, _vp]
    lib.oracle_model_config.restype = None
    lib.oracle_model_config.argtypes = [_vp, _vp]
    lib.oracle_model_block_table.restype = _i
    lib.oracle_model_block_table.argtypes = [_vp, _i, _vp, _i]
    lib.oracle_model_last_attn_out.restype = None
    lib.oracle_model_last_attn_out.argtypes = [_vp, _vp, _i]
    lib.oracle_greedy_sample.restype = _i
    lib.oracle_greedy_sample.argtypes = [_vp, _i]
    return lib


def _load_ref(path):
    lib = C.CDLL(path)
    lib.ref_paged_attention.restype = None
    lib.ref_paged_attention.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]
    lib.ref_standard_attention.restype = None
    lib.ref_standard_attention.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]
    lib.ref_paged_attention_batch.restype = None
    lib.ref_paged_attention_batch.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _i]
    if hasattr(lib, "ref_chunked_attention"):
        lib.ref_chunked_attention.restype = None
        lib.ref_chunked_attention.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]
    if hasattr(lib, "ref_apply_rope"):
        lib.ref_apply_rope.restype = None
        lib.ref_apply_rope.argtypes = [_vp, _vp, _i, _i, _i, _i, C.c_float]
    for name in ("ref_rms_norm_rows", "ref_batch_rms_norm"):
        if hasattr(lib, name):
            getattr(lib, name).restype = None
            getattr(lib, name).argtypes = [_vp, _vp, _vp, _i, _i]
    for name in ("ref_matmul_rows", "ref_batch_matmul"):
        if hasattr(lib, name):
            getattr(lib, name).restype = None
            getattr(lib, name).argtypes = [_vp, _vp, _vp, _i, _i, _i]
    if hasattr(lib, "ref_tokenizer_create"):
        lib.ref_tokenizer_create.restype = _vp
        lib.ref_tokenizer_create.argtypes = [C.c_char_p, _i]
        lib.ref_tokenizer_destroy.restype = None
        lib.ref_tokenizer_destroy.argtypes = [_vp]
        lib.ref_tokenizer_encode.restype = _i
        lib.ref_tokenizer_encode.argtypes = [_vp, C.c_char_p, _i, _i, _i, _vp, _i]
        lib.ref_tokenizer_decode.restype = _i
        lib.ref_tokenizer_decode.argtypes = [_vp, _i, C.c_char_p, _i]
    lib.ref_bm_create.restype = _vp
    lib.ref_bm_create.argtypes = [_i, _i]
    lib.ref_bm_destroy.restype = None
    lib.ref_bm_destroy.argtypes = [_vp]
    for name, args in {
        "ref_bm_allocate_block": [_vp],
        "ref_bm_free_block": [_vp, _i],
        "ref_bm_allocate_sequence": [_vp, _i, _vp, _i],
        "ref_bm_free_sequence": [_vp, _vp, _i],
        "ref_bm_num_free": [_vp],
        "ref_bm_num_blocks": [_vp],
        "ref_bm_block_size": [_vp],
        "ref_bm_is_free": [_vp, _i],
        "ref_bm_allocate_for_request": [_vp, _i, _i, _vp, _i],
        "ref_bm_allocate_block_for_request": [_vp, _i],
        "ref_bm_get_request_blocks": [_vp, _i, _vp, _i],
        "ref_bm_get_request_block_count": [_vp, _i],
        "ref_bm_num_active_requests": [_vp],
        "ref_bm_allocate_block_safe": [_vp],
    }.items():
        getattr(lib, name).restype = _i
        getattr(lib, name).argtypes = args
    lib.ref_bm_free_request.restype = None
    lib.ref_bm_free_request.argtypes = [_vp, _i]
    lib.ref_bm_free_block_safe.restype = None
    lib.ref_bm_free_block_safe.argtypes = [_vp, _i]
    lib.ref_bm_utilization.restype = C.c_float
    lib.ref_bm_utilization.argtypes = [_vp]
    lib.ref_model_create.restype = _vp
    lib.ref_model_create.argtypes = [C.c_char_p, _i, _i, _i]
    lib.ref_model_destroy.restype = None
    lib.ref_model_destroy.argtypes = [_vp]
    lib.ref_model_config.restype = None
    lib.ref_model_config.argtypes = [_vp, _vp]
    lib.ref_model_forward.restype = _i
    lib.ref_model_forward.argtypes = [_vp, _i, _i, _vp]
    lib.ref_model_block_table.restype = _i
    lib.ref_model_block_table.argtypes = [_vp, _i, _vp, _i]
    lib.ref_model_last_attn_out.restype = None
    lib.ref_model_last_attn_out.argtypes = [_vp, _vp, _i]
    lib.ref_greedy_sample.restype = _i
    lib.ref_greedy_sample.argtypes = [_vp, _i]
    return lib


ORACLE = _load_oracle(ORACLE_PATH) if os.path.exists(ORACLE_PATH) else None
REF = _load_ref(REF_PATH) if os.path.exists(REF_PATH) else None


def load_fast_oracle():
    return _load_oracle(ORACLE_FAST_PATH) if os.path.exists(ORACLE_FAST_PATH) else None


def load_fast_ref():
    return _load_ref(REF_FAST_PATH) if os.path.exists(REF_FAST_PATH) else None


# ---- numpy-level helpers ------------------------------------------------------
def paged_attention(lib_fn, q, k_cache, v_cache, table, num_tokens, block_size, head_dim, n_heads, n_kv_heads):
    """Call {oracle,ref}_paged_attention on numpy arrays; returns out [n_heads, head_dim]."""
    q = np.ascontiguousarray(q, np.float32)
    k_cache = np.ascontiguousarray(k_cache, np.float32)
    v_cache = np.ascontiguousarray(v_cache, np.float32)
    table = np.ascontiguousarray(table, np.int32)
    out = np.full((n_heads, head_dim), np.nan, np.float32)
    scratch = np.empty(max(n_heads * max(num_tokens, 1), 1), np.float32)
    lib_fn(_p(out), _p(q), _p(k_cache), _p(v_cache), _p(table), _p(scratch), int(num_tokens), int(block_size),
           int(head_dim), int(n_heads), int(n_kv_heads))
    return out


def oracle_paged_attention(q, k_cache, v_cache, table, num_tokens, block_size, head_dim, n_heads, n_kv_heads):
    return paged_attention(ORACLE.oracle_paged_attention, q, k_cache, v_cache, table, num_tokens, block_size,
                           head_dim, n_heads, n_kv_heads)


def ref_paged_attention(q, k_cache, v_cache, table, num_tokens, block_size, head_dim, n_heads, n_kv_heads):
    return paged_attention(REF.ref_paged_attention, q, k_cache, v_cache, table, num_tokens, block_size, head_dim,
                           n_heads, n_kv_heads)


def standard_attention(lib_fn, q, k_cache, v_cache, pos, head_dim, n_heads, n_kv_heads, max_seq_len):
    q = np.ascontiguousarray(q, np.float32)
    k_cache = np.ascontiguousarray(k_cache, np.float32)
    v_cache = np.ascontiguousarray(v_cache, np.float32)
    out = np.full((n_heads, head_dim), np.nan, np.float32)
    scratch = np.empty(n_heads * max_seq_len, np.float32)
    lib_fn(_p(out), _p(q), _p(k_cache), _p(v_cache), _p(scratch), int(pos), int(head_dim), int(n_heads),
           int(n_kv_heads), int(max_seq_len))
    return out


def oracle_decode_batch(q, k_cache, v_cache, tables, ctx_lens, block_size, n_kv_heads, n_threads=1, lib=None):
    """Batched checker: q [n, n_heads, head_dim] -> out, one oracle call per sequence."""
    lib = lib or ORACLE
    q = np.ascontiguousarray(q, np.float32)
    n, n_heads, head_dim = q.shape
    tables = np.ascontiguousarray(tables, np.int32)
    ctx_lens = np.ascontiguousarray(ctx_lens, np.int32)
    k_cache = np.ascontiguousarray(k_cache, np.float32)
    v_cache = np.ascontiguousarray(v_cache, np.float32)
    out = np.full_like(q, np.nan)
    rc = lib.oracle_decode_batch(_p(out), _p(q), _p(k_cache), _p(v_cache), _p(tables), int(tables.shape[1]),
                                 _p(ctx_lens), n, int(block_size), head_dim, n_heads, int(n_kv_heads),
                                 int(n_threads))
    assert rc == 0
    return out


def oracle_fill_uniform(seed, first, count):
    out = np.empty(count, np.float32)
    ORACLE.oracle_fill_uniform(_p(out), int(seed), int(first), int(count))
    return out


def apply_rope(lib_fn, q, k, pos, head_dim, n_heads, n_kv_heads, theta=10000.0):
    """{oracle,ref}_apply_rope on copies: returns (q_rot [n_heads, hd], k_rot [n_kv_heads, hd])."""
    q = np.array(q, np.float32, copy=True).reshape(n_heads, head_dim)
    k = np.array(k, np.float32, copy=True).reshape(n_kv_heads, head_dim)
    lib_fn(_p(q), _p(k), int(pos), int(head_dim), int(n_heads), int(n_kv_heads), C.c_float(theta))
    return q, k


def oracle_chunked_attention(q, k_lin, v_lin, start_pos, n_kv_heads):
    """q [C, H, hd] at positions start_pos + b over a contiguous cache [max_seq_len, KV, hd] -> out [C, H, hd]."""
    q = np.ascontiguousarray(q, np.float32)
    C_, H, hd = q.shape
    k_lin = np.ascontiguousarray(k_lin, np.float32)
    v_lin = np.ascontiguousarray(v_lin, np.float32)
    max_seq = k_lin.shape[0]
    out = np.full_like(q, np.nan)
    att = np.empty(C_ * H * max_seq, np.float32)
    ORACLE.oracle_chunked_attention(_p(out), _p(q), _p(k_lin), _p(v_lin), _p(att), C_, int(start_pos), hd, H,
                                    int(n_kv_heads), max_seq)
    return out


def ref_chunked_attention(q, k_lin, v_lin, start_pos, n_kv_heads):
    q = np.ascontiguousarray(q, np.float32)
    C_, H, hd = q.shape
    k_lin = np.ascontiguousarray(k_lin, np.float32)
    v_lin = np.ascontiguousarray(v_lin, np.float32)
    out = np.full_like(q, np.nan)
    REF.ref_chunked_attention(_p(out), _p(q), _p(k_lin), _p(v_lin), C_, int(start_pos), hd, H, int(n_kv_heads),
                              k_lin.shape[0])
    return out


def oracle_paged_prefill(q, k_cache, v_cache, table, start_pos, block_size, n_kv_heads):
    """q [C, H, hd]: query b is paged_attention with num_tokens = start_pos + b + 1 over `table`."""
    q = np.ascontiguousarray(q, np.float32)
    C_, H, hd = q.shape
    k_cache = np.ascontiguousarray(k_cache, np.float32)
    v_cache = np.ascontiguousarray(v_cache, np.float32)
    table = np.ascontiguousarray(table, np.int32)
    out = np.full_like(q, np.nan)
    att = np.empty(H * (start_pos + C_ + 1), np.float32)
    ORACLE.oracle_paged_prefill_attention(_p(out), _p(q), _p(k_cache), _p(v_cache), _p(table), _p(att), C_,
                                          int(start_pos), int(block_size), hd, H, int(n_kv_heads))
    return out


def batch_rms_norm(fn, x, weight):
    """fn = ORACLE.oracle_batch_rms_norm / REF.ref_batch_rms_norm / REF.ref_rms_norm_rows; x [batch, dim]."""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(weight, np.float32)
    out = np.empty_like(x)
    fn(_p(out), _p(x), _p(w), x.shape[0], x.shape[1])
    return out


def batch_matmul(fn, x, weight):
    """fn = ORACLE.oracle_batch_matmul / REF.ref_batch_matmul / REF.ref_matmul_rows; x [batch, in], weight [out, in]."""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(weight, np.float32)
    out = np.empty((x.shape[0], w.shape[0]), np.float32)
    fn(_p(out), _p(x), _p(w), x.shape[0], x.shape[1], w.shape[0])
    return out


def ref_tokenizer_encode(tok, text: bytes, bos=True, eos=False):
    """Token ids of the reference's Tokenizer::encode (tok = REF.ref_tokenizer_create(...))."""
    cap = len(text) + 8
    ids = (C.c_int * cap)()
    n = REF.ref_tokenizer_encode(tok, text, len(text), int(bos), int(eos), ids, cap)
    return list(ids[:n])


def ref_tokenizer_decode(tok, token: int) -> bytes:
    buf = C.create_string_buffer(256)
    n = REF.ref_tokenizer_decode(tok, int(token), buf, 256)
    return buf.raw[:n]
