"""Paged chunked prefill: the multi-token companion of the decode path.

The reference's experimental chunked prefill (``experimental/chunked_prefill/model_chunked.hpp``) appends
a chunk of new tokens to the contiguous cache (``:79-94``) and runs causal attention of the chunk over it
(``chunked_attention``, ``:167-224``).  These are the same two steps through block tables: row (s, b) of
the result equals ``Attention::paged_attention`` with ``num_tokens = chunk_starts[s] + b + 1``."""
from __future__ import annotations

import numpy as np

from . import _capi
from .kv_pool import PagedKVPool, _devptr, _f32, _i32, _ptr

PREFILL_AUTO = _capi.PREFILL_AUTO
PREFILL_ROWWISE = _capi.PREFILL_ROWWISE
PREFILL_TILED = _capi.PREFILL_TILED


def _chunks(block_tables, chunk_starts, chunk_lens):
    tabs = _i32(block_tables)
    if tabs.ndim != 2:
        raise ValueError("block_tables must be [num_seqs, max_blocks]")
    starts = _i32(chunk_starts).reshape(-1)
    lens = _i32(chunk_lens).reshape(-1)
    if starts.size != tabs.shape[0] or lens.size != tabs.shape[0]:
        raise ValueError("chunk_starts / chunk_lens must be [num_seqs]")
    return tabs, starts, lens, int(lens.astype(np.int64).sum())


def rope_append_chunks(pool: PagedKVPool, layer: int, q, k_rows, v_rows, block_tables, chunk_starts, chunk_lens,
                       rope_theta: float = 10000.0) -> np.ndarray:
    """RoPE + append of every chunk token (host arrays); returns the rotated q [total_q, n_heads, head_dim]."""
    tabs, starts, lens, total = _chunks(block_tables, chunk_starts, chunk_lens)
    qh = np.array(q, dtype=np.float32, order="C", copy=True)
    if qh.ndim != 3 or qh.shape[0] != total or qh.shape[2] != pool.head_dim:
        raise ValueError("q must be [sum(chunk_lens), n_heads, head_dim]")
    k = _f32(k_rows).reshape(total, pool.row_len)
    v = _f32(v_rows).reshape(total, pool.row_len)
    _capi.check(_capi.lib.nvk_pool_rope_append_chunks(pool.handle, int(layer), tabs.shape[0], int(qh.shape[1]), _ptr(qh),
                                                      _ptr(k), _ptr(v), _ptr(tabs), tabs.shape[1], _ptr(starts),
                                                      _ptr(lens), float(rope_theta), _capi.PTR_HOST, None))
    return qh


def prefill(pool: PagedKVPool, layer: int, q, block_tables, chunk_starts, chunk_lens,
            flags: int = PREFILL_AUTO) -> np.ndarray:
    """Causal attention of each sequence's chunk over the paged pool, host arrays:
    q [sum(chunk_lens), n_heads, head_dim] -> out of the same shape."""
    tabs, starts, lens, total = _chunks(block_tables, chunk_starts, chunk_lens)
    qh = _f32(q)
    if qh.ndim != 3 or qh.shape[0] != total or qh.shape[2] != pool.head_dim:
        raise ValueError("q must be [sum(chunk_lens), n_heads, head_dim]")
    out = np.empty_like(qh)
    _capi.check(_capi.lib.nvk_prefill(pool.handle, int(layer), tabs.shape[0], int(qh.shape[1]), _ptr(qh), _ptr(tabs),
                                      tabs.shape[1], _ptr(starts), _ptr(lens), _ptr(out), int(flags)))
    return out


def prefill_device(pool: PagedKVPool, layer: int, n_heads: int, q_dev, block_tables_dev, num_seqs: int,
                   max_blocks: int, chunk_starts, chunk_lens, out_dev, flags: int = PREFILL_AUTO, stream=None) -> None:
    """Asynchronous form: q / tables / out on the device, the chunk list on the host."""
    starts = _i32(chunk_starts).reshape(-1)
    lens = _i32(chunk_lens).reshape(-1)
    _capi.check(_capi.lib.nvk_prefill_device(pool.handle, int(layer), int(num_seqs), int(n_heads), _devptr(q_dev),
                                             _devptr(block_tables_dev), int(max_blocks), _ptr(starts), _ptr(lens),
                                             _devptr(out_dev), int(flags), _devptr(stream)))
