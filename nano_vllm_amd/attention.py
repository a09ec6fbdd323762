"""Paged-attention decode: host-side mirror of the reference's attention interface.

``paged_attention`` keeps the argument list of the reference's
``Attention::paged_attention`` (``include/core/attention.hpp:82-92``) and routes
it through the C ABI drop-in ``nvk_paged_attention``; ``decode`` /
``decode_device`` are the batched layer-call the reference's scheduler has the
shape for (``ScheduledBatch.decode_requests``, ``include/scheduler/scheduler.hpp:24-43``)
but never executes.  Everything here runs on the GPU through
``libnvk_paged_attn.so``; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _capi
from .kv_pool import PagedKVPool, _devptr, _f32, _i32, _ptr

AUTO = _capi.DECODE_AUTO
SINGLE_PASS = _capi.DECODE_SINGLE_PASS
SPLIT_K = _capi.DECODE_SPLIT_K
# modifiers, OR-ed onto a mode (include/nvk_paged_attn.h)
MERGE_IN_KERNEL = _capi.DECODE_MERGE_IN_KERNEL
MERGE_SEPARATE = _capi.DECODE_MERGE_SEPARATE
UNIFORM = _capi.DECODE_UNIFORM
NO_PLAN = _capi.DECODE_NO_PLAN
VECTOR_ONLY = _capi.DECODE_VECTOR_ONLY
NO_PAIRS = _capi.DECODE_NO_PAIRS
FREE_RUNNING = _capi.DECODE_FREE_RUNNING
ROW_PAIRS = _capi.DECODE_ROW_PAIRS


def bind_pool(pool: Optional[PagedKVPool], layer: int = 0) -> None:
    """Make (pool, layer) the cache ``paged_attention`` reads on this thread (None unbinds).

    Plays the role of the layer slicing in ``LlamaModel::attention``
    (``include/core/model.hpp:400-403``)."""
    _capi.check(_capi.lib.nvk_bind_pool(pool.handle if pool is not None else None, int(layer)))


def set_host_cache_policy(append_only: bool) -> None:
    """How the host-slab forms of :func:`paged_attention` / :func:`standard_attention` stage K/V:
    re-stage every call (default, always correct) or keep a device mirror per slab under the promise that
    the engine only appends rows between calls, as ``LlamaModel::forward`` does (``model.hpp:190-207``)."""
    _capi.check(_capi.lib.nvk_set_host_cache_policy(_capi.HOST_CACHE_APPEND_ONLY if append_only
                                                    else _capi.HOST_CACHE_RESTAGE))


def host_cache_staged_bytes(reset: bool = False) -> int:
    return int(_capi.lib.nvk_host_cache_staged_bytes(1 if reset else 0))


def paged_attention(out: np.ndarray, q, key_cache, value_cache, block_table, att_scores, num_tokens: int,
                    block_size: int, head_dim: int, n_heads: int, n_kv_heads: int) -> None:
    """Drop-in for ``Attention::paged_attention`` (same eleven arguments, same meaning).

    ``out`` ([n_heads*head_dim] float32, C-contiguous) is written in place.
    ``key_cache`` / ``value_cache`` are one layer's host slabs
    ``[num_blocks, block_size, n_kv_heads, head_dim]``, or ``None`` when a device
    pool is bound with :func:`bind_pool`.  ``att_scores`` is accepted and ignored."""
    if not (isinstance(out, np.ndarray) and out.dtype == np.float32 and out.flags.c_contiguous):
        raise TypeError("out must be a C-contiguous float32 ndarray (it is written in place)")
    if out.size != n_heads * head_dim:
        raise ValueError("out must hold n_heads*head_dim floats")
    qh = _f32(q).reshape(-1)
    if qh.size != n_heads * head_dim:
        raise ValueError("q must hold n_heads*head_dim floats")
    kc = None if key_cache is None else _f32(key_cache)
    vc = None if value_cache is None else _f32(value_cache)
    tab = _i32(block_table).reshape(-1)
    need = (int(num_tokens) + int(block_size) - 1) // int(block_size)
    if tab.size < need:
        raise ValueError(f"block_table has {tab.size} entries, {need} needed for {num_tokens} tokens")
    if kc is not None and need > 0:
        blk = int(block_size) * int(n_kv_heads) * int(head_dim)
        top = int(tab[:need].max())
        if (top + 1) * blk > kc.size or (top + 1) * blk > vc.size:
            raise ValueError("block_table names a block beyond the host cache")
    _capi.check(_capi.lib.nvk_paged_attention(_ptr(out), _ptr(qh), _ptr(kc), _ptr(vc), _ptr(tab), None,
                                              int(num_tokens), int(block_size), int(head_dim), int(n_heads),
                                              int(n_kv_heads)))


def standard_attention(out: np.ndarray, q, key_cache, value_cache, att_scores, pos: int, head_dim: int,
                       n_heads: int, n_kv_heads: int, max_seq_len: int) -> None:
    """Drop-in for ``Attention::standard_attention`` (``include/core/attention.hpp:16-25``): the same
    maths over one layer's contiguous ``[max_seq_len, n_kv_heads, head_dim]`` host cache, tokens 0..pos."""
    if not (isinstance(out, np.ndarray) and out.dtype == np.float32 and out.flags.c_contiguous):
        raise TypeError("out must be a C-contiguous float32 ndarray (it is written in place)")
    if out.size != n_heads * head_dim:
        raise ValueError("out must hold n_heads*head_dim floats")
    qh = _f32(q).reshape(-1)
    if qh.size != n_heads * head_dim:
        raise ValueError("q must hold n_heads*head_dim floats")
    kc, vc = _f32(key_cache), _f32(value_cache)
    if min(kc.size, vc.size) < int(max_seq_len) * int(n_kv_heads) * int(head_dim):
        raise ValueError("caches must hold max_seq_len*n_kv_heads*head_dim floats")
    _capi.check(_capi.lib.nvk_standard_attention(_ptr(out), _ptr(qh), _ptr(kc), _ptr(vc), None, int(pos),
                                                 int(head_dim), int(n_heads), int(n_kv_heads), int(max_seq_len)))


def decode(pool: PagedKVPool, layer: int, q, block_tables, ctx_lens, flags: int = AUTO,
           chunk_tokens: int = 0) -> np.ndarray:
    """One layer-call from host arrays: q [num_seqs, n_heads, head_dim] -> out of the same shape."""
    qh = _f32(q)
    if qh.ndim != 3 or qh.shape[2] != pool.head_dim:
        raise ValueError("q must be [num_seqs, n_heads, head_dim]")
    n, n_heads, _ = qh.shape
    tabs = _i32(block_tables)
    if tabs.ndim != 2 or tabs.shape[0] != n:
        raise ValueError("block_tables must be [num_seqs, max_blocks]")
    ctx = _i32(ctx_lens).reshape(-1)
    if ctx.size != n:
        raise ValueError("ctx_lens must be [num_seqs]")
    out = np.empty_like(qh)
    _capi.check(_capi.lib.nvk_decode(pool.handle, int(layer), n, n_heads, _ptr(qh), _ptr(tabs), tabs.shape[1],
                                     _ptr(ctx), _ptr(out), int(flags), int(chunk_tokens)))
    return out


def decode_device(pool: PagedKVPool, layer: int, num_seqs: int, n_heads: int, q_dev, block_tables_dev,
                  max_blocks: int, ctx_lens_dev, max_ctx: int, out_dev, flags: int = AUTO, chunk_tokens: int = 0,
                  stream=None) -> None:
    """Asynchronous layer-call on device pointers (ints, or torch tensors on the pool's GPU)."""
    _capi.check(_capi.lib.nvk_decode_device(pool.handle, int(layer), int(num_seqs), int(n_heads), _devptr(q_dev),
                                            _devptr(block_tables_dev), int(max_blocks), _devptr(ctx_lens_dev),
                                            int(max_ctx), _devptr(out_dev), int(flags), int(chunk_tokens),
                                            _devptr(stream)))


def decode_device_noalloc(pool: PagedKVPool, layer: int, num_seqs: int, n_heads: int, q_dev, block_tables_dev,
                          max_blocks: int, ctx_lens_dev, max_ctx: int, out_dev, flags: int = AUTO,
                          chunk_tokens: int = 0, stream=None) -> None:
    """:func:`decode_device` for callers that must not allocate: work list, split-K partials and arrival counters
    live in the library's static device array (``noalloc_scratch_bytes()``); larger batches are refused."""
    _capi.check(_capi.lib.nvk_decode_device_noalloc(pool.handle, int(layer), int(num_seqs), int(n_heads),
                                                    _devptr(q_dev), _devptr(block_tables_dev), int(max_blocks),
                                                    _devptr(ctx_lens_dev), int(max_ctx), _devptr(out_dev), int(flags),
                                                    int(chunk_tokens), _devptr(stream)))


def noalloc_scratch_bytes() -> int:
    return int(_capi.lib.nvk_noalloc_scratch_bytes())


class DecodePlan:
    """A batch's decode work list, reusable across the layers of one model step (``nvk_plan_*``)."""

    def __init__(self, pool: PagedKVPool):
        h = C.c_void_p()
        _capi.check(_capi.lib.nvk_plan_create(pool.handle, C.byref(h)))
        self._h, self.pool = h, pool

    def close(self) -> None:
        if getattr(self, "_h", None):
            _capi.lib.nvk_plan_destroy(self._h)
            self._h = None

    __del__ = close

    def build(self, num_seqs: int, n_heads: int, ctx_lens_dev, max_ctx: int, flags: int = AUTO,
              chunk_tokens: int = 0, stream=None) -> "DecodePlan":
        _capi.check(_capi.lib.nvk_plan_build(self._h, int(num_seqs), int(n_heads), _devptr(ctx_lens_dev),
                                             int(max_ctx), int(flags), int(chunk_tokens), _devptr(stream)))
        return self

    def info(self) -> dict:
        out = (C.c_int * 4)()
        _capi.check(_capi.lib.nvk_plan_info(self._h, out))
        return {"chunk_tokens": out[0], "max_chunks": out[1], "workgroup_bound": out[2], "waves": out[3]}

    def run(self, layer: int, q_dev, block_tables_dev, max_blocks: int, out_dev, stream=None) -> None:
        _capi.check(_capi.lib.nvk_decode_planned(self.pool.handle, int(layer), self._h, _devptr(q_dev),
                                                 _devptr(block_tables_dev), int(max_blocks), _devptr(out_dev),
                                                 _devptr(stream)))


def decode_bench(pool: PagedKVPool, layer: int, num_seqs: int, n_heads: int, q_dev, block_tables_dev,
                 max_blocks: int, ctx_lens_dev, max_ctx: int, out_dev, flags: int = AUTO, chunk_tokens: int = 0,
                 warmup: int = 5, iters: int = 20, want_attn: bool = True,
                 replan_every: int = 1, rotate_layers: int = 1) -> Tuple[float, Optional[float]]:
    """HIP-event timing on the pool's stream: (ms per layer-call, ms per attention kernel).  With rotate_layers > 1
    call i reads layer ``layer + i % rotate_layers`` (cold rotation over several slabs)."""
    total = C.c_float(0.0)
    attn = C.c_float(0.0)
    _capi.check(_capi.lib.nvk_decode_bench_layers(pool.handle, int(layer), int(rotate_layers), int(num_seqs),
                                                  int(n_heads), _devptr(q_dev), _devptr(block_tables_dev),
                                                  int(max_blocks), _devptr(ctx_lens_dev), int(max_ctx),
                                                  _devptr(out_dev), int(flags), int(chunk_tokens), int(replan_every),
                                                  int(warmup), int(iters), C.byref(total),
                                                  C.byref(attn) if want_attn else None))
    return float(total.value), (float(attn.value) if want_attn else None)


def decode_bench_graph(pool: PagedKVPool, layer: int, num_seqs: int, n_heads: int, q_dev, block_tables_dev,
                       max_blocks: int, ctx_lens_dev, max_ctx: int, out_dev, flags: int = AUTO, chunk_tokens: int = 0,
                       layers_per_graph: int = 32, warmup: int = 2, iters: int = 5, rotate_layers: int = 1) -> float:
    """ms per layer-call when ``layers_per_graph`` layer-calls (one plan, as for the layers of a model step) are
    captured into ONE hipGraph and replayed ``iters`` times on the pool's stream."""
    total = C.c_float(0.0)
    _capi.check(_capi.lib.nvk_decode_bench_graph(pool.handle, int(layer), int(rotate_layers), int(num_seqs),
                                                 int(n_heads), _devptr(q_dev), _devptr(block_tables_dev),
                                                 int(max_blocks), _devptr(ctx_lens_dev), int(max_ctx),
                                                 _devptr(out_dev), int(flags), int(chunk_tokens),
                                                 int(layers_per_graph), int(warmup), int(iters), C.byref(total)))
    return float(total.value)


def decode_kernel_name(pool: PagedKVPool, num_seqs: int, n_heads: int, max_ctx: int, flags: int = AUTO,
                       chunk_tokens: int = 0) -> str:
    """The attention kernel instantiation a launch of this shape runs, as rocprofv3 names it."""
    buf = C.create_string_buffer(128)
    _capi.check(_capi.lib.nvk_decode_kernel_name(pool.handle, int(num_seqs), int(n_heads), int(max_ctx), int(flags),
                                                 int(chunk_tokens), buf, 128))
    return buf.value.decode()
