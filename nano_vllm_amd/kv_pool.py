"""PagedKVPool: the device-resident paged KV cache behind the C ABI.

Mirrors what the reference keeps on ``LlamaModel`` -- ``config.{block_size,
num_blocks}`` (``include/core/model.hpp:33-36``), ``state.paged_key_cache`` /
``paged_value_cache`` (``:94-97``, layout ``[n_layers, num_blocks, block_size,
n_kv_heads, head_dim]`` fp32) and the append at ``:190-207`` -- with the slabs in
HBM instead of ``std::vector<float>``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _capi


def _f32(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a: Optional[np.ndarray]) -> C.c_void_p:
    return C.c_void_p(None) if a is None else C.c_void_p(a.ctypes.data)


def _devptr(x) -> C.c_void_p:
    """int address, ctypes pointer, or anything with .data_ptr() (a torch tensor on the GPU)."""
    if x is None:
        return C.c_void_p(None)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(int(x.data_ptr()))
    if isinstance(x, C.c_void_p):
        return x
    return C.c_void_p(int(x))


class PagedKVPool:
    def __init__(self, n_layers: int, num_blocks: int, block_size: int, n_kv_heads: int, head_dim: int,
                 device: int = 0):
        handle = C.c_void_p()
        _capi.check(_capi.lib.nvk_pool_create(int(n_layers), int(num_blocks), int(block_size), int(n_kv_heads),
                                              int(head_dim), int(device), C.byref(handle)))
        self._h = handle
        self.n_layers, self.num_blocks, self.block_size = int(n_layers), int(num_blocks), int(block_size)
        self.n_kv_heads, self.head_dim, self.device = int(n_kv_heads), int(head_dim), int(device)

    # -- lifetime -----------------------------------------------------------
    def close(self) -> None:
        h = getattr(self, "_h", None)
        if h:
            _capi.lib.nvk_pool_destroy(h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self) -> C.c_void_p:
        if not self._h:
            raise RuntimeError("pool is closed")
        return self._h

    @property
    def row_len(self) -> int:
        return self.n_kv_heads * self.head_dim

    @property
    def block_floats(self) -> int:
        return self.block_size * self.row_len

    @property
    def layer_floats(self) -> int:
        return self.num_blocks * self.block_floats

    def layer_ptrs(self, layer: int) -> Tuple[int, int]:
        k, v = C.c_void_p(), C.c_void_p()
        _capi.check(_capi.lib.nvk_pool_layer_ptrs(self.handle, int(layer), C.byref(k), C.byref(v)))
        return int(k.value), int(v.value)

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        _capi.check(_capi.lib.nvk_pool_stream(self.handle, C.byref(s)))
        return int(s.value or 0)

    def synchronize(self) -> None:
        _capi.check(_capi.lib.nvk_pool_synchronize(self.handle))

    # -- filling ------------------------------------------------------------
    def append(self, layer: int, physical_block: int, block_offset: int, k_row, v_row) -> None:
        k = _f32(k_row).reshape(-1)
        v = _f32(v_row).reshape(-1)
        if k.size != self.row_len or v.size != self.row_len:
            raise ValueError(f"rows must hold n_kv_heads*head_dim={self.row_len} floats")
        _capi.check(_capi.lib.nvk_pool_append(self.handle, int(layer), int(physical_block), int(block_offset),
                                              _ptr(k), _ptr(v)))

    def append_batch(self, layer: int, k_rows, v_rows, block_tables, positions) -> None:
        tabs = _i32(block_tables)
        if tabs.ndim != 2:
            raise ValueError("block_tables must be [num_seqs, max_blocks]")
        n, max_blocks = tabs.shape
        k = _f32(k_rows).reshape(n, self.row_len)
        v = _f32(v_rows).reshape(n, self.row_len)
        pos = _i32(positions).reshape(n)
        _capi.check(_capi.lib.nvk_pool_append_batch(self.handle, int(layer), n, _ptr(k), _ptr(v), _ptr(tabs),
                                                    max_blocks, _ptr(pos), _capi.PTR_HOST))

    def append_batch_device(self, layer: int, num_seqs: int, k_rows_dev, v_rows_dev, block_tables_dev,
                            max_blocks: int, positions_dev) -> None:
        _capi.check(_capi.lib.nvk_pool_append_batch(self.handle, int(layer), int(num_seqs), _devptr(k_rows_dev),
                                                    _devptr(v_rows_dev), _devptr(block_tables_dev), int(max_blocks),
                                                    _devptr(positions_dev), _capi.PTR_DEVICE))

    def rope_append_batch(self, layer: int, q, k_rows, v_rows, block_tables, positions,
                          rope_theta: float = 10000.0) -> np.ndarray:
        """RoPE + append for one new token per sequence, host arrays (``Ops::apply_rope`` at
        ``model.hpp:169-175`` then the append at ``:190-207``).  Returns the rotated q
        [num_seqs, n_heads, head_dim]; the rotated K rows and the V rows are in the pool."""
        tabs = _i32(block_tables)
        if tabs.ndim != 2:
            raise ValueError("block_tables must be [num_seqs, max_blocks]")
        n, max_blocks = tabs.shape
        qh = np.array(q, dtype=np.float32, order="C", copy=True)
        if qh.ndim != 3 or qh.shape[0] != n or qh.shape[2] != self.head_dim:
            raise ValueError("q must be [num_seqs, n_heads, head_dim]")
        k = _f32(k_rows).reshape(n, self.row_len)
        v = _f32(v_rows).reshape(n, self.row_len)
        pos = _i32(positions).reshape(n)
        _capi.check(_capi.lib.nvk_pool_rope_append_batch(self.handle, int(layer), n, int(qh.shape[1]), _ptr(qh),
                                                         _ptr(k), _ptr(v), _ptr(tabs), max_blocks, _ptr(pos),
                                                         float(rope_theta), _capi.PTR_HOST, None))
        return qh

    def set_rope_theta(self, rope_theta: float) -> None:
        """Prebuild the RoPE frequency table for another theta (needed before capturing a stream)."""
        _capi.check(_capi.lib.nvk_pool_set_rope_theta(self.handle, float(rope_theta)))

    def rope_append_batch_device(self, layer: int, num_seqs: int, n_heads: int, q_dev, k_rows_dev, v_rows_dev,
                                 block_tables_dev, max_blocks: int, positions_dev, rope_theta: float = 10000.0,
                                 stream=None) -> None:
        _capi.check(_capi.lib.nvk_pool_rope_append_batch(self.handle, int(layer), int(num_seqs), int(n_heads),
                                                         _devptr(q_dev), _devptr(k_rows_dev), _devptr(v_rows_dev),
                                                         _devptr(block_tables_dev), int(max_blocks),
                                                         _devptr(positions_dev), float(rope_theta), _capi.PTR_DEVICE,
                                                         _devptr(stream)))

    def upload_block(self, layer: int, physical_block: int, k_block, v_block) -> None:
        shape = (self.block_size, self.n_kv_heads, self.head_dim)
        k = _f32(k_block).reshape(shape)
        v = _f32(v_block).reshape(shape)
        _capi.check(_capi.lib.nvk_pool_upload_block(self.handle, int(layer), int(physical_block), _ptr(k), _ptr(v)))

    def download_block(self, layer: int, physical_block: int) -> Tuple[np.ndarray, np.ndarray]:
        shape = (self.block_size, self.n_kv_heads, self.head_dim)
        k = np.empty(shape, np.float32)
        v = np.empty(shape, np.float32)
        _capi.check(_capi.lib.nvk_pool_download_block(self.handle, int(layer), int(physical_block), _ptr(k), _ptr(v)))
        return k, v

    def gather(self, layer: int, block_table, num_tokens: int, kv_head: int) -> Tuple[np.ndarray, np.ndarray]:
        tab = _i32(block_table).reshape(-1)
        k = np.empty((num_tokens, self.head_dim), np.float32)
        v = np.empty((num_tokens, self.head_dim), np.float32)
        _capi.check(_capi.lib.nvk_pool_gather(self.handle, int(layer), _ptr(tab), int(num_tokens), int(kv_head),
                                              _ptr(k), _ptr(v)))
        return k, v

    def fill_uniform(self, layer: int, seed_k: int, seed_v: int) -> None:
        _capi.check(_capi.lib.nvk_pool_fill_uniform(self.handle, int(layer), int(seed_k), int(seed_v)))

    def reserve_workspace(self, num_seqs: int, n_heads: int, max_ctx: int, chunk_tokens: int = 0) -> None:
        _capi.check(_capi.lib.nvk_pool_reserve_workspace(self.handle, int(num_seqs), int(n_heads), int(max_ctx),
                                                         int(chunk_tokens)))
