"""DecodeBatch: per-request block tables on the device and one append + attention call per layer for all
DECODING requests -- what the reference's scheduler has the types for (``ScheduledBatch::decode_requests``,
``include/scheduler/scheduler.hpp:24-43``; ``Request::block_ids``, ``include/scheduler/request.hpp:74``) and
never executes (``include/scheduler/batched_runner.hpp:84-85``).  Thin mirror of ``nvk_batch_*``."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

from . import _capi
from .block_manager import BlockManager
from .kv_pool import PagedKVPool, _devptr


class DecodeBatch:
    def __init__(self, pool: PagedKVPool, block_manager: BlockManager, max_seqs: int, max_blocks_per_seq: int):
        h = C.c_void_p()
        _capi.check(_capi.lib.nvk_batch_create(pool.handle, block_manager._h, int(max_seqs), int(max_blocks_per_seq),
                                               C.byref(h)))
        self._h, self.pool, self.block_manager = h, pool, block_manager  # keeps both alive

    def close(self) -> None:
        if getattr(self, "_h", None):
            _capi.lib.nvk_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def add_request(self, request_id: int) -> None:
        _capi.check(_capi.lib.nvk_batch_add_request(self._h, int(request_id)))

    def finish_request(self, request_id: int) -> None:
        _capi.check(_capi.lib.nvk_batch_finish_request(self._h, int(request_id)))

    def num_requests(self) -> int:
        return int(_capi.lib.nvk_batch_num_requests(self._h))

    def request_len(self, request_id: int) -> int:
        return _capi.check(_capi.lib.nvk_batch_request_len(self._h, int(request_id)))

    def request_blocks(self, request_id: int) -> List[int]:
        n = _capi.check(_capi.lib.nvk_batch_request_blocks(self._h, int(request_id), None, 0))
        buf = (C.c_int * max(n, 1))()
        _capi.check(_capi.lib.nvk_batch_request_blocks(self._h, int(request_id), buf, n))
        return list(buf[:n])

    def begin_step(self, request_ids: Sequence[int], n_heads: int, stream=None) -> None:
        ids = (C.c_int * max(len(request_ids), 1))(*[int(r) for r in request_ids])
        _capi.check(_capi.lib.nvk_batch_begin_step(self._h, ids, len(request_ids), int(n_heads), _devptr(stream)))

    def begin_step_chunks(self, request_ids: Sequence[int], chunk_lens: Sequence[int], n_heads: int,
                          stream=None) -> int:
        """A step where request i takes ``chunk_lens[i]`` tokens (decode requests, one token each, listed first;
        prefill chunks behind them).  Returns the number of q/k/v/out rows the step's layer calls expect."""
        n = len(request_ids)
        ids = (C.c_int * max(n, 1))(*[int(r) for r in request_ids])
        lens = (C.c_int * max(n, 1))(*[int(c) for c in chunk_lens])
        _capi.check(_capi.lib.nvk_batch_begin_step_chunks(self._h, ids, lens, n, int(n_heads), _devptr(stream)))
        return _capi.check(_capi.lib.nvk_batch_step_rows(self._h))

    def layer(self, layer: int, q_dev, k_rows_dev, v_rows_dev, out_dev, rope_theta: float = 10000.0,
              stream=None) -> None:
        _capi.check(_capi.lib.nvk_batch_layer(self._h, int(layer), _devptr(q_dev), _devptr(k_rows_dev),
                                              _devptr(v_rows_dev), _devptr(out_dev), float(rope_theta),
                                              _devptr(stream)))

    def layer_hidden(self, layer: int, x_dev, dim: int, rms_weight_dev, wq_dev, wk_dev, wv_dev, out_dev,
                     rope_theta: float = 10000.0, rms_eps: float = 1e-5, stream=None) -> None:
        """The layer from the hidden state (model.hpp:161-221): rms_norm + q/k/v projections + RoPE + append +
        attention; x [rows, dim] on the device, weights [.., dim] row-major on the device."""
        _capi.check(_capi.lib.nvk_batch_layer_hidden(self._h, int(layer), _devptr(x_dev), int(dim),
                                                     _devptr(rms_weight_dev), _devptr(wq_dev), _devptr(wk_dev),
                                                     _devptr(wv_dev), float(rms_eps), _devptr(out_dev),
                                                     float(rope_theta), _devptr(stream)))

    def end_step(self) -> None:
        _capi.check(_capi.lib.nvk_batch_end_step(self._h))

    def stats(self) -> dict:
        out = (C.c_longlong * 2)()
        _capi.check(_capi.lib.nvk_batch_stats(self._h, out))
        return {"full_uploads": int(out[0]), "patch_steps": int(out[1])}
