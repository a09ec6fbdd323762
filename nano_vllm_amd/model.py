"""A llama2.c checkpoint resident on the GPU and its forward pass as batched steps.

Mirror of the reference's ``LlamaModel`` (``include/core/model.hpp:104-433``): ``load`` reads the same file
(``:118-148, 316-365``), ``forward(token, pos)`` is ``LlamaModel::forward`` for one sequence, and ``step`` /
``step_chunks`` run it for many requests at once -- the decode batch the reference's scheduler forms but never
executes (``include/scheduler/batched_runner.hpp:84-85``).  Greedy sampling only (``sampler.hpp:27-29``)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _capi


class LlamaModel:
    def __init__(self, path: str, block_size: int = 16, num_blocks: int = 256, max_seqs: int = 1, device: int = 0):
        """block_size / num_blocks default to the reference's Config (model.hpp:35-36)."""
        h = C.c_void_p()
        _capi.check(_capi.lib.nvk_model_load(os.fsencode(path), int(block_size), int(num_blocks), int(max_seqs),
                                             int(device), C.byref(h)))
        self._h = h
        cfg = (C.c_int * 8)()
        _capi.check(_capi.lib.nvk_model_config(self._h, cfg))
        (self.dim, self.hidden_dim, self.n_layers, self.n_heads, self.n_kv_heads, self.vocab_size, self.max_seq_len,
         shared) = (int(v) for v in cfg)
        self.weights_shared = bool(shared)
        self.head_dim = self.dim // self.n_heads

    @classmethod
    def load(cls, path: str, **kw) -> "LlamaModel":
        return cls(path, **kw)

    def close(self) -> None:
        if getattr(self, "_h", None):
            _capi.lib.nvk_model_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    __del__ = close

    def add_request(self, request_id: int) -> None:
        _capi.check(_capi.lib.nvk_model_add_request(self._h, int(request_id)))

    def finish_request(self, request_id: int) -> None:
        _capi.check(_capi.lib.nvk_model_finish_request(self._h, int(request_id)))

    def request_len(self, request_id: int) -> int:
        return _capi.check(_capi.lib.nvk_model_request_len(self._h, int(request_id)))

    def use_graphs(self, on: bool) -> None:
        """Replay pure decode steps as a hipGraph (default) or launch every kernel directly; same results."""
        _capi.check(_capi.lib.nvk_model_use_graphs(self._h, int(bool(on))))

    def use_linked_norms(self, on: bool) -> None:
        """Single-row steps carry their norms between the GEMVs (default) or run them as kernels (a few ulp apart)."""
        _capi.check(_capi.lib.nvk_model_use_linked_norms(self._h, int(bool(on))))

    @property
    def free_blocks(self) -> int:
        return int(_capi.lib.nvk_model_free_blocks(self._h))

    def step_chunks(self, request_ids: Sequence[int], chunks: Sequence[Sequence[int]],
                    want_logits: bool = False) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """Request i takes the tokens chunks[i] (a prompt chunk, or one token = decode).  Returns the greedy next
        token of every request and, if asked, the logits after its last token [n, vocab_size]."""
        n = len(request_ids)
        ids = (C.c_int * n)(*[int(r) for r in request_ids])
        lens = (C.c_int * n)(*[len(c) for c in chunks])
        flat = [int(t) for c in chunks for t in c]
        toks = (C.c_int * max(len(flat), 1))(*flat)
        nxt = (C.c_int * max(n, 1))()
        logits = np.empty((n, self.vocab_size), np.float32) if want_logits else None
        _capi.check(_capi.lib.nvk_model_step_chunks(
            self._h, ids, lens, toks, n, nxt,
            logits.ctypes.data_as(C.POINTER(C.c_float)) if logits is not None else None))
        return np.array(nxt[:n], np.int32), logits

    def step(self, request_ids: Sequence[int], tokens: Sequence[int], want_logits: bool = False):
        return self.step_chunks(request_ids, [[int(t)] for t in tokens], want_logits)

    def forward(self, token: int, pos: int) -> np.ndarray:
        """LlamaModel::forward(token, pos) for one sequence: the logits [vocab_size]."""
        logits = np.empty(self.vocab_size, np.float32)
        _capi.check(_capi.lib.nvk_model_forward(self._h, int(token), int(pos),
                                                logits.ctypes.data_as(C.POINTER(C.c_float))))
        return logits


class LlamaModelGroup:
    """The model sequence-sharded over several GPUs (``nvk_model_group_*``): one ``LlamaModel``'s worth of state per
    entry of ``devices``, each driven by its own host thread inside the library; a request is placed on one member at
    admission (fewest promised tokens first) and stays there, nothing is exchanged between members.  ``devices`` may
    repeat a device (two members on one GPU: a rehearsal of two GPUs).  Same step interface as :class:`LlamaModel`."""

    def __init__(self, path: str, devices: Sequence[int], block_size: int = 16, num_blocks: int = 256,
                 max_seqs_per_member: int = 1):
        h = C.c_void_p()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        _capi.check(_capi.lib.nvk_model_group_load(os.fsencode(path), int(block_size), int(num_blocks),
                                                   int(max_seqs_per_member), len(devices), devs, C.byref(h)))
        self._h = h
        cfg = (C.c_int * 8)()
        _capi.check(_capi.lib.nvk_model_group_config(self._h, cfg))
        (self.dim, self.hidden_dim, self.n_layers, self.n_heads, self.n_kv_heads, self.vocab_size, self.max_seq_len,
         _) = (int(v) for v in cfg)
        self.size = int(_capi.lib.nvk_model_group_size(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            _capi.lib.nvk_model_group_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    __del__ = close

    def add_request(self, request_id: int, expected_tokens: int) -> int:
        """Places the request; returns the index of the member that took it."""
        return _capi.check(_capi.lib.nvk_model_group_add_request(self._h, int(request_id), int(expected_tokens)))

    def member_of(self, request_id: int) -> int:
        return int(_capi.lib.nvk_model_group_member_of(self._h, int(request_id)))

    def finish_request(self, request_id: int) -> None:
        _capi.check(_capi.lib.nvk_model_group_finish_request(self._h, int(request_id)))

    def request_len(self, request_id: int) -> int:
        return _capi.check(_capi.lib.nvk_model_group_request_len(self._h, int(request_id)))

    def use_graphs(self, on: bool) -> None:
        _capi.check(_capi.lib.nvk_model_group_use_graphs(self._h, int(bool(on))))

    def step_chunks(self, request_ids: Sequence[int], chunks: Sequence[Sequence[int]],
                    want_logits: bool = False) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        n = len(request_ids)
        ids = (C.c_int * max(n, 1))(*[int(r) for r in request_ids])
        lens = (C.c_int * max(n, 1))(*[len(c) for c in chunks])
        flat = [int(t) for c in chunks for t in c]
        toks = (C.c_int * max(len(flat), 1))(*flat)
        nxt = (C.c_int * max(n, 1))()
        logits = np.empty((n, self.vocab_size), np.float32) if want_logits else None
        _capi.check(_capi.lib.nvk_model_group_step_chunks(
            self._h, ids, lens, toks, n, nxt,
            logits.ctypes.data_as(C.POINTER(C.c_float)) if logits is not None else None))
        return np.array(nxt[:n], np.int32), logits

    def step(self, request_ids: Sequence[int], tokens: Sequence[int], want_logits: bool = False):
        return self.step_chunks(request_ids, [[int(t)] for t in tokens], want_logits)
