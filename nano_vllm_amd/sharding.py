"""Partitioning the decode layer-call over the GPUs of one node.

The reference is single-process; its attention touches one table, one q and
one out per sequence (``include/core/attention.hpp:82-159``) with the q-head loop
outermost (``:102``) and kv-head ``h / kv_mul`` (``:105``), so a batch splits two ways
without changing the result (bit for bit by sequence; by head group up to the
association of fp32 sums, because a shard with few kv-heads spreads each chunk's
tokens over its spare waves):

* by **sequence** -- rank r owns a private pool, block manager and batch; nothing
  is exchanged (:func:`partition_sequences`, :func:`contiguous_ranges`; no collective);
* by **head group** -- rank g holds kv-heads ``[g*KV/G, (g+1)*KV/G)`` of every
  sequence and their q-heads, and the per-rank ``out`` slices are joined by ONE
  all-gather per layer-call (:class:`HeadShard`; RCCL over xGMI on the GPUs,
  gloo in the CPU tests).

One process per GPU.  The collective has two carriers: ``torch.distributed``
(:meth:`HeadShard.all_gather_out`; gloo in the CPU tests) and the library's own
C ABI, ``nvk_shard_*`` (:class:`NvkShard`: RCCL called directly, the gather on a
stream of its own under the next layer-call's attention) -- what a C++ engine uses.
The partitioning helpers are host logic only and run without a GPU.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np


# --------------------------------------------------------------------------- by sequence
def partition_sequences(ctx_lens: Sequence[int], world: int) -> List[np.ndarray]:
    """Sequence ids per rank, balanced by context tokens (the HBM bytes a sequence costs).

    Longest-processing-time first: sequences by decreasing length, each to the
    rank with the least tokens so far (ties: lowest rank, then lowest id, so every
    rank computes the same partition without talking).  Ids inside a rank stay in
    ascending order."""
    ctx = np.asarray(ctx_lens, dtype=np.int64).reshape(-1)
    if world <= 0:
        raise ValueError("world must be positive")
    order = np.lexsort((np.arange(ctx.size), -ctx))
    load = np.zeros(world, np.int64)
    owner = np.empty(ctx.size, np.int64)
    for s in order.tolist():
        r = int(np.argmin(load))  # first minimum = lowest rank
        owner[s] = r
        load[r] += ctx[s]
    return [np.nonzero(owner == r)[0] for r in range(world)]


def contiguous_ranges(num_seqs: int, world: int) -> List[Tuple[int, int]]:
    """[lo, hi) per rank for batches of equal-length sequences (configs 5a/5b: 4096 -> 512 per GPU)."""
    bounds = np.linspace(0, num_seqs, world + 1).astype(np.int64)
    return [(int(bounds[r]), int(bounds[r + 1])) for r in range(world)]


def merge_by_sequence(parts: Sequence[np.ndarray], owners: Sequence[np.ndarray]) -> np.ndarray:
    """Inverse of :func:`partition_sequences` for host arrays: rows of ``parts[r]`` go back to ids ``owners[r]``."""
    n = int(sum(len(o) for o in owners))
    first = next(p for p in parts if len(p))
    out = np.empty((n,) + tuple(first.shape[1:]), first.dtype)
    for p, o in zip(parts, owners):
        out[np.asarray(o, np.int64)] = p
    return out


# --------------------------------------------------------------------------- by head group
@dataclass(frozen=True)
class HeadShard:
    """Rank ``rank`` of ``world`` in the head-sharded layout.

    ``world`` must divide ``n_kv_heads`` (G <= n_kv_heads: 8 for 32q/8kv means one
    kv-head and four q-heads per GPU at G = 8)."""
    rank: int
    world: int
    n_heads: int
    n_kv_heads: int

    def __post_init__(self):
        if self.world <= 0 or not (0 <= self.rank < self.world):
            raise ValueError("bad rank/world")
        if self.n_kv_heads % self.world != 0:
            raise ValueError(f"world={self.world} must divide n_kv_heads={self.n_kv_heads}")
        if self.n_heads % self.n_kv_heads != 0:
            raise ValueError("n_heads must be a multiple of n_kv_heads")

    @property
    def kv_mul(self) -> int:
        return self.n_heads // self.n_kv_heads

    @property
    def local_kv_heads(self) -> int:
        return self.n_kv_heads // self.world

    @property
    def local_heads(self) -> int:
        return self.n_heads // self.world

    @property
    def kv_range(self) -> Tuple[int, int]:
        return self.rank * self.local_kv_heads, (self.rank + 1) * self.local_kv_heads

    @property
    def q_range(self) -> Tuple[int, int]:
        return self.rank * self.local_heads, (self.rank + 1) * self.local_heads

    # host-array slicing (what a rank keeps of full-size inputs)
    def local_q(self, q):
        """q [num_seqs, n_heads, head_dim] -> this rank's q-heads, contiguous."""
        lo, hi = self.q_range
        part = q[:, lo:hi]
        return part.contiguous() if hasattr(part, "contiguous") else np.ascontiguousarray(part)

    def local_kv_block(self, block):
        """One pool block [block_size, n_kv_heads, head_dim] -> this rank's kv-heads
        (the rank's pool is [num_blocks, block_size, n_kv_heads/G, head_dim]; tables are replicated)."""
        lo, hi = self.kv_range
        return np.ascontiguousarray(block[:, lo:hi])

    # the one collective
    def all_gather_out(self, out_local, gathered=None, group=None):
        """ONE all-gather per layer-call: out_local [num_seqs, n_heads/G, head_dim] on every rank ->
        gathered [G, num_seqs, n_heads/G, head_dim] on every rank (transpose-free; see :func:`assemble_heads`).

        ``out_local`` is a torch tensor (GPU: backend "nccl" = RCCL over xGMI; CPU tests: gloo)."""
        import torch
        import torch.distributed as dist

        if gathered is None:
            gathered = torch.empty((self.world,) + tuple(out_local.shape), dtype=out_local.dtype,
                                   device=out_local.device)
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            gathered[0].copy_(out_local)
            return gathered
        # concatenation form ([G*num_seqs, ...] is the same memory as [G, num_seqs, ...]): accepted by RCCL and gloo alike
        dist.all_gather_into_tensor(gathered.view((-1,) + tuple(out_local.shape[1:])), out_local.contiguous(),
                                    group=group)
        return gathered


def c_head_range(rank: int, world: int, n_heads: int, n_kv_heads: int) -> Tuple[int, int, int, int]:
    """(local q-heads, local kv-heads, first q-head, first kv-head) from the C ABI's ``nvk_shard_head_range``
    (host only; raises ValueError where :class:`HeadShard` would)."""
    import ctypes as C

    from . import _capi

    out = (C.c_int * 4)()
    if _capi.lib.nvk_shard_head_range(int(rank), int(world), int(n_heads), int(n_kv_heads), out) != 0:
        raise ValueError(_capi.lib.nvk_last_error().decode())
    return tuple(out)


class NvkShard:
    """One rank of the head-sharded split over the C ABI (``nvk_shard_*``, RCCL called directly).

    ``unique_id`` is the 128-byte id rank 0 got from :meth:`new_unique_id`, handed to the other ranks by any
    channel (``bench.py`` broadcasts it with torch.distributed).  Creation is collective."""

    def __init__(self, rank: int, world: int, unique_id: bytes, device: int, n_heads: int, n_kv_heads: int):
        import ctypes as C

        from . import _capi

        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of nvk_shard_unique_id")
        self._capi, self._C = _capi, C
        buf = C.create_string_buffer(bytes(unique_id), 128)
        h = C.c_void_p()
        _capi.check(_capi.lib.nvk_shard_create(int(rank), int(world), buf, int(device), int(n_heads), int(n_kv_heads),
                                               C.byref(h)))
        self._h = h
        self.rank, self.world, self.n_heads, self.n_kv_heads = rank, world, n_heads, n_kv_heads

    @staticmethod
    def new_unique_id() -> bytes:
        import ctypes as C

        from . import _capi

        buf = C.create_string_buffer(128)
        _capi.check(_capi.lib.nvk_shard_unique_id(buf))
        return buf.raw

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._capi.lib.nvk_shard_destroy(self._h)
            self._h = None

    __del__ = close

    def info(self) -> dict:
        out = (self._C.c_int * 6)()
        self._capi.check(self._capi.lib.nvk_shard_info(self._h, out))
        return dict(zip(("rank", "world", "local_heads", "local_kv_heads", "first_head", "first_kv_head"), out))

    def all_gather_out(self, out_local_dev, gathered_dev, num_seqs: int, head_dim: int, stream=None) -> None:
        from .kv_pool import _devptr

        self._capi.check(self._capi.lib.nvk_shard_all_gather_out(self._h, _devptr(out_local_dev), _devptr(gathered_dev),
                                                                 int(num_seqs), int(head_dim), _devptr(stream)))

    def decode_gather(self, pool, layer: int, num_seqs: int, q_local_dev, block_tables_dev, max_blocks: int,
                      ctx_lens_dev, max_ctx: int, out_local_dev, gathered_dev, buffer: int, flags: int = 0,
                      chunk_tokens: int = 0, compute_stream=None) -> None:
        """Attention of this rank's heads on ``compute_stream``, then the all-gather on the shard's own stream."""
        from .kv_pool import _devptr

        self._capi.check(self._capi.lib.nvk_shard_decode_gather(
            self._h, pool.handle, int(layer), int(num_seqs), _devptr(q_local_dev), _devptr(block_tables_dev),
            int(max_blocks), _devptr(ctx_lens_dev), int(max_ctx), _devptr(out_local_dev), _devptr(gathered_dev),
            int(buffer), int(flags), int(chunk_tokens), _devptr(compute_stream)))

    def wait_gather(self, buffer: int, stream=None) -> None:
        from .kv_pool import _devptr

        self._capi.check(self._capi.lib.nvk_shard_wait_gather(self._h, int(buffer), _devptr(stream)))

    GATHER_AUTO, GATHER_OVERLAPPED, GATHER_INLINE = 0, 1, 2

    def set_gather_mode(self, mode: int) -> None:
        """Where decode_gather runs the collective: its own stream (hidden under the next layer-call; the hand-over
        costs the compute stream 8-9 us), or in line on the compute stream; AUTO = overlapped iff world > 1."""
        self._capi.check(self._capi.lib.nvk_shard_set_gather_mode(self._h, int(mode)))

    @property
    def gather_mode(self) -> str:
        return {1: "overlapped", 2: "inline"}[self._capi.check(self._capi.lib.nvk_shard_gather_mode(self._h))]


def assemble_heads(gathered):
    """[G, num_seqs, n_heads/G, head_dim] -> [num_seqs, n_heads, head_dim] (the reference's ``out`` layout).

    A consumer that can index the gathered layout directly (the ``wo`` GEMV reads
    head h at ``gathered[h // (H/G), s, h % (H/G)]``) skips this permute."""
    if hasattr(gathered, "permute"):
        g, n, hl, hd = gathered.shape
        return gathered.permute(1, 0, 2, 3).reshape(n, g * hl, hd)
    g, n, hl, hd = gathered.shape
    return np.ascontiguousarray(np.transpose(gathered, (1, 0, 2, 3))).reshape(n, g * hl, hd)


def all_gather_bytes(num_seqs: int, n_heads: int, head_dim: int, world: int) -> Tuple[int, int]:
    """(bytes each rank contributes, bytes each rank ends up holding) for the head-sharded all-gather, fp32."""
    shard = num_seqs * (n_heads // world) * head_dim * 4
    return shard, shard * world
