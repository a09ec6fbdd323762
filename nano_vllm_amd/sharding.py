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

One process per GPU, ``torch.distributed`` for the collective.  The attention
itself is whatever callable the caller passes (``nano_vllm_amd.decode_device`` in
``bench.py``); this module is host logic only and runs without a GPU.
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
