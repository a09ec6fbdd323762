"""Synthetic decode workloads of BASELINE.json / SURVEY.md section 8d (host logic, numpy only).

Values come from the counter-based generator ``value(seed, i) = 2*u - 1`` with
``u`` the top 24 bits of ``splitmix64(seed ^ i)`` -- the same function as
``nvk_fill_kernel`` on the device and ``oracle_fill_uniform`` in the checker --
so the GPU pool, the CPU checker and these arrays agree bit for bit without
shipping gigabytes of fixtures.  Seeds: K = 1, V = 2, q = 3.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np

SEED_K, SEED_V, SEED_Q = 1, 2, 3

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, index) -> np.ndarray:
    """splitmix64 finaliser of (seed ^ index) + golden, elementwise over uint64 ``index``."""
    with np.errstate(over="ignore"):
        x = (np.asarray(index, dtype=np.uint64) ^ np.uint64(seed)) + _GOLDEN
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


def uniform(seed: int, first_index: int, count: int) -> np.ndarray:
    """float32 U(-1, 1) values for element indices [first_index, first_index + count)."""
    idx = np.arange(first_index, first_index + count, dtype=np.uint64)
    bits = splitmix64(seed, idx) >> np.uint64(40)
    return (bits.astype(np.float32) * np.float32(1.0 / 16777216.0) * np.float32(2.0) - np.float32(1.0)).astype(
        np.float32)


def uniform_block(seed: int, physical_block: int, block_floats: int) -> np.ndarray:
    """The values block ``physical_block`` of a layer slab holds after ``fill_uniform(seed)``."""
    return uniform(seed, physical_block * block_floats, block_floats)


@dataclass
class Geometry:
    n_heads: int = 32
    n_kv_heads: int = 8
    head_dim: int = 128
    block_size: int = 16

    @property
    def row_len(self) -> int:
        return self.n_kv_heads * self.head_dim

    @property
    def block_floats(self) -> int:
        return self.block_size * self.row_len


G_LLAMA = Geometry(32, 8, 128, 16)
G_MHA = Geometry(32, 32, 128, 16)
G_TINY = Geometry(6, 6, 48, 16)


@dataclass
class Workload:
    name: str
    geom: Geometry
    num_blocks: int
    ctx_lens: np.ndarray      # [num_seqs] int32
    block_tables: np.ndarray  # [num_seqs, max_blocks] int32, rows padded with 0
    fragmented: bool = False
    notes: str = ""
    extra: dict = field(default_factory=dict)

    @property
    def num_seqs(self) -> int:
        return int(self.ctx_lens.size)

    @property
    def max_ctx(self) -> int:
        return int(self.ctx_lens.max()) if self.ctx_lens.size else 0

    @property
    def max_blocks(self) -> int:
        return int(self.block_tables.shape[1])

    @property
    def total_tokens(self) -> int:
        return int(self.ctx_lens.astype(np.int64).sum())

    def algorithmic_bytes(self) -> int:
        """B of SURVEY.md 8d: K and V once, q read + out written, table entries read."""
        g = self.geom
        ctx = self.ctx_lens.astype(np.int64)
        kv = 2 * ctx.sum() * g.n_kv_heads * g.head_dim * 4
        qo = 2 * self.num_seqs * g.n_heads * g.head_dim * 4
        tab = ((ctx + g.block_size - 1) // g.block_size).sum() * 4
        return int(kv + qo + tab)

    def q(self) -> np.ndarray:
        g = self.geom
        return uniform(SEED_Q, 0, self.num_seqs * g.n_heads * g.head_dim).reshape(self.num_seqs, g.n_heads,
                                                                                    g.head_dim)

    def blocks_of(self, seq: int) -> np.ndarray:
        n = (int(self.ctx_lens[seq]) + self.geom.block_size - 1) // self.geom.block_size
        return self.block_tables[seq, :n]


def _uniform_tables(num_seqs: int, blocks_per_seq: int) -> np.ndarray:
    return (np.arange(num_seqs, dtype=np.int64)[:, None] * blocks_per_seq
            + np.arange(blocks_per_seq, dtype=np.int64)[None, :]).astype(np.int32)


def uniform_workload(name: str, num_seqs: int, ctx: int, geom: Geometry = G_LLAMA) -> Workload:
    """Sequence s owns blocks [s*n, s*n + n) in order (configs 2, 4, 5b)."""
    per = (ctx + geom.block_size - 1) // geom.block_size
    return Workload(name, geom, num_seqs * per, np.full(num_seqs, ctx, np.int32), _uniform_tables(num_seqs, per))


def shuffled_permutation(n: int, seed: int) -> np.ndarray:
    """Fisher-Yates driven by splitmix64(seed, i): portable between numpy and C."""
    perm = np.arange(n, dtype=np.int64)
    draws = splitmix64(seed, np.arange(n, dtype=np.uint64))
    for i in range(n - 1, 0, -1):
        j = int(draws[i] % np.uint64(i + 1))
        perm[i], perm[j] = perm[j], perm[i]
    return perm


def fragmented_workload(name: str, ctx_lens: np.ndarray, geom: Geometry = G_LLAMA, slack: float = 0.25,
                        perm_seed: int = 4321) -> Workload:
    """Tables dealt round-robin from a seeded permutation of the whole pool, so a
    sequence's consecutive logical blocks are never physically adjacent (config 3)."""
    ctx_lens = np.asarray(ctx_lens, np.int32)
    need = (ctx_lens.astype(np.int64) + geom.block_size - 1) // geom.block_size
    used = int(need.sum())
    num_blocks = used + int(used * slack)
    perm = shuffled_permutation(num_blocks, perm_seed)
    max_blocks = int(need.max()) if need.size else 1
    tables = np.zeros((ctx_lens.size, max(max_blocks, 1)), np.int32)
    cursor = 0
    for b in range(max_blocks):
        takers = np.nonzero(need > b)[0]
        tables[takers, b] = perm[cursor:cursor + takers.size]
        cursor += takers.size
    return Workload(name, geom, num_blocks, ctx_lens, tables, fragmented=True)


def config2(geom: Geometry = G_LLAMA) -> Workload:
    """64 seqs x 2K ctx, uniform block tables (BASELINE.json configs[1])."""
    return uniform_workload("decode_64x2048_uniform", 64, 2048, geom)


def config3(num_seqs: int = 256, geom: Geometry = G_LLAMA) -> Workload:
    """256 seqs, ctx 128..8192, fragmented tables (configs[2])."""
    s = np.arange(num_seqs, dtype=np.uint64)
    ctx = (128 + (splitmix64(1234, s) % np.uint64(8192 - 128 + 1)).astype(np.int64)).astype(np.int32)
    return fragmented_workload(f"decode_{num_seqs}xragged128-8192_fragmented", ctx, geom)


def config4(geom: Geometry = G_LLAMA) -> Workload:
    """16 seqs x 32K ctx, uniform tables, split-K (configs[3])."""
    return uniform_workload("decode_16x32768_uniform", 16, 32768, geom)


def config5a(num_seqs: int = 4096, geom: Geometry = G_LLAMA, first_seq: int = 0) -> Workload:
    """stress-test-like short contexts: ctx = 64 + splitmix64(99, s) % 129 (configs[4], 5a)."""
    s = np.arange(first_seq, first_seq + num_seqs, dtype=np.uint64)
    ctx = (64 + (splitmix64(99, s) % np.uint64(129)).astype(np.int64)).astype(np.int32)
    w = fragmented_workload(f"decode_{num_seqs}xshort64-192", ctx, geom, slack=0.0)
    w.fragmented = True
    return w


def config5b(num_seqs: int = 512, geom: Geometry = G_LLAMA) -> Workload:
    """The bandwidth-scaling point: 2K-ctx sequences, 512 per GPU (configs[4], 5b)."""
    return uniform_workload(f"decode_{num_seqs}x2048_uniform", num_seqs, 2048, geom)


def host_blocks_for(work: Workload, seqs: List[int], seed_k: int = SEED_K, seed_v: int = SEED_V):
    """Compact host copies of just the blocks the given sequences touch.

    Returns (k_pool, v_pool, tables, ctx) where the pools hold the touched blocks
    back to back (values exactly as ``fill_uniform(seed_k, seed_v)`` leaves them in the full
    device pool) and ``tables`` is remapped onto them -- what a CPU checker needs
    to verify sequences of a pool too large to materialise on the host."""
    g = work.geom
    bf = g.block_floats
    ids: List[int] = []
    remap = {}
    for s in seqs:
        for pb in work.blocks_of(s).tolist():
            if pb not in remap:
                remap[pb] = len(ids)
                ids.append(pb)
    k_pool = np.empty((max(len(ids), 1), g.block_size, g.n_kv_heads, g.head_dim), np.float32)
    v_pool = np.empty_like(k_pool)
    for i, pb in enumerate(ids):
        k_pool[i] = uniform_block(seed_k, pb, bf).reshape(k_pool.shape[1:])
        v_pool[i] = uniform_block(seed_v, pb, bf).reshape(k_pool.shape[1:])
    tables = np.zeros((len(seqs), work.max_blocks), np.int32)
    for r, s in enumerate(seqs):
        row = work.blocks_of(s)
        tables[r, :row.size] = [remap[int(pb)] for pb in row]
    ctx = work.ctx_lens[np.asarray(seqs, dtype=np.int64)].astype(np.int32)
    return k_pool, v_pool, tables, ctx
