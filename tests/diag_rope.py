"""Diagnostic: where do the fused RoPE kernel's q values differ from the checker's (per pair index, per pos)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import nano_vllm_amd as nv
import oracle_lib as ol
from nano_vllm_amd import workloads as wl

H, KV, hd, bs = 6, 6, 48, 16
n, max_blocks = 96, 6
nb = n * max_blocks
tabs = np.arange(nb, dtype=np.int32).reshape(n, max_blocks)
pos = np.arange(n, dtype=np.int32)
q = (wl.uniform(31, 0, n * H * hd) * np.float32(2)).reshape(n, H, hd)
k = (wl.uniform(32, 0, n * KV * hd) * np.float32(2)).reshape(n, KV, hd)
v = wl.uniform(33, 0, n * KV * hd).reshape(n, KV, hd)
with nv.PagedKVPool(1, nb, bs, KV, hd) as pool:
    q_rot = pool.rope_append_batch(0, q, k, v, tabs, pos, 10000.0)
want = np.stack([ol.apply_rope(ol.ORACLE.oracle_apply_rope, q[s], k[s], int(pos[s]), hd, H, KV)[0] for s in range(n)])
neq = q_rot.view(np.uint32) != want.view(np.uint32)
print("mismatch frac", neq.mean(), "max abs", np.abs(q_rot - want).max())
print("by pair index:", neq.reshape(n, H, hd // 2, 2).mean(axis=(0, 1, 3)).round(2))
print("by pos (first 24):", neq.mean(axis=(1, 2))[:24].round(2))
ulp = np.abs(q_rot.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
print("ulp histogram:", np.bincount(np.minimum(ulp.reshape(-1), 8)))
