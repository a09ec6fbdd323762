"""Rank body for tests/test_sharding_gloo.py (importable by spawned processes).  Test infrastructure:
the attention of each rank is computed by the CPU checker, the partitioning and the collective are
the product's (nano_vllm_amd/sharding.py) over gloo."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def make_case():
    from nano_vllm_amd import workloads as wl

    g = wl.Geometry(8, 4, 32, 8)
    ctx = np.array([5, 130, 0, 64, 77, 17, 129, 1, 250, 33, 8], np.int32)
    work = wl.fragmented_workload("gloo_case", ctx, g, slack=0.2)
    k, v, tabs, c = wl.host_blocks_for(work, list(range(work.num_seqs)))
    return work, g, k, v, tabs, c


def run_rank(rank, world, port, result_path, c_abi_ranges=False):
    import torch
    import torch.distributed as dist

    import oracle_lib as ol
    from nano_vllm_amd import sharding as sh

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        work, g, k, v, tabs, ctx = make_case()
        q = work.q()
        whole = ol.oracle_decode_batch(q, k, v, tabs, ctx, g.block_size, g.n_kv_heads)

        # ---- sequence-sharded: no data-path collective; the gather below is the test's own check
        owners = sh.partition_sequences(ctx, world)
        mine = owners[rank]
        local = ol.oracle_decode_batch(q[mine], k, v, tabs[mine], ctx[mine], g.block_size, g.n_kv_heads)
        parts = [None] * world
        dist.all_gather_object(parts, local)
        merged = sh.merge_by_sequence(parts, owners)
        seq_ok = bool(np.array_equal(merged.view(np.uint32), whole.view(np.uint32)))
        loads = [int(ctx[o].sum()) for o in owners]

        # ---- head-sharded: one all-gather of out per layer-call
        hs = sh.HeadShard(rank, world, g.n_heads, g.n_kv_heads)
        lo, hi = hs.kv_range
        if c_abi_ranges:  # the slicing a C++ engine would do from nvk_shard_head_range
            lh, lkv, h0, kv0 = sh.c_head_range(rank, world, g.n_heads, g.n_kv_heads)
            assert (kv0, kv0 + lkv) == (lo, hi) and (h0, h0 + lh) == hs.q_range
            lo, hi = kv0, kv0 + lkv
        k_loc = np.ascontiguousarray(k[:, :, lo:hi])
        v_loc = np.ascontiguousarray(v[:, :, lo:hi])
        out_loc = ol.oracle_decode_batch(hs.local_q(q), k_loc, v_loc, tabs, ctx, g.block_size, hs.local_kv_heads)
        gathered = hs.all_gather_out(torch.from_numpy(out_loc))
        full = sh.assemble_heads(gathered).numpy()
        head_ok = bool(np.array_equal(full.view(np.uint32), whole.view(np.uint32)))
        shape_ok = tuple(gathered.shape) == (world, work.num_seqs, g.n_heads // world, g.head_dim)

        ok = torch.tensor([int(seq_ok), int(head_ok), int(shape_ok)])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if rank == 0:
            with open(result_path, "w") as fh:
                fh.write(f"{ok.tolist()} {loads}\n")
    finally:
        dist.destroy_process_group()
