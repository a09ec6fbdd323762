#!/usr/bin/env python3
"""Does every XCD stream at the same rate?  (diagnostic build with -DNVK_STAMPS only)
    NVK_LIB_PATH=build/variants/stamps.so python tools/xcd_balance.py --workload config2 [--launches 8]
Workgroup ids go round-robin over the 8 XCDs (id % 8) and the hardware never moves a workgroup to another XCD, so an
XCD that streams slower than the others ends the launch late whatever its CUs' dispatcher does.  Per stamped launch:
the median main-loop time (stamp 2 - stamp 1) and the last main-loop end of the workgroups of each XCD, and the
launch's end."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import nano_vllm_amd as nv  # noqa: E402
from bench import make_workload  # noqa: E402
from hipmem import DeviceArray  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="config2")
ap.add_argument("--geom", default="llama")
ap.add_argument("--flags", type=lambda x: int(x, 0), default=0)
ap.add_argument("--launches", type=int, default=8)
ap.add_argument("--uniform", action="store_true", help="NVK_DECODE_UNIFORM: no plan, the dealing bench.py's uniform workloads get")
ap.add_argument("--seed", type=int, default=0, help="another seed = another placement of the blocks in the pool")
a = ap.parse_args()
work = make_workload(a.workload, a.seed, a.geom)
g = work.geom
pool = nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim)
pool.fill_uniform(0, 1, 2)
q = work.q()
qd, td, cd = DeviceArray(q), DeviceArray(work.block_tables), DeviceArray(work.ctx_lens)
out = DeviceArray(np.zeros_like(q))
nwg = 1 << 14
lib = nv._capi.lib
flags = a.flags | (nv.UNIFORM if a.uniform else 0)


def launch(st):
    lib.nvk_debug_set_stamps(C.c_void_p(st.data_ptr() if st is not None else 0))
    nv.decode_device(pool, 0, work.num_seqs, g.n_heads, qd, td, work.max_blocks, cd, work.max_ctx, out, flags, 0)
    pool.synchronize()


for _ in range(3):
    launch(None)
print(f"{work.name} geom={a.geom} flags={flags:#x}: main loop per XCD (median us | last end us), {a.launches} launches")
rows = []
for it in range(a.launches):
    st = DeviceArray(np.zeros((nwg, 16), np.uint64))
    launch(st)
    s = st.numpy().astype(np.int64)
    ids = np.nonzero(s[:, 0] > 0)[0]
    s = s[ids]
    t0 = s[:, 0].min()
    ok = s[:, 2] > 0
    med = [float(np.median((s[ok & (ids % 8 == x), 2] - s[ok & (ids % 8 == x), 1]) / 100.0)) for x in range(8)]
    end = [float(((s[ok & (ids % 8 == x), 2] - t0) / 100.0).max()) for x in range(8)]
    n = [int((ids % 8 == x).sum()) for x in range(8)]
    last = float((np.maximum.reduce([s[:, i] for i in range(16)], axis=0).max() - t0) / 100.0)
    rows.append(med)
    print(f"  launch {it}: " + "  ".join(f"{m:6.1f}|{e:6.1f}" for m, e in zip(med, end)) + f"   launch end {last:6.1f}   wgs/xcd {n[0]}")
r = np.array(rows)
print("  mean over launches:   " + "  ".join(f"{m:13.1f}" for m in r.mean(axis=0)))
print(f"  slowest XCD / mean of XCDs, per launch: " + " ".join(f"{x:.3f}" for x in r.max(axis=1) / r.mean(axis=1)))
