#!/usr/bin/env python3
"""Per-workgroup timeline of one nvk_decode_kernel launch (diagnostic build with -DNVK_STAMPS only):
    NVK_LIB_PATH=build/variants/stamps.so python tools/wg_timeline.py --workload config2 [--geom llama] [bench flags]
Stamps (100 MHz): 0 entry, 1 table staged, 2 main loop done (wave 0), 3 partial stored, 4 workgroup met,
5 arrival known, 6 combine done."""
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
ap.add_argument("--uniform", action="store_true")
a = ap.parse_args()
work = make_workload(a.workload, 0, a.geom)
g = work.geom
pool = nv.PagedKVPool(1, work.num_blocks, g.block_size, g.n_kv_heads, g.head_dim)
pool.fill_uniform(0, 1, 2)
q = work.q()
qd, td, cd = DeviceArray(q), DeviceArray(work.block_tables), DeviceArray(work.ctx_lens)
out = DeviceArray(np.zeros_like(q))
nwg = 1 << 16
st = DeviceArray(np.zeros((nwg, 16), np.uint64))
lib = nv._capi.lib
flags = a.flags | (nv.UNIFORM if a.uniform else 0)
for it in range(4):
    lib.nvk_debug_set_stamps(C.c_void_p(st.data_ptr() if it == 3 else 0))
    nv.decode_device(pool, 0, work.num_seqs, g.n_heads, qd, td, work.max_blocks, cd, work.max_ctx, out, flags, 0)
    pool.synchronize()
s = st.numpy().astype(np.int64)
live = s[:, 0] > 0
s = s[live]
t0 = s[:, 0].min()
us = lambda x: (x - t0) / 100.0
print(f"{work.name}: {live.sum()} workgroups stamped")
def dist(name, v):
    v = np.sort(v)
    print(f"  {name:34s} min {v[0]:8.2f}  p10 {v[len(v)//10]:8.2f}  med {v[len(v)//2]:8.2f}  p90 {v[(9*len(v))//10]:8.2f}  max {v[-1]:8.2f} us")
dist("entry", us(s[:, 0]))
ok = s[:, 1] > 0
dist("table staged - entry", (s[ok, 1] - s[ok, 0]) / 100.0)
ok = s[:, 2] > 0
dist("main loop (2 - 1)", (s[ok, 2] - s[ok, 1]) / 100.0)
dist("main loop end", us(s[ok, 2]))
ok = s[:, 4] > 0
if ok.any():
    dist("store+drain+barrier (4 - 2)", (s[ok, 4] - s[ok, 2]) / 100.0)
    ok7 = ok & (s[:, 7] > 0)
    if ok7.any():  # token-split shapes: wave 0's own join (shuffles, records to LDS), then the parts' meeting and the stores
        dist("  wave 0's slots joined, records in LDS (7 - 2)", (s[ok7, 7] - s[ok7, 2]) / 100.0)
        dist("  parts met, merged, partial stored (3 - 7)", (s[ok7, 3] - s[ok7, 7]) / 100.0)
    ok3 = ok & (s[:, 3] > 0)
    if ok3.any():
        dist("  wave join + partial stored (3 - 2)", (s[ok3, 3] - s[ok3, 2]) / 100.0)
        dist("  drain + barrier (4 - 3)", (s[ok3, 4] - s[ok3, 3]) / 100.0)
    dist("arrival known (5 - 4)", (s[ok, 5] - s[ok, 4]) / 100.0)
    red = s[:, 6] > 0
    dist("combine (6 - 5), reducers", (s[red, 6] - s[red, 5]) / 100.0)
    dist("reducer end", us(s[red, 6]))
last = np.maximum.reduce([s[:, i] for i in range(16)], axis=0)
dist("workgroup end", us(last))
print("  per-wave main-loop time (wave end - table staged), median over workgroups:")
for w in range(8):
    ok = s[:, 8 + w] > 0
    if ok.any():
        d = np.sort((s[ok, 8 + w] - s[ok, 1]) / 100.0)
        print(f"    wave {w}: min {d[0]:8.2f}  med {d[len(d)//2]:8.2f}  max {d[-1]:8.2f} us")
ids = np.nonzero(live)[0]
ok = s[:, 2] > 0
print("  main loop (2 - 1) by workgroup id % 8 (XCD), median / max:")
for x in range(8):
    sel = ok & (ids % 8 == x)
    if sel.any():
        d = (s[sel, 2] - s[sel, 1]) / 100.0
        print(f"    xcd {x}: n {sel.sum():4d}  med {np.median(d):8.2f}  max {d.max():8.2f}   by id // 8 quartiles: "
              + " ".join(f"{np.median(q):7.2f}" for q in np.array_split(d, 4)))
# how many workgroups are inside their main loop at every 5 us mark (256 CUs: a launch streams at full rate while ~256 are)
ok = s[:, 2] > 0
a0, a1 = us(s[ok, 1]), us(s[ok, 2])
end = float(us(last).max())
marks = np.arange(0.0, end + 5.0, 5.0)
print("  workgroups in their main loop at t (us):  " + "  ".join(f"{t:.0f}:{int(((a0 <= t) & (a1 > t)).sum())}" for t in marks))
busy = float((a1 - a0).sum())
print(f"  main-loop time summed over workgroups {busy:.0f} us = {busy / 256.0:.1f} us per CU of a {end:.1f} us launch "
      f"({100.0 * busy / 256.0 / end:.0f} % of the launch's CU-time is spent streaming)")
