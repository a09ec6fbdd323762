import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import nano_vllm_amd as nv
from nano_vllm_amd import workloads as wl
H, KV, hd, bs = 32, 8, 128, 16
nb = 32768 // bs + 2
with nv.PagedKVPool(1, nb, bs, KV, hd) as pool:
    pool.fill_uniform(0, 1, 2)
    q = wl.uniform(3, 0, H * hd).reshape(1, H, hd)
    for T in (2048, 8192, 32768):
        tab = np.arange((T + bs - 1) // bs, dtype=np.int32).reshape(1, -1)
        for chunk in (0, 128, 256, 512, 1024, 2048):
            flags = nv.AUTO if chunk == 0 else nv.SPLIT_K
            for _ in range(5):
                nv.decode(pool, 0, q, tab, np.array([T], np.int32), flags, chunk)
            t0 = time.perf_counter()
            for _ in range(100):
                nv.decode(pool, 0, q, tab, np.array([T], np.int32), flags, chunk)
            print(f"T {T:6d} chunk {chunk:5d}: {(time.perf_counter() - t0) / 100 * 1e6:7.1f} us", flush=True)
