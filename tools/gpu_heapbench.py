"""Micro-benchmark of the heapsort branch of the std::sort emulation (run on the GPU box):
median-of-3 killer inputs send ~the whole array through make_heap + sort_heap in one segment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakid_amd import capi


def killer(n, div=1):
    k = n // 2
    a = np.zeros(n, np.int64)
    i = np.arange(k)
    a[:k] = np.where(i % 2 == 0, i + 1, k + i + (1 if k % 2 == 0 else 0))
    a[k:2 * k] = 2 * (i + 1)
    return (np.concatenate([[0], a]) // div).astype(np.uint32)


ctx = capi.Context([("chr1", 1000)])
rng = np.random.default_rng(1)
for n in [int(x) for x in (sys.argv[1:] or ["1000", "18000", "100000", "400000"])]:
    key = killer(n)
    off = np.array([0, len(key)], np.uint64)
    rnd = rng.integers(0, 1 << 30, len(key)).astype(np.uint32)
    ctx.debug_std_sort(key, off)
    t0 = time.perf_counter(); ctx.debug_std_sort(key, off); t1 = time.perf_counter()
    ctx.debug_std_sort(rnd, off)
    t2 = time.perf_counter(); ctx.debug_std_sort(rnd, off); t3 = time.perf_counter()
    d = (t1 - t0) - (t3 - t2)
    print("n=%d heap %.3f ms (random input %.3f ms) -> %.3f us per element" % (len(key), (t1 - t0) * 1e3, (t3 - t2) * 1e3, d * 1e6 / len(key)), flush=True)
