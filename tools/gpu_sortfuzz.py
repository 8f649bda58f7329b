"""Fuzz of the std::sort emulation against libstdc++'s std::sort (through the oracle's unit hook), GPU box.
python tools/gpu_sortfuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakid_amd import capi
from oracle import pyoracle


def killer(n, div):
    k = n // 2
    a = np.zeros(n, np.int64)
    i = np.arange(k)
    a[:k] = np.where(i % 2 == 0, i + 1, k + i + (1 if k % 2 == 0 else 0))
    a[k:2 * k] = 2 * (i + 1)
    return (np.concatenate([[0], a]) // div).astype(np.uint32)


def gen(rng, n):
    if n == 0:
        return np.zeros(0, np.int64)
    kind = int(rng.integers(0, 9))
    if kind == 0:
        return rng.integers(0, max(2, int(rng.choice([2, 5, 100, 10_000, 1 << 30]))), n)
    if kind == 1:
        y = np.sort(rng.integers(0, 200_000_000, n))
        return (rng.random(n) * y).astype(np.int64) // int(rng.choice([1, 1, 50, 1000]))
    if kind == 2:
        return np.sort(rng.integers(0, 1 << 20, n))
    if kind == 3:
        return np.sort(rng.integers(0, 1 << 20, n))[::-1]
    if kind == 4:
        h = n // 2
        return np.concatenate([np.arange(h), np.arange(n - h)[::-1]])  # organ pipe
    if kind == 5:
        return killer(max(n, 4), int(rng.choice([1, 2, 3, 7, 40])))[:n]
    if kind == 6:
        return np.full(n, 7)
    if kind == 7:
        a = np.sort(rng.integers(0, 1 << 24, n))
        sw = rng.integers(0, n, max(1, n // 50))
        a[sw] = rng.integers(0, 1 << 24, len(sw))  # nearly sorted
        return a
    return (rng.integers(0, 1 << 16, n) << 8) | rng.integers(0, 3, n)


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = capi.Context([("chr1", 1000)])
bad = 0
for case in range(cases):
    ng = int(rng.integers(1, 40))
    sizes = []
    for g in range(ng):
        cls = int(rng.integers(0, 6))
        hi = [20, 300, 3000, 30_000, 70_000, 250_000][cls]
        if rng.random() < 0.25:  # class boundaries of the emulation: 16/17, finisher 256 / 2048, heaps 1024 / 4096 / 20000 / 40947 / 65534
            sizes.append(max(0, int(rng.choice([16, 17, 32, 33, 256, 257, 1024, 1025, 2048, 2049, 20000, 20001, 40000, 40001, 40947, 40948, 65534, 65535, 65536, 65537])) + int(rng.integers(-2, 3))))
        else:
            sizes.append(int(rng.integers(0, hi)))
    parts = [np.asarray(gen(rng, s), dtype=np.int64).astype(np.uint32) for s in sizes]
    key = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
    off = np.cumsum([0] + sizes).astype(np.uint64)
    if len(key) == 0:
        continue
    got = ctx.debug_std_sort(key, off)
    exp = pyoracle.unit_std_sort(key, off)
    if not np.array_equal(got, exp):
        bad += 1
        print("MISMATCH case", case, "sizes", sizes, "first diff", np.nonzero(got != exp)[0][:5], flush=True)
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez_compressed("gpurun_out/sortfuzz_fail_%d.npz" % case, key=key, off=off, got=got, exp=exp)
print("sort fuzz done: %d cases, %d bad" % (cases, bad), flush=True)
