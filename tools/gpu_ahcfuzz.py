"""Fuzz of the AHC kernel against the oracle's util_cluster.cc restatement (unit hooks), GPU box.
python tools/gpu_ahcfuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakid_amd import capi
from oracle import pyoracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = capi.Context([("chr1", 1000)])
bad = 0
for case in range(cases):
    n = int(rng.choice([2, 3, 5, 17, 64, 65, 130, 200, 256, 257, 300, 513, 700, 1000]))
    span = int(rng.choice([3, 8, 40, 300, 5000, 100000]))
    T = int(rng.choice([1, 2, 5, 30, 400, 3000]))   # the reference's threshold is an int; the C ABI takes w (double)
    frac = float(rng.choice([0.0, 0.25, 0.75]))
    x = rng.integers(0, span, n)
    y = rng.integers(0, span, n)
    mode = int(rng.integers(0, 5))
    if mode == 0:   # duplicates
        k = max(1, n // 5)
        src = rng.integers(0, n, k); dst = rng.integers(0, n, k)
        x[dst] = x[src]; y[dst] = y[src]
    elif mode == 1:  # lattice: many equal distances
        x = (x // 4) * 4; y = (y // 4) * 4
    elif mode == 2:  # several far-apart components interleaved in x order
        y = y + (np.arange(n) % int(rng.integers(2, 6))) * 1_000_000
    elif mode == 3:  # a line
        y = np.zeros(n, np.int64)
    order = np.argsort(x, kind="stable")
    x, y = x[order].astype(np.uint32), y[order].astype(np.uint32)
    gi, gc = ctx.debug_ahc(x, y, T + frac)
    nodes = pyoracle.unit_ahc(x, y, T)
    idx, cl, k = [], [], 0
    for is_root, npts, _, _, pts in nodes:
        if is_root and npts >= 2:
            idx += pts
            cl += [k] * npts
            k += 1
    ei, ec = np.asarray(idx, np.uint32), np.asarray(cl, np.int32)
    if not (np.array_equal(gi, ei) and np.array_equal(gc, ec)):
        bad += 1
        print("MISMATCH case", case, "n", n, "span", span, "T", T, "mode", mode, len(gi), len(ei), flush=True)
print("ahc fuzz done: %d cases, %d bad" % (cases, bad), flush=True)
