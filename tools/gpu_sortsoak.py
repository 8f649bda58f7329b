"""Repeats the sort emulation on inputs that drive every heap class (LDS, LDS + global tail, global) many times each and
compares with libstdc++ every time: timing-dependent faults do not show in a single run.
    python tools/gpu_sortsoak.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakid_amd import capi
from oracle import pyoracle
from tools.gpu_sortfuzz import killer

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = capi.Context([("chr1", 1000)])
rng = np.random.default_rng(5)
cases = [("killer 233512/3", [killer(233512, 3)]), ("killer 190000/1", [killer(190000, 1)]), ("killer 170000/2 + 90000/5", [killer(170000, 2), killer(90000, 5)]),
         ("killer 300000/2", [killer(300000, 2)]), ("killer 120000/7 x3", [killer(120000, 7), killer(118000, 3), killer(126000, 1)]),
         ("killer 700000/4", [killer(700000, 4)]), ("killer 65600/1 + 81000/2", [killer(65600, 1), killer(81000, 2)]),
         ("sorted-with-noise 400000", [(np.sort(rng.integers(0, 1 << 20, 400000)) + rng.integers(0, 3, 400000)).astype(np.uint32)])]
bad = 0
for name, parts in cases:
    key = np.concatenate(parts).astype(np.uint32)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    exp = pyoracle.unit_std_sort(key, off)
    b = 0
    for r in range(reps):
        got = ctx.debug_std_sort(key, off)
        if not np.array_equal(got, exp):
            b += 1
            print("  %s rep %d: MISMATCH at %d positions, a permutation: %s" % (name, r, int((got != exp).sum()), bool(np.array_equal(np.sort(got), np.arange(len(key))))), flush=True)
    print("%-32s %d reps, %d bad" % (name, reps, b), flush=True)
    bad += b
print("SORTSOAK %d bad" % bad, flush=True)
sys.exit(1 if bad else 0)
