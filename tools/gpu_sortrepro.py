"""Repeats one saved case of tools/gpu_sortfuzz.py (an .npz with key / off / exp): python tools/gpu_sortrepro.py <npz> [reps]
BK_DEBUG=sortcheck makes std_sort_groups say after which phase the payloads stop being a permutation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakid_amd import capi

d = np.load(sys.argv[1])
key, off, exp = d["key"], d["off"], d["exp"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = capi.Context([("chr1", 1000)])
bad = 0
for r in range(reps):
    got = ctx.debug_std_sort(key, off)
    if not np.array_equal(got, exp):
        bad += 1
        diff = np.nonzero(got != exp)[0]
        perm = np.array_equal(np.sort(got), np.arange(len(key)))
        print("rep %d: MISMATCH at %d positions (%d..%d), a permutation: %s" % (r, len(diff), diff[0], diff[-1], perm), flush=True)
print("repro done: %d reps, %d bad" % (reps, bad), flush=True)
