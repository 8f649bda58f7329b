import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakid_amd import capi
from oracle import pyoracle

def near_diag(rng, n_loci, per, noise, span=240_000_000, jitter=400):
    pa = rng.integers(5000, span, n_loci); pb = pa + 50_000 + rng.integers(0, 1_000_000, n_loci)
    x = (pa[:, None] + rng.integers(-jitter, jitter + 1, (n_loci, per))).ravel()
    y = (pb[:, None] + rng.integers(-jitter, jitter + 1, (n_loci, per))).ravel()
    u1 = rng.integers(0, span, noise); u2 = rng.integers(0, span, noise)
    x = np.concatenate([x, np.minimum(u1, u2)]); y = np.concatenate([y, np.maximum(u1, u2)])
    order = np.argsort(y, kind="stable")
    return x[order].astype(np.uint32)

rng = np.random.default_rng(4)
ctx = capi.Context([("chr1", 1000)])
for name, key in [("near_diag_4k", near_diag(rng, 4_000, 50, 10_000)), ("near_diag_16k", near_diag(rng, 16_000, 50, 60_000)),
                  ("near_diag_40k", near_diag(rng, 40_000, 50, 100_000))]:
    off = np.array([0, len(key)], np.uint64)
    got = ctx.debug_std_sort(key, off)
    exp = pyoracle.unit_std_sort(key, off)
    print(name, len(key), "equal", np.array_equal(got, exp), flush=True)
