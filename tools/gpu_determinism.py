"""Runs the hot path R times on one resident table and compares every stage's output with the first run's: a race anywhere in
the path shows up as a run that differs (the std::sort emulation's heap loops are timing sensitive by construction).
    python tools/gpu_determinism.py [records] [runs] [mode fast|ahc]"""
import hashlib, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("BREAKID_GROUP_LANES", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from breakid_amd import abi, capi, synth_gpu

n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
fast = (sys.argv[3] if len(sys.argv) > 3 else "fast") == "fast"
dev = torch.device("cuda", 0)
contigs, cols = synth_gpu.make_wgs(n_rec, 4711, dev)
torch.cuda.synchronize()
torch.cuda.empty_cache()
ctx = capi.Context(contigs)
ptrs = abi.device_ptrs(cols)
stages = [abi.STAGE_GROUP_KEYS, abi.STAGE_ISO, abi.STAGE_CLUSTERED, abi.STAGE_CLUSTERS, abi.STAGE_SPLITS]


def digest():
    out = []
    for st in stages:
        a, goff = ctx.fetch(st)
        h = hashlib.sha256(np.ascontiguousarray(a).tobytes())
        if goff is not None:
            h.update(np.ascontiguousarray(goff).tobytes())
        out.append(h.hexdigest()[:16])
    return out


first, bad = None, 0
for r in range(runs):
    ctx.attach_device(ptrs, cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])
    w, nv = ctx.run(qual=20, fast=fast)
    d = digest() + [repr(w), str(nv)]
    if first is None:
        first = d
        print("run 0:", d, flush=True)
    elif d != first:
        bad += 1
        print("run %d DIFFERS:" % r, [i for i, (x, y) in enumerate(zip(d, first)) if x != y], d, flush=True)
print("determinism: %d runs of %d records, %d differ from the first" % (runs, cols["n"], bad), flush=True)
sys.exit(1 if bad else 0)
