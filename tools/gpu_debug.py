import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakid_amd import abi, capi, synth
from oracle import pyoracle
from tests import refdump
name = sys.argv[1] if len(sys.argv) > 1 else "g1"
fast = True
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
contigs, cols = refdump.load_soa(gd, name)
o = pyoracle.Oracle(contigs, cols)
ow, _ = o.run(20, fast)
ctx = capi.Context(contigs)
ctx.upload(cols)
mean, sd = ctx.isize_stats(); print("isize", mean, sd, o.isize_stats(), flush=True)
w = capi.w_from(mean, sd)
print("pairs", ctx.discordant_pairs(20, w), flush=True)
a, ao = ctx.fetch(abi.STAGE_SCAN); b, bo = o.fetch(abi.STAGE_SCAN)
print("scan equal", np.array_equal(a, b), np.array_equal(ao, bo), len(a), len(b), flush=True)
print("cluster", ctx.mask_and_cluster(w, fast), flush=True)
for st, nm in ((abi.STAGE_ISO, "iso"), (abi.STAGE_CLUSTERED, "clustered")):
    a, ao = ctx.fetch(st); b, bo = o.fetch(st)
    print(nm, "equal", np.array_equal(a, b), np.array_equal(ao, bo), len(a), len(b), flush=True)
print("splits", ctx.split_evidence(), flush=True)
a, _ = ctx.fetch(abi.STAGE_SPLITS); b, _ = o.fetch(abi.STAGE_SPLITS)
print("splits equal", np.array_equal(a, b), len(a), len(b), flush=True)
print("summary", ctx.cluster_summary(w), flush=True)
print("bp", ctx.split_breakpoints(w), flush=True)
a, _ = ctx.fetch(abi.STAGE_CLUSTERS); b, _ = o.fetch(abi.STAGE_CLUSTERS)
print("clusters equal", np.array_equal(a, b), len(a), len(b), flush=True)
print(a[:3]); print(b[:3])
