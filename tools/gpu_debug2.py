import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from breakid_amd import abi, capi, synth_gpu
from oracle import pyoracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dev = torch.device("cuda", 0)
contigs, cols = synth_gpu.make_wgs(n, 12346, dev)
host = synth_gpu.to_numpy_cols(cols)
o = pyoracle.Oracle(contigs, host)
ow, _ = o.run(20, True)
ctx = capi.Context(contigs)
ctx.attach_device({k: cols[k].data_ptr() for k, _ in abi.SOA_COLS}, cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])
mean, sd = ctx.isize_stats(); print("isize", (mean, sd) == o.isize_stats(), flush=True)
w = capi.w_from(mean, sd)
print("pairs", ctx.discordant_pairs(20, w), flush=True)
a, ao = ctx.fetch(abi.STAGE_SCAN); b, bo = o.fetch(abi.STAGE_SCAN)
print("scan equal", np.array_equal(a, b), np.array_equal(ao, bo), len(a), len(b), "max group", int(np.diff(bo).max()), flush=True)
try:
    print("cluster", ctx.mask_and_cluster(w, True), flush=True)
except Exception as e:
    print("ERR", e, flush=True); sys.exit(0)
for st, nm in ((abi.STAGE_ISO, "iso"), (abi.STAGE_CLUSTERED, "clustered")):
    a, ao = ctx.fetch(st); b, bo = o.fetch(st)
    print(nm, "equal", np.array_equal(a, b), np.array_equal(ao, bo), len(a), len(b), flush=True)
ctx.split_evidence(); ctx.cluster_summary(w); print("bp", ctx.split_breakpoints(w), flush=True)
for st, nm in ((abi.STAGE_SPLITS, "splits"), (abi.STAGE_CLUSTERS, "clusters")):
    a, _ = ctx.fetch(st); b, _ = o.fetch(st)
    print(nm, "equal", np.array_equal(a, b), len(a), len(b), flush=True)
