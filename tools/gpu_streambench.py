"""k_stream timing on a WGS-shape table (run on the GPU box): prints the average launch time from bk_timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from breakid_amd import abi, capi, synth_gpu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000
dev = torch.device("cuda", 0)
contigs, cols = synth_gpu.make_wgs(n, 12346, dev)
ctx = capi.Context(contigs)
ptrs = abi.device_ptrs(cols)
ctx.timing_enable(True)
for it in range(4):
    ctx.attach_device(ptrs, cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])
    ctx.isize_stats()
    ctx.sync()
t = [(nm, ms, by) for nm, ms, by in ctx.timing() if nm == "k_stream"]
ms = sum(x[1] for x in t[1:]) / max(1, len(t) - 1)
print("k_stream n=%d: %.3f ms, %.1f GB/s algorithmic" % (cols["n"], ms, t[-1][2] / ms / 1e6), flush=True)
