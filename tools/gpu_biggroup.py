"""Run time of the hot path when the discordant pairs fall into few, large chr-pair groups (GPU box).
usage: gpu_biggroup.py <records> <number of hg19 contigs>; BK_DEBUG=sort prints the sort emulation's per-sort statistics,
BK_DEBUG_SORT_DUMP=<prefix> writes the keys and group offsets of the first five sorts."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from breakid_amd import abi, capi, synth_gpu
dev = torch.device("cuda", 0)
n = int(sys.argv[1]); nc = int(sys.argv[2])
contigs = synth_gpu.HG19[:nc]
contigs2, cols = synth_gpu.make_wgs(n, 11, dev, contigs=contigs)
ptrs = abi.device_ptrs(cols)
ctx = capi.Context(contigs2)
for rep in range(2):
    ctx.attach_device(ptrs, cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w, nv = ctx.run(qual=20, fast=True)
    ctx.sync(); t1 = time.perf_counter()
    print("records %d contigs %d: run %.1f ms, valid %d" % (cols["n"], nc, (t1 - t0) * 1e3, nv), flush=True)
