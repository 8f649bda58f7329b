import sys, time, torch
sys.path.insert(0, "/root/repo")
from breakid_amd import abi, capi, synth_gpu
dev = torch.device("cuda", 0)
n = int(sys.argv[1]); nc = int(sys.argv[2])
contigs = synth_gpu.HG19[:nc]
contigs2, cols = synth_gpu.make_wgs(n, 11, dev, contigs=contigs)
ptrs = {k: cols[k].data_ptr() for k, _ in abi.SOA_COLS}
ctx = capi.Context(contigs2)
for rep in range(2):
    ctx.attach_device(ptrs, cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w, nv = ctx.run(qual=20, fast=True)
    ctx.sync(); t1 = time.perf_counter()
    print("records %d contigs %d: run %.1f ms, valid %d" % (cols["n"], nc, (t1 - t0) * 1e3, nv), flush=True)
