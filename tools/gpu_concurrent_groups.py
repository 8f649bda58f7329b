"""Experiment (GPU box): K contexts over the same resident table, each masking + clustering the chr-pair groups an
LPT split gives it, driven from K host threads.  Compares the wall time with one context doing all groups."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from breakid_amd import abi, capi, synth_gpu
from breakid_amd.sharded import lpt_owner

n = int(sys.argv[1]) if len(sys.argv) > 1 else 620_000_000
dev = torch.device("cuda", 0)
contigs, cols = synth_gpu.make_wgs(n, 12346, dev)
ptrs = abi.device_ptrs(cols)


def prep():
    ctx = capi.Context(contigs)
    ctx.attach_device(ptrs, cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])
    mean, sd = ctx.isize_stats()
    w = capi.w_from(mean, sd)
    ctx.discordant_pairs(20, w)
    return ctx, w


KS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4)
for K in KS:
    ctxs = [prep() for _ in range(K)]
    L = ctxs[0][0].L
    starts, ng = C.POINTER(C.c_uint64)(), C.c_uint32()
    ctxs[0][0]._check(L.bk_shard_group_sizes(ctxs[0][0].h, C.byref(starts), C.byref(ng)))
    sizes = [int(starts[g + 1] - starts[g]) for g in range(ng.value)]
    owner = lpt_owner(sizes, K)
    for k, (ctx, w) in enumerate(ctxs):
        if K > 1:
            own = np.asarray([1 if o == k else 0 for o in owner], dtype=np.uint8)
            ctx._check(L.bk_shard_own_groups(ctx.h, own.ctypes.data, ng.value))
    for rep in range(3):
        for ctx, w in ctxs:
            ctx.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=lambda c=ctx, ww=w: (c.mask_and_cluster(ww, True), c.sync())) for ctx, w in ctxs]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        print("K=%d rep %d: mask_and_cluster wall %.1f ms" % (K, rep, dt * 1e3), flush=True)
    for ctx, w in ctxs:
        ctx.close()
