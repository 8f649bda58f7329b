"""Long fuzz run (GPU box): random small tables, GPU vs oracle on every stage.  python tools/gpu_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakid_amd import abi, capi, synth
from oracle import pyoracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
contig_sets = [[("chr1", 400_000), ("chr2", 300_000)], [("chr1", 2_000_000), ("chr2", 1_500_000), ("chr3", 900_000), ("chrX", 700_000), ("chrM", 60_000)],
               [("chr%d" % i, 3_000_000) for i in range(1, 23)] + [("chrX", 2_000_000), ("chrY", 500_000)]]
bad = 0
for case in range(cases):
    fast = bool(rng.integers(0, 2))
    contigs = contig_sets[int(rng.integers(0, 3))]
    loci = int(rng.integers(1, 60))
    ppl = int(rng.integers(2, 300 if fast else 70))
    noise = int(rng.integers(0, 3000))
    jitter = int(rng.choice([0, 1, 3, 30, 250, 900]))
    n = 2 * (loci * ppl + noise) + 3 * 12 * loci + int(rng.integers(200, 60_000))
    args = dict(split_every=int(rng.integers(1, 4)), splits_per_locus=int(rng.integers(0, 12)), jitter=jitter, read_len=int(rng.choice([50, 100, 150])),
                same_chr_frac=float(rng.choice([0.0, 0.3, 1.0])), partner_flag=int(rng.choice([0x100, 0x800])), dup_frac=float(rng.choice([0.0, 0.01, 0.3])),
                lowq_frac=float(rng.choice([0.0, 0.02, 0.4])))
    seed = int(rng.integers(1, 1 << 30))
    ds = synth.make_cfg(seed, contigs, n, loci, ppl, noise, **args)
    cols = ds.to_soa()
    qual = int(rng.choice([0, 20, 30]))
    ctx = capi.Context(contigs)
    ctx.upload(cols)
    ok = True
    try:
        w, nv = ctx.run(qual=qual, fast=fast)
        o = pyoracle.Oracle(contigs, cols)
        ow, rc = o.run(qual, fast=fast)
        ok = rc == 0 and ow == w
        for st in (abi.STAGE_SCAN, abi.STAGE_ISO, abi.STAGE_CLUSTERED, abi.STAGE_SPLITS, abi.STAGE_CLUSTERS):
            a, _ = ctx.fetch(st)
            b, _ = o.fetch(st)
            if len(a) != len(b) or not np.array_equal(a, b):
                ok = False
                print("MISMATCH stage", st, len(a), len(b))
        o.close()
    except capi.BreakIDError as e:
        print("ERROR", e)
        ok = False
    ctx.close()
    if not ok:
        bad += 1
        print("case", case, "seed", seed, "fast", fast, "contigs", len(contigs), loci, ppl, noise, n, args, qual, flush=True)
print("fuzz done: %d cases, %d bad" % (cases, bad), flush=True)
