#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REAL REFERENCE (oracle/_ref/*,
built from /root/reference by `make -C oracle ref`).  Only inputs (synthetic, seeded) and the
reference's outputs are written; no reference source is copied.

    python tools/make_golden.py            # regenerate everything (needs /root/reference)
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from breakid_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")


def run(cmd, inp=None, env=None):
    r = subprocess.run(cmd, input=inp, capture_output=True, text=True, env=env)
    if r.returncode != 0:
        raise RuntimeError("%s failed (%d): %s" % (cmd, r.returncode, r.stderr[-2000:]))
    return r.stdout


def datasets():
    small = [("chr1", 3_000_000), ("chr2", 2_000_000), ("chr3", 1_500_000)]
    yield "g1", synth.make_g1(), synth.G1_REFGENE
    yield "g2", synth.make_g1(partner_flag=0x800), synth.G1_REFGENE
    yield "small", synth.make_cfg(101, small, 24_000, 40, 24, 150, jitter=300, read_len=100), SMALL_REFGENE
    # heavy x/y key ties (H2): tiny jitter, panel-like
    yield "edge", synth.make_edge(), EDGE_REFGENE
    yield "ties", synth.make_cfg(202, small[:2], 12_000, 12, 60, 40, jitter=6, read_len=100, split_every=1,
                                 splits_per_locus=5), SMALL_REFGENE


EDGE_REFGENE = [
    "0\tNM_200001\tchr1\t+\t50000\t350000\t50500\t349000\t3\t50000,99000,200000,\t90000,150000,350000,\t0\tEA\tcmpl\tcmpl\t0,0,0,",
    "0\tNM_200002\tchr2\t-\t100000\t390000\t100500\t389000\t2\t100000,199000,\t190000,390000,\t0\tEB\tcmpl\tcmpl\t0,0,",
    "0\tNM_200003\tchr3\t+\t10000\t200000\t10500\t199000\t1\t10000,\t200000,\t0\tEC\tcmpl\tcmpl\t0,",
]

SMALL_REFGENE = [
    "0\tNM_100001\tchr1\t+\t100000\t2900000\t100500\t2899000\t3\t100000,1000000,2000000,\t500000,1500000,2900000,\t0\tGA\tcmpl\tcmpl\t0,0,0,",
    "0\tNM_100002\tchr2\t-\t100000\t1900000\t100500\t1899000\t2\t100000,1000000,\t900000,1900000,\t0\tGB\tcmpl\tcmpl\t0,0,",
    "0\tNR_100003\tchr3\t+\t100000\t1400000\t100500\t100500\t1\t100000,\t1400000,\t0\tGC\tunk\tunk\t-1,",
    "0\tNM_100004\tchr3\t+\t200000\t1400000\t200500\t1399000\t2\t200000,900000,\t800000,1400000,\t0\tGD\tcmpl\tcmpl\t0,0,",
]


def make_dataset_golden(name, ds, refgene):
    with tempfile.TemporaryDirectory() as tmp:
        bam = os.path.join(tmp, name + ".bam")
        ds.write_bam(bam)
        side = synth.write_side_files(ds, tmp, refgene_lines=refgene)
        run([os.path.join(REF, "ref_index"), bam])
        env = dict(os.environ, BREAKID_REF_INSTALLDIR=side["install"])
        soa = ds.to_soa()
        np.savez_compressed(os.path.join(GOLD, name + ".soa.npz"), names=np.array([n for n, _ in ds.contigs]), **soa)
        for mode, flag in (("ahc", "0"), ("fast", "1")):
            out = os.path.join(GOLD, "%s.%s.stages.txt" % (name, mode))
            run([os.path.join(REF, "ref_harness"), "stages", bam, side["nib"], "20", flag, out], env=env)
            prefix = os.path.join(tmp, "out_" + mode)
            cmd = [os.path.join(REF, "BreakID_ref"), "-i", bam, "-o", prefix, "-n", side["nib"], "-all"]
            if mode == "fast":
                cmd.append("-fast")
            run(cmd, env=env)
            for suffix in ("_fusion.txt", "_fusion_all.txt"):
                with open(prefix + suffix) as f, open(os.path.join(GOLD, "%s.%s%s" % (name, mode, suffix)), "w") as g:
                    g.write(f.read())
            with open(prefix + "_params.txt") as f, open(os.path.join(GOLD, "%s.%s_params.txt" % (name, mode)), "w") as g:
                g.write(f.read().replace(tmp, "<TMP>"))
            # _performance.txt (BreakID.cc:175-191): header + the five deterministic columns (the other four are clock() times)
            with open(prefix + "_performance.txt") as f, open(os.path.join(GOLD, "%s.%s_perf5.txt" % (name, mode)), "w") as g:
                lines = f.read().split("\n")
                g.write(lines[0] + "\n" + "\t".join(lines[1].split("\t")[:5]) + "\n")
        # a few raw region / depth queries (find_sa_reads / cal_single_base_depth) incl. edge regions
        queries = []
        rng = np.random.default_rng(5)
        for k in range(12):
            t = int(rng.integers(0, len(ds.contigs)))
            s = int(rng.integers(0, ds.contigs[t][1] - 5000))
            queries.append((ds.contigs[t][0], s, s + int(rng.integers(1, 4000))))
        clusters_txt = open(os.path.join(GOLD, "%s.ahc.stages.txt" % name)).read().split("\n")
        for line in clusters_txt:
            f = line.split()
            if len(f) == 19 and f[1].startswith("chr"):
                queries.append((f[1], max(0, int(f[3]) - 1331), int(f[3]) + 1331))
                queries.append((f[2], max(0, int(f[4]) - 1331), int(f[4]) + 1331))
        res = []
        for chrom, s, e in queries[:40]:
            sa = run([os.path.join(REF, "ref_harness"), "sa", bam, chrom, str(s), str(e)])
            dp = run([os.path.join(REF, "ref_harness"), "depth", bam, chrom, str(max(1, s))]).strip()
            res.append({"chr": chrom, "start": s, "end": e, "sa": sa, "depth_at_start": dp})
        with open(os.path.join(GOLD, name + ".regions.json"), "w") as f:
            json.dump(res, f, indent=0)


def unit_vectors():
    rng = np.random.default_rng(77)
    out = {"ahc": [], "cigar": [], "points": [], "vote": []}
    # ---- AHC: random + adversarial tie sets ---------------------------------------------------
    sets = []
    sets.append(([0, 0, 1, 2, 10, 11, 11, 30], [0, 0, 1, 2, 10, 10, 11, 30], 2))
    sets.append(([5, 0, 10, 5, 5], [0, 5, 5, 10, 5], 6))                 # centre last-but..., equal distances
    sets.append(([100, 5, 0, 10, 5, 5], [100, 0, 5, 5, 10, 5], 6))
    for n, span, T in ((30, 40, 6), (60, 30, 5), (120, 200, 30), (200, 3000, 400), (300, 100, 8), (64, 8, 3)):
        x = rng.integers(0, span, n)
        y = rng.integers(0, span, n)
        sets.append((x.tolist(), y.tolist(), T))
    # collinear integers + duplicates
    sets.append((list(range(0, 40, 2)) + [10, 10, 20], [7] * 20 + [7, 7, 7], 3))
    sets.append(([1000, 1000, 1000, 1000, 2000, 2000, 2000], [50, 50, 50, 50, 90, 90, 90], 10))
    # two far-apart components with interleaved indices (cross-component tail rule)
    sets.append(([0, 10000, 3, 10003, 3, 10006, 6, 10003], [0, 0, 4, 4, 0, 0, 4, 0], 5))
    sets.append((rng.integers(0, 25, 150).tolist() + (rng.integers(0, 25, 150) + 100000).tolist(),
                 rng.integers(0, 25, 300).tolist(), 4))
    for x, y, T in sets:
        txt = "%d\n" % len(x) + "".join("%d %d\n" % (a, b) for a, b in zip(x, y))
        res = run([os.path.join(REF, "ref_units"), "ahc", str(T)], inp=txt)
        out["ahc"].append({"x": x, "y": y, "T": T, "ref": res})
    # ---- CIGAR table ------------------------------------------------------------------------------
    rows = []
    texts = ["60M40S", "60S40M", "30S30S", "30M30M40S", "10H50M40S", "50M10I40S", "100M", "5M95S", "95S5M", "0M60M40S",
             "60M0S40S", "60M40S10H", "20S30M50S", "60M40X", "60=40S", "60X40S", "1M1S", "45M55S", "55S45M", "70M30S",
             "30M70S", "3S97M", "60M5D40S", "60M5N35M", "60M40P", "M", "60MS", "060M040S", "60M40S0M"]
    for a in texts:
        for b in ["60S40M", "60M40S", "40S60M", "50M50M", "30S30S40M", "060S040M", "55S45M", "100M", "10S90M", "*", "60S40"]:
            rows.append("t %s %s 10" % (a, b))
    for e in (0, 1, 5):
        rows.append("t 60M40S 55S45M %d" % e)
        rows.append("t 60M40S 61S39M %d" % e)
    words = [[(60 << 4) | 0, (40 << 4) | 4], [(60 << 4) | 7, (40 << 4) | 4], [(60 << 4) | 8, (40 << 4) | 4],
             [(30 << 4) | 0, (30 << 4) | 8, (40 << 4) | 4], [(10 << 4) | 5, (50 << 4) | 0, (40 << 4) | 4],
             [(60 << 4) | 0, (40 << 4) | 9], [(0 << 4) | 0, (60 << 4) | 0, (40 << 4) | 4], [(60 << 4) | 0, (5 << 4) | 2, (40 << 4) | 4],
             [(60 << 4) | 0, (5 << 4) | 3, (35 << 4) | 0], [(60 << 4) | 4, (40 << 4) | 0], [(100 << 4) | 4]]
    for w in words:
        for b in ["60S40M", "60M40S", "40S60M"]:
            rows.append("b %s %s 10" % (",".join(str(v) for v in w), b))
    res = run([os.path.join(REF, "ref_units"), "cigar"], inp="\n".join(rows) + "\n")
    out["cigar"] = {"rows": rows, "ref": res}
    # ---- mask / remove_isolated / fast on point sets (np in 0..4, ties, last-element loss) -----------
    psets = []
    for n in (0, 1, 2, 3, 4, 5, 8):
        psets.append((rng.integers(0, 30, n).tolist(), rng.integers(0, 30, n).tolist(), 10.5))
    for n, span, w in ((40, 50, 6.5), (200, 400, 12.0), (500, 3000, 40.25), (300, 30, 3.0), (1000, 200000, 1331.5), (64, 10, 2.0)):
        psets.append((rng.integers(0, span, n).tolist(), rng.integers(0, span, n).tolist(), w))
    # clustered loci with duplicate coordinates
    xs, ys = [], []
    for c in range(12):
        cx, cy = int(rng.integers(0, 10 ** 6)), int(rng.integers(0, 10 ** 6))
        m = int(rng.integers(2, 40))
        xs += (cx + rng.integers(0, 12, m)).tolist()
        ys += (cy + rng.integers(0, 12, m)).tolist()
    perm = rng.permutation(len(xs))
    psets.append((np.asarray(xs)[perm].tolist(), np.asarray(ys)[perm].tolist(), 20.0))
    for x, y, w in psets:
        txt = "%d\n" % len(x) + "".join("%d %d\n" % (a, b) for a, b in zip(x, y))
        entry = {"x": x, "y": y, "w": w}
        entry["mask"] = run([os.path.join(REF, "ref_harness"), "mask", str(int(w))], inp=txt)
        entry["iso"] = run([os.path.join(REF, "ref_harness"), "iso", repr(w)], inp=txt)
        xs_sorted = sorted(zip(x, y))  # fast expects an x-sorted vector (it follows remove_isolated)
        txt2 = "%d\n" % len(x) + "".join("%d %d\n" % (a, b) for a, b in xs_sorted)
        entry["fast_in"] = [list(t) for t in xs_sorted]
        entry["fast"] = run([os.path.join(REF, "ref_harness"), "fast", repr(w)], inp=txt2) if len(x) >= 2 else ""
        out["points"].append(entry)
    # ---- vote (find_bp_pair): tie-break by key string order, +-2 neighbourhood, uint wrap near 0 --------
    def tup(q, sec, pc, ps, pe, pcg, pb, sc, ss, se, scg, sb):
        return "%s %d %s %d %d %s %d %s %d %d %s %d" % (q, sec, pc, ps, pe, pcg, pb, sc, ss, se, scg, sb)
    votes = []
    base = [("r%d" % i, 50040, 50099, 80200, 80239) for i in range(3)]
    s1 = [tup(q, 0, "chr1", a, b, "60M40S", b, "chr2", c, d, "60S40M", c) for q, a, b, c, d in base]
    s2 = [tup(q, 1, "chr1", a, b, "60M40S", b, "chr2", c, d, "60S40M", c) for q, a, b, c, d in base]
    votes.append((s1, s2, "chr1", "chr2"))
    # string-order tie-break: "100,5" < "99,5"
    s1 = [tup("a", 0, "chr1", 41, 100, "60M40S", 100, "chr2", 5, 44, "60S40M", 5),
          tup("b", 0, "chr1", 40, 99, "60M40S", 99, "chr2", 5, 44, "60S40M", 5)]
    s2 = [tup("a", 1, "chr1", 41, 100, "60M40S", 100, "chr2", 5, 44, "60S40M", 5),
          tup("b", 1, "chr1", 40, 99, "60M40S", 99, "chr2", 5, 44, "60S40M", 5)]
    votes.append((s1, s2, "chr1", "chr2"))
    # primary on the p2 side (swap), duplicates inside one side, +-2 window with far outlier, bp at 1 (uint wrap)
    s1 = [tup("a", 1, "chr2", 300, 359, "60M40S", 359, "chr1", 1, 40, "60S40M", 1),
          tup("a", 1, "chr2", 300, 359, "60M40S", 359, "chr1", 1, 40, "60S40M", 1),
          tup("c", 1, "chr2", 302, 361, "60M40S", 361, "chr1", 3, 42, "60S40M", 3),
          tup("d", 1, "chr2", 900, 959, "60M40S", 959, "chr1", 500, 539, "60S40M", 500)]
    s2 = [tup("a", 0, "chr2", 300, 359, "60M40S", 359, "chr1", 1, 40, "60S40M", 1),
          tup("c", 0, "chr2", 302, 361, "60M40S", 361, "chr1", 3, 42, "60S40M", 3),
          tup("d", 0, "chr2", 900, 959, "60M40S", 959, "chr1", 500, 539, "60S40M", 500),
          tup("e", 0, "chr2", 300, 359, "60M40S", 359, "chr1", 1, 40, "60S40M", 1)]
    votes.append((s1, s2, "chr1", "chr2"))
    # same secondary flag on both sides -> no pairs; mismatching cigar text
    s1 = [tup("a", 0, "chr1", 41, 100, "60M40S", 100, "chr2", 5, 44, "60S40M", 5)]
    s2 = [tup("a", 0, "chr1", 41, 100, "60M40S", 100, "chr2", 5, 44, "60S40M", 5),
          tup("a", 1, "chr1", 41, 100, "060M40S", 100, "chr2", 5, 44, "60S40M", 5)]
    votes.append((s1, s2, "chr1", "chr2"))
    for trial in range(6):
        n = int(rng.integers(3, 14))
        a, b = [], []
        for i in range(n):
            p = 1000 + int(rng.integers(0, 6))
            s = 7000 + int(rng.integers(0, 6))
            q = "q%d" % int(rng.integers(0, 6))
            a.append(tup(q, 0, "chr1", p - 59, p, "60M40S", p, "chr1", s, s + 39, "60S40M", s))
            b.append(tup(q, 1, "chr1", p - 59, p, "60M40S", p, "chr1", s, s + 39, "60S40M", s))
        rng.shuffle(b)
        votes.append((a, b[: int(rng.integers(1, n + 1))], "chr1", "chr1"))
    for s1, s2, c1, c2 in votes:
        txt = "%d\n%s\n%d\n%s\n%s %s\n" % (len(s1), "\n".join(s1), len(s2), "\n".join(s2), c1, c2)
        res = run([os.path.join(REF, "ref_harness"), "vote"], inp=txt)
        out["vote"].append({"s1": s1, "s2": s2, "p1_chr": c1, "p2_chr": c2, "ref": res.strip()})
    with open(os.path.join(GOLD, "units.json"), "w") as f:
        json.dump(out, f, indent=0)


def make_poison_golden():
    ds = synth.make_poison()
    with tempfile.TemporaryDirectory() as tmp:
        bam = os.path.join(tmp, "poison.bam")
        ds.write_bam(bam)
        side = synth.write_side_files(ds, tmp, refgene_lines=EDGE_REFGENE)
        run([os.path.join(REF, "ref_index"), bam])
        env = dict(os.environ, BREAKID_REF_INSTALLDIR=side["install"])
        r = subprocess.run([os.path.join(REF, "BreakID_ref"), "-i", bam, "-o", os.path.join(tmp, "o"), "-n", side["nib"], "-fast"],
                           env=env, capture_output=True, text=True)
        soa = ds.to_soa()
        np.savez_compressed(os.path.join(GOLD, "poison.soa.npz"), names=np.array([n for n, _ in ds.contigs]), **soa)
        with open(os.path.join(GOLD, "poison.json"), "w") as f:
            json.dump({"returncode": r.returncode, "stderr_tail": r.stderr[-200:]}, f)
        print("golden: poison exit", r.returncode, repr(r.stderr[-60:]))


def main():
    os.makedirs(GOLD, exist_ok=True)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    unit_vectors()
    make_poison_golden()
    for name, ds, refgene in datasets():
        make_dataset_golden(name, ds, refgene)
        print("golden:", name, len(ds.recs), "records")


if __name__ == "__main__":
    main()
