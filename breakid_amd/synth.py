"""Synthetic paired-end inputs shaped like BASELINE.json's configs.

Two products:
  * a *record table* (python/numpy) that can be written as a coordinate-sorted BAM (+ nib dir,
    ref_names.txt, refGene.txt) for the reference / oracle / CLI, and
  * the same records as the columnar SoA the C-ABI takes (see include/breakid_hip.h), so the bench
    and the parity tests need no BAM round trip.

Everything is seeded; nothing here reads /root/reference.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import bamio

HG19 = [("chr1", 249250621), ("chr2", 243199373), ("chr3", 198022430), ("chr4", 191154276),
        ("chr5", 180915260), ("chr6", 171115067), ("chr7", 159138663), ("chr8", 146364022),
        ("chr9", 141213431), ("chr10", 135534747), ("chr11", 135006516), ("chr12", 133851895),
        ("chr13", 115169878), ("chr14", 107349540), ("chr15", 102531392), ("chr16", 90354753),
        ("chr17", 81195210), ("chr18", 78077248), ("chr19", 59128983), ("chr20", 63025520),
        ("chr21", 48129895), ("chr22", 51304566), ("chrX", 155270560), ("chrY", 59373566)]


def fnv1a64(name: bytes) -> int:
    """qname hash used for the qhash column (must match csrc/bam_reader.cc: bk_qname_hash)."""
    h = 0xCBF29CE484222325
    for b in name:
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    # final avalanche (splitmix64 finaliser) so that radix digits are well mixed
    h ^= h >> 30
    h = (h * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    h ^= h >> 27
    h = (h * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    h ^= h >> 31
    return h


def qname_check(name: bytes) -> int:
    """second, independent 32-bit hash of a read name (csrc/bk_hash.h: qname_check32 = bk_qname_check); never 0"""
    M = 0xFFFFFFFF
    h = (0x811C9DC5 ^ ((len(name) * 0x9E3779B1) & M)) & M
    for b in name:
        h = ((h ^ b) * 0x01000193) & M
        h = ((h << 13) | (h >> 19)) & M
        h = (h * 5 + 0xE6546B64) & M
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M
    h ^= h >> 16
    return h or 1


@dataclass
class Rec:
    qname: str
    flag: int
    tid: int
    pos: int  # 0-based
    mapq: int
    cigar: str
    mtid: int
    mpos: int
    isize: int
    sa: str = ""
    oc: str = ""


@dataclass
class Dataset:
    contigs: List[Tuple[str, int]]
    recs: List[Rec] = field(default_factory=list)

    def sort(self) -> None:
        big = 1 << 40
        # coordinate order as `samtools sort` would give (unmapped tid=-1 last); stable on ties
        self.recs.sort(key=lambda r: ((r.tid if r.tid >= 0 else big), r.pos))

    # ---- BAM + side files -------------------------------------------------------------
    def write_bam(self, path: str, aligned: bool = False) -> None:
        def gen():
            for r in self.recs:
                aux = []
                if r.sa:
                    aux.append(("SA", r.sa))
                if r.oc:
                    aux.append(("OC", r.oc))
                yield bamio.encode_record(r.qname, r.flag, r.tid, r.pos, r.mapq, bamio.parse_cigar(r.cigar),
                                          r.mtid, r.mpos, r.isize, aux)
        bamio.write_bam(path, self.contigs, gen(), aligned=aligned)

    # ---- SoA ---------------------------------------------------------------------------
    def to_soa(self) -> Dict[str, np.ndarray]:
        n = len(self.recs)
        soa = {
            "tid": np.empty(n, np.int32), "pos": np.empty(n, np.int32), "mtid": np.empty(n, np.int32),
            "mpos": np.empty(n, np.int32), "isize": np.empty(n, np.int32), "flag": np.empty(n, np.uint16),
            "mapq": np.empty(n, np.uint8), "qhash": np.empty(n, np.uint64), "qcheck": np.empty(n, np.uint32),
            "cigar_off": np.zeros(n + 1, np.uint32), "aux_off": np.zeros(n + 1, np.uint32),
        }
        cig: List[int] = []
        aux = bytearray()
        for i, r in enumerate(self.recs):
            soa["tid"][i] = r.tid
            soa["pos"][i] = r.pos
            soa["mtid"][i] = r.mtid
            soa["mpos"][i] = r.mpos
            soa["isize"][i] = r.isize
            soa["flag"][i] = r.flag
            soa["mapq"][i] = r.mapq
            soa["qhash"][i] = fnv1a64(r.qname.encode())
            soa["qcheck"][i] = qname_check(r.qname.encode())
            cig.extend(bamio.parse_cigar(r.cigar))
            soa["cigar_off"][i + 1] = len(cig)
            aux += encode_aux(r.sa, r.oc)
            soa["aux_off"][i + 1] = len(aux)
        soa["cigar"] = np.asarray(cig, dtype=np.uint32)
        soa["aux"] = np.frombuffer(bytes(aux), dtype=np.uint8).copy()
        soa["target_len"] = np.asarray([l for _, l in self.contigs], dtype=np.uint32)
        return soa


def fnv1a64_fixed(names) -> "np.ndarray":
    """fnv1a64() of many equal-length byte strings at once: `names` is a (n, L) uint8 array."""
    with np.errstate(over="ignore"):
        h = np.full(names.shape[0], 0xCBF29CE484222325, np.uint64)
        for j in range(names.shape[1]):
            h = (h ^ names[:, j].astype(np.uint64)) * np.uint64(0x100000001B3)
        h ^= h >> np.uint64(30)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(27)
        h *= np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(31)
    return h


def name_records(cols):
    """Give every record of a generated table (whose qhash column is only a pair id hash) a real read name, the 16 hex
    digits of that hash, and re-derive qhash from the name the way the BAM decoders do.  Returns (cols', names) with names
    an (n, 16) uint8 array; mates keep equal names.  This is how synth_gpu tables are written as BAM for the reference."""
    q = np.ascontiguousarray(cols["qhash"]).view(np.uint64)
    digits = np.frombuffer(b"0123456789abcdef", np.uint8)
    names = np.empty((len(q), 16), np.uint8)
    for j in range(16):
        names[:, j] = digits[((q >> np.uint64(60 - 4 * j)) & np.uint64(15)).astype(np.int64)]
    out = dict(cols)
    out["qhash"] = fnv1a64_fixed(names)
    out["qcheck"] = qname_check_fixed(names)
    return out, names


def qname_check_fixed(names) -> "np.ndarray":
    """qname_check() of many equal-length byte strings at once: `names` is a (n, L) uint8 array."""
    with np.errstate(over="ignore"):
        n, L = names.shape
        h = np.full(n, (0x811C9DC5 ^ ((L * 0x9E3779B1) & 0xFFFFFFFF)) & 0xFFFFFFFF, np.uint32)
        for j in range(L):
            h = (h ^ names[:, j].astype(np.uint32)) * np.uint32(0x01000193)
            h = (h << np.uint32(13)) | (h >> np.uint32(19))
            h = h * np.uint32(5) + np.uint32(0xE6546B64)
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    return np.where(h == 0, np.uint32(1), h).astype(np.uint32)


def encode_aux(sa: str, oc: str) -> bytes:
    """Per-record aux blob of the SoA: '' (no SA), SA text, or OC text + '\\t' + SA text.
    Records without an SA tag carry nothing (the path never looks at OC then)."""
    if not sa:
        return b""
    if oc:
        return oc.encode() + b"\t" + sa.encode()
    return sa.encode()


# ----------------------------------------------------------------------------------------------
def write_side_files(ds, root: str, seed: int = 7, refgene_lines: Sequence[str] = (), max_nib_len: int = 5_000_000) -> Dict[str, str]:
    """nib dir (+ref_names.txt) and INSTALLDIR/ref_files/refGene.txt for the reference / CLI.  `ds` is a Dataset or a
    contig list; contigs longer than `max_nib_len` get no sequence file."""
    contigs = ds.contigs if hasattr(ds, "contigs") else list(ds)
    nib = os.path.join(root, "nib")
    inst = os.path.join(root, "install")
    os.makedirs(nib, exist_ok=True)
    os.makedirs(os.path.join(inst, "ref_files"), exist_ok=True)
    rng = np.random.default_rng(seed)
    with open(os.path.join(nib, "ref_names.txt"), "w") as f:
        for name, _ in contigs:
            f.write(name + "\n")
    code_of = np.asarray([2, 1, 3, 0], np.uint8)  # "ACGT"[i] -> nib code (T=0 C=1 A=2 G=3, nibtools.cc:18-26)
    for name, ln in contigs:
        p = os.path.join(nib, "hg19_%s.nib" % name)
        if ln <= max_nib_len:
            vals = code_of[rng.integers(0, 4, ln)]
            if ln & 1:
                vals = np.concatenate([vals, np.zeros(1, np.uint8)])
            with open(p, "wb") as f:
                f.write(np.asarray([0x6BE93D3A, ln], "<u4").tobytes())
                f.write(((vals[0::2] << 4) | vals[1::2]).astype(np.uint8).tobytes())
    with open(os.path.join(inst, "ref_files", "refGene.txt"), "w") as f:
        for l in refgene_lines:
            f.write(l + "\n")
    return {"nib": nib, "install": inst}


# ----------------------------------------------------------------------------------------------
def _proper_pair(rng, i, tid, lo, hi, read_len, ins_mean, ins_sd, prefix="p"):
    ins = max(read_len + 1, int(round(rng.normal(ins_mean, ins_sd))))
    s = int(rng.integers(lo, max(lo + 1, hi - ins)))
    e = s + ins - read_len
    q = "%s%d" % (prefix, i)
    c = "%dM" % read_len
    return [Rec(q, 0x1 | 0x2 | 0x20 | 0x40, tid, s, 60, c, tid, e, ins),
            Rec(q, 0x1 | 0x2 | 0x10 | 0x80, tid, e, 60, c, tid, s, -ins)]


def _discordant_pair(q, ta, pa, tb, pb, read_len, rev_a=False, rev_b=True, mapq=60):
    c = "%dM" % read_len
    fa = 0x1 | 0x40 | (0x10 if rev_a else 0) | (0x20 if rev_b else 0)
    fb = 0x1 | 0x80 | (0x10 if rev_b else 0) | (0x20 if rev_a else 0)
    isz = 0 if ta != tb else (pb - pa + read_len)
    return [Rec(q, fa, ta, pa, mapq, c, tb, pb, isz), Rec(q, fb, tb, pb, mapq, c, ta, pa, -isz)]


def _split_pair(q, names, ta, bpa, tb, bpb, m1=60, m2=40, partner_flag=0x100):
    """Primary m1M m2S ending at 1-based bpa on A; partner m1S m2M starting at 1-based bpb on B."""
    pos_a = bpa - m1  # 0-based start so that 1-based end == bpa
    pos_b = bpb - 1
    c1 = "%dM%dS" % (m1, m2)
    c2 = "%dS%dM" % (m1, m2)
    sa1 = "%s,%d,+,%s,60,0;" % (names[tb], pos_b + 1, c2)
    sa2 = "%s,%d,+,%s,60,0;" % (names[ta], pos_a + 1, c1)
    prim = Rec(q, 0x1 | 0x2 | 0x40 | 0x20, ta, pos_a, 60, c1, ta, pos_a + 200, 300, sa=sa1)
    part = Rec(q, 0x1 | 0x40 | 0x20 | partner_flag, tb, pos_b, 60, c2, ta, pos_a + 200, 0, sa=sa2)
    mate = Rec(q, 0x1 | 0x2 | 0x80 | 0x10, ta, pos_a + 200, 60, "100M", ta, pos_a, -300)
    return [prim, part, mate]


def make_g1(partner_flag: int = 0x100, seed: int = 12345) -> Dataset:
    """24 x 200 kb contigs, 6000 proper pairs, 12 discordant pairs chr1->chr2,
    6 split reads (60M40S at chr1 / 0x100 partner 60S40M at chr2:80200)."""
    rng = np.random.default_rng(seed)
    names = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
    ds = Dataset([(n, 200_000) for n in names])
    for i in range(6000):
        tid = int(rng.integers(0, 24))
        ds.recs += _proper_pair(rng, i, tid, 1000, 199_000, 100, 350, 40)
    for i in range(12):
        pa = 49_700 + int(rng.integers(0, 200))
        pb = 80_050 + int(rng.integers(0, 150))
        ds.recs += _discordant_pair("d%d" % i, 0, pa, 1, pb, 100)
    for i in range(6):
        ds.recs += _split_pair("s%d" % i, names, 0, 50_099, 1, 80_200, partner_flag=partner_flag)
    ds.sort()
    return ds


def random_refgene(contigs, n_genes: int, seed: int) -> List[str]:
    """Synthetic refGene.txt rows (UCSC table layout read by RefSeqTranscript.cc:203-260): genes with 2-6 exons spread over
    the contigs, both strands, a few NR_ (skipped by the reference) among them."""
    rng = np.random.default_rng(seed)
    rows = []
    for g in range(n_genes):
        t = int(rng.integers(0, len(contigs)))
        name, ln = contigs[t]
        span = int(rng.integers(20_000, max(20_001, min(ln // 4, 2_000_000))))
        s = int(rng.integers(1000, ln - span - 1000))
        ne = int(rng.integers(2, 7))
        cuts = np.sort(rng.choice(np.arange(s + 100, s + span - 100), 2 * ne - 2, replace=False))
        starts = [s] + [int(v) for v in cuts[1::2]]
        ends = [int(v) for v in cuts[0::2]] + [s + span]
        acc = ("NR_%06d" if g % 11 == 10 else "NM_%06d") % (300000 + g)
        rows.append("0\t%s\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%s\t%s\t0\tSG%d\tcmpl\tcmpl\t%s" % (
            acc, name, "+-"[int(rng.integers(0, 2))], s, s + span, s + 50, s + span - 50, ne,
            "".join("%d," % v for v in starts), "".join("%d," % v for v in ends), g, "0," * ne))
    return rows


G1_REFGENE = [
    "0\tNM_000001\tchr1\t+\t40000\t60000\t40500\t59500\t3\t40000,50000,58000,\t41000,51000,60000,\t0\tGENEA\tcmpl\tcmpl\t0,0,0,",
    "0\tNM_000002\tchr2\t-\t70000\t90000\t70500\t89500\t2\t70000,80100,\t71000,90000,\t0\tGENEB\tcmpl\tcmpl\t0,0,",
]


def make_cfg(seed: int, contigs: Sequence[Tuple[str, int]], n_records: int, n_loci: int, pairs_per_locus: int,
             noise_pairs: int, split_every: int = 2, splits_per_locus: int = 8, jitter: int = 400,
             read_len: int = 150, ins_mean: float = 350, ins_sd: float = 40, same_chr_frac: float = 0.3,
             partner_flag: int = 0x100, dup_frac: float = 0.01, lowq_frac: float = 0.02) -> Dataset:
    """BASELINE config-1 shape, scalable: `n_records` total, discordant mass concentrated in
    `n_loci` loci x `pairs_per_locus` pairs (+- jitter) plus `noise_pairs` uniform pairs; split reads
    (primary + `partner_flag` partner) at every `split_every`-th locus."""
    rng = np.random.default_rng(seed)
    names = [n for n, _ in contigs]
    lens = [l for _, l in contigs]
    nt = len(contigs)
    ds = Dataset(list(contigs))
    used = 0

    def rand_site(margin=5000):
        t = int(rng.integers(0, nt))
        return t, int(rng.integers(margin, lens[t] - margin))

    for li in range(n_loci):
        ta, pa = rand_site()
        if rng.random() < same_chr_frac:
            tb = ta
            pb = int(rng.integers(5000, lens[tb] - 5000))
            if abs(pb - pa) < 20000:
                pb = (pa + 50000) % (lens[tb] - 10000) + 5000
        else:
            tb, pb = rand_site()
        rev_a, rev_b = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        for k in range(pairs_per_locus):
            ja = pa + int(rng.integers(-jitter, jitter + 1))
            jb = pb + int(rng.integers(-jitter, jitter + 1))
            mq = 60
            if rng.random() < lowq_frac:
                mq = int(rng.integers(0, 20))
            recs = _discordant_pair("L%d_%d" % (li, k), ta, ja, tb, jb, read_len, rev_a, rev_b, mq)
            if rng.random() < dup_frac:
                recs[0].flag |= 0x400
            ds.recs += recs
            used += 2
        if split_every and li % split_every == 0:
            for k in range(splits_per_locus):
                ds.recs += _split_pair("S%d_%d" % (li, k), names, ta, pa + 30, tb, pb + 30, 90, 60, partner_flag)
                used += 3
    for i in range(noise_pairs):
        ta, pa = rand_site()
        tb, pb = rand_site()
        ds.recs += _discordant_pair("N%d" % i, ta, pa, tb, pb, read_len, bool(rng.integers(0, 2)),
                                    bool(rng.integers(0, 2)))
        used += 2
    n_prop = max(0, (n_records - used) // 2)
    tids = rng.integers(0, nt, n_prop)
    for i in range(n_prop):
        t = int(tids[i])
        ds.recs += _proper_pair(rng, i, t, 1000, lens[t] - 1000, read_len, ins_mean, ins_sd)
    ds.sort()
    return ds


def make_edge(seed: int = 4321) -> Dataset:
    """Edge cases of the mate join (H3) and of the per-read evidence (A12/A13): three records per read name,
    a mate below the mapq threshold, mate coordinates that disagree, OC tags, hard clips, =/X ops, SA text with
    several entries / leading zeros / empty fields, left-clipped primaries, 0x800 partners, duplicates."""
    rng = np.random.default_rng(seed)
    names = ["chr%d" % i for i in range(1, 5)]
    ds = Dataset([(n, 400_000) for n in names])
    for i in range(3000):
        ds.recs += _proper_pair(rng, i, int(rng.integers(0, 4)), 1000, 399_000, 100, 330, 35)
    # three loci with well formed discordant pairs
    loci = [(0, 100_000, 1, 200_000), (0, 250_000, 2, 120_000), (1, 60_000, 1, 300_000)]
    for li, (ta, pa, tb, pb) in enumerate(loci):
        for k in range(14):
            ja, jb = pa + int(rng.integers(-200, 200)), pb + int(rng.integers(-200, 200))
            ds.recs += _discordant_pair("e%d_%d" % (li, k), ta, ja, tb, jb, 100, bool(k & 1), bool(k & 2))
    # --- join quirks ---
    # (1) third record with the same name (supplementary 0x800 is not filtered): buffered again, pairs with a fourth
    a, b = _discordant_pair("trip", 0, 100_050, 1, 200_050, 100)
    sup = Rec("trip", 0x1 | 0x40 | 0x800, 2, 50_000, 60, "50S50M", 1, 200_050, 0)
    late = Rec("trip", 0x1 | 0x80, 3, 70_000, 60, "100M", 2, 50_000, 0)
    ds.recs += [a, b, sup, late]
    # (2) mate fails mapq: never pairs; (3) mate's mpos differs from the mate's pos; (4) same chr, closer than w
    a, b = _discordant_pair("lowq", 0, 100_060, 1, 200_060, 100)
    b.mapq = 3
    ds.recs += [a, b]
    a, b = _discordant_pair("skew", 0, 100_070, 1, 200_070, 100)
    b.mpos = 150_000
    a.mpos = 210_000
    ds.recs += [a, b]
    a, b = _discordant_pair("near", 2, 30_000, 2, 30_400, 100)
    ds.recs += [a, b]
    a, b = _discordant_pair("dupl", 0, 100_080, 1, 200_080, 100)
    a.flag |= 0x400
    ds.recs += [a, b]
    # --- split-read variants at locus 0 (chr1:100000 / chr2:200000) ---
    def sp(q, cig1, cig2, sa1_extra="", oc1="", oc2="", flag2=0x100, pos_a=99_950, pos_b=199_999, sa1=None, sa2=None, dup=False):
        s1 = sa1 if sa1 is not None else "chr2,%d,+,%s,60,0;%s" % (pos_b + 1, cig2, sa1_extra)
        s2 = sa2 if sa2 is not None else "chr1,%d,+,%s,60,0;" % (pos_a + 1, cig1)
        prim = Rec(q, 0x1 | 0x2 | 0x40 | 0x20 | (0x400 if dup else 0), 0, pos_a, 60, cig1, 0, pos_a + 150, 250, sa=s1, oc=oc1)
        part = Rec(q, 0x1 | 0x40 | 0x20 | flag2, 1, pos_b, 60, cig2, 0, pos_a + 150, 0, sa=s2, oc=oc2)
        mate = Rec(q, 0x1 | 0x2 | 0x80 | 0x10, 0, pos_a + 150, 60, "100M", 0, pos_a, -250)
        return [prim, part, mate]
    for i in range(4):
        ds.recs += sp("ok%d" % i, "60M40S", "60S40M")
    ds.recs += sp("multi", "60M40S", "60S40M", sa1_extra="chr3,500,-,30M70S,20,1;")
    ds.recs += sp("left", "40S60M", "40M60S", pos_a=100_010, pos_b=199_950)
    ds.recs += sp("left2", "40S60M", "40M60S", pos_a=100_010, pos_b=199_950)
    ds.recs += sp("eqx", "30=30X40S", "60S40M")
    ds.recs += sp("hard", "60M40H", "60H40M")
    ds.recs += sp("ocp", "55M45S", "60S40M", oc1="60M40S")               # OC replaces the BAM cigar of the primary
    ds.recs += sp("ocs", "60M40S", "58S42M", oc2="60S40M")               # ... and of the 0x100 partner
    ds.recs += sp("ocmerge", "50M50S", "60S40M", oc1="30M30M40S")
    ds.recs += sp("zeros", "60M40S", "60S40M", sa1="chr2,200000,+,060S040M,60,0;")
    # (an SA text with fewer than 4 comma fields makes the reference index past the end of a vector: undefined, not pinned)
    ds.recs += sp("empt", "60M40S", "60S40M", sa1="chr2,,200000,+,,60S40M,60,0;")
    ds.recs += sp("supp", "60M40S", "60S40M", flag2=0x800)
    ds.recs += sp("dupp", "60M40S", "60S40M", dup=True)
    ds.recs += sp("three", "50M10I40S", "60S40M")
    ds.recs += sp("nocomp", "90M10S", "60S40M")
    ds.recs += sp("ins", "60M40S", "60S10I30M")
    ds.recs += sp("unk", "60M40S", "60S40M", sa1="chrUn_x,77,+,60S40M,60,0;", sa2="chr1,99951,+,60M40S,60,0;")
    ds.sort()
    return ds


def make_poison() -> Dataset:
    """A complementary pair whose SA cigar rolls to a single clip-free op (50M50M -> 100M): the reference prints
    "error cigar" and exits with -1 (BreakID.cc:954-968) once the read is inside a queried region."""
    ds = make_edge()
    bad = Rec("bad", 0x1 | 0x2 | 0x40 | 0x20, 0, 99_960, 60, "5M95S", 0, 100_100, 250, sa="chr2,200001,+,50M50M,60,0;")
    ds.recs.append(bad)
    ds.sort()
    return ds


# ----------------------------------------------------------------------------------------------
def median3_killer(n: int, div: int = 1) -> np.ndarray:
    """Musser's median-of-3 killer shifted by one element (libstdc++ samples first+1, mid, last-1): drives
    introsort into its depth limit, so std::sort heapsorts segments of up to ~n elements.  `div` adds ties."""
    k = n // 2
    a = np.zeros(n, np.int64)
    i = np.arange(k)
    a[:k] = np.where(i % 2 == 0, i + 1, k + i + (1 if k % 2 == 0 else 0))
    a[k:2 * k] = 2 * (i + 1)
    return (np.concatenate([[0], a]) // div).astype(np.uint32)


DEEP_GROUPS = [(0, 0, 1000, 1), (0, 1, 12000, 2), (1, 1, 20000, 3), (0, 2, 30000, 1), (1, 2, 39000, 3), (2, 2, 50000, 1),
               (0, 3, 64000, 5), (1, 3, 65596, 1), (2, 3, 100000, 2), (3, 3, 300000, 2)]


def make_deep(seed: int = 31337, groups=DEEP_GROUPS, n_proper: int = 20000, read_len: int = 100):
    """Chromosome-pair groups whose pairs are DISCOVERED in median-of-3-killer order of the p1 coordinate, so that the
    reference's first std::sort of every group (remove_isolated_pairs, BreakID.cc:1274) runs into introsort's depth limit
    and heapsorts segments of ~n/2 elements: every heap size class of the sort emulation, pinned by the reference binary.
    Discovery order = file order of the later mate (p2), so p2 positions ascend (ties keep generation order) while the p1
    positions follow the killer sequence; everything stays within the mask distance, so the later sorts see all pairs.
    Returns (contigs, cols, names): numpy SoA + (n, 16) uint8 read names (see name_records)."""
    rng = np.random.default_rng(seed)
    contigs = [("chr%d" % (i + 1), 5_000_000) for i in range(6)]
    parts = {k: [] for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "pid")}
    pid0 = 1 << 40
    for gi, (ta, tb, n, div) in enumerate(groups):
        kv = median3_killer(n, div).astype(np.int64)
        n = len(kv)
        T = max(1, (n + 799) // 800)
        xa = 100_000 + 400_000 * (gi % 3)
        yb = 2_500_000 + 600_000 * (gi % 4)
        x = xa + kv
        y = yb + np.arange(n) // T
        ra, rb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        fa = 0x1 | 0x40 | (0x10 if ra else 0) | (0x20 if rb else 0)
        fb = 0x1 | 0x80 | (0x10 if rb else 0) | (0x20 if ra else 0)
        isz = (y - x + read_len) if ta == tb else np.zeros(n, np.int64)
        pid = pid0 + np.arange(n)
        pid0 += n
        for tid, pos, mtid, mpos, iz, fl in ((ta, x, tb, y, isz, fa), (tb, y, ta, x, -isz, fb)):
            parts["tid"].append(np.full(n, tid)); parts["pos"].append(pos); parts["mtid"].append(np.full(n, mtid))
            parts["mpos"].append(mpos); parts["isize"].append(iz); parts["flag"].append(np.full(n, fl)); parts["pid"].append(pid)
    t = rng.integers(4, 6, n_proper)
    ins = np.maximum(read_len + 1, np.rint(rng.normal(350, 40, n_proper)).astype(np.int64))
    s = rng.integers(1000, 4_900_000, n_proper)
    e = s + ins - read_len
    pid = np.arange(n_proper)
    for tid, pos, mpos, iz, fl in ((t, s, e, ins, 0x63), (t, e, s, -ins, 0x93)):
        parts["tid"].append(tid); parts["pos"].append(pos); parts["mtid"].append(tid); parts["mpos"].append(mpos)
        parts["isize"].append(iz); parts["flag"].append(np.full(n_proper, fl)); parts["pid"].append(pid)
    c = {k: np.concatenate(v) for k, v in parts.items()}
    order = np.argsort((c["tid"].astype(np.int64) << 32) | c["pos"].astype(np.int64), kind="stable")
    n = len(order)
    with np.errstate(over="ignore"):
        q = (c["pid"][order].astype(np.uint64) ^ np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
        q ^= q >> np.uint64(29)
        q *= np.uint64(0xBF58476D1CE4E5B9)
        q ^= q >> np.uint64(32)
    cols = {"tid": c["tid"][order].astype(np.int32), "pos": c["pos"][order].astype(np.int32), "mtid": c["mtid"][order].astype(np.int32),
            "mpos": c["mpos"][order].astype(np.int32), "isize": c["isize"][order].astype(np.int32), "flag": c["flag"][order].astype(np.uint16),
            "mapq": np.full(n, 60, np.uint8), "qhash": q, "cigar_off": np.arange(n + 1, dtype=np.uint32),
            "cigar": np.full(n, (read_len << 4), np.uint32), "aux_off": np.zeros(n + 1, np.uint32), "aux": np.zeros(0, np.uint8),
            "target_len": np.asarray([l for _, l in contigs], np.uint32)}
    cols, names = name_records(cols)
    return contigs, cols, names
