"""WGS-shape synthetic record tables generated directly in HBM (torch is only used as an array
library here).  Shape = BASELINE.json config 2: hg19 contigs, 2x150 bp, insert
N(350,40), ~5 % of records discordant (80 % in loci of 50 pairs with +-400 bp jitter, 20 % uniform
noise), 8 split reads (primary + 0x100 partner + mate) at every second locus.  Coordinate sorted.

The result is the columnar table of include/breakid_hip.h as torch tensors on `device`, so bench.py can
hand device pointers to the C ABI (BK_MEM_DEVICE) and the timed region starts with inputs resident."""
from __future__ import annotations

import numpy as np
import torch

from .synth import HG19


def _np_special(rng, contigs, n_loci, pairs_per_locus, noise_pairs, split_every, splits_per_locus, read_len, jitter):
    """Discordant pairs + split triplets as numpy columns (vectorised)."""
    lens = np.asarray([l for _, l in contigs], dtype=np.int64)
    names = np.asarray([n for n, _ in contigs])
    nt = len(contigs)

    def sites(k, margin=5000):
        t = rng.integers(0, nt, k)
        p = (rng.random(k) * (lens[t] - 2 * margin)).astype(np.int64) + margin
        return t, p

    cols = {k: [] for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "mapq", "pairid", "c0", "c1", "sa")}

    def add(tid, pos, mtid, mpos, isize, flag, mapq, pairid, c0, c1=None, sa=None):
        n = len(tid)
        cols["tid"].append(tid.astype(np.int32)); cols["pos"].append(pos.astype(np.int32))
        cols["mtid"].append(mtid.astype(np.int32)); cols["mpos"].append(mpos.astype(np.int32))
        cols["isize"].append(isize.astype(np.int32)); cols["flag"].append(flag.astype(np.uint16))
        cols["mapq"].append(mapq.astype(np.uint8)); cols["pairid"].append(pairid.astype(np.int64))
        cols["c0"].append(np.full(n, c0, np.uint32) if np.isscalar(c0) else c0.astype(np.uint32))
        cols["c1"].append(np.zeros(n, np.uint32) if c1 is None else np.full(n, c1, np.uint32))
        cols["sa"].append(np.full(n, "", dtype="U48") if sa is None else sa.astype("U48"))

    M = lambda n: (n << 4) | 0
    S = lambda n: (n << 4) | 4
    # loci
    la_t, la_p = sites(n_loci)
    lb_t, lb_p = sites(n_loci)
    same = rng.random(n_loci) < 0.3
    lb_t = np.where(same, la_t, lb_t)
    lb_p = np.where(same, (la_p + 50000 + (rng.random(n_loci) * 1e6).astype(np.int64)) % (lens[lb_t] - 10000) + 5000, lb_p)
    rev_a = rng.integers(0, 2, n_loci).astype(bool)
    rev_b = rng.integers(0, 2, n_loci).astype(bool)
    k = n_loci * pairs_per_locus
    li = np.repeat(np.arange(n_loci), pairs_per_locus)
    ta, tb = la_t[li], lb_t[li]
    pa = la_p[li] + rng.integers(-jitter, jitter + 1, k)
    pb = lb_p[li] + rng.integers(-jitter, jitter + 1, k)
    ra, rb = rev_a[li], rev_b[li]
    mq = np.where(rng.random(k) < 0.02, rng.integers(0, 20, k), 60)
    dup = (rng.random(k) < 0.01)
    pid = np.arange(k, dtype=np.int64) + (1 << 40)
    fa = 0x1 | 0x40 | np.where(ra, 0x10, 0) | np.where(rb, 0x20, 0) | np.where(dup, 0x400, 0)
    fb = 0x1 | 0x80 | np.where(rb, 0x10, 0) | np.where(ra, 0x20, 0)
    isz = np.where(ta == tb, pb - pa + read_len, 0)
    add(ta, pa, tb, pb, isz, fa, mq, pid, M(read_len))
    add(tb, pb, ta, pa, -isz, fb, mq, pid, M(read_len))
    # noise pairs
    if noise_pairs:
        ta, pa = sites(noise_pairs)
        tb, pb = sites(noise_pairs)
        ra = rng.integers(0, 2, noise_pairs).astype(bool)
        rb = rng.integers(0, 2, noise_pairs).astype(bool)
        pid = np.arange(noise_pairs, dtype=np.int64) + (2 << 40)
        fa = 0x1 | 0x40 | np.where(ra, 0x10, 0) | np.where(rb, 0x20, 0)
        fb = 0x1 | 0x80 | np.where(rb, 0x10, 0) | np.where(ra, 0x20, 0)
        isz = np.where(ta == tb, pb - pa + read_len, 0)
        m60 = np.full(noise_pairs, 60)
        add(ta, pa, tb, pb, isz, fa, m60, pid, M(read_len))
        add(tb, pb, ta, pa, -isz, fb, m60, pid, M(read_len))
    # split triplets at every `split_every`-th locus
    sl = np.arange(0, n_loci, split_every) if split_every else np.zeros(0, np.int64)
    if len(sl) and splits_per_locus:
        m1, m2 = 90, 60
        li = np.repeat(sl, splits_per_locus)
        k = len(li)
        ta, tb = la_t[li], lb_t[li]
        bpa, bpb = la_p[li] + 30, lb_p[li] + 30
        pos_a, pos_b = bpa - m1, bpb - 1
        pid = np.arange(k, dtype=np.int64) + (3 << 40)
        c1t, c2t = "%dM%dS" % (m1, m2), "%dS%dM" % (m1, m2)
        sa1 = np.char.add(np.char.add(np.char.add(names[tb], ","), (pos_b + 1).astype("U12")), ",+,%s,60,0;" % c2t)
        sa2 = np.char.add(np.char.add(np.char.add(names[ta], ","), (pos_a + 1).astype("U12")), ",+,%s,60,0;" % c1t)
        m60 = np.full(k, 60)
        add(ta, pos_a, ta, pos_a + 200, np.full(k, 300), np.full(k, 0x1 | 0x2 | 0x40 | 0x20), m60, pid, M(m1), S(m2), sa1)
        add(tb, pos_b, ta, pos_a + 200, np.zeros(k), np.full(k, 0x1 | 0x40 | 0x20 | 0x100), m60, pid, S(m1), M(m2), sa2)
        add(ta, pos_a + 200, ta, pos_a, np.full(k, -300), np.full(k, 0x1 | 0x2 | 0x80 | 0x10), m60, pid, M(100))
    return {k: np.concatenate(v) if v else np.zeros(0) for k, v in cols.items()}


def _mix64(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (two's complement wraparound, logical shifts emulated)."""
    def lsr(v, s):
        return (v >> s) & ((1 << (64 - s)) - 1)
    x = x ^ lsr(x, 30)
    x = x * (-4658895280553007687)  # 0xBF58476D1CE4E5B9
    x = x ^ lsr(x, 27)
    x = x * (-7723592293110705685)  # 0x94D049BB133111EB
    x = x ^ lsr(x, 31)
    return x


def _qcheck_of(pid: torch.Tensor, salt: int) -> torch.Tensor:
    """the qcheck column of a generated table: a second hash of the pair id (mates share it), never 0; int32 bits == uint32"""
    q = (_mix64(pid ^ (salt ^ 0x5851F42D4C957F2D)) >> 32) & 0xFFFFFFFF
    q = torch.where(q == 0, torch.ones_like(q), q)
    return torch.where(q >= (1 << 31), q - (1 << 32), q).to(torch.int32)


def make_wgs(n_records: int, seed: int, device, contigs=HG19, read_len=150, disc_frac=0.05, pairs_per_locus=50,
             split_every=2, splits_per_locus=8, jitter=400, ins_mean=350.0, ins_sd=40.0):
    """Returns (contigs, cols) with cols a dict of torch tensors on `device` (see abi.SOA_COLS) plus
    'n', 'n_cigar_words', 'n_aux_bytes'."""
    rng = np.random.default_rng(seed)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    disc_records = int(n_records * disc_frac)
    disc_pairs = disc_records // 2
    n_loci = max(1, int(disc_pairs * 0.8) // pairs_per_locus)
    noise_pairs = max(0, disc_pairs - n_loci * pairs_per_locus)
    sp = _np_special(rng, contigs, n_loci, pairs_per_locus, noise_pairs, split_every, splits_per_locus, read_len, jitter)
    n_special = len(sp["tid"])
    P = max(0, (n_records - n_special) // 2)
    lens = torch.tensor([l for _, l in contigs], dtype=torch.int64, device=device)
    prefix = torch.cumsum(lens, 0) - lens
    G = int(lens.sum().item())
    # proper pairs in HBM
    gs = (torch.rand(P, generator=g, device=device, dtype=torch.float64) * (G - 1)).to(torch.int64)
    tid = torch.searchsorted(prefix, gs, right=True) - 1
    pos = gs - prefix[tid]
    ins = torch.clamp(torch.round(torch.randn(P, generator=g, device=device) * ins_sd + ins_mean).to(torch.int64), min=read_len + 1)
    pos = torch.minimum(pos, lens[tid] - ins - 1).clamp_(min=0)
    mpos = pos + ins - read_len
    pid = torch.arange(P, device=device, dtype=torch.int64)
    del gs

    def cat(a, b, special, dt):
        return torch.cat([a.to(dt), b.to(dt), torch.from_numpy(np.ascontiguousarray(special)).to(device).to(dt)])

    c_tid = cat(tid, tid, sp["tid"], torch.int32)
    c_pos = cat(pos, mpos, sp["pos"], torch.int32)
    c_mtid = cat(tid, tid, sp["mtid"], torch.int32)
    c_mpos = cat(mpos, pos, sp["mpos"], torch.int32)
    c_isize = cat(ins, -ins, sp["isize"], torch.int32)
    f1 = torch.full((P,), 0x63, device=device, dtype=torch.int32)
    f2 = torch.full((P,), 0x93, device=device, dtype=torch.int32)
    c_flag = cat(f1, f2, sp["flag"].astype(np.int32), torch.int32)
    m = torch.full((P,), 60, device=device, dtype=torch.uint8)
    c_mapq = cat(m, m, sp["mapq"], torch.uint8)
    salt = ((seed & 0x7FFF) << 48) ^ 0x1E3779B97F4A7C15
    c_pid = cat(pid, pid, sp["pairid"], torch.int64)
    c_qh = _mix64(c_pid ^ salt)
    c_qc = _qcheck_of(c_pid, salt)
    del c_pid
    w150 = torch.full((P,), (read_len << 4), device=device, dtype=torch.int64)
    c_c0 = cat(w150, w150, sp["c0"].astype(np.int64), torch.int64)
    zeros = torch.zeros(P, device=device, dtype=torch.int64)
    c_c1 = cat(zeros, zeros, sp["c1"].astype(np.int64), torch.int64)
    del tid, pos, mpos, ins, pid, f1, f2, m, w150, zeros
    n = c_tid.numel()
    # coordinate sort (stable, so equal keys keep generation order like `samtools sort`)
    key = (c_tid.to(torch.int64) << 32) | c_pos.to(torch.int64)
    perm = torch.sort(key, stable=True)[1]
    del key

    def take(t):
        return t[perm].contiguous()

    out = {"tid": take(c_tid), "pos": take(c_pos), "mtid": take(c_mtid), "mpos": take(c_mpos), "isize": take(c_isize)}
    del c_tid, c_pos, c_mtid, c_mpos, c_isize
    out["flag"] = take(c_flag).to(torch.int16)  # same bits as uint16
    out["mapq"] = take(c_mapq)
    out["qhash"] = take(c_qh)  # int64 bits == uint64 hash
    out["qcheck"] = take(c_qc)
    del c_flag, c_mapq, c_qh, c_qc
    c0 = take(c_c0)
    c1 = take(c_c1)
    ncig = 1 + (c1 != 0).to(torch.int64)
    coff = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(ncig, 0, out=coff[1:])
    nwords = int(coff[-1].item())
    cigar = torch.zeros(max(nwords, 1), dtype=torch.int64, device=device)
    cigar[coff[:-1]] = c0
    two = (c1 != 0).nonzero().squeeze(1)
    cigar[coff[two] + 1] = c1[two]
    out["cigar_off"] = coff.to(torch.int32)
    out["cigar"] = cigar.to(torch.int32)
    del c0, c1, c_c0, c_c1, ncig, cigar
    # aux blobs (SA text) for the special records that carry one, in sorted order
    sa = sp["sa"]
    has = np.nonzero(sa != "")[0]
    aux_len_special = np.zeros(n_special, np.int64)
    sab = np.char.encode(sa[has], "ascii") if len(has) else np.zeros(0, "S1")
    lens_sa = np.char.str_len(sab).astype(np.int64) if len(has) else np.zeros(0, np.int64)
    aux_len_special[has] = lens_sa
    aux_len = torch.cat([torch.zeros(2 * P, dtype=torch.int64, device=device), torch.from_numpy(aux_len_special).to(device)])
    aux_len = aux_len[perm]
    aoff = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(aux_len, 0, out=aoff[1:])
    nbytes = int(aoff[-1].item())
    # order of SA-bearing records after the sort
    src = perm[(aux_len > 0).nonzero().squeeze(1)] - 2 * P  # index into the special arrays, sorted order
    src_np = src.cpu().numpy()
    rank_in_has = np.searchsorted(has, src_np)
    if len(has):
        width = int(lens_sa.max())
        mat = np.frombuffer(sab.astype("S%d" % width).tobytes(), dtype=np.uint8).reshape(len(has), width)[rank_in_has]
        mask = np.arange(width)[None, :] < lens_sa[rank_in_has][:, None]
        blob = mat[mask]
    else:
        blob = np.zeros(0, np.uint8)
    assert len(blob) == nbytes
    out["aux_off"] = aoff.to(torch.int32)
    out["aux"] = torch.from_numpy(np.ascontiguousarray(blob) if nbytes else np.zeros(1, np.uint8)).to(device)
    out["n"] = n
    out["n_cigar_words"] = nwords
    out["n_aux_bytes"] = nbytes
    del perm, aux_len, aoff
    out["side"] = side_rows(out)
    return list(contigs), out


def side_rows(cols):
    """The bk_side layout of a table (include/breakid_hip.h): qhash, mtid, mpos, qcheck of every record in one 32-byte row, as a
    producer of device tables may hand it over next to (or instead of) those four columns.  int64 [n, 4]: the hash, mtid in the
    low and mpos in the high half of the second word, qcheck in the low half of the third."""
    n = cols["tid"].numel()
    side = torch.zeros((n, 4), dtype=torch.int64, device=cols["tid"].device)
    side[:, 0] = cols["qhash"]
    side[:, 1] = (cols["mtid"].to(torch.int64) & 0xFFFFFFFF) | (cols["mpos"].to(torch.int64) << 32)
    if "qcheck" in cols:
        side[:, 2] = cols["qcheck"].to(torch.int64) & 0xFFFFFFFF
    return side


def to_numpy_cols(cols):
    """Device table -> host numpy dict in the abi.SOA_COLS layout (for the oracle / cpu_baseline leg)."""
    view = {"flag": np.uint16, "qhash": np.uint64, "cigar_off": np.uint32, "cigar": np.uint32, "aux_off": np.uint32, "qcheck": np.uint32}
    out = {}
    for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "mapq", "qhash", "cigar_off", "cigar", "aux_off", "aux") + (("qcheck",) if "qcheck" in cols else ()):
        a = cols[k].cpu().numpy()
        out[k] = a.view(view[k]) if k in view else a
    out["cigar"] = out["cigar"][: cols["n_cigar_words"]]
    out["aux"] = out["aux"][: cols["n_aux_bytes"]]
    return out


# ---------------------------------------------------------------------------------------------------------------
# One sample sharded over `world` ranks: rank r generates only the records whose genome-wide coordinate falls into
# its contiguous range, from counter-based random numbers, so that mates / split partners that live on other
# shards carry consistent positions without any rank materialising the whole sample.
def _h64(*parts):
    """splitmix64 of a tuple of broadcastable uint64 arrays / ints (vectorised, wraparound arithmetic)."""
    with np.errstate(over="ignore"):
        x = np.uint64(0x9E3779B97F4A7C15)
        for p in parts:
            x = (x ^ np.asarray(p, dtype=np.uint64)) * np.uint64(0xBF58476D1CE4E5B9)
            x = x ^ (x >> np.uint64(29))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _site(seed, salt, idx, lens, prefix, margin=5000):
    """deterministic (tid, pos, gpos) of site `idx`."""
    G = int(prefix[-1] + lens[-1])
    g = (_h64(seed, salt, idx) % np.uint64(G)).astype(np.int64)
    t = np.searchsorted(prefix, g, side="right") - 1
    p = np.clip(g - prefix[t], margin, lens[t] - margin)
    return t, p, prefix[t] + p


def make_wgs_shard(n_per_rank: int, seed: int, device, rank: int, world: int, contigs=HG19, read_len=150, disc_frac=0.05,
                   pairs_per_locus=50, split_every=2, splits_per_locus=8, jitter=400, ins_mean=350.0, ins_sd=40.0):
    """Records of rank `rank` (genome range [rank, rank+1) * G / world) of one world*n_per_rank-record sample."""
    lens = np.asarray([l for _, l in contigs], dtype=np.int64)
    names = np.asarray([n for n, _ in contigs])
    prefix = np.cumsum(lens) - lens
    G = int(lens.sum())
    lo, hi = rank * G // world, (rank + 1) * G // world
    n_total = n_per_rank * world
    disc_pairs = int(n_total * disc_frac) // 2
    n_loci = max(1, int(disc_pairs * 0.8) // pairs_per_locus)
    noise_pairs = max(0, disc_pairs - n_loci * pairs_per_locus)
    sd = np.uint64(seed)
    cols = {k: [] for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "mapq", "pairid", "c0", "c1", "sa")}

    def add(mask, tid, pos, mtid, mpos, isize, flag, mapq, pairid, c0, c1=None, sa=None):
        k = int(mask.sum())
        if k == 0:
            return
        f = lambda a: (np.full(len(mask), a) if np.isscalar(a) else np.asarray(a))[mask]
        cols["tid"].append(f(tid).astype(np.int32)); cols["pos"].append(f(pos).astype(np.int32))
        cols["mtid"].append(f(mtid).astype(np.int32)); cols["mpos"].append(f(mpos).astype(np.int32))
        cols["isize"].append(f(isize).astype(np.int32)); cols["flag"].append(f(flag).astype(np.uint16))
        cols["mapq"].append(f(mapq).astype(np.uint8)); cols["pairid"].append(f(pairid).astype(np.int64))
        cols["c0"].append(f(c0).astype(np.uint32))
        cols["c1"].append(np.zeros(k, np.uint32) if c1 is None else f(c1).astype(np.uint32))
        cols["sa"].append(np.full(k, "", dtype="U48") if sa is None else np.asarray(sa)[mask].astype("U48"))

    M = lambda n: (n << 4) | 0
    S = lambda n: (n << 4) | 4
    own = lambda t, p: ((prefix[t] + p) >= lo) & ((prefix[t] + p) < hi)
    li_all = np.arange(n_loci, dtype=np.int64)
    la_t, la_p, la_g = _site(sd, 1, li_all, lens, prefix)
    lb_t, lb_p, lb_g = _site(sd, 2, li_all, lens, prefix)
    same = (_h64(sd, 3, li_all) % np.uint64(10)) < 3
    lb_t = np.where(same, la_t, lb_t)
    lb_p = np.where(same, (la_p + 50000 + (_h64(sd, 4, li_all) % np.uint64(1000000)).astype(np.int64)) % (lens[lb_t] - 10000) + 5000, lb_p)
    rev_a = (_h64(sd, 5, li_all) & np.uint64(1)).astype(bool)
    rev_b = (_h64(sd, 6, li_all) & np.uint64(1)).astype(bool)
    span = jitter + 10
    near = lambda t, p: ((prefix[t] + p + span) >= lo) & ((prefix[t] + p - span) < hi)
    mine = np.nonzero(near(la_t, la_p) | near(lb_t, lb_p))[0]
    # loci pairs
    li = np.repeat(mine, pairs_per_locus)
    kk = np.tile(np.arange(pairs_per_locus, dtype=np.int64), len(mine))
    ta, tb = la_t[li], lb_t[li]
    pa = la_p[li] + (_h64(sd, 7, li, kk) % np.uint64(2 * jitter + 1)).astype(np.int64) - jitter
    pb = lb_p[li] + (_h64(sd, 8, li, kk) % np.uint64(2 * jitter + 1)).astype(np.int64) - jitter
    r9 = _h64(sd, 9, li, kk)
    mq = np.where((r9 % np.uint64(100)) < 2, ((r9 >> np.uint64(8)) % np.uint64(20)).astype(np.int64), 60)
    dup = ((r9 >> np.uint64(20)) % np.uint64(100)) < 1
    ra, rb = rev_a[li], rev_b[li]
    pid = li * 64 + kk + (1 << 40)
    fa = 0x1 | 0x40 | np.where(ra, 0x10, 0) | np.where(rb, 0x20, 0) | np.where(dup, 0x400, 0)
    fb = 0x1 | 0x80 | np.where(rb, 0x10, 0) | np.where(ra, 0x20, 0)
    isz = np.where(ta == tb, pb - pa + read_len, 0)
    add(own(ta, pa), ta, pa, tb, pb, isz, fa, mq, pid, M(read_len))
    add(own(tb, pb), tb, pb, ta, pa, -isz, fb, mq, pid, M(read_len))
    # noise pairs (every rank evaluates the cheap site hashes of all of them and keeps its own)
    if noise_pairs:
        ni = np.arange(noise_pairs, dtype=np.int64)
        ta, pa, _ = _site(sd, 11, ni, lens, prefix)
        tb, pb, _ = _site(sd, 12, ni, lens, prefix)
        ra = (_h64(sd, 13, ni) & np.uint64(1)).astype(bool)
        rb = (_h64(sd, 14, ni) & np.uint64(1)).astype(bool)
        pid = ni + (2 << 40)
        fa = 0x1 | 0x40 | np.where(ra, 0x10, 0) | np.where(rb, 0x20, 0)
        fb = 0x1 | 0x80 | np.where(rb, 0x10, 0) | np.where(ra, 0x20, 0)
        isz = np.where(ta == tb, pb - pa + read_len, 0)
        add(own(ta, pa), ta, pa, tb, pb, isz, fa, 60, pid, M(read_len))
        add(own(tb, pb), tb, pb, ta, pa, -isz, fb, 60, pid, M(read_len))
    # split triplets
    sl = mine[mine % split_every == 0] if split_every else mine[:0]
    if len(sl) and splits_per_locus:
        m1, m2 = 90, 60
        li = np.repeat(sl, splits_per_locus)
        kk = np.tile(np.arange(splits_per_locus, dtype=np.int64), len(sl))
        ta, tb = la_t[li], lb_t[li]
        bpa, bpb = la_p[li] + 30, lb_p[li] + 30
        pos_a, pos_b = bpa - m1, bpb - 1
        pid = li * 64 + kk + (3 << 40)
        c1t, c2t = "%dM%dS" % (m1, m2), "%dS%dM" % (m1, m2)
        sa1 = np.char.add(np.char.add(np.char.add(names[tb], ","), (pos_b + 1).astype("U12")), ",+,%s,60,0;" % c2t)
        sa2 = np.char.add(np.char.add(np.char.add(names[ta], ","), (pos_a + 1).astype("U12")), ",+,%s,60,0;" % c1t)
        add(own(ta, pos_a), ta, pos_a, ta, pos_a + 200, 300, 0x1 | 0x2 | 0x40 | 0x20, 60, pid, M(m1), S(m2), sa1)
        add(own(tb, pos_b), tb, pos_b, ta, pos_a + 200, 0, 0x1 | 0x40 | 0x20 | 0x100, 60, pid, S(m1), M(m2), sa2)
        add(own(ta, pos_a + 200), ta, pos_a + 200, ta, pos_a, -300, 0x1 | 0x2 | 0x80 | 0x10, 60, pid, M(100))
    sp = {k: (np.concatenate(v) if v else np.zeros(0, dtype=(np.int64 if k != "sa" else "U48"))) for k, v in cols.items()}
    n_special = len(sp["tid"])
    # proper pairs: both mates inside this rank's range
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1000 + rank)
    P = max(0, (n_per_rank - n_special) // 2)
    tlens = torch.tensor(lens, dtype=torch.int64, device=device)
    tprefix = torch.tensor(prefix, dtype=torch.int64, device=device)
    gs = lo + (torch.rand(P, generator=g, device=device, dtype=torch.float64) * max(1, hi - lo - 1)).to(torch.int64)
    tid = torch.searchsorted(tprefix, gs, right=True) - 1
    pos = gs - tprefix[tid]
    ins = torch.clamp(torch.round(torch.randn(P, generator=g, device=device) * ins_sd + ins_mean).to(torch.int64), min=read_len + 1)
    ub = torch.minimum(tlens[tid], hi - tprefix[tid]) - ins - 1   # keep the mate on the contig and inside the range
    lb = torch.clamp(lo - tprefix[tid], min=0)
    pos = torch.maximum(torch.minimum(pos, ub), lb)
    mpos = pos + ins - read_len
    pid = torch.arange(P, device=device, dtype=torch.int64) + (rank << 34)
    return _assemble(contigs, device, seed, P, tid, pos, mpos, ins, pid, sp, read_len)


def _assemble(contigs, device, seed, P, tid, pos, mpos, ins, pid, sp, read_len):
    """Shared tail of the generators: concatenate proper pairs (device) with the special records (numpy), sort by
    coordinate, build cigar/aux columns."""
    n_special = len(sp["tid"])

    def cat(a, b, special, dt):
        return torch.cat([a.to(dt), b.to(dt), torch.from_numpy(np.ascontiguousarray(special)).to(device).to(dt)])

    c_tid = cat(tid, tid, sp["tid"], torch.int32)
    c_pos = cat(pos, mpos, sp["pos"], torch.int32)
    c_mtid = cat(tid, tid, sp["mtid"], torch.int32)
    c_mpos = cat(mpos, pos, sp["mpos"], torch.int32)
    c_isize = cat(ins, -ins, sp["isize"], torch.int32)
    f1 = torch.full((P,), 0x63, device=device, dtype=torch.int32)
    f2 = torch.full((P,), 0x93, device=device, dtype=torch.int32)
    c_flag = cat(f1, f2, sp["flag"].astype(np.int32), torch.int32)
    m = torch.full((P,), 60, device=device, dtype=torch.uint8)
    c_mapq = cat(m, m, sp["mapq"].astype(np.uint8), torch.uint8)
    salt = ((seed & 0x7FFF) << 48) ^ 0x1E3779B97F4A7C15
    c_pid = cat(pid, pid, sp["pairid"].astype(np.int64), torch.int64)
    c_qh = _mix64(c_pid ^ salt)
    c_qc = _qcheck_of(c_pid, salt)
    del c_pid
    w150 = torch.full((P,), (read_len << 4), device=device, dtype=torch.int64)
    c_c0 = cat(w150, w150, sp["c0"].astype(np.int64), torch.int64)
    zeros = torch.zeros(P, device=device, dtype=torch.int64)
    c_c1 = cat(zeros, zeros, sp["c1"].astype(np.int64), torch.int64)
    n = c_tid.numel()
    key = (c_tid.to(torch.int64) << 32) | c_pos.to(torch.int64)
    perm = torch.sort(key, stable=True)[1]
    del key
    take = lambda t: t[perm].contiguous()
    out = {"tid": take(c_tid), "pos": take(c_pos), "mtid": take(c_mtid), "mpos": take(c_mpos), "isize": take(c_isize)}
    out["flag"] = take(c_flag).to(torch.int16)
    out["mapq"] = take(c_mapq)
    out["qhash"] = take(c_qh)
    out["qcheck"] = take(c_qc)
    c0 = take(c_c0)
    c1 = take(c_c1)
    ncig = 1 + (c1 != 0).to(torch.int64)
    coff = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(ncig, 0, out=coff[1:])
    nwords = int(coff[-1].item()) if n else 0
    cigar = torch.zeros(max(nwords, 1), dtype=torch.int64, device=device)
    if n:
        cigar[coff[:-1]] = c0
        two = (c1 != 0).nonzero().squeeze(1)
        cigar[coff[two] + 1] = c1[two]
    out["cigar_off"] = coff.to(torch.int32)
    out["cigar"] = cigar.to(torch.int32)
    sa = sp["sa"]
    has = np.nonzero(sa != "")[0] if n_special else np.zeros(0, np.int64)
    aux_len_special = np.zeros(n_special, np.int64)
    sab = np.char.encode(sa[has], "ascii") if len(has) else np.zeros(0, "S1")
    lens_sa = np.char.str_len(sab).astype(np.int64) if len(has) else np.zeros(0, np.int64)
    aux_len_special[has] = lens_sa
    aux_len = torch.cat([torch.zeros(2 * P, dtype=torch.int64, device=device), torch.from_numpy(aux_len_special).to(device)])[perm]
    aoff = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(aux_len, 0, out=aoff[1:])
    nbytes = int(aoff[-1].item()) if n else 0
    src_np = (perm[(aux_len > 0).nonzero().squeeze(1)] - 2 * P).cpu().numpy()
    if len(has):
        rank_in_has = np.searchsorted(has, src_np)
        width = int(lens_sa.max())
        mat = np.frombuffer(sab.astype("S%d" % width).tobytes(), dtype=np.uint8).reshape(len(has), width)[rank_in_has]
        blob = mat[np.arange(width)[None, :] < lens_sa[rank_in_has][:, None]]
    else:
        blob = np.zeros(0, np.uint8)
    assert len(blob) == nbytes
    out["aux_off"] = aoff.to(torch.int32)
    out["aux"] = torch.from_numpy(np.ascontiguousarray(blob) if nbytes else np.zeros(1, np.uint8)).to(device)
    out["n"] = n
    out["n_cigar_words"] = nwords
    out["n_aux_bytes"] = nbytes
    out["side"] = side_rows(out)
    return list(contigs), out


# ---------------------------------------------------------------------------------------------------------------
# Targeted-panel shape (BASELINE.json configs[3]): all reads pile up over `n_loci` fusion loci at `depth`x, a large
# share of them split reads (primary + supplementary partner with SA tags) whose clip points scatter +-2 bp around
# the breakpoint (exercises the A15 vote), plus discordant pairs bridging the two sides of every fusion.
def make_panel(seed: int, device, n_loci=500, depth=2000, window=600, split_frac=0.2, disc_frac=0.1, contigs=HG19,
               read_len=150, ins_mean=350.0, ins_sd=40.0):
    rng = np.random.default_rng(seed)
    lens = np.asarray([l for _, l in contigs], dtype=np.int64)
    names = np.asarray([n for n, _ in contigs])
    nt = len(contigs)
    per_locus = depth * window // read_len            # records piled on each side of a locus
    n_split = int(per_locus * split_frac) // 3        # triplets
    n_disc = int(per_locus * disc_frac) // 2          # pairs
    n_proper = max(0, (per_locus - 3 * n_split - 2 * n_disc) // 2)
    cols = {k: [] for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "mapq", "pairid", "c0", "c1", "sa")}

    def add(tid, pos, mtid, mpos, isize, flag, mapq, pairid, c0, c1=None, sa=None):
        n = len(tid)
        b = lambda a: np.full(n, a) if np.isscalar(a) else np.asarray(a)
        cols["tid"].append(b(tid).astype(np.int32)); cols["pos"].append(b(pos).astype(np.int32))
        cols["mtid"].append(b(mtid).astype(np.int32)); cols["mpos"].append(b(mpos).astype(np.int32))
        cols["isize"].append(b(isize).astype(np.int32)); cols["flag"].append(b(flag).astype(np.uint16))
        cols["mapq"].append(b(mapq).astype(np.uint8)); cols["pairid"].append(b(pairid).astype(np.int64))
        cols["c0"].append(b(c0).astype(np.uint32))
        cols["c1"].append(np.zeros(n, np.uint32) if c1 is None else b(c1).astype(np.uint32))
        cols["sa"].append(np.full(n, "", dtype="U48") if sa is None else np.asarray(sa).astype("U48"))

    def sites(k, margin=20000):
        t = rng.integers(0, nt, k)
        p = (rng.random(k) * (lens[t] - 2 * margin)).astype(np.int64) + margin
        return t, p

    la_t, la_p = sites(n_loci)
    lb_t, lb_p = sites(n_loci)
    same = rng.random(n_loci) < 0.3
    lb_t = np.where(same, la_t, lb_t)
    lb_p = np.where(same, (la_p + 50000 + (rng.random(n_loci) * 1e6).astype(np.int64)) % (lens[lb_t] - 40000) + 20000, lb_p)
    rev_a = rng.integers(0, 2, n_loci).astype(bool)
    rev_b = rng.integers(0, 2, n_loci).astype(bool)
    # discordant pairs bridging A and B
    if n_disc:
        li = np.repeat(np.arange(n_loci), n_disc)
        k = len(li)
        ta, tb = la_t[li], lb_t[li]
        pa = la_p[li] - rng.integers(read_len, window // 2 + read_len, k)
        pb = lb_p[li] + rng.integers(0, window // 2, k)
        ra, rb = rev_a[li], rev_b[li]
        mq = np.where(rng.random(k) < 0.03, rng.integers(0, 30, k), 60)
        pid = np.arange(k, dtype=np.int64) + (1 << 40)
        fa = 0x1 | 0x40 | np.where(ra, 0x10, 0) | np.where(rb, 0x20, 0) | np.where(rng.random(k) < 0.02, 0x400, 0)
        fb = 0x1 | 0x80 | np.where(rb, 0x10, 0) | np.where(ra, 0x20, 0)
        isz = np.where(ta == tb, pb - pa + read_len, 0)
        add(ta, pa, tb, pb, isz, fa, mq, pid, (read_len << 4))
        add(tb, pb, ta, pa, -isz, fb, mq, pid, (read_len << 4))
    # split triplets: primary clipped at A, supplementary at B, mate upstream of A
    if n_split:
        li = np.repeat(np.arange(n_loci), n_split)
        k = len(li)
        ta, tb = la_t[li], lb_t[li]
        m1 = rng.integers(35, read_len - 35, k)
        m2 = read_len - m1
        wob = rng.choice(np.asarray([0, 0, 0, 0, 0, 1, -1, 2, -2, 3]), k)
        bpa = la_p[li] + wob
        bpb = lb_p[li] + rng.choice(np.asarray([0, 0, 0, 0, 1, -1, 2]), k)
        pos_a, pos_b = bpa - m1, bpb
        mate = pos_a - rng.integers(60, 250, k)
        pid = np.arange(k, dtype=np.int64) + (3 << 40)
        cig = lambda a, ca, b, cb: np.char.add(np.char.add(a.astype("U4"), ca), np.char.add(b.astype("U4"), cb))
        c1t, c2t = cig(m1, "M", m2, "S"), cig(m1, "S", m2, "M")
        mqs = np.where(rng.random(k) < 0.05, rng.integers(0, 25, k), 60)
        head = lambda nm, p: np.char.add(np.char.add(nm, ","), np.char.add((p + 1).astype("U12"), ",+,"))
        sa1 = np.char.add(np.char.add(head(names[tb], pos_b), c2t), np.char.add(np.char.add(",", mqs.astype("U3")), ",0;"))
        sa2 = np.char.add(np.char.add(head(names[ta], pos_a), c1t), ",60,0;")
        add(ta, pos_a, ta, mate, -(pos_a + m1 - mate), 0x1 | 0x2 | 0x80 | 0x10, 60, pid, (m1 << 4), (m2 << 4) | 4, sa1)
        part = np.where(rng.random(k) < 0.85, 0x100, 0x800)   # bwa mem -M marks the partner 0x100; 0x800 never votes
        add(tb, pos_b, ta, mate, 0, 0x1 | 0x80 | 0x10 | part, mqs, pid, (m1 << 4) | 4, (m2 << 4), sa2)
        add(ta, mate, ta, pos_a, pos_a + m1 - mate, 0x1 | 0x2 | 0x40 | 0x20, 60, pid, (read_len << 4))
    sp = {k: (np.concatenate(v) if v else np.zeros(0, dtype=(np.int64 if k != "sa" else "U48"))) for k, v in cols.items()}
    # proper pairs piled over both sides of each locus
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    P = 2 * n_loci * n_proper
    side_t = torch.from_numpy(np.concatenate([la_t, lb_t])).to(device)
    side_p = torch.from_numpy(np.concatenate([la_p, lb_p])).to(device)
    si = torch.arange(P, device=device, dtype=torch.int64) // max(n_proper, 1)
    tid = side_t[si] if P else torch.zeros(0, dtype=torch.int64, device=device)
    ins = torch.clamp(torch.round(torch.randn(P, generator=g, device=device) * ins_sd + ins_mean).to(torch.int64), min=read_len + 1)
    off = (torch.rand(P, generator=g, device=device) * (window + 400)).to(torch.int64) - (window + 400) // 2 - 175
    pos = (side_p[si] + off) if P else torch.zeros(0, dtype=torch.int64, device=device)
    mpos = pos + ins - read_len
    pid = torch.arange(P, device=device, dtype=torch.int64)
    return _assemble(contigs, device, seed, P, tid, pos, mpos, ins, pid, sp, read_len)
