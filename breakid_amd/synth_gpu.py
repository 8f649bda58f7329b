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
    c_qh = _mix64(cat(pid, pid, sp["pairid"], torch.int64) ^ salt)
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
    del c_flag, c_mapq, c_qh
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
    return list(contigs), out


def to_numpy_cols(cols):
    """Device table -> host numpy dict in the abi.SOA_COLS layout (for the oracle / cpu_baseline leg)."""
    view = {"flag": np.uint16, "qhash": np.uint64, "cigar_off": np.uint32, "cigar": np.uint32, "aux_off": np.uint32}
    out = {}
    for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "mapq", "qhash", "cigar_off", "cigar", "aux_off", "aux"):
        a = cols[k].cpu().numpy()
        out[k] = a.view(view[k]) if k in view else a
    out["cigar"] = out["cigar"][: cols["n_cigar_words"]]
    out["aux"] = out["aux"][: cols["n_aux_bytes"]]
    return out
