"""Minimal BGZF/BAM *writer* used by the synthetic-input generator and the tests.

This is tooling (fixtures, bench inputs), not the product's feed path: the product decodes BAM in
C++ (breakid_amd/csrc/bam_reader.cc).  Format follows the SAM/BAM specification; the fields the
reference consumes are the ones listed at /root/reference/thirdparty/.../htslib/sam.h:148-181.
"""
from __future__ import annotations

import struct
import zlib
from typing import Iterable, List, Optional, Sequence, Tuple

CIGAR_OPS = "MIDNSHP=X"
_OP_CODE = {c: i for i, c in enumerate(CIGAR_OPS)}

_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def parse_cigar(text: str) -> List[int]:
    """'60M40S' -> BAM words (len<<4|op)."""
    if text in ("", "*"):
        return []
    out, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | _OP_CODE[ch])
            num = ""
    return out


def cigar_reflen(words: Sequence[int]) -> int:
    n = 0
    for w in words:
        if (w & 15) in (0, 2, 3, 7, 8):
            n += w >> 4
    return n


def reg2bin(beg: int, end: int) -> int:
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


class BgzfWriter:
    def __init__(self, path: str, level: int = 1):
        self.f = open(path, "wb")
        self.buf = bytearray()
        self.level = level

    def write(self, data: bytes) -> None:
        self.buf += data
        while len(self.buf) >= 0xFF00:
            self._flush_block(bytes(self.buf[:0xFF00]))
            del self.buf[:0xFF00]

    def flush(self) -> None:
        """ends the current block (htslib's bgzf_flush)"""
        if self.buf:
            self._flush_block(bytes(self.buf))
            self.buf.clear()

    def write_record(self, rec: bytes) -> None:
        """htslib's bam_write1: bgzf_flush_try first, so that no record straddles two blocks"""
        if len(self.buf) + len(rec) > 0xFF00:
            self.flush()
        self.write(rec)

    def _flush_block(self, data: bytes) -> None:
        c = zlib.compressobj(self.level, zlib.DEFLATED, -15)
        comp = c.compress(data) + c.flush()
        bsize = len(comp) + 25
        hdr = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, bsize)
        self.f.write(hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))

    def close(self) -> None:
        if self.buf:
            self._flush_block(bytes(self.buf))
            self.buf.clear()
        self.f.write(_BGZF_EOF)
        self.f.close()


def encode_record(qname: str, flag: int, tid: int, pos: int, mapq: int, cigar: Sequence[int],
                  mtid: int, mpos: int, isize: int, aux: Sequence[Tuple[str, str]] = (),
                  seq_len: int = 0) -> bytes:
    """One BAM alignment record (pos/mpos 0-based).  aux = [(tag, string)] written as Z."""
    name = qname.encode() + b"\0"
    if flag & 4 or not cigar:
        end = pos + 1
    else:
        end = pos + cigar_reflen(cigar)
    b = reg2bin(max(pos, 0), max(end, pos + 1) if pos >= 0 else 1) if pos >= 0 else 4680
    body = struct.pack("<iiBBHHHIiii", tid, pos, len(name), mapq, b, len(cigar), flag, seq_len, mtid, mpos, isize)
    body += name
    body += struct.pack("<%dI" % len(cigar), *cigar) if cigar else b""
    if seq_len:
        body += b"\x11" * ((seq_len + 1) // 2) + b"\x1e" * seq_len
    for tag, val in aux:
        body += tag.encode() + b"Z" + val.encode() + b"\0"
    return struct.pack("<i", len(body)) + body


def write_bam(path: str, contigs: Sequence[Tuple[str, int]], records: Iterable[bytes],
              header_text: Optional[str] = None, aligned: bool = False) -> None:
    """records: iterable of encode_record() bytes, already coordinate sorted.  aligned=True writes blocks the way
    htslib does (header flushed, no record across a block boundary); the default cuts blocks at fixed 0xFF00 bytes."""
    if header_text is None:
        header_text = "@HD\tVN:1.4\tSO:coordinate\n" + "".join(
            "@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in contigs)
    w = BgzfWriter(path)
    t = header_text.encode()
    hdr = b"BAM\1" + struct.pack("<i", len(t)) + t + struct.pack("<i", len(contigs))
    for n, l in contigs:
        nb = n.encode() + b"\0"
        hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    w.write(hdr)
    if aligned:
        w.flush()
        for r in records:
            w.write_record(r)
        w.close()
        return
    chunk = bytearray()
    for r in records:
        chunk += r
        if len(chunk) > (1 << 20):
            w.write(bytes(chunk))
            chunk.clear()
    if chunk:
        w.write(bytes(chunk))
    w.close()


def write_nib(path: str, seq: str) -> None:
    """UCSC .nib as read by /root/reference/src/nibtools.cc:18-58 (T=0 C=1 A=2 G=3 N=4, high nibble first)."""
    code = {"T": 0, "C": 1, "A": 2, "G": 3, "N": 4}
    n = len(seq)
    out = bytearray(struct.pack("<II", 0x6BE93D3A, n))
    vals = [code.get(c, 4) for c in seq.upper()]
    if n & 1:
        vals.append(0)
    for i in range(0, len(vals), 2):
        out.append((vals[i] << 4) | vals[i + 1])
    with open(path, "wb") as f:
        f.write(bytes(out))


def write_bam_from_soa(path: str, contigs: Sequence[Tuple[str, int]], cols, qnames: Sequence[bytes], aligned: bool = True) -> None:
    """Columnar table (numpy dict, abi.SOA_COLS layout) -> coordinate-sorted BAM; qnames[i] is record i's read name.
    The aux blob becomes SA:Z (and OC:Z when the blob is 'OC \\t SA').  Tooling for fixtures: this is how tables made by
    breakid_amd.synth_gpu reach the reference binary."""
    tid, pos, mtid, mpos, isize = (cols[k].tolist() for k in ("tid", "pos", "mtid", "mpos", "isize"))
    flag, mapq = cols["flag"].tolist(), cols["mapq"].tolist()
    coff, aoff = cols["cigar_off"].tolist(), cols["aux_off"].tolist()
    cig = cols["cigar"].tolist()
    aux = cols["aux"].tobytes()
    pack_core = struct.Struct("<iiiBBHHHIiii").pack

    def gen():
        for i in range(len(tid)):
            c = cig[coff[i]:coff[i + 1]]
            p = pos[i]
            if flag[i] & 4 or not c:
                end = p + 1
            else:
                end = p + cigar_reflen(c)
            b = reg2bin(p, max(end, p + 1)) if p >= 0 else 4680
            name = qnames[i] + b"\0"
            body = name
            if c:
                body += struct.pack("<%dI" % len(c), *c)
            a = aux[aoff[i]:aoff[i + 1]]
            if a:
                if b"\t" in a:
                    oc, a = a.split(b"\t", 1)
                    body += b"SAZ" + a + b"\0OCZ" + oc + b"\0"
                else:
                    body += b"SAZ" + a + b"\0"
            yield pack_core(32 + len(body), tid[i], p, len(name), mapq[i], b, len(c), flag[i], 0, mtid[i], mpos[i], isize[i]) + body
    write_bam(path, contigs, gen(), aligned=aligned)


def write_bam_from_soa_fast(path: str, contigs: Sequence[Tuple[str, int]], cols, names, threads: int = 0, chunk: int = 4_000_000) -> None:
    """write_bam_from_soa(aligned=True) for tables of 10^7..10^8 records: the same bytes (same record layout, the same greedy
    block cuts as htslib's bam_write1 / bgzf_flush_try, the same zlib calls per block), assembled with numpy and deflated on
    `threads` threads.  `names` is an (n, L) uint8 array of fixed-length read names (synth.name_records)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    n = len(cols["tid"])
    L = names.shape[1]
    coff = np.asarray(cols["cigar_off"], np.int64)
    aoff = np.asarray(cols["aux_off"], np.int64)
    cig = np.asarray(cols["cigar"], np.uint32)
    aux = np.asarray(cols["aux"], np.uint8)
    threads = threads or min(16, os.cpu_count() or 1)
    w = BgzfWriter(path)
    t = ("@HD\tVN:1.4\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (nm, l) for nm, l in contigs)).encode()
    hdr = b"BAM\1" + struct.pack("<i", len(t)) + t + struct.pack("<i", len(contigs))
    for nm, l in contigs:
        nb = nm.encode() + b"\0"
        hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    w.write(hdr)
    w.flush()
    f = w.f

    def deflate(data: bytes) -> bytes:
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        comp = c.compress(data) + c.flush()
        return (struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, len(comp) + 25) + comp
                + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))

    carry = b""  # the open block of the chunk before (bgzf keeps it until a record does not fit)
    reflen_op = np.zeros(16, bool)
    reflen_op[[0, 2, 3, 7, 8]] = True
    with ThreadPoolExecutor(threads) as pool:
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            m = hi - lo
            nc = (coff[lo + 1:hi + 1] - coff[lo:hi])
            al = (aoff[lo + 1:hi + 1] - aoff[lo:hi])
            # aux bytes as BAM writes them: SA:Z:<sa>\0 (and OC:Z:<oc>\0 behind it when the blob is 'OC \t SA')
            has = np.nonzero(al)[0]
            blob_parts, blob_len = [], np.zeros(m, np.int64)
            for i in has.tolist():
                a = aux[aoff[lo + i]:aoff[lo + i + 1]].tobytes()
                if b"\t" in a:
                    oc, a = a.split(b"\t", 1)
                    b = b"SAZ" + a + b"\0OCZ" + oc + b"\0"
                else:
                    b = b"SAZ" + a + b"\0"
                blob_parts.append(b)
                blob_len[i] = len(b)
            rec = 36 + (L + 1) + 4 * nc + blob_len          # bytes of a record incl. its block_size word
            off = np.zeros(m + 1, np.int64)
            np.cumsum(rec, out=off[1:])
            buf = np.zeros(int(off[-1]), np.uint8)
            pos = np.asarray(cols["pos"][lo:hi], np.int64)
            flag = np.asarray(cols["flag"][lo:hi], np.int64)
            words = cig[coff[lo]:coff[hi]]
            owner = np.repeat(np.arange(m), nc)
            rl = np.bincount(owner, weights=np.where(reflen_op[words & 15], words >> 4, 0), minlength=m).astype(np.int64)
            end = np.where(((flag & 4) != 0) | (nc == 0), pos + 1, pos + rl)
            end = np.maximum(end, pos + 1) - 1
            beg = np.maximum(pos, 0)
            bn = np.zeros(m, np.int64)
            done = np.zeros(m, bool)
            for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
                hit = ~done & ((beg >> shift) == (end >> shift))
                bn[hit] = base + (beg[hit] >> shift)
                done |= hit
            bn[pos < 0] = 4680
            core = np.zeros(m, np.dtype([("bs", "<i4"), ("tid", "<i4"), ("pos", "<i4"), ("l_name", "u1"), ("mapq", "u1"), ("bin", "<u2"), ("ncig", "<u2"),
                                         ("flag", "<u2"), ("lseq", "<u4"), ("mtid", "<i4"), ("mpos", "<i4"), ("isize", "<i4")]))
            core["bs"] = rec - 4
            for k in ("tid", "pos", "mtid", "mpos", "isize", "mapq", "flag"):
                core[k] = cols[k][lo:hi]
            core["l_name"], core["bin"], core["ncig"] = L + 1, bn, nc
            fixed = np.zeros((m, 36 + L + 1), np.uint8)
            fixed[:, :36] = core.view(np.uint8).reshape(m, 36)
            fixed[:, 36:36 + L] = names[lo:hi]
            idx = off[:-1, None] + np.arange(36 + L + 1)[None, :]
            buf[idx.ravel()] = fixed.ravel()
            del idx, fixed
            if len(words):
                wstart = off[:-1] + 36 + L + 1
                within = np.arange(len(words)) - np.repeat(coff[lo:hi] - coff[lo], nc)
                wpos = np.repeat(wstart, nc) + 4 * within
                wb = words.astype("<u4").view(np.uint8).reshape(-1, 4)
                for j in range(4):
                    buf[wpos + j] = wb[:, j]
            for i, b in zip(has.tolist(), blob_parts):
                s = int(off[i]) + 36 + L + 1 + 4 * int(nc[i])
                buf[s:s + len(b)] = np.frombuffer(b, np.uint8)
            assert int(rec.max()) <= 0xFF00, "record longer than a BGZF block: use write_bam_from_soa"
            # greedy cuts: a record that does not fit the open block closes it (bam_write1 -> bgzf_flush_try)
            data = memoryview(buf)
            blocks, start = [], 0
            while True:
                r = int(np.searchsorted(off, off[start] + 0xFF00 - len(carry), side="right")) - 1
                if r >= m:
                    carry += bytes(data[off[start]:off[m]])
                    break
                blocks.append(carry + bytes(data[off[start]:off[r]]))
                carry, start = b"", r
            for comp in pool.map(deflate, blocks):
                f.write(comp)
    if carry:
        f.write(deflate(carry))
    f.write(_BGZF_EOF)
    f.close()


def write_bai(bam_path: str, bai_path: Optional[str] = None) -> str:
    """Index of a coordinate-sorted BAM in the BAI format of the SAM specification (5.2: bins with their chunk lists, the 16 kb
    linear index, n_no_coor), made by reading the file back: what `samtools index` leaves next to it.  The reference refuses to
    call breakpoints without a loadable index (BreakID.cc:411-416), so the command-line tests give it a real one."""
    raw = open(bam_path, "rb").read()
    # BGZF blocks -> one inflated stream + the file offset / inflated offset of every block
    blocks, data, off = [], bytearray(), 0
    while off < len(raw):
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        blocks.append((off, len(data)))
        data += zlib.decompress(raw[off + 18:off + bsize - 8], -15)
        off += bsize
    starts = [b[1] for b in blocks]

    import bisect

    def voffset(p: int) -> int:
        i = bisect.bisect_right(starts, p) - 1
        # a position at the very end of a block belongs to the start of the next one, as htslib's bgzf_tell reports it
        while i + 1 < len(blocks) and blocks[i + 1][1] == p and p > blocks[i][1]:
            i += 1
        return (blocks[i][0] << 16) | (p - blocks[i][1])

    l_text = struct.unpack_from("<i", data, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]
    p += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, p)[0]
        p += 8 + l_name
    bins = [dict() for _ in range(n_ref)]
    lin = [dict() for _ in range(n_ref)]
    n_no_coor = 0
    while p + 4 <= len(data):
        bs = struct.unpack_from("<i", data, p)[0]
        tid, pos, l_name, _mq, _bin, n_cig, flag = struct.unpack_from("<iiBBHHH", data, p + 4)
        beg_v, end_v = voffset(p), voffset(p + 4 + bs)
        if tid < 0:
            n_no_coor += 1
        else:
            cig = struct.unpack_from("<%dI" % n_cig, data, p + 36 + l_name) if n_cig else ()
            rl = cigar_reflen(cig) if cig and not flag & 4 else 0
            end = pos + (rl if rl > 0 else 1)
            b = reg2bin(pos, end)
            ch = bins[tid].setdefault(b, [])
            if ch and ch[-1][1] == beg_v:
                ch[-1][1] = end_v
            else:
                ch.append([beg_v, end_v])
            for wdw in range(pos >> 14, ((end - 1) >> 14) + 1):
                lin[tid].setdefault(wdw, beg_v)
        p += 4 + bs
    out = bytearray(b"BAI\1" + struct.pack("<i", n_ref))
    for t in range(n_ref):
        out += struct.pack("<i", len(bins[t]))
        for b in sorted(bins[t]):
            out += struct.pack("<Ii", b, len(bins[t][b]))
            for c in bins[t][b]:
                out += struct.pack("<QQ", c[0], c[1])
        n_intv = (max(lin[t]) + 1) if lin[t] else 0
        out += struct.pack("<i", n_intv)
        last = 0
        for wdw in range(n_intv):
            last = lin[t].get(wdw, last)
            out += struct.pack("<Q", last)
    out += struct.pack("<Q", n_no_coor)
    bai_path = bai_path or bam_path + ".bai"
    with open(bai_path, "wb") as f:
        f.write(out)
    return bai_path
