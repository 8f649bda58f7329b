"""GPU feed (bk_bam_decode_device: BGZF inflate + BAM record decode on the device) against the host decoder and the
generator; the inflate kernel alone against zlib."""
import ctypes as C
import os
import struct
import tempfile
import zlib

import numpy as np
import pytest

from breakid_amd import abi, capi, synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def _device_cols(table):
    import torch
    from breakid_amd.sharded import tensor_from_ptr
    dev = torch.device("cuda", 0)
    s = table.soa
    n = s.n
    sizes = {"cigar_off": n + 1, "aux_off": n + 1, "cigar": s.n_cigar_words, "aux": s.n_aux_bytes}
    out = {}
    for name, dt in abi.SOA_COLS_ALL:
        cnt = sizes.get(name, n)
        nb = cnt * np.dtype(dt).itemsize
        out[name] = tensor_from_ptr(getattr(s, name), nb, dev).cpu().numpy().view(dt).copy() if nb else np.zeros(0, dt)
    return out


def _dataset():
    contigs = [("chr1", 3_000_000), ("chr2", 2_000_000), ("chrX", 900_000)]
    ds = synth.make_cfg(9, contigs, 60_000, 40, 30, 300, jitter=200, read_len=100)
    for i in range(0, len(ds.recs), 311):
        ds.recs[i].sa = "chr2,%d,+,40S60M,60,0;" % (100 + i)
        if i % 2:
            ds.recs[i].oc = "60M40S"
    return contigs, ds


def test_device_decode_matches_generator_and_pipeline():
    contigs, ds = _dataset()
    ref = ds.to_soa()
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "a.bam")
        ds.write_bam(p, aligned=True)   # blocks as htslib writes them
        table = capi.decode_bam_device(p)
        host_contigs, host_cols = capi.decode_bam(p)
    assert table.contigs == contigs == host_contigs
    got = _device_cols(table)
    for k, _ in abi.SOA_COLS_ALL:
        assert np.array_equal(got[k], ref[k]), k
        assert np.array_equal(got[k], host_cols[k]), k
    # the device table feeds the pipeline in place
    ctx = capi.Context(contigs)
    ctx.attach_device_table(table)
    w, nv = ctx.run(qual=20, fast=True)
    o = pyoracle.Oracle(contigs, ref)
    ow, rc = o.run(20, fast=True)
    a, _ = ctx.fetch(abi.STAGE_CLUSTERS)
    b, _ = o.fetch(abi.STAGE_CLUSTERS)
    assert rc == 0 and w == ow and np.array_equal(a, b) and len(a) > 0
    ctx.close()
    o.close()
    table.close()


@pytest.fixture(params=["chunks", "batch"])
def packed_variant(request, monkeypatch):
    """files whose records run across BGZF blocks are decoded in chunks (the default: a chunk starts with the record carried
    over from the one before) or, BREAKID_FEED_PACKED_BATCH=1, as one batch (also what the chunked variant falls back to when a
    record longer than 8 MiB crosses a chunk)"""
    if request.param == "batch":
        monkeypatch.setenv("BREAKID_FEED_PACKED_BATCH", "1")
    return request.param


def test_device_decode_of_records_across_blocks_and_of_the_golden_shapes(packed_variant):
    contigs, ds = _dataset()
    ref = ds.to_soa()
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "u.bam")
        ds.write_bam(p)                 # fixed-size blocks: records straddle them (htsjdk / Picard style)
        table = capi.decode_bam_device(p)   # record boundaries guessed per block, verified to chain
        got = _device_cols(table)
        assert table.contigs == contigs
        for k, _ in abi.SOA_COLS_ALL:
            assert np.array_equal(got[k], ref[k]), k
        table.close()
        # a cut in the middle of the stream / a file that is no BAM: an error, not a table
        data = open(p, "rb").read()
        open(p, "wb").write(data[: len(data) // 2])
        with pytest.raises(capi.BreakIDError):
            capi.decode_bam_device(p)
        open(p, "wb").write(b"not a bam")
        with pytest.raises(capi.BreakIDError):
            capi.decode_bam_device(p)
    # the golden generators (join quirks, OC tags, clips, SA variants), written with aligned blocks at another zlib level
    for make in (synth.make_g1, synth.make_edge):
        g = make()
        ref = g.to_soa()
        with tempfile.TemporaryDirectory() as t:
            up = os.path.join(t, "u.bam")
            g.write_bam(up)
            ut = capi.decode_bam_device(up)     # records across blocks
            ugot = _device_cols(ut)
            for k, _ in abi.SOA_COLS_ALL:
                assert np.array_equal(ugot[k], ref[k]), (make.__name__, "across blocks", k)
            ut.close()
            raw = b"".join(_inflate_blocks(open(up, "rb").read()))
            ap = os.path.join(t, "a.bam")
            _rewrite_aligned(raw, ap)
            table = capi.decode_bam_device(ap)
            got = _device_cols(table)
            assert table.contigs == g.contigs
            for k, _ in abi.SOA_COLS_ALL:
                assert np.array_equal(got[k], ref[k]), (make.__name__, k)
            table.close()


def _inflate_blocks(data):
    off = 0
    while off < len(data):
        xlen = struct.unpack_from("<H", data, off + 10)[0]
        bsize = struct.unpack_from("<H", data, off + 16)[0]
        yield zlib.decompress(data[off + 12 + xlen: off + bsize + 1 - 8], -15)
        off += bsize + 1


def _rewrite_aligned(raw, path):
    """re-blocks an inflated BAM stream the way htslib does"""
    from breakid_amd import bamio
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, p)[0]
        p += 4 + l_name + 4
    w = bamio.BgzfWriter(path, level=6)
    w.write(raw[:p])
    w.flush()
    while p < len(raw):
        bs = struct.unpack_from("<i", raw, p)[0]
        w.write_record(raw[p:p + 4 + bs])
        p += 4 + bs
    w.close()


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_gpu_inflate_matches_zlib(level):
    rng = np.random.default_rng(level)
    # a mix that exercises literals, short and long matches, far matches (> 16 KiB back), runs (distance 1), stored blocks
    parts = [rng.integers(0, 256, 40_000, dtype=np.uint8).tobytes(), b"ACGT" * 30_000, bytes(70_000), rng.integers(65, 70, 200_000, dtype=np.uint8).tobytes()]
    big = rng.integers(0, 256, 20_000, dtype=np.uint8).tobytes()
    parts += [big, rng.integers(0, 4, 9_000, dtype=np.uint8).tobytes(), big, big[:777] * 50]
    raw = b"".join(parts)
    blocks = []
    for off in range(0, len(raw), 0xFF00):
        blk = raw[off:off + 0xFF00]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(blk) + c.flush()
        blocks.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(blk), len(blk)))
    blocks.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    data = b"".join(blocks)
    L = capi.lib()
    import torch  # noqa: F401
    src = np.frombuffer(data, np.uint8)
    out = np.zeros(len(raw) + 16, np.uint8)
    olen, ms, err = C.c_uint64(), C.c_float(), C.create_string_buffer(256)
    rc = L.bk_debug_bgzf_inflate(src.ctypes.data, len(data), out.ctypes.data, len(out), C.byref(olen), C.byref(ms), err, 256)
    assert rc == 0, err.value
    assert olen.value == len(raw) and out[:len(raw)].tobytes() == raw
    # a corrupted stream is reported, not mis-decoded silently
    bad = bytearray(data)
    bad[40] ^= 0x55
    src2 = np.frombuffer(bytes(bad), np.uint8)
    rc = L.bk_debug_bgzf_inflate(src2.ctypes.data, len(bad), out.ctypes.data, len(out), C.byref(olen), C.byref(ms), err, 256)
    assert rc != 0 or out[:len(raw)].tobytes() != raw


def _bgzf(blocks_raw, compress):
    out = []
    for blk in blocks_raw:
        comp = compress(blk)
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(blk), len(blk)))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(out)


def _gpu_inflate(data, n_out):
    L = capi.lib()
    import torch  # noqa: F401
    src = np.frombuffer(data, np.uint8)
    out = np.zeros(n_out + 16, np.uint8)
    olen, ms, err = C.c_uint64(), C.c_float(), C.create_string_buffer(256)
    rc = L.bk_debug_bgzf_inflate(src.ctypes.data, len(data), out.ctypes.data, len(out), C.byref(olen), C.byref(ms), err, 256)
    return rc, olen.value, out[:n_out].tobytes(), err.value


def test_gpu_inflate_block_kinds_and_code_shapes():
    """deflate streams the BAM writers do not usually produce: fixed-Huffman blocks, several deflate blocks per BGZF
    block (sync flushes, a stored block between two compressed ones), skewed alphabets whose rare symbols get 12-15 bit
    codes, maximal-length matches and distance-1 runs across the whole block, one-byte and empty blocks"""
    rng = np.random.default_rng(5)
    # geometric byte frequencies: code lengths up to 15 for the tail of the alphabet
    p = 0.5 ** (np.arange(256) / 6.0)
    skew = rng.choice(256, 0xFF00, p=p / p.sum()).astype(np.uint8).tobytes()
    texty = (b"chr1\t100\t+\t60M40S\t60\t0;" * 3000)[:0xFF00]
    runs = bytes(0xFF00)
    mixed = rng.integers(0, 256, 3000, dtype=np.uint8).tobytes() + b"A" * 20000 + rng.integers(0, 4, 30000, dtype=np.uint8).tobytes()
    for name, blks in (("skewed", [skew]), ("text", [texty]), ("runs", [runs]), ("mixed", [mixed, skew[:777], b"x", b"", texty[:5000]])):
        raw = b"".join(blks)
        # (a) one dynamic block per BGZF block, best compression
        def dyn(b):
            c = zlib.compressobj(9, zlib.DEFLATED, -15)
            return c.compress(b) + c.flush()
        # (b) fixed Huffman codes only
        def fixed(b):
            c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
            return c.compress(b) + c.flush()
        # (c) several deflate blocks in one BGZF block: sync flushes (each adds an empty stored block) and a level-0 piece
        def multi(b):
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            out = b""
            step = max(1, len(b) // 5)
            for i in range(0, len(b), step):
                out += c.compress(b[i:i + step]) + c.flush(zlib.Z_SYNC_FLUSH if (i // step) % 2 else zlib.Z_FULL_FLUSH)
            return out + c.flush()
        # (d) Huffman only (no matches) and RLE (distance 1 only)
        def huff(b):
            c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_HUFFMAN_ONLY)
            return c.compress(b) + c.flush()
        def rle(b):
            c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_RLE)
            return c.compress(b) + c.flush()
        for kind, fn in (("dyn", dyn), ("fixed", fixed), ("multi", multi), ("huff", huff), ("rle", rle)):
            data = _bgzf(blks, fn)
            rc, n, got, err = _gpu_inflate(data, len(raw))
            assert rc == 0 and n == len(raw) and got == raw, (name, kind, rc, err)
    # truncated / corrupted streams: flagged, never a hang or a wrong "success"
    data = bytearray(_bgzf([texty], lambda b: zlib.compress(b, 6)[2:-4]))
    for at in (30, 100, len(data) // 2):
        bad = bytearray(data)
        bad[at] ^= 0xA5
        rc, n, got, err = _gpu_inflate(bytes(bad), len(texty))
        assert rc != 0 or n == len(texty)   # (a flipped bit may still be a valid stream of the same length)


def test_gpu_inflate_streams_that_never_synchronise_and_corrupted_ones():
    """The decoder walks 64 parts of a Huffman block from guessed bits and relies on the streams' self-synchronisation
    (bgzf_gpu.hip, lanes_block); where every code has the same length a wrong guess stays wrong for ever and the block is
    handed to the decoder that takes 64 bit positions per round.  Uniform random bytes without matches (8-bit codes only),
    two-letter alphabets (1-bit codes: 1024 symbols per part), and both glued to ordinary text inside one BGZF block; then 80
    corrupted copies of a file: an error or the right length, never a hang."""
    rng = np.random.default_rng(11)
    def huff(b):
        c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_HUFFMAN_ONLY)
        return c.compress(b) + c.flush()
    def dyn(b):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        return c.compress(b) + c.flush()
    flat = rng.integers(0, 256, 0xFF00, dtype=np.uint8).tobytes()
    two = rng.integers(0, 2, 0xFF00, dtype=np.uint8).tobytes()
    text = (b"read_%08d\tchr7\t55249071\t60\t151M\t=\t55249300\t380\n" * 1200)[:30000]
    for name, blks, fn in (("flat", [flat, flat[:1000], flat[:70]], huff), ("two", [two, two[:4097]], huff), ("glued", [text + flat[:20000] + two[:15000], flat[:5000] + text], dyn),
                           ("glued-huff", [text[:20000] + flat[:20000] + text[:20000]], huff)):
        raw = b"".join(blks)
        rc, n, got, err = _gpu_inflate(_bgzf(blks, fn), len(raw))
        assert rc == 0 and n == len(raw) and got == raw, (name, rc, err)
    blks = [text + two[:9000], flat[:3000] + text[:20000], text]
    raw = b"".join(blks)
    data = _bgzf(blks, dyn)
    for it in range(80):
        bad = bytearray(data)
        at = int(rng.integers(18, len(data) - 40))
        bad[at] ^= int(rng.integers(1, 256))
        rc, n, got, err = _gpu_inflate(bytes(bad), len(raw))
        assert rc != 0 or n == len(raw), (it, at)


def test_device_decode_in_chunks_equals_one_chunk():
    """the streaming feed (bk_bam_decode_device takes the file in chunks; three in flight) with chunks of a few blocks:
    same table, including the growth of the columns when the first chunk under-estimates the rest"""
    contigs, ds = _dataset()
    # denser tail: the first chunk's records-per-byte under-estimates the file
    for i in range(len(ds.recs) // 2, len(ds.recs), 3):
        ds.recs[i].sa = "chr1,%d,-,30M70S,50,1;" % (5 + i)
    ref = ds.to_soa()
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "a.bam")
        ds.write_bam(p, aligned=True)
        size = os.path.getsize(p)
        # chunk size: the default, a fifth and a ninth of the file (more than three chunks: the staging pool runs), a few blocks per chunk
        for mb in (None, size / 5 / 1048576.0, size / 9 / 1048576.0, 0.07):
            env = {"BREAKID_FEED_CHUNK_MB": None if mb is None else repr(mb)}
            for k, v in env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            try:
                table = capi.decode_bam_device(p)
            finally:
                for k in env:
                    os.environ.pop(k, None)
            got = _device_cols(table)
            assert table.contigs == contigs
            for k, _ in abi.SOA_COLS_ALL:
                assert np.array_equal(got[k], ref[k]), (mb, k)
            table.close()


def test_feed_keeps_its_buffers_between_files_and_gives_them_back():
    """the second and third file of a process go through the staging buffers, copy threads and slots the first one left
    (every chunk staged, header hops on the producer side): same table; bk_feed_release_caches() gives everything back and
    the next file starts from nothing again: same table; both layouts (records inside / across BGZF blocks)"""
    contigs, ds = _dataset()
    ref = ds.to_soa()
    with tempfile.TemporaryDirectory() as t:
        for aligned in (True, False):
            p = os.path.join(t, "a%d.bam" % aligned)
            ds.write_bam(p, aligned=aligned)
            os.environ["BREAKID_FEED_CHUNK_MB"] = repr(os.path.getsize(p) / 7 / 1048576.0)  # more than three chunks: the staging pool runs
            try:
                for rep in range(5):
                    if rep == 3:
                        capi.lib().bk_feed_release_caches()
                    table = capi.decode_bam_device(p)
                    got = _device_cols(table)
                    assert table.contigs == contigs
                    for k, _ in abi.SOA_COLS_ALL:
                        assert np.array_equal(got[k], ref[k]), (aligned, rep, k)
                    table.close()
            finally:
                os.environ.pop("BREAKID_FEED_CHUNK_MB", None)
    capi.lib().bk_feed_release_caches()


@pytest.mark.parametrize("parts", [2, 3, 7, 64])
def test_parts_of_a_file_tile_its_record_table(parts):
    """bk_bam_decode_device_part: the blocks that start in the k-th part of the file's bytes, for every k - concatenated in order
    they are the table of the whole file (what the ranks of a sharded run decode on their own GPUs); more parts than the file has
    blocks leave some of them empty.  A file whose records run across blocks (round 4): a part takes the records that START in its
    blocks, every part guesses its first boundary and verifies its chain into the next part's first block"""
    contigs, ds = _dataset()
    ref = ds.to_soa()
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "a.bam")
        ds.write_bam(p, aligned=True)
        os.environ["BREAKID_FEED_CHUNK_MB"] = repr(os.path.getsize(p) / 11 / 1048576.0)  # several chunks per part, parts that start inside a chunk
        try:
            tables = [capi.decode_bam_device_part(p, k, parts) for k in range(parts)]
        finally:
            os.environ.pop("BREAKID_FEED_CHUNK_MB", None)
        def check(tables):
            got = [_device_cols(tb) for tb in tables]
            assert all(tb.contigs == contigs for tb in tables)
            assert sum(tb.soa.n for tb in tables) == len(ds.recs)
            assert sum(1 for tb in tables if tb.soa.n) >= min(parts, 2)
            for k, _ in abi.SOA_COLS_ALL:
                if k in ("cigar_off", "aux_off"):
                    # offsets restart with every part: compare the lengths they describe
                    lens = np.concatenate([np.diff(g[k].astype(np.int64)) for g in got])
                    assert np.array_equal(lens, np.diff(ref[k].astype(np.int64))), k
                else:
                    assert np.array_equal(np.concatenate([g[k] for g in got]), ref[k]), k
            for tb in tables:
                tb.close()
        check(tables)
        q = os.path.join(t, "b.bam")
        ds.write_bam(q, aligned=False)
        check([capi.decode_bam_device_part(q, k, parts) for k in range(parts)])
        with pytest.raises(capi.BreakIDError) as e:
            capi.decode_bam_device_part(p, 2, 2)
        assert e.value.code == abi.BK_ERR_ARG


def test_device_decode_of_records_longer_than_a_block(packed_variant):
    """long reads: a record spans several BGZF blocks, some blocks hold no record start at all"""
    from breakid_amd import bamio
    contigs = [("chr1", 3_000_000), ("chr2", 2_000_000)]
    ds = synth.make_cfg(21, contigs, 600, 3, 6, 4, jitter=100, read_len=100)
    ref = ds.to_soa()

    def gen():
        for i, r in enumerate(ds.recs):
            aux = ([("SA", r.sa)] if r.sa else []) + ([("OC", r.oc)] if r.oc else [])
            # every 7th record carries 150-300 kb of sequence + qualities (2 to 7 blocks), the others 0 to 400 bases
            seq_len = 100_000 + 13_007 * (i % 11) if i % 7 == 3 else (i * 37) % 400
            yield bamio.encode_record(r.qname, r.flag, r.tid, r.pos, r.mapq, bamio.parse_cigar(r.cigar), r.mtid, r.mpos, r.isize, aux, seq_len=seq_len)

    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "long.bam")
        bamio.write_bam(p, contigs, gen())
        host_contigs, host_cols = capi.decode_bam(p)
        table = capi.decode_bam_device(p)
        got = _device_cols(table)
        assert table.contigs == contigs == host_contigs
        for k, _ in abi.SOA_COLS_ALL:
            assert np.array_equal(got[k], ref[k]), k
            assert np.array_equal(got[k], host_cols[k]), k
        table.close()


def test_device_decode_of_records_across_blocks_in_chunks():
    """the variant for files too large for one batch: chunks start at the record carried over from the chunk before;
    short records, records of 100-230 kB (2-7 blocks) and chunks of a few blocks"""
    from breakid_amd import bamio
    contigs = [("chr1", 3_000_000), ("chr2", 2_000_000)]
    ds = synth.make_cfg(23, contigs, 3000, 4, 8, 6, jitter=100, read_len=100)
    for i in range(0, len(ds.recs), 97):
        ds.recs[i].sa = "chr2,%d,+,40S60M,60,0;" % (100 + i)
    ref = ds.to_soa()
    rng = np.random.default_rng(4)

    def gen(long_every):
        for i, r in enumerate(ds.recs):
            aux = ([("SA", r.sa)] if r.sa else []) + ([("OC", r.oc)] if r.oc else [])
            seq_len = 100_000 + 13_007 * (i % 11) if long_every and i % long_every == 3 else (i * 37) % 400
            rec = bytearray(bamio.encode_record(r.qname, r.flag, r.tid, r.pos, r.mapq, bamio.parse_cigar(r.cigar), r.mtid, r.mpos, r.isize, aux, seq_len=seq_len))
            if seq_len:   # incompressible bases / qualities: chunks of a few blocks really are a few blocks
                at = len(rec) - sum(3 + len(v) + 1 for _, v in aux) - ((seq_len + 1) // 2 + seq_len)
                rec[at:at + (seq_len + 1) // 2 + seq_len] = rng.integers(0, 256, (seq_len + 1) // 2 + seq_len, dtype=np.uint8).tobytes()
            yield bytes(rec)

    os.environ["BREAKID_FEED_PACKED_CHUNKS"] = "1"
    try:
        for long_every, mb in ((0, 0.07), (0, 0.3), (29, 0.07), (29, 0.5), (7, 1.0)):
            with tempfile.TemporaryDirectory() as t:
                p = os.path.join(t, "x.bam")
                bamio.write_bam(p, contigs, gen(long_every))
                os.environ["BREAKID_FEED_CHUNK_MB"] = repr(mb)
                table = capi.decode_bam_device(p)
                got = _device_cols(table)
                assert table.contigs == contigs
                for k, _ in abi.SOA_COLS_ALL:
                    assert np.array_equal(got[k], ref[k]), (long_every, mb, k)
                table.close()
    finally:
        os.environ.pop("BREAKID_FEED_PACKED_CHUNKS", None)
        os.environ.pop("BREAKID_FEED_CHUNK_MB", None)


@pytest.mark.parametrize("chunk_mb", ["0.25", "64"])
def test_stream_pass_overlapped_with_the_feed_matches_oracle(monkeypatch, capfd, chunk_mb):
    """bk_bam_decode_device_ctx: one read of the file, k_stream running on the chunks already decoded while the rest of the file
    arrives (the reference makes two sequential passes, BreakID.cc:1929, :1414).  Many small chunks: most records
    go through k_stream piecewise, over column buffers that grow (and move) meanwhile; the calls must equal the oracle's."""
    from oracle import pyoracle
    contigs = [("chr1", 6_000_000), ("chr2", 5_000_000), ("chrX", 3_000_000)]
    ds = synth.make_cfg(31, contigs, 400_000, 60, 40, 600, jitter=250, read_len=100)
    ref = ds.to_soa()
    monkeypatch.setenv("BREAKID_FEED_CHUNK_MB", chunk_mb)
    monkeypatch.setenv("BK_DEBUG", "feed")
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "a.bam")
        ds.write_bam(p, aligned=True)
        ctx, table = capi.decode_bam_device_ctx(p, qual=20)
    err = capfd.readouterr().err
    assert "[feed/stream] stream pass overlapped" in err, err[-600:]
    done, total = [int(v) for v in err.split("[feed/stream] stream pass overlapped:")[1].split("records")[0].replace("of", " ").split()]
    assert total == len(ref["tid"]) and (done > total // 2 if chunk_mb == "0.25" else True), (done, total)
    assert ctx.contigs == contigs
    w, n_valid = ctx.run(qual=20, fast=True)
    o = pyoracle.Oracle(contigs, ref)
    ow, rc = o.run(20, fast=True)
    assert rc == 0 and w == ow
    for st in (abi.STAGE_SCAN, abi.STAGE_ISO, abi.STAGE_CLUSTERED, abi.STAGE_SPLITS, abi.STAGE_CLUSTERS):
        a, _ = ctx.fetch(st)
        b, _ = o.fetch(st)
        assert np.array_equal(a, b), st
    assert n_valid > 0
    # a second run with another mapq threshold on the same context re-filters the candidates (full pass)
    w2, _ = ctx.run(qual=30, fast=True)
    o.run(30, fast=True)
    a, _ = ctx.fetch(abi.STAGE_CLUSTERS)
    b, _ = o.fetch(abi.STAGE_CLUSTERS)
    assert np.array_equal(a, b)
    ctx.close()
    table.close()
    o.close()
