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
    for name, dt in abi.SOA_COLS:
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
    for k, _ in abi.SOA_COLS:
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


def test_device_decode_rejects_unaligned_blocks_and_takes_the_golden_shapes_when_aligned():
    contigs, ds = _dataset()
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "u.bam")
        ds.write_bam(p)                 # fixed-size blocks: records straddle them
        with pytest.raises(capi.BreakIDError) as e:
            capi.decode_bam_device(p)
        assert e.value.code == abi.BK_ERR_IO and "aligned" in str(e.value)
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
            raw = b"".join(_inflate_blocks(open(up, "rb").read()))
            ap = os.path.join(t, "a.bam")
            _rewrite_aligned(raw, ap)
            table = capi.decode_bam_device(ap)
            got = _device_cols(table)
            assert table.contigs == g.contigs
            for k, _ in abi.SOA_COLS:
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
