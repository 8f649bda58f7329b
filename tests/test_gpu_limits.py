"""Documented limits and the hash-collision detection: each case ends in the documented error code, never in a call the
reference would not make.  (The reference compares read names and CIGAR texts as strings: BreakID.cc:1424, :627-637.)"""
import ctypes as C

import numpy as np
import pytest

from breakid_amd import abi, capi, synth
from tests import refdump

pytestmark = pytest.mark.gpu


def test_forged_read_name_collision_in_the_mate_join_is_detected(golden_dir):
    """two DIFFERENT read names given the same 64-bit qhash: their second hashes (qcheck) differ -> BK_ERR_COLLISION"""
    contigs, cols = refdump.load_soa(golden_dir, "edge")
    cols = {k: v.copy() for k, v in cols.items()}
    f = cols["flag"]
    cand = np.nonzero((cols["mapq"] >= 20) & ((f & 0x400) == 0) & ((f & 0x100) == 0) & ((f & 1) != 0) & ((f & 2) == 0))[0]
    a = int(cand[0])
    b = int(next(i for i in cand if cols["qhash"][i] != cols["qhash"][a]))
    assert cols["qcheck"][a] != cols["qcheck"][b]
    cols["qhash"][b] = cols["qhash"][a]
    ctx = capi.Context(contigs)
    ctx.upload(cols)
    with pytest.raises(capi.BreakIDError) as e:
        ctx.run(qual=20, fast=True)
    assert e.value.code == abi.BK_ERR_COLLISION and "read names" in str(e.value)
    # without the qcheck column the same table cannot be checked: the collision goes through (the declared behaviour of
    # tables that carry no names, e.g. generated ones)
    del cols["qcheck"]
    ctx.upload(cols)
    ctx.run(qual=20, fast=True)
    ctx.close()


def test_forged_read_name_collision_in_the_breakpoint_vote_is_detected(golden_dir):
    contigs, cols = refdump.load_soa(golden_dir, "g1")
    cols = {k: v.copy() for k, v in cols.items()}
    sa = np.nonzero(np.diff(cols["aux_off"].astype(np.int64)) > 0)[0]
    hashes = sorted(set(int(cols["qhash"][i]) for i in sa))
    assert len(hashes) >= 2
    victim = cols["qhash"] == np.uint64(hashes[1])
    cols["qhash"][victim] = np.uint64(hashes[0])   # split read s1 now "is" s0 by its 64-bit hash; qcheck still tells them apart
    ctx = capi.Context(contigs)
    ctx.upload(cols)
    with pytest.raises(capi.BreakIDError) as e:
        ctx.run(qual=20, fast=True)
    assert e.value.code == abi.BK_ERR_COLLISION
    ctx.close()


def test_cigar_text_codes_are_exact_for_gate_passing_texts():
    """bk_split.prim_cigar / sec_cigar: <n><M|S><n><M|S> texts are encoded exactly (bit 63), so two different texts that can
    reach a tuple never compare equal; anything else is a 63-bit hash with bit 63 clear"""
    from oracle import pyoracle
    L = pyoracle.lib()
    texts = ["60M40S", "60S40M", "060M40S", "60M040S", "0060M40S", "6M040S", "60M4S", "604M0S", "0M1S", "00M1S", "268435455M1S"]
    codes = [L.ora_text_hash(t.encode(), len(t)) for t in texts]
    assert all(c >> 63 for c in codes) and len(set(codes)) == len(codes)
    for t in ["00000M1S", "268435456M1S", "60M40S1M", "60M", "", "6X4S", "60M40H"]:
        assert (L.ora_text_hash(t.encode(), len(t)) >> 63) == 0, t


def test_more_than_2_32_records_is_refused():
    ctx = capi.Context([("chr1", 1000)])
    s = abi.Soa()
    s.n = 0xFFFFFFF1
    with pytest.raises(capi.BreakIDError) as e:
        ctx._check(ctx.L.bk_upload_records(ctx.h, C.byref(s), abi.BK_MEM_HOST))
    assert e.value.code == abi.BK_ERR_LIMIT and "2^32" in str(e.value)
    ctx.close()


def test_device_table_with_a_misaligned_column_is_refused():
    """BK_MEM_DEVICE tables are used in place and read as 16-byte vectors: a column that does not start on 16 bytes (a slice
    of a larger buffer, say) is an argument error, not a slow or faulting run"""
    import torch
    from breakid_amd import synth_gpu
    dev = torch.device("cuda", 0)
    contigs, cols = synth_gpu.make_wgs(200_000, 5, dev)
    ctx = capi.Context(contigs)
    ptrs = abi.device_ptrs(cols)
    ctx.attach_device(ptrs, cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])  # as generated: fine
    shifted = torch.empty(cols["n"] + 1, dtype=torch.int32, device=dev)
    shifted[1:] = cols["isize"]
    bad = dict(ptrs, isize=shifted[1:].data_ptr())
    assert bad["isize"] % 16 == 4
    with pytest.raises(capi.BreakIDError) as e:
        ctx.attach_device(bad, cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])
    assert e.value.code == abi.BK_ERR_ARG and "16-byte aligned" in str(e.value)
    ctx.close()


def test_read_name_run_longer_than_4096_candidates_matches_the_oracle():
    """4200 discordant records under ONE read name (the reference pairs them up in arrival order, BreakID.cc:1419-1436): the
    in-run selection of the mate join is quadratic and gives up at 4096, the join then sorts every run by record index and walks
    it as it lies - no limit on the length of a run"""
    from oracle import pyoracle
    contigs = [("chr1", 10_000_000), ("chr2", 10_000_000)]
    ds = synth.Dataset(contigs)
    rng = np.random.default_rng(3)
    for i in range(400):
        ds.recs += synth._proper_pair(rng, i, 0, 1000, 9_000_000, 100, 350, 40)
    for i in range(2100):
        ds.recs += synth._discordant_pair("same", 0, 100_000 + 7 * i, 1, 200_000 + 5 * i, 100)
    for i in range(30):  # and ordinary names beside it
        ds.recs += synth._discordant_pair("other%d" % i, 0, 3_000_000 + 11 * i, 1, 4_000_000 + 13 * i, 100)
    ds.sort()
    cols = ds.to_soa()
    ctx = capi.Context(contigs)
    ctx.upload(cols)
    w, _ = ctx.run(qual=20, fast=True)
    o = pyoracle.Oracle(contigs, cols)
    ow, rc = o.run(20, fast=True)
    assert rc == 0 and w == ow
    for st in (abi.STAGE_SCAN, abi.STAGE_ISO, abi.STAGE_CLUSTERED, abi.STAGE_CLUSTERS):
        a, ao = ctx.fetch(st)
        b, bo = o.fetch(st)
        assert np.array_equal(a, b), st
    assert len(ctx.fetch(abi.STAGE_SCAN)[0]) >= 30  # (the 4200 same-name records pair up in arrival order, mostly within one chromosome and closer than w)
    ctx.close()
    o.close()


def test_read_name_run_of_4096_candidates_matches_the_oracle():
    from oracle import pyoracle
    contigs = [("chr1", 20_000_000), ("chr2", 20_000_000)]
    ds = synth.Dataset(contigs)
    rng = np.random.default_rng(4)
    for i in range(400):
        ds.recs += synth._proper_pair(rng, i, 0, 1000, 9_000_000, 100, 350, 40)
    for i in range(2048):
        ds.recs += synth._discordant_pair("same", 0, 100_000 + 5000 * i, 1, 200_000 + 5000 * i, 100)
    ds.sort()
    cols = ds.to_soa()
    ctx = capi.Context(contigs)
    ctx.upload(cols)
    w, _ = ctx.run(qual=20, fast=True)
    o = pyoracle.Oracle(contigs, cols)
    ow, rc = o.run(20, fast=True)
    assert rc == 0 and w == ow
    for st in (abi.STAGE_SCAN, abi.STAGE_ISO, abi.STAGE_CLUSTERED, abi.STAGE_CLUSTERS):
        a, _ = ctx.fetch(st)
        b, _ = o.fetch(st)
        assert np.array_equal(a, b), st
    assert len(ctx.fetch(abi.STAGE_SCAN)[0]) == 2048  # arrival order pairs neighbours on one chromosome, 5 kb apart
    ctx.close()
    o.close()


def test_ahc_component_beyond_the_pool_budget_is_refused():
    """the reference needs an N x N matrix of doubles for such a group (80 GB at 100 000 points) and hours of insert_sorted;
    the exact replay refuses it instead of running on"""
    n = 100_000
    x = np.full(n, 5000, np.uint32)
    y = np.full(n, 7000, np.uint32)
    ctx = capi.Context([("chr1", 1_000_000)])
    with pytest.raises(capi.BreakIDError) as e:
        ctx.debug_ahc(x, y, 10.5)
    assert e.value.code == abi.BK_ERR_LIMIT and "component" in str(e.value)
    ctx.close()


def test_record_longer_than_8_MiB_across_a_feed_chunk_takes_the_host_decoder(monkeypatch):
    """GPU feed: a BAM record longer than 8 MiB that crosses a feed chunk is refused (BK_ERR_LIMIT, records across BGZF blocks in
    chunks); the host decoder reads the same file, which is what the command line falls back to"""
    import tempfile
    from breakid_amd import bamio
    contigs = [("chr1", 50_000_000)]
    recs = []
    for i in range(200):
        recs.append(bamio.encode_record("r%d" % i, 0x63, 0, 1000 + 10 * i, 60, [100 << 4], 0, 1300 + 10 * i, 400, seq_len=100))
    huge = bytearray(bamio.encode_record("huge", 0x63, 0, 5000, 60, [100 << 4], 0, 5300, 400, seq_len=9_000_000))   # 13.5 MB record
    huge[-9_000_000:] = np.random.default_rng(1).integers(0, 64, 9_000_000, dtype=np.uint8).tobytes()  # qualities that do not compress away
    recs.append(bytes(huge))
    for i in range(200):
        recs.append(bamio.encode_record("s%d" % i, 0x63, 0, 6000 + 10 * i, 60, [100 << 4], 0, 6300 + 10 * i, 400, seq_len=100))
    with tempfile.TemporaryDirectory() as t:
        p = t + "/huge.bam"
        bamio.write_bam(p, contigs, recs, aligned=False)   # fixed-size blocks: the records run across BGZF blocks
        monkeypatch.setenv("BREAKID_FEED_PACKED_CHUNKS", "1")
        monkeypatch.setenv("BREAKID_FEED_CHUNK_MB", "1")
        with pytest.raises(capi.BreakIDError) as e:
            capi.decode_bam_device(p)
        assert e.value.code == abi.BK_ERR_LIMIT and "8 MiB" in str(e.value)
        contigs2, cols = capi.decode_bam(p)
    assert contigs2 == contigs and len(cols["tid"]) == 401 and int(cols["pos"][200]) == 5000


@pytest.mark.parametrize("names_per_half", [3, 10**9])
def test_read_names_sharing_the_upper_half_of_their_hash_are_still_joined_by_name(golden_dir, names_per_half):
    """the candidates are sorted by the upper 32 bits of the name hash only; runs of equal upper halves that hold several names
    are put in order by the full hash (k_join_fix_runs: up to 64 candidates per run), longer ones send the join back to a
    sort on all 64 bits - either way the pairs must be the oracle's"""
    from oracle import pyoracle
    contigs, cols = refdump.load_soa(golden_dir, "g2")
    cols = {k: v.copy() for k, v in cols.items()}
    q = cols["qhash"]
    names = np.unique(q)
    # groups of `names_per_half` names share their upper 32 bits (3 names = 6 candidates: fixed in place; all names in one run: fallback)
    remap = {}
    for j, h in enumerate(names):
        remap[int(h)] = ((0x9E3779B9 + j // names_per_half) << 32) | (j * 2654435761 & 0xFFFFFFFF)
    assert len(set(remap.values())) == len(names)
    cols["qhash"] = np.array([remap[int(h)] for h in q], dtype=np.uint64)
    ctx = capi.Context(contigs)
    ctx.upload(cols)
    w, nv = ctx.run(qual=20, fast=True)
    o = pyoracle.Oracle(contigs, cols)
    ow, rc = o.run(20, fast=True)
    assert rc == 0 and w == ow
    for st in (abi.STAGE_SCAN, abi.STAGE_ISO, abi.STAGE_CLUSTERED, abi.STAGE_CLUSTERS):
        a, ao = ctx.fetch(st)
        b, bo = o.fetch(st)
        assert np.array_equal(a, b), st
    ctx.close()
    o.close()


def test_sa_contig_names_outside_the_header_carry_a_second_hash():
    """an SA:Z text may name a contig that is neither in the header nor chr1..22,X,Y: it travels as a 30-bit hash id in
    prim_chr / sec_chr, and its second hash (bk_qname_check of the text) rides in bk_split.reserved, so that two such names are only
    taken for one when 62 bits agree (the reference compares the strings, BreakID.cc:627-637); known names leave the field 0"""
    from oracle import pyoracle
    ds = synth.make_g1()
    weird = 0
    for r in ds.recs:
        if r.sa and r.qname in ("s0", "s1", "s2"):
            f = r.sa.split(",")
            f[0] = {"s0": "GL000207.1", "s1": "HLA-A*01:01", "s2": "decoy_" + "x" * 40}[r.qname]
            r.sa = ",".join(f)
            weird += 1
    assert weird >= 6
    cols = ds.to_soa()
    ctx = capi.Context(ds.contigs)
    ctx.upload(cols)
    w, _ = ctx.run(qual=20, fast=True)
    o = pyoracle.Oracle(ds.contigs, cols)
    ow, rc = o.run(20, fast=True)
    assert rc == 0 and w == ow
    got, _ = ctx.fetch(abi.STAGE_SPLITS)
    exp, _ = o.fetch(abi.STAGE_SPLITS)
    assert np.array_equal(got, exp)
    unknown = ((got["prim_chr"] & 0x40000000) != 0) | ((got["sec_chr"] & 0x40000000) != 0)
    assert unknown.sum() >= 6 and np.all(got["reserved"][unknown] != 0) and np.all(got["reserved"][~unknown] == 0)
    L = capi.lib()
    for name in (b"GL000207.1", b"HLA-A*01:01"):
        assert int(L.bk_qname_check(name, len(name))) in set(int(v) for v in got["reserved"][unknown])
    a, _ = ctx.fetch(abi.STAGE_CLUSTERS)
    b, _ = o.fetch(abi.STAGE_CLUSTERS)
    assert np.array_equal(a, b)
    ctx.close()
    o.close()
