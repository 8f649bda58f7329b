"""GPU parity at the sizes SURVEY 8(c) names, against the REAL reference's outputs (tools/make_golden_big.py): G3 =
BASELINE.json configs[0] shape (1 004 800 records, 2 x 50 Mb), G5 = panel shape, and the inputs whose sorts take
libstdc++'s heapsort branch inside the reference binary.  Through the C ABI and through bin/BreakID."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from breakid_amd import abi, capi, synth
from breakid_amd import bamio
from tests import bigcases

pytestmark = pytest.mark.gpu
BIN = os.path.join(bigcases.ROOT, "breakid_amd", "bin", "BreakID")


def _gpu(fx, fast):
    ctx = capi.Context(fx.contigs)
    ctx.upload(fx.cols)
    mean, sd = ctx.isize_stats()
    w = capi.w_from(mean, sd)
    ctx.discordant_pairs(20, w)
    ctx.mask_and_cluster(w, fast)
    ctx.split_evidence()
    ctx.cluster_summary(w)
    ctx.split_breakpoints(w)
    return ctx, mean, sd, w


@pytest.mark.parametrize("name,mode", [("deep", "fast"), ("deepw", "fast"), ("panel", "fast"), ("panel", "ahc"), ("g3", "fast"), ("g3", "ahc"), ("panelfull", "fast"), ("wgs100", "fast")])
def test_stages_match_reference_on_large_inputs(name, mode):
    """wgs100 = BASELINE.json configs[1]'s shape at 100 M records, the largest hg19-shaped table the REAL reference was run on (-fast,
    755 s; its own std::sort heapsorts 4 558 segments of up to 22 852 elements there): every stage and every call of all 300 groups."""
    fx, meta = bigcases.load(name)
    if fx is None or not any(os.path.exists(os.path.join(bigcases.GOLD, "%s.%s.%s" % (name, mode, s))) for s in ("stages.txt.gz", "digest.json")):
        pytest.skip("golden %s/%s not generated" % (name, mode))
    ctx, mean, sd, w = _gpu(fx, mode == "fast")
    bigcases.check(name, mode, ctx.fetch, mean, sd, w)
    ctx.close()


@pytest.mark.parametrize("name,mode", [("g3", "fast"), ("g3", "ahc"), ("panel", "fast"), ("panel", "ahc")])
def test_cli_on_large_inputs_matches_reference_txt(name, mode):
    if not os.path.exists(os.path.join(bigcases.GOLD, "%s.%s_fusion_all.txt" % (name, mode))):
        pytest.skip("golden %s/%s not generated" % (name, mode))
    fx, meta = bigcases.load(name)
    with tempfile.TemporaryDirectory() as tmp:
        bam = os.path.join(tmp, name + ".bam")
        fx.write_bam(bam)
        bamio.write_bai(bam)  # the reference loads the index before it calls breakpoints (BreakID.cc:411-416)
        side = synth.write_side_files(fx.contigs, tmp, refgene_lines=fx.refgene, max_nib_len=60_000_000)
        prefix = os.path.join(tmp, "out")
        cmd = [BIN, "-i", bam, "-o", prefix, "-n", side["nib"], "-all"] + (["-fast"] if mode == "fast" else [])
        r = subprocess.run(cmd, env=dict(os.environ, BREAKID_INSTALLDIR=side["install"]), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        for suffix in ("_fusion.txt", "_fusion_all.txt"):
            got = open(prefix + suffix).read()
            exp = open(os.path.join(bigcases.GOLD, "%s.%s%s" % (name, mode, suffix))).read()
            assert got == exp, (suffix, got[:600], exp[:600])
        # _performance.txt (BreakID.cc:175-191): header + the five deterministic columns against the reference's file
        perf = open(prefix + "_performance.txt").read().split("\n")
        exp = open(os.path.join(bigcases.GOLD, "%s.%s_perf5.txt" % (name, mode))).read().split("\n")
        assert perf[0] == exp[0] and perf[1].split("\t")[:5] == exp[1].split("\t") and len(perf[1].split("\t")) == 9, (perf, exp)


def test_cli_on_the_full_size_panel_matches_the_reference_files():
    """BASELINE.json configs[3] at its full size (500 loci x 2000x, 6 799 500 records, -fast: the reference takes 72 s): the txt
    files of bin/BreakID against the sha256 of the reference's, and the five deterministic columns of _performance.txt"""
    import hashlib
    if not os.path.exists(os.path.join(bigcases.GOLD, "panelfull.fast.digest.json")):
        pytest.skip("golden panelfull not generated")
    sha, rows, perf5 = bigcases.expected_txt("panelfull", "fast")
    fx, meta = bigcases.load("panelfull")
    with tempfile.TemporaryDirectory() as tmp:
        bam = os.path.join(tmp, "panelfull.bam")
        fx.write_bam(bam)
        bamio.write_bai(bam)
        side = synth.write_side_files(fx.contigs, tmp, refgene_lines=fx.refgene, max_nib_len=60_000_000)
        prefix = os.path.join(tmp, "out")
        r = subprocess.run([BIN, "-i", bam, "-o", prefix, "-n", side["nib"], "-all", "-fast"], env=dict(os.environ, BREAKID_INSTALLDIR=side["install"]), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        for suffix in ("_fusion.txt", "_fusion_all.txt"):
            got = open(prefix + suffix, "rb").read()
            assert got.count(b"\n") == rows[suffix] and hashlib.sha256(got).hexdigest() == sha[suffix], (suffix, got[:600])
        perf = open(prefix + "_performance.txt").read().split("\n")
        assert perf[0] == perf5[0] and "\t".join(perf[1].split("\t")[:5]) == perf5[1], (perf, perf5)


@pytest.mark.skipif(not os.environ.get("BREAKID_BIG_TESTS"), reason="writes a 12.6 GB BAM under the temporary directory: BREAKID_BIG_TESTS=1 asks for it (run and logged once per round: profiles/)")
def test_cli_on_the_100M_record_wgs_table_matches_the_reference_files():
    """configs[1]'s shape at 100 M records through bin/BreakID (GPU feed from the BAM file, annotation, writers): the txt files against
    the sha256 of the REAL reference's, and the five deterministic columns of _performance.txt"""
    import hashlib
    sha, rows, perf5 = bigcases.expected_txt("wgs100", "fast")
    fx, meta = bigcases.load("wgs100")
    with tempfile.TemporaryDirectory() as tmp:
        bam = os.path.join(tmp, "wgs100.bam")
        fx.write_bam(bam)
        bamio.write_bai(bam)
        side = synth.write_side_files(fx.contigs, tmp, refgene_lines=fx.refgene, max_nib_len=fx.max_nib_len)
        prefix = os.path.join(tmp, "out")
        r = subprocess.run([BIN, "-i", bam, "-o", prefix, "-n", side["nib"], "-all", "-fast"], env=dict(os.environ, BREAKID_INSTALLDIR=side["install"]), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        for suffix in ("_fusion.txt", "_fusion_all.txt"):
            got = open(prefix + suffix, "rb").read()
            assert got.count(b"\n") == rows[suffix] and hashlib.sha256(got).hexdigest() == sha[suffix], (suffix, got[:600])
        perf = open(prefix + "_performance.txt").read().split("\n")
        assert perf[0] == perf5[0] and "\t".join(perf[1].split("\t")[:5]) == perf5[1], (perf, perf5)
