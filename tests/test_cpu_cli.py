"""CPU: the product's HOST code - command line (breakid_main.cc: options, fatal paths, refGene / nib annotation, writers,
_performance.txt) and host BAM decoder (bam_reader.cc) - linked over the CPU oracle by oracle/cpu_shim.cc (test
infrastructure, `make -C oracle cpucli asan ubsan`), plain and under AddressSanitizer / UBSan, against the REFERENCE's
golden txt files.  The same command-line checks run against the real GPU binary in tests/test_gpu_cli.py."""
import os
import subprocess
import sys
import tempfile

import pytest

from breakid_amd import synth
from breakid_amd import bamio
from tools import make_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "oracle", "_san")


@pytest.fixture(scope="module")
def binaries():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-j3", "san"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return {k: os.path.join(SAN, "BreakID_cpu" + ("" if k == "plain" else "_" + k)) for k in ("plain", "asan", "ubsan")}


def _dataset(name):
    for n, ds, refgene in make_golden.datasets():
        if n == name:
            return ds, refgene
    raise KeyError(name)


def run_cli(binary, name, mode, golden_dir, aligned=True):
    ds, refgene = _dataset(name)
    with tempfile.TemporaryDirectory() as tmp:
        bam = os.path.join(tmp, name + ".bam")
        ds.write_bam(bam, aligned=aligned)
        bamio.write_bai(bam)  # the reference loads the index before it calls breakpoints (BreakID.cc:411-416)
        side = synth.write_side_files(ds, tmp, refgene_lines=refgene)
        prefix = os.path.join(tmp, "out")
        cmd = [binary, "-i", bam, "-o", prefix, "-n", side["nib"], "-all"] + (["-fast"] if mode == "fast" else [])
        env = dict(os.environ, BREAKID_INSTALLDIR=side["install"], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
        for suffix in ("_fusion.txt", "_fusion_all.txt"):
            got = open(prefix + suffix).read()
            exp = open(os.path.join(golden_dir, "%s.%s%s" % (name, mode, suffix))).read()
            assert got == exp, (suffix, got[:600], exp[:600])
        got = open(prefix + "_params.txt").read().replace(tmp, "<TMP>").replace("out_file\t<TMP>/out", "out_file\t<TMP>/out_" + mode)
        assert got == open(os.path.join(golden_dir, "%s.%s_params.txt" % (name, mode))).read()
        perf = open(prefix + "_performance.txt").read().split("\n")
        exp = open(os.path.join(golden_dir, "%s.%s_perf5.txt" % (name, mode))).read().split("\n")
        assert perf[0] == exp[0] and perf[1].split("\t")[:5] == exp[1].split("\t") and len(perf[1].split("\t")) == 9, (perf, exp)


@pytest.mark.parametrize("name", ["g1", "g2", "small", "ties", "edge"])
@pytest.mark.parametrize("mode", ["fast", "ahc"])
def test_host_code_reproduces_reference_txt(binaries, golden_dir, name, mode):
    run_cli(binaries["plain"], name, mode, golden_dir, aligned=(mode == "fast"))


@pytest.mark.parametrize("san", ["asan", "ubsan"])
@pytest.mark.parametrize("name,mode", [("edge", "fast"), ("ties", "ahc"), ("small", "fast")])
def test_host_code_clean_under_sanitizers(binaries, golden_dir, san, name, mode):
    run_cli(binaries[san], name, mode, golden_dir, aligned=(name != "ties"))


@pytest.mark.parametrize("san", ["asan", "ubsan"])
def test_oracle_unit_entry_points_clean_under_sanitizers(binaries, san):
    """the oracle's unit entry points (CIGAR table, AHC, masks, votes, regions) against the reference's vectors, sanitized"""
    env = dict(os.environ, BREAKID_ORACLE_LIB=os.path.join(SAN, "liboracle_%s.so" % san), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    if san == "asan":
        env["LD_PRELOAD"] = subprocess.check_output(["g++", "-print-file-name=libasan.so"], text=True).strip()
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py")],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and "runtime error" not in (r.stdout + r.stderr), (r.stdout[-2000:], r.stderr[-2000:])


def fatal_cases(tmp, ds, refgene):
    """(description, argv tail, env patch, expected exit code, expected stderr text) - BreakID.cc:78-91, :1917-1921, :1399-1404,
    :411-416, RefSeqTranscript.cc:212-216"""
    bam = os.path.join(tmp, "in.bam")
    ds.write_bam(bam, aligned=True)
    bamio.write_bai(bam)  # the reference loads the index before it calls breakpoints (BreakID.cc:411-416)
    side = synth.write_side_files(ds, tmp, refgene_lines=refgene)
    noidx = os.path.join(tmp, "noidx.bam")
    ds.write_bam(noidx, aligned=True)
    # an index that is there but does not load (hts_idx_load_local: short read of the magic / foreign magic / file ends inside the
    # bin count of a reference) is the same fatal path as no index at all; an index named <stem>.bai is found like <bam>.bai
    bad = {}
    for what, content in (("emptyidx", b""), ("foreignidx", b"XYZ\1" + bytes(64)), ("shortidx", b"BAI\1" + (len(ds.contigs)).to_bytes(4, "little") + bytes(2))):
        bad[what] = os.path.join(tmp, what + ".bam")
        ds.write_bam(bad[what], aligned=True)
        open(bad[what] + ".bai", "wb").write(content)
    stem = os.path.join(tmp, "stem.bam")
    ds.write_bam(stem, aligned=True)
    bamio.write_bai(stem, os.path.join(tmp, "stem.bai"))
    empty_nib = os.path.join(tmp, "empty_nib")
    os.makedirs(empty_nib)
    no_inst = os.path.join(tmp, "no_install")
    os.makedirs(no_inst)
    ok_env = {"BREAKID_INSTALLDIR": side["install"]}
    return [
        ("help", ["-h"], ok_env, 1, "Usage"),
        ("no output prefix", ["-i", bam], ok_env, 1, "Error: input- and output file is required."),
        ("no nib dir", ["-i", bam, "-o", os.path.join(tmp, "o")], ok_env, 1, "Error: nib file's root dir is required."),
        ("missing bam", ["-i", os.path.join(tmp, "nope.bam"), "-o", os.path.join(tmp, "o"), "-n", side["nib"]], ok_env, 1,
         "Error: can not open bam-file: " + os.path.join(tmp, "nope.bam")),
        ("missing ref_names.txt", ["-i", bam, "-o", os.path.join(tmp, "o"), "-n", empty_nib], ok_env, 1, "Error: cannot open reference names file."),
        ("missing index", ["-i", noidx, "-o", os.path.join(tmp, "o"), "-n", side["nib"], "-fast"], ok_env, 1, "Error: please index bam-file first:\t" + noidx),
        ("empty index", ["-i", bad["emptyidx"], "-o", os.path.join(tmp, "o"), "-n", side["nib"], "-fast"], ok_env, 1, "Error: please index bam-file first:\t" + bad["emptyidx"]),
        ("foreign index", ["-i", bad["foreignidx"], "-o", os.path.join(tmp, "o"), "-n", side["nib"], "-fast"], ok_env, 1, "Error: please index bam-file first:\t" + bad["foreignidx"]),
        ("truncated index", ["-i", bad["shortidx"], "-o", os.path.join(tmp, "o"), "-n", side["nib"], "-fast"], ok_env, 1, "Error: please index bam-file first:\t" + bad["shortidx"]),
        ("index named after the stem", ["-i", stem, "-o", os.path.join(tmp, "o2"), "-n", side["nib"], "-fast"], ok_env, 0, ""),
        ("missing refGene.txt", ["-i", bam, "-o", os.path.join(tmp, "o"), "-n", side["nib"], "-fast"], {"BREAKID_INSTALLDIR": no_inst}, 1,
         "Error: cannot open \t" + os.path.join(no_inst, "ref_files", "refGene.txt")),
    ]


def check_fatal_paths(binary):
    ds, refgene = _dataset("g1")
    with tempfile.TemporaryDirectory() as tmp:
        for what, argv, envp, code, text in fatal_cases(tmp, ds, refgene):
            r = subprocess.run([binary] + argv, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", **envp), capture_output=True, text=True)
            assert r.returncode == code and text in r.stderr, (what, r.returncode, r.stderr[-400:])


@pytest.mark.parametrize("which", ["plain", "asan", "ubsan"])
def test_fatal_paths_match_the_reference_messages(binaries, which):
    check_fatal_paths(binaries[which])
