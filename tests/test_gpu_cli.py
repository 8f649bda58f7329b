"""End-to-end drop-in check: the BreakID command line (same flags, same txt files as the reference) on the
GPU against the reference's own output files kept as golden fixtures (tools/make_golden.py)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from breakid_amd import synth
from breakid_amd import bamio
from tests import refdump
from tools import make_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "breakid_amd", "bin", "BreakID")


def _dataset(name):
    for n, ds, refgene in make_golden.datasets():
        if n == name:
            return ds, refgene
    raise KeyError(name)


@pytest.mark.parametrize("name", ["g1", "g2", "small", "ties", "edge"])
@pytest.mark.parametrize("mode", ["fast", "ahc"])
def test_cli_txt_outputs_match_reference(golden_dir, name, mode):
    ds, refgene = _dataset(name)
    with tempfile.TemporaryDirectory() as tmp:
        bam = os.path.join(tmp, name + ".bam")
        aligned = mode == "fast"   # blocks as htslib writes them -> the streaming GPU decoder; fixed-size blocks (records across
        ds.write_bam(bam, aligned=aligned)  # blocks, as htsjdk writes them) -> its one-batch variant, or the host decoder when forced
        host = name in ("g2", "ties") and not aligned
        bamio.write_bai(bam)  # the reference loads the index before it calls breakpoints (BreakID.cc:411-416)
        side = synth.write_side_files(ds, tmp, refgene_lines=refgene)
        prefix = os.path.join(tmp, "out")
        cmd = [BIN, "-i", bam, "-o", prefix, "-n", side["nib"], "-all"] + (["-fast"] if mode == "fast" else [])
        env = dict(os.environ, BREAKID_INSTALLDIR=side["install"], BK_DEBUG="feed")
        if host:
            env["BREAKID_HOST_DECODE"] = "1"
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("[feed/gpu]" in r.stderr) == (not host) and ("[feed]" in r.stderr) == host, r.stderr[-500:]
        assert ("records across blocks" in r.stderr) == (not host and not aligned), r.stderr[-500:]
        for suffix in ("_fusion.txt", "_fusion_all.txt"):
            got = open(prefix + suffix).read()
            exp = open(os.path.join(golden_dir, "%s.%s%s" % (name, mode, suffix))).read()
            assert got == exp, (suffix, got[:600], exp[:600])
        got = open(prefix + "_params.txt").read().replace(tmp, "<TMP>").replace("out_file\t<TMP>/out", "out_file\t<TMP>/out_" + mode)
        exp = open(os.path.join(golden_dir, "%s.%s_params.txt" % (name, mode))).read()
        assert got == exp, (got, exp)
        # _performance.txt (BreakID.cc:175-191): header + the five deterministic columns; four clock() columns follow
        perf = open(prefix + "_performance.txt").read().split("\n")
        exp = open(os.path.join(golden_dir, "%s.%s_perf5.txt" % (name, mode))).read().split("\n")
        assert perf[0] == exp[0] and perf[1].split("\t")[:5] == exp[1].split("\t") and len(perf[1].split("\t")) == 9, (perf, exp)


def test_cli_usage_errors():
    r = subprocess.run([BIN, "-h"], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr
    r = subprocess.run([BIN, "-i", "x.bam"], capture_output=True, text=True)
    assert r.returncode == 1 and "input- and output file is required" in r.stderr
    r = subprocess.run([BIN, "-i", "x.bam", "-o", "p"], capture_output=True, text=True)
    assert r.returncode == 1 and "nib file's root dir is required" in r.stderr


def test_cli_fatal_paths_match_the_reference_messages():
    """missing BAM / ref_names.txt / index / refGene.txt: the reference's messages and exit codes (BreakID.cc:1917-1921, :1399-1404,
    :411-416, RefSeqTranscript.cc:212-216), same table as the CPU build of the host code (tests/test_cpu_cli.py)"""
    from tests.test_cpu_cli import check_fatal_paths
    check_fatal_paths(BIN)


@pytest.mark.parametrize("name,mode,ranks,comm,feed", [("small", "fast", 1, "rccl", "gpu"), ("edge", "ahc", 1, "rccl", "gpu"), ("small", "fast", 2, "local", "gpu"),
                                                        ("ties", "ahc", 2, "local", "gpu"), ("edge", "fast", 3, "local", "gpu"), ("g1", "fast", 4, "local", "gpu"),
                                                        ("g1", "fast", 5, "local", "gpu-small-chunks"), ("small", "fast", 2, "local", "host"), ("edge", "fast", 3, "local", "across"),
                                                        ("ties", "ahc", 1, "rccl", "host")])
def test_cli_sharded_run_from_cpp_matches_reference(golden_dir, name, mode, ranks, comm, feed):
    """`BreakID -gpus N`: one sample over N contexts, orchestrated in C++ (csrc/multi_gpu.hip, include/breakid_multi.h).
    -comm rccl issues the collectives through librccl directly (world size 1 on this one-GPU box: ncclAllGather / grouped
    ncclBroadcast / ncclSend+ncclRecv / ncclAllReduce all run); -comm local puts N contexts on one GPU and moves the same
    tables by device-to-device copies - the N-rank code path (record ranges, routed all-to-alls, LPT group ownership, vote
    slices) against the reference's txt files.  feed: "gpu" = every rank decodes its part of the file on the GPU
    (bk_multi_run_bam / bk_bam_decode_device_part; "-small-chunks": several feed chunks per part), "host" = the host decoder's
    table cut into record ranges (bk_multi_run, BREAKID_HOST_DECODE=1), "across" = a file whose records run across BGZF blocks:
    every rank decodes its part on the GPU all the same (round 4: boundaries guessed per part and verified into the next part)."""
    ds, refgene = _dataset(name)
    with tempfile.TemporaryDirectory() as tmp:
        bam = os.path.join(tmp, name + ".bam")
        ds.write_bam(bam, aligned=feed != "across")
        bamio.write_bai(bam)  # the reference loads the index before it calls breakpoints (BreakID.cc:411-416)
        side = synth.write_side_files(ds, tmp, refgene_lines=refgene)
        prefix = os.path.join(tmp, "out")
        cmd = [BIN, "-i", bam, "-o", prefix, "-n", side["nib"], "-all", "-gpus", str(ranks), "-comm", comm] + (["-fast"] if mode == "fast" else [])
        env = dict(os.environ, BREAKID_INSTALLDIR=side["install"])
        if feed == "host":
            env["BREAKID_HOST_DECODE"] = "1"
        if feed == "gpu-small-chunks":
            env["BREAKID_FEED_CHUNK_MB"] = repr(os.path.getsize(bam) / 23 / 1048576.0)
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        for suffix in ("_fusion.txt", "_fusion_all.txt"):
            got = open(prefix + suffix).read()
            exp = open(os.path.join(golden_dir, "%s.%s%s" % (name, mode, suffix))).read()
            assert got == exp, (suffix, got[:600], exp[:600])
        # the sharded run writes the same _performance.txt columns (per-group counters summed over the ranks: bk_multi_stats) and
        # the same stdout lines as the single-GPU path
        perf = open(prefix + "_performance.txt").read().split("\n")
        exp = open(os.path.join(golden_dir, "%s.%s_perf5.txt" % (name, mode))).read().split("\n")
        assert perf[0] == exp[0] and perf[1].split("\t")[:5] == exp[1].split("\t") and len(perf[1].split("\t")) == 9, (perf, exp)
        assert "the insert size mean: " in r.stdout and "Scanning discordant read pairs done." in r.stdout


@pytest.mark.parametrize("comm", ["local"] + (["rccl"] if False else []))
def test_cli_sharded_run_one_rank_fails_and_every_rank_stops(comm):
    """a failure of ONE rank (its part of the file ends in a corrupt BGZF block) must end the whole run with that rank's error -
    never leave the others waiting in an exchange (the rank threads meet on the host in front of every collective)"""
    ds, refgene = _dataset("small")
    with tempfile.TemporaryDirectory() as tmp:
        bam = os.path.join(tmp, "bad.bam")
        ds.write_bam(bam, aligned=True)
        raw = bytearray(open(bam, "rb").read())
        # damage the deflate stream of a block in the last third of the file (its header stays intact: the cut points are found)
        off, starts = 0, []
        while off < len(raw):
            starts.append(off)
            off += int.from_bytes(raw[off + 16:off + 18], "little") + 1
        victim = starts[-3]
        for k in range(40, 80):
            raw[victim + k] ^= 0x5A
        open(bam, "wb").write(bytes(raw))
        bamio_ok = False
        try:
            bamio.write_bai(bam)
            bamio_ok = True
        except Exception:
            open(bam + ".bai", "wb").write(b"BAI\1" + (len(ds.contigs)).to_bytes(4, "little") + bytes(8 * len(ds.contigs)))
        side = synth.write_side_files(ds, tmp, refgene_lines=refgene)
        cmd = [BIN, "-i", bam, "-o", os.path.join(tmp, "out"), "-n", side["nib"], "-all", "-fast", "-gpus", "3", "-comm", comm]
        r = subprocess.run(cmd, env=dict(os.environ, BREAKID_INSTALLDIR=side["install"]), capture_output=True, text=True, timeout=120)
        assert r.returncode != 0, (r.stdout[-500:], r.stderr[-500:], bamio_ok)


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_device_count() < 2, reason="needs >= 2 GPUs: RCCL takes one device per rank")
@pytest.mark.parametrize("name,mode,feed", [("small", "fast", "gpu"), ("edge", "ahc", "gpu"), ("g1", "fast", "host")])
def test_cli_two_rccl_ranks_match_reference(golden_dir, name, mode, feed):
    """`BreakID -gpus 2 -comm rccl` on a node with two devices: librccl with two ranks (threads of one process, one GPU each)"""
    test_cli_sharded_run_from_cpp_matches_reference(golden_dir, name, mode, 2, "rccl", feed)
