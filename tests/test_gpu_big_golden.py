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


@pytest.mark.parametrize("name,mode", [("deep", "fast"), ("deepw", "fast"), ("panel", "fast"), ("panel", "ahc"), ("g3", "fast"), ("g3", "ahc")])
def test_stages_match_reference_on_large_inputs(name, mode):
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
