"""Seeded inputs of the LARGE golden vectors (tests/golden/*.digest.json, g3.*): too big to commit as tables, so the
tests regenerate them and check the table's sha256 against the one the reference saw (tools/make_golden_big.py).

Every entry: contigs, cols (numpy SoA, abi.SOA_COLS layout + target_len), write_bam(path), refgene rows, and whether the
contigs get sequence (.nib) files (only needed when breakpoints are annotated)."""
from __future__ import annotations

import numpy as np

from . import bamio, synth

G3_CONTIGS = [("chr1", 50_000_000), ("chr2", 50_000_000)]
PANEL_CONTIGS = [("chr%d" % (i + 1), 5_000_000) for i in range(8)]


class Fixture:
    def __init__(self, name, contigs, cols, bam_writer, refgene=(), nib=False):
        self.name, self.contigs, self.cols, self.write_bam, self.refgene, self.nib = name, contigs, cols, bam_writer, list(refgene), nib


def _from_table(name, contigs, cols, refgene=(), nib=False):
    cols, names = synth.name_records(cols)
    cols["target_len"] = np.asarray([l for _, l in contigs], np.uint32)
    qn = [bytes(r) for r in names]
    return Fixture(name, contigs, cols, lambda path: bamio.write_bam_from_soa(path, contigs, cols, qn), refgene, nib)


def g3():
    """SURVEY 8(c) G3 = BASELINE.json configs[0]: 1 004 800 records, 2 x 50 Mb, 400 loci x 50 pairs, 5 000 noise pairs,
    8 split reads at every second locus."""
    ds = synth.make_cfg(12346, G3_CONTIGS, 1_004_800, 400, 50, 5000)
    return Fixture("g3", G3_CONTIGS, ds.to_soa(), ds.write_bam, synth.random_refgene(G3_CONTIGS, 60, 3), nib=True)


def panel():
    """SURVEY 8(c) G5 = BASELINE.json configs[3] at test size: 40 fusion loci x 600x over 8 x 5 Mb contigs, ~20 % split
    reads with clip points scattered around the breakpoint, heavy coordinate ties."""
    import torch
    from . import synth_gpu
    contigs, cols = synth_gpu.make_panel(77, torch.device("cpu"), n_loci=40, depth=600, window=600, contigs=PANEL_CONTIGS)
    return _from_table("panel", contigs, synth_gpu.to_numpy_cols(cols), synth.random_refgene(PANEL_CONTIGS, 80, 5), nib=True)


def deep():
    """Pairs discovered in median-of-3-killer order: the reference's own std::sort heapsorts segments of 10^3..3*10^5."""
    contigs, cols, names = synth.make_deep()
    qn = [bytes(r) for r in names]
    return Fixture("deep", contigs, cols, lambda path: bamio.write_bam_from_soa(path, contigs, cols, qn))


def deepw():
    """hg19-shaped table with half of the records discordant: the same/other-chromosome mixture that drives the LATER sorts
    of remove_isolated_pairs / the fast clustering (by the mate coordinate) into the depth limit."""
    import torch
    from . import synth_gpu
    contigs, cols = synth_gpu.make_wgs(2_000_000, 12346, torch.device("cpu"), disc_frac=0.5)
    return _from_table("deepw", contigs, synth_gpu.to_numpy_cols(cols))


def panelfull():
    """BASELINE.json configs[3] at its full size: 500 fusion loci x 2000x (6 799 500 records, 20 % split reads, heavy coordinate
    ties) - the table tools/time_reference.py times the reference on ("config 4")."""
    import torch
    from . import synth_gpu
    contigs, cols = synth_gpu.make_panel(12349, torch.device("cpu"), n_loci=500, depth=2000, window=600, contigs=PANEL_CONTIGS)
    return _from_table("panelfull", contigs, synth_gpu.to_numpy_cols(cols), synth.random_refgene(PANEL_CONTIGS, 80, 5), nib=True)


def wgs100():
    """BASELINE.json configs[1] shape at 100 M records: the hg19 table of tests/test_gpu_parity.py::test_wgs_shape_100M... (same
    generator, same seed, made on the CPU so that it is the same table on every machine) - the largest hg19-shaped input the
    REAL reference was run on (-fast, tools/make_golden_big.py wgs100).  Same-chromosome groups of ~27 K pairs whose later sorts
    run into libstdc++'s depth limit inside the reference binary."""
    import torch
    from . import synth_gpu
    contigs, cols = synth_gpu.make_wgs(100_000_000, 2024, torch.device("cpu"))
    cols, names = synth.name_records(synth_gpu.to_numpy_cols(cols))
    cols["target_len"] = np.asarray([l for _, l in contigs], np.uint32)
    fx = Fixture("wgs100", contigs, cols, lambda path: bamio.write_bam_from_soa_fast(path, contigs, cols, names), synth.random_refgene(contigs, 400, 11), nib=True)
    fx.max_nib_len = 300_000_000
    return fx


ALL = {"g3": g3, "panel": panel, "deep": deep, "deepw": deepw, "panelfull": panelfull, "wgs100": wgs100}
