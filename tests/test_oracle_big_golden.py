"""CPU: the oracle port against the reference's outputs on the LARGE inputs (tools/make_golden_big.py): configs[0] shape
(G3, 1 M records), panel shape (G5; `panelfull` = BASELINE.json configs[3] at its full size, 6.8 M records, through the calls) and the two inputs on which the reference binary's own std::sort takes its heapsort
branch (`deep`: segments up to 3*10^5 elements; `deepw`: the WGS same/other-chromosome mixture)."""
import json
import os

import pytest

from oracle import pyoracle
from tests import bigcases

CASES = [("deep", "fast"), ("deepw", "fast"), ("panel", "fast"), ("panel", "ahc"), ("g3", "fast"), ("panelfull", "fast")]


@pytest.mark.parametrize("name,mode", CASES)
def test_oracle_matches_reference_on_large_inputs(name, mode):
    fx, meta = bigcases.load(name)
    if fx is None:
        pytest.skip("golden for %s not generated" % name)
    o = pyoracle.Oracle(fx.contigs, fx.cols)
    mean, sd = o.isize_stats()
    w, rc = o.run(20, fast=(mode == "fast"))
    assert rc == 0
    bigcases.check(name, mode, o.fetch, mean, sd, w)
    o.close()


@pytest.mark.skipif(not os.environ.get("BREAKID_BIG_TESTS"), reason="100 M records on the CPU: ~7 minutes and ~20 GB here; BREAKID_BIG_TESTS=1 asks for it (run and logged once per round: profiles/r04_wgs100_oracle_cpu.log)")
def test_oracle_matches_reference_on_the_100M_record_wgs_table():
    """configs[1]'s shape at 100 M records (`wgs100`): the oracle port against the REAL reference's stages and calls of all 300 groups"""
    fx, meta = bigcases.load("wgs100")
    o = pyoracle.Oracle(fx.contigs, fx.cols)
    mean, sd = o.isize_stats()
    w, rc = o.run(20, fast=True)
    assert rc == 0
    assert bigcases.check("wgs100", "fast", o.fetch, mean, sd, w) == "digest+calls"
    o.close()


def test_deep_inputs_really_reach_the_heapsort_branch():
    """The point of `deep`/`deepw`: libstdc++'s introsort runs out of depth on them (replayed with the library's own
    partition step inside the oracle), in every heap size class of the product's sort emulation."""
    for name, min_max in (("deep", 250_000), ("deepw", 3_000)):
        meta = json.load(open(os.path.join(bigcases.GOLD, name + ".meta.json")))
        hb = meta["heapsort_branch"]
        assert hb["heap_segments"] > 100 and hb["max_heap"] >= min_max, (name, hb)
