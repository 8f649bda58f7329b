"""Shared helpers for the LARGE golden vectors (tools/make_golden_big.py): regenerate the seeded input, prove it is the
table the reference saw, and compare stage results with the reference's dumps / digests."""
import functools
import json
import os

from breakid_amd import fixtures
from tests import refdump

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@functools.lru_cache(maxsize=2)
def load(name):
    path = os.path.join(GOLD, name + ".meta.json")
    if not os.path.exists(path):
        return None, None
    meta = json.load(open(path))
    fx = fixtures.ALL[name]()
    assert refdump.soa_sha(fx.cols) == meta["soa_sha256"], "regenerated %s differs from the table the reference was run on" % name
    return fx, meta


def check(name, mode, fetch, mean, sd, w):
    """fetch(stage) -> (array, group_off) of the implementation under test."""
    fx, meta = load(name)
    names = [n for n, _ in fx.contigs]
    full = os.path.join(GOLD, "%s.%s.stages.txt.gz" % (name, mode))
    if os.path.exists(full):
        refdump.compare_with_dump(refdump.parse_stages(full), names, fetch, mean, sd, w)
        return "dump"
    exp = json.load(open(os.path.join(GOLD, "%s.%s.digest.json" % (name, mode))))
    # a digest taken from a run that went on to the calls also holds every cluster row (breakpoints, counts, depths, type)
    calls = any("clusters" in g for g in exp["groups"].values())
    got = refdump.digest_from_fetch(names, fetch, mean, sd, w, with_clusters=calls)
    assert got["order"] == exp["order"]
    assert (got["mean"], got["sd"], got["w"]) == (exp["mean"], exp["sd"], exp["w"])
    for key in exp["order"]:
        assert got["groups"][key] == exp["groups"][key], (name, mode, key, str(got["groups"][key])[:600], str(exp["groups"][key])[:600])
    return "digest+calls" if calls else "digest"


def expected_txt(name, mode):
    """sha256 / row counts of the reference's txt files and its five deterministic _performance.txt columns, where the digest holds them"""
    exp = json.load(open(os.path.join(GOLD, "%s.%s.digest.json" % (name, mode))))
    return exp.get("txt_sha256"), exp.get("txt_rows"), exp.get("perf5")
