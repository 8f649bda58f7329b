"""Pins the CPU oracle (oracle/oracle.cc) to golden vectors produced by the REAL reference
(tools/make_golden.py -> oracle/_ref/ref_harness, ref_units).  CPU only."""
import json
import os

import numpy as np
import pytest

from breakid_amd import abi
from oracle import pyoracle
from tests import refdump

DATASETS = ["g1", "g2", "small", "ties", "edge"]


@pytest.mark.parametrize("name", DATASETS)
@pytest.mark.parametrize("mode", ["fast", "ahc"])
def test_stages_match_reference(golden_dir, name, mode):
    contigs, cols = refdump.load_soa(golden_dir, name)
    dump = refdump.parse_stages(os.path.join(golden_dir, "%s.%s.stages.txt" % (name, mode)))
    o = pyoracle.Oracle(contigs, cols)
    mean, sd = o.isize_stats()
    w, rc = o.run(20, fast=(mode == "fast"))
    assert rc == 0
    refdump.compare_with_dump(dump, [n for n, _ in contigs], o.fetch, mean, sd, w)
    o.close()


def test_units_ahc(golden_dir):
    u = json.load(open(os.path.join(golden_dir, "units.json")))
    for case in u["ahc"]:
        nodes = pyoracle.unit_ahc(case["x"], case["y"], case["T"])
        lines = case["ref"].strip().split("\n")
        assert lines[0] == "nodes %d" % len(nodes)
        for i, (ln, nd) in enumerate(zip(lines[1:], nodes)):
            head, pts = ln.split("|")
            h = [int(v) for v in head.split()]
            assert h == [i, nd[0], nd[1], nd[2], nd[3]], (i, h, nd)
            assert [int(v) for v in pts.split()] == nd[4]


def test_units_cigar(golden_dir):
    u = json.load(open(os.path.join(golden_dir, "units.json")))["cigar"]
    ref = u["ref"].strip().split("\n")
    assert len(ref) == len(u["rows"])
    for row, exp in zip(u["rows"], ref):
        kind, c1, c2, e = row.split()
        s, vals = pyoracle.unit_cigar(kind, c1, c2, int(e))
        e_f = exp.split()
        assert [s] + [str(v) for v in vals] == e_f, (row, s, vals, exp)


def _parse_points(txt, with_cluster):
    lines = txt.strip().split("\n")
    n = int(lines[0].split()[1])
    rows = [l.split() for l in lines[1:1 + n]]
    if with_cluster:
        return [(int(r[0]), int(r[1])) for r in rows]
    return [int(r[0]) for r in rows]


def test_units_points(golden_dir):
    u = json.load(open(os.path.join(golden_dir, "units.json")))["points"]
    for case in u:
        ids, _, _ = pyoracle.unit_points("mask", case["x"], case["y"], case["w"])
        assert list(ids) == _parse_points(case["mask"], False)
        ids, _, _ = pyoracle.unit_points("iso", case["x"], case["y"], case["w"])
        assert list(ids) == _parse_points(case["iso"], False)
        if case["fast"]:
            fx = [p[0] for p in case["fast_in"]]
            fy = [p[1] for p in case["fast_in"]]
            ids, cl, k = pyoracle.unit_points("fast", fx, fy, case["w"])
            lines = case["fast"].strip().split("\n")
            assert lines[0] == "k %d" % k
            assert list(zip(ids.tolist(), cl.tolist())) == _parse_points("\n".join(lines[1:]), True)


def _tuples(rows, name_id):
    arr = np.zeros(len(rows), abi.SPLIT)
    L = pyoracle.lib()
    for i, r in enumerate(rows):
        f = r.split()
        arr[i]["qhash"] = L.ora_text_hash(f[0].encode(), len(f[0]))
        arr[i]["flags"] = int(f[1])
        arr[i]["prim_chr"] = name_id[f[2]]
        arr[i]["prim_start"], arr[i]["prim_end"] = int(f[3]), int(f[4])
        arr[i]["prim_cigar"] = L.ora_text_hash(f[5].encode(), len(f[5]))
        arr[i]["prim_bp"] = int(f[6])
        arr[i]["sec_chr"] = name_id[f[7]]
        arr[i]["sec_start"], arr[i]["sec_end"] = int(f[8]), int(f[9])
        arr[i]["sec_cigar"] = L.ora_text_hash(f[10].encode(), len(f[10]))
        arr[i]["sec_bp"] = int(f[11])
    return arr


def test_units_vote(golden_dir):
    u = json.load(open(os.path.join(golden_dir, "units.json")))["vote"]
    ids = {"chr1": 0, "chr2": 1}
    for case in u:
        got = pyoracle.unit_vote(_tuples(case["s1"], ids), _tuples(case["s2"], ids), ids[case["p1_chr"]])
        assert list(got) == [int(v) for v in case["ref"].split()], (case, got)


@pytest.mark.parametrize("name", DATASETS)
def test_region_queries_match_reference(golden_dir, name):
    """find_sa_reads / cal_single_base_depth on raw regions (htslib overlap predicate, region verdict, H6)."""
    contigs, cols = refdump.load_soa(golden_dir, name)
    names = [n for n, _ in contigs]
    o = pyoracle.Oracle(contigs, cols)
    L = pyoracle.lib()
    regions = json.load(open(os.path.join(golden_dir, name + ".regions.json")))
    assert regions
    n_nonempty = 0
    for r in regions:
        tid = names.index(r["chr"])
        got = o.region(tid, r["start"], r["end"])
        lines = r["sa"].strip().split("\n")
        assert lines[0] == "tuples %d" % len(got), (r["chr"], r["start"], r["end"], lines[0], len(got))
        exp = []
        for ln in lines[1:]:
            f = ln.split()
            exp.append((int(f[2]), o.name_id(f[3]), int(f[4]), int(f[5]), L.ora_text_hash(f[6].encode(), len(f[6])), int(f[7]),
                        o.name_id(f[8]), int(f[9]), int(f[10]), L.ora_text_hash(f[11].encode(), len(f[11])), int(f[12])))
        mine = [(int(t["flags"] & 1), int(t["prim_chr"]), int(t["prim_start"]), int(t["prim_end"]), int(t["prim_cigar"]), int(t["prim_bp"]),
                 int(t["sec_chr"]), int(t["sec_start"]), int(t["sec_end"]), int(t["sec_cigar"]), int(t["sec_bp"])) for t in got]
        assert sorted(exp) == sorted(mine)
        n_nonempty += bool(exp)
        assert float.fromhex(r["depth_at_start"]) == float(o.depth(tid, max(1, r["start"])))
    if name in ("g1", "small", "ties"):
        assert n_nonempty > 0
    o.close()


def test_error_cigar_exit_matches_reference(golden_dir):
    """the reference exits with -1 ("error cigar") on a clip-free complementary pair inside a queried region"""
    exp = json.load(open(os.path.join(golden_dir, "poison.json")))
    assert exp["returncode"] == 255 and "error cigar" in exp["stderr_tail"]
    contigs, cols = refdump.load_soa(golden_dir, "poison")
    o = pyoracle.Oracle(contigs, cols)
    w, rc = o.run(20, fast=True)
    assert rc == abi.BK_ERR_CIGAR
    o.close()
