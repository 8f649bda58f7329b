"""Parser for the stage dumps written by oracle/harness/ref_harness.cc (`stages` sub-command) and
helpers that turn oracle / product stage arrays into the same comparable form."""
import json
import os

import numpy as np

TYPE_NAMES = [(8, "Deletion"), (4, "Duplication"), (2, "Inversion"), (1, "Translocation")]  # BreakID.cc:1888-1907


def fusion_type(mask):
    for bit, name in TYPE_NAMES:
        if mask & bit:
            return name
    return "Unknown"


def parse_stages(path):
    out = {"groups": {}, "order": []}
    cur = None
    if path.endswith(".gz"):
        import gzip
        with gzip.open(path, "rt") as f:
            lines = f.read().split("\n")
    else:
        with open(path) as f:
            lines = f.read().split("\n")
    i = 0
    while i < len(lines):
        t = lines[i].split()
        i += 1
        if not t:
            continue
        if t[0] == "insert":
            out["mean"], out["sd"], out["w"] = (float.fromhex(v) for v in t[1:4])
        elif t[0] in ("scan", "iso", "clustered"):
            key, n = t[1], int(t[2])
            g = out["groups"].setdefault(key, {})
            if key not in out["order"]:
                out["order"].append(key)
            rows = []
            for _ in range(n):
                r = lines[i].split()
                i += 1
                rows.append(r)
            if t[0] == "scan":
                g["scan"] = [dict(id=r[0], p1_flag=int(r[1]), p2_flag=int(r[2]), p1_chr=r[3], p2_chr=r[4],
                                  p1_pos=int(r[5]), p2_pos=int(r[6]), p1_mapq=int(r[7]), p2_mapq=int(r[8]),
                                  s1=r[9], s2=r[10], x=int(r[11]), y=int(r[12]), qname=r[13]) for r in rows]
            elif t[0] == "iso":
                g["iso"] = [int(r[0].split("_")[-1]) for r in rows]
            else:
                g["clustered"] = [(int(r[0].split("_")[-1]), int(r[13])) for r in rows]
        elif t[0] == "roots":
            pass
        elif t[0] == "clusters":
            key, n = t[1], int(t[2])
            g = out["groups"].setdefault(key, {})
            rows = []
            for _ in range(n):
                r = lines[i].split()
                i += 1
                rows.append(dict(id=int(r[0]), p1_chr=r[1], p2_chr=r[2], p1_mean=int(r[3]), p2_mean=int(r[4]),
                                 p1_min=int(r[5]), p1_max=int(r[6]), p2_min=int(r[7]), p2_max=int(r[8]),
                                 p1_exact=int(r[9]), p2_exact=int(r[10]), n_drp=int(r[11]), n_sr=int(r[12]),
                                 type=r[13], depth1=float.fromhex(r[14]), depth2=float.fromhex(r[15]),
                                 af1=float.fromhex(r[16]), af2=float.fromhex(r[17])))
            g["clusters"] = rows
    return out


def tid_name(names, tid):
    return "*" if tid < 0 else names[tid]


def compare_with_dump(dump, names, fetch, mean, sd, w):
    """fetch(stage) -> (array, group_off).  Raises AssertionError on the first difference."""
    from breakid_amd import abi
    assert (mean, sd, w) == (dump["mean"], dump["sd"], dump["w"]), ((mean, sd, w), dump["mean"], dump["sd"], dump["w"])
    keys, _ = fetch(abi.STAGE_GROUP_KEYS)
    gnames = [tid_name(names, int(k["p1_tid"])) + "_" + tid_name(names, int(k["p2_tid"])) for k in keys]
    assert gnames == dump["order"], (gnames, dump["order"])
    scan, soff = fetch(abi.STAGE_SCAN)
    iso, ioff = fetch(abi.STAGE_ISO)
    clu, coff = fetch(abi.STAGE_CLUSTERED)
    for g, key in enumerate(gnames):
        ref = dump["groups"][key]
        rows = scan[soff[g]:soff[g + 1]]
        assert len(rows) == len(ref["scan"]), (key, len(rows), len(ref["scan"]))
        for k, (a, b) in enumerate(zip(rows, ref["scan"])):
            got = dict(p1_flag=int(a["p1_flag"]), p2_flag=int(a["p2_flag"]), p1_chr=tid_name(names, int(a["p1_tid"])),
                       p2_chr=tid_name(names, int(a["p2_tid"])), p1_pos=int(a["p1_pos"]), p2_pos=int(a["p2_pos"]),
                       p1_mapq=int(a["p1_mapq"]), p2_mapq=int(a["p2_mapq"]), s1="-" if a["p1_rev"] else "+",
                       s2="-" if a["p2_rev"] else "+", x=int(a["x"]), y=int(a["y"]))
            exp = {k2: b[k2] for k2 in got}
            assert got == exp, (key, k, got, exp)
            assert int(a["id"]) == k and int(a["group"]) == g
        got_iso = [int(v) for v in iso[ioff[g]:ioff[g + 1]]["id"]]
        assert got_iso == ref.get("iso", []), (key, "iso", got_iso[:20], ref.get("iso", [])[:20])
        got_cl = [(int(r["id"]), int(r["cluster"])) for r in clu[coff[g]:coff[g + 1]]]
        assert got_cl == ref.get("clustered", []), (key, "clustered", got_cl[:20], ref.get("clustered", [])[:20])
    cl, _ = fetch(abi.STAGE_CLUSTERS)
    valid = cl[(cl["flags"] & 2) != 0]
    exp_rows = []
    for g, key in enumerate(gnames):
        for r in dump["groups"][key].get("clusters", []):
            exp_rows.append((g, r))
    assert len(valid) == len(exp_rows), (len(valid), len(exp_rows))
    for a, (g, r) in zip(valid, exp_rows):
        got = dict(id=int(a["id"]), p1_chr=tid_name(names, int(a["p1_tid"])), p2_chr=tid_name(names, int(a["p2_tid"])),
                   p1_mean=int(a["p1_mean"]), p2_mean=int(a["p2_mean"]), p1_min=int(a["p1_min"]), p1_max=int(a["p1_max"]),
                   p2_min=int(a["p2_min"]), p2_max=int(a["p2_max"]), p1_exact=int(a["p1_exact"]), p2_exact=int(a["p2_exact"]),
                   n_drp=int(a["n_drp"]), n_sr=int(a["n_sr"]), type=fusion_type(int(a["type_mask"])),
                   depth1=float(a["depth1"]), depth2=float(a["depth2"]),
                   af1=float(np.float32(a["n_sr"]) / np.float32(a["depth1"])),
                   af2=float(np.float32(a["n_sr"]) / np.float32(a["depth2"])))
        assert int(a["group"]) == g and got == r, (got, r)
    return True


def load_soa(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".soa.npz"))
    cols = {k: z[k] for k in z.files if k not in ("names",)}
    names = [str(s) for s in z["names"]]
    contigs = list(zip(names, [int(v) for v in cols["target_len"]]))
    return contigs, cols


# ---- digests of large stage dumps (tests/golden/*.digest.json, made by tools/make_golden_big.py) ----------------
def _sha(rows):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(np.asarray(rows, dtype=np.int64)).tobytes()).hexdigest()


SCAN_FIELDS = ("p1_flag", "p2_flag", "p1_pos", "p2_pos", "p1_mapq", "p2_mapq", "x", "y")


def digest_from_dump(dump):
    """Canonical per-group digests of a parsed reference stage dump (what a *.digest.json holds)."""
    out = {"mean": float(dump["mean"]).hex(), "sd": float(dump["sd"]).hex(), "w": float(dump["w"]).hex(), "order": list(dump["order"]), "groups": {}}
    for key in dump["order"]:
        g = dump["groups"][key]
        scan = [[r[k] for k in SCAN_FIELDS] + [int(r["s1"] == "-"), int(r["s2"] == "-")] for r in g["scan"]]
        d = {"n_scan": len(scan), "scan": _sha(scan), "n_iso": len(g.get("iso", [])), "iso": _sha(g.get("iso", [])),
             "n_clustered": len(g.get("clustered", [])), "clustered": _sha(g.get("clustered", []))}
        if "clusters" in g:
            d["clusters"] = [[r["id"], r["p1_mean"], r["p2_mean"], r["p1_min"], r["p1_max"], r["p2_min"], r["p2_max"], r["p1_exact"], r["p2_exact"],
                              r["n_drp"], r["n_sr"], r["type"], int(r["depth1"]), int(r["depth2"])] for r in g["clusters"]]
        out["groups"][key] = d
    return out


def digest_from_fetch(names, fetch, mean, sd, w, with_clusters=False):
    """The same digests from product / oracle stage arrays (fetch(stage) -> (array, group_off))."""
    from breakid_amd import abi
    keys, _ = fetch(abi.STAGE_GROUP_KEYS)
    gnames = [tid_name(names, int(k["p1_tid"])) + "_" + tid_name(names, int(k["p2_tid"])) for k in keys]
    out = {"mean": float(mean).hex(), "sd": float(sd).hex(), "w": float(w).hex(), "order": gnames, "groups": {}}
    scan, soff = fetch(abi.STAGE_SCAN)
    iso, ioff = fetch(abi.STAGE_ISO)
    clu, coff = fetch(abi.STAGE_CLUSTERED)
    cl = None
    if with_clusters:
        cl, _ = fetch(abi.STAGE_CLUSTERS)
        cl = cl[(cl["flags"] & 2) != 0]
    for g, key in enumerate(gnames):
        a = scan[soff[g]:soff[g + 1]]
        rows = np.stack([a[k].astype(np.int64) for k in SCAN_FIELDS] + [a["p1_rev"].astype(np.int64), a["p2_rev"].astype(np.int64)], axis=1) if len(a) else []
        i = iso[ioff[g]:ioff[g + 1]]
        c = clu[coff[g]:coff[g + 1]]
        d = {"n_scan": len(a), "scan": _sha(rows), "n_iso": len(i), "iso": _sha(i["id"].astype(np.int64)),
             "n_clustered": len(c), "clustered": _sha(np.stack([c["id"].astype(np.int64), c["cluster"].astype(np.int64)], axis=1) if len(c) else [])}
        if with_clusters:
            rows = cl[cl["group"] == g]
            if len(i) >= 2:
                d["clusters"] = [[int(r["id"]), int(r["p1_mean"]), int(r["p2_mean"]), int(r["p1_min"]), int(r["p1_max"]), int(r["p2_min"]), int(r["p2_max"]),
                                  int(r["p1_exact"]), int(r["p2_exact"]), int(r["n_drp"]), int(r["n_sr"]), fusion_type(int(r["type_mask"])),
                                  int(r["depth1"]), int(r["depth2"])] for r in rows]
        out["groups"][key] = d
    return out


def soa_sha(cols):
    """Digest of the input table itself: the regenerated input must be the one the reference saw."""
    import hashlib
    h = hashlib.sha256()
    for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "mapq", "qhash", "cigar_off", "cigar", "aux_off", "aux"):
        h.update(np.ascontiguousarray(cols[k]).tobytes())
    return h.hexdigest()
