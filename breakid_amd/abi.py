"""numpy / ctypes mirror of include/breakid_hip.h (POD layouts and stage ids)."""
import ctypes as C

import numpy as np

BK_OK = 0
BK_ERR_ARG, BK_ERR_HIP, BK_ERR_NO_DEVICE, BK_ERR_UNSORTED, BK_ERR_CIGAR, BK_ERR_IO, BK_ERR_LIMIT, BK_ERR_COLLISION = -1, -2, -3, -4, -5, -6, -7, -8
BK_MEM_HOST, BK_MEM_DEVICE = 0, 1
STAGE_SCAN, STAGE_ISO, STAGE_CLUSTERED, STAGE_SPLITS, STAGE_CLUSTERS, STAGE_GROUP_KEYS = range(6)

PAIR = np.dtype([("x", "<u4"), ("y", "<u4"), ("p1_pos", "<u4"), ("p2_pos", "<u4"), ("p1_tid", "<i4"),
                 ("p2_tid", "<i4"), ("p1_flag", "<u2"), ("p2_flag", "<u2"), ("p1_mapq", "u1"), ("p2_mapq", "u1"),
                 ("p1_rev", "u1"), ("p2_rev", "u1"), ("rec", "<u8"), ("id", "<u4"), ("cluster", "<i4"),
                 ("group", "<u4"), ("reserved", "<u4")])
SPLIT = np.dtype([("rec", "<u8"), ("tid", "<i4"), ("pos", "<i4"), ("endpos", "<i4"), ("reserved", "<u4"), ("qhash", "<u8"),
                  ("prim_chr", "<i4"), ("sec_chr", "<i4"), ("prim_start", "<u4"), ("prim_end", "<u4"),
                  ("prim_bp", "<u4"), ("sec_start", "<u4"), ("sec_end", "<u4"), ("sec_bp", "<u4"),
                  ("prim_cigar", "<u8"), ("sec_cigar", "<u8"), ("flags", "<u4"), ("qcheck", "<u4")])
CLUSTER = np.dtype([("group", "<u4"), ("id", "<i4"), ("p1_tid", "<i4"), ("p2_tid", "<i4"), ("p1_mean", "<u4"),
                    ("p2_mean", "<u4"), ("p1_min", "<u4"), ("p1_max", "<u4"), ("p2_min", "<u4"), ("p2_max", "<u4"),
                    ("p1_exact", "<u4"), ("p2_exact", "<i4"), ("n_drp", "<u4"), ("n_sr", "<u4"), ("depth1", "<u4"),
                    ("depth2", "<u4"), ("type_mask", "<u4"), ("flags", "<u4")])
GROUP_KEY = np.dtype([("p1_tid", "<i4"), ("p2_tid", "<i4")])
assert PAIR.itemsize == 56 and SPLIT.itemsize == 88 and CLUSTER.itemsize == 72

STAGE_DTYPE = {STAGE_SCAN: PAIR, STAGE_ISO: PAIR, STAGE_CLUSTERED: PAIR, STAGE_SPLITS: SPLIT,
               STAGE_CLUSTERS: CLUSTER, STAGE_GROUP_KEYS: GROUP_KEY}


class Soa(C.Structure):
    _fields_ = [("n", C.c_uint64), ("tid", C.c_void_p), ("pos", C.c_void_p), ("mtid", C.c_void_p),
                ("mpos", C.c_void_p), ("isize", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p),
                ("qhash", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("aux_off", C.c_void_p), ("aux", C.c_void_p), ("n_cigar_words", C.c_uint64),
                ("n_aux_bytes", C.c_uint64), ("qcheck", C.c_void_p), ("side", C.c_void_p)]


SOA_COLS = [("tid", np.int32), ("pos", np.int32), ("mtid", np.int32), ("mpos", np.int32), ("isize", np.int32),
            ("flag", np.uint16), ("mapq", np.uint8), ("qhash", np.uint64), ("cigar_off", np.uint32),
            ("cigar", np.uint32), ("aux_off", np.uint32), ("aux", np.uint8)]


SOA_COLS_ALL = SOA_COLS + [("qcheck", np.uint32)]  # with the optional second read-name hash (the BAM decoders fill it)


def soa_from_numpy(cols) -> Soa:
    """cols: dict of contiguous numpy arrays (host).  The dict must outlive the returned struct."""
    s = Soa()
    s.n = len(cols["tid"])
    for name, dt in SOA_COLS:
        a = cols[name]
        assert a.dtype == dt and a.flags["C_CONTIGUOUS"], name
        setattr(s, name, a.ctypes.data if a.size else 0)
    assert len(cols["cigar_off"]) == s.n + 1 and len(cols["aux_off"]) == s.n + 1
    s.n_cigar_words = int(cols["cigar_off"][-1]) if s.n else 0
    s.n_aux_bytes = int(cols["aux_off"][-1]) if s.n else 0
    q = cols.get("qcheck")  # optional column: second hash of the read names
    if q is not None and s.n:
        assert q.dtype == np.uint32 and q.flags["C_CONTIGUOUS"] and len(q) == s.n
        s.qcheck = q.ctypes.data
    return s


def fetch_array(lib, handle, fn, stage):
    """Generic `*_fetch` caller -> (structured array copy, group offsets or None)."""
    data = C.c_void_p()
    count = C.c_uint64()
    goff = C.POINTER(C.c_uint64)()
    ng = C.c_uint32()
    rc = fn(handle, stage, C.byref(data), C.byref(count), C.byref(goff), C.byref(ng))
    if rc != 0:
        raise RuntimeError("fetch(stage=%d) failed rc=%d" % (stage, rc))
    dt = STAGE_DTYPE[stage]
    n = count.value
    if n:
        buf = (C.c_char * (n * dt.itemsize)).from_address(data.value)
        arr = np.frombuffer(buf, dtype=dt, count=n).copy()
    else:
        arr = np.zeros(0, dt)
    off = None
    if ng.value or bool(goff):
        off = np.ctypeslib.as_array(goff, shape=(ng.value + 1,)).copy() if bool(goff) else None
    return arr, off


class ShardStats(C.Structure):
    _fields_ = [("isize_sum", C.c_uint64), ("isize_n", C.c_uint64), ("sumsq", C.c_double), ("vmax", C.c_uint32),
                ("max_span", C.c_uint32), ("n_cand", C.c_uint64), ("n_split", C.c_uint64)]


BUF_CANDIDATES, BUF_TUPLES, BUF_CLUSTERS = 0, 1, 2


def device_ptrs(cols):
    """torch column dict (breakid_amd.synth_gpu) -> name -> device pointer, incl. the optional qcheck column"""
    p = {k: cols[k].data_ptr() for k, _ in SOA_COLS}
    if "qcheck" in cols:
        p["qcheck"] = cols["qcheck"].data_ptr()
    if "side" in cols:  # bk_side rows (include/breakid_hip.h): qhash, mtid, mpos, qcheck of a record in one 32-byte row
        p["side"] = cols["side"].data_ptr()
    return p
