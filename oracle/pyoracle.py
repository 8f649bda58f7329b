"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (see oracle/oracle.cc header).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

from breakid_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "port"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("BREAKID_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")  # sanitizer builds: oracle/_san/liboracle_*.so
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.ora_new.restype = C.c_void_p
        L.ora_new.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int]
        L.ora_free.argtypes = [C.c_void_p]
        L.ora_last_error.restype = C.c_char_p
        L.ora_last_error.argtypes = [C.c_void_p]
        L.ora_set_records.argtypes = [C.c_void_p, C.POINTER(abi.Soa)]
        L.ora_isize_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ora_discordant_pairs.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.ora_mask_and_cluster.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.ora_split_evidence.argtypes = [C.c_void_p]
        L.ora_cluster_summary.argtypes = [C.c_void_p, C.c_double]
        L.ora_split_breakpoints.argtypes = [C.c_void_p, C.c_double]
        L.ora_fetch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint32)]
        L.ora_unit_cigar.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_int,
                                     C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        L.ora_unit_ahc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_long, C.c_void_p, C.c_void_p]
        L.ora_unit_points.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_void_p,
                                      C.c_void_p, C.POINTER(C.c_int)]
        L.ora_unit_vote.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.ora_unit_std_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.ora_unit_region.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.ora_unit_depth.restype = C.c_uint32
        L.ora_unit_depth.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.ora_text_hash.restype = C.c_uint64
        L.ora_text_hash.argtypes = [C.c_char_p, C.c_size_t]
        L.ora_name_id.argtypes = [C.c_void_p, C.c_char_p]
        _LIB = L
    return _LIB


class Oracle:
    """Stage-by-stage CPU oracle over a SoA dict (see breakid_amd.synth.Dataset.to_soa)."""

    def __init__(self, contigs, cols):
        L = lib()
        self.L = L
        self.cols = {k: np.ascontiguousarray(v) for k, v in cols.items()}
        lens = np.asarray([l for _, l in contigs], dtype=np.uint32)
        names = (C.c_char_p * len(contigs))(*[n.encode() for n, _ in contigs])
        self.h = L.ora_new(lens.ctypes.data, names, len(contigs))
        self.soa = abi.soa_from_numpy(self.cols)
        L.ora_set_records(self.h, C.byref(self.soa))
        self.w = None

    def close(self):
        if self.h:
            self.L.ora_free(self.h)
            self.h = None

    def isize_stats(self):
        m, s = C.c_double(), C.c_double()
        self.L.ora_isize_stats(self.h, C.byref(m), C.byref(s))
        return m.value, s.value

    @staticmethod
    def w_from(mean, sd):
        import math
        times = 2
        return times * math.sqrt(times) * (mean + 3 * sd)  # BreakID.cc:103

    def discordant_pairs(self, qual, w):
        self.L.ora_discordant_pairs(self.h, qual, w)

    def mask_and_cluster(self, w, fast):
        self.L.ora_mask_and_cluster(self.h, w, int(fast))

    def split_evidence(self):
        self.L.ora_split_evidence(self.h)

    def cluster_summary(self, w):
        self.L.ora_cluster_summary(self.h, w)

    def split_breakpoints(self, w):
        rc = self.L.ora_split_breakpoints(self.h, w)
        return rc

    def run(self, qual=20, fast=True):
        mean, sd = self.isize_stats()
        w = self.w_from(mean, sd)
        self.w = w
        self.discordant_pairs(qual, w)
        self.mask_and_cluster(w, fast)
        self.split_evidence()
        self.cluster_summary(w)
        rc = self.split_breakpoints(w)
        return w, rc

    def fetch(self, stage):
        return abi.fetch_array(self.L, self.h, self.L.ora_fetch, stage)

    def region(self, tid, start, end, cap=4096):
        out = np.zeros(cap, abi.SPLIT)
        n = self.L.ora_unit_region(self.h, tid, start, end, out.ctypes.data, cap)
        return out[:min(n, cap)].copy()

    def depth(self, tid, pos):
        return int(self.L.ora_unit_depth(self.h, tid, pos))

    def name_id(self, name):
        return self.L.ora_name_id(self.h, name.encode())


def unit_cigar(kind, c1, c2, e):
    L = lib()
    out = (C.c_int * 5)()
    buf = C.create_string_buffer(4096)
    if kind == "t":
        L.ora_unit_cigar(0, c1.encode(), None, 0, c2.encode(), e, out, buf, 4096)
    else:
        w = np.asarray([int(t) for t in c1.split(",")], dtype=np.uint32)
        L.ora_unit_cigar(1, b"", w.ctypes.data, len(w), c2.encode(), e, out, buf, 4096)
    return buf.value.decode(), list(out)


def unit_ahc(x, y, T):
    L = lib()
    x = np.ascontiguousarray(x, np.uint32)
    y = np.ascontiguousarray(y, np.uint32)
    n = len(x)
    nodes = np.zeros((2 * n + 1, 5), np.int32)
    pts = np.zeros(n * (2 * n + 2) + 4, np.int32) if n < 3000 else np.zeros(n * 64, np.int32)
    nn = L.ora_unit_ahc(x.ctypes.data, y.ctypes.data, n, T, nodes.ctypes.data, pts.ctypes.data)
    nodes = nodes[:nn]
    out, p = [], 0
    for r in nodes:
        out.append((int(r[1]), int(r[2]), int(r[3]), int(r[4]), [int(v) for v in pts[p:p + r[2]]]))
        p += r[2]
    return out


def unit_points(mode, x, y, w):
    L = lib()
    x = np.ascontiguousarray(x, np.uint32)
    y = np.ascontiguousarray(y, np.uint32)
    n = len(x)
    ids = np.zeros(max(n, 1), np.uint32)
    cl = np.zeros(max(n, 1), np.int32)
    k = C.c_int()
    m = L.ora_unit_points({"mask": 0, "iso": 1, "fast": 2}[mode], x.ctypes.data, y.ctypes.data, n, float(w),
                          ids.ctypes.data, cl.ctypes.data, C.byref(k))
    return ids[:m].copy(), cl[:m].copy(), k.value


def unit_vote(s1, s2, p1_chr):
    L = lib()
    s1 = np.ascontiguousarray(s1, abi.SPLIT)
    s2 = np.ascontiguousarray(s2, abi.SPLIT)
    out = np.zeros(3, np.int32)
    L.ora_unit_vote(s1.ctypes.data, len(s1), s2.ctypes.data, len(s2), p1_chr, out.ctypes.data)
    return tuple(int(v) for v in out)


def unit_std_sort(key, group_off):
    L = lib()
    key = np.ascontiguousarray(key, np.uint32)
    group_off = np.ascontiguousarray(group_off, np.uint64)
    perm = np.zeros(len(key), np.uint32)
    L.ora_unit_std_sort(key.ctypes.data, group_off.ctypes.data, len(group_off) - 1, perm.ctypes.data)
    return perm


def sort_probe(on=True):
    """Diagnostic: count the segments libstdc++'s introsort hands to its heapsort branch in the oracle's sorts."""
    lib().ora_sort_probe(int(on))


def sort_stats():
    out = (C.c_uint64 * 4)()
    lib().ora_sort_stats(out)
    return {"sorts": int(out[0]), "heap_segments": int(out[1]), "heap_elems": int(out[2]), "max_heap": int(out[3])}
