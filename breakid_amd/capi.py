"""ctypes binding of libbreakid_hip.so (include/breakid_hip.h).  The host-side mirror of the
reference's stage functions: names and argument meaning follow BreakID.cc's free functions
(get_mean_insert_size, scan_discordant_pairs, remove_isolated_pairs + find_cluster_pairs_enspan_*,
findClusterBreakPointInfoSaTag).  There is NO CPU fallback: without the built extension or without a
gfx950 device every entry point raises."""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

from . import abi

# the lanes of bk_mask_and_cluster and the chunk streams of the GPU feed want more hardware queues than ROCm's default of 4; the HIP
# runtime reads the variable when it starts, i.e. at the process's first GPU call - importing this module before that is enough
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbreakid_hip.so")
_LIB = None


class BreakIDError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libbreakid_hip error %d: %s" % (code, msg))
        self.code = code


def build(verbose=False):
    """Compile the HIP extension in-tree for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    r = subprocess.run(cmd, capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libbreakid_hip.so failed:\n" + (r.stdout or "")[-4000:] + (r.stderr or "")[-4000:])


EXPORTS = ["bk_init", "bk_prepare_process", "bk_free", "bk_last_error", "bk_set_stream", "bk_sync", "bk_get_stream", "bk_upload_records", "bk_isize_stats",
           "bk_discordant_pairs", "bk_mask_and_cluster", "bk_split_evidence", "bk_cluster_summary",
           "bk_split_breakpoints", "bk_run", "bk_fetch", "bk_timing", "bk_timing_enable", "bk_timing_touched", "bk_group_stats", "bk_qname_hash", "bk_qname_check",
           "bk_bam_open", "bk_bam_header", "bk_bam_decode", "bk_bam_close", "bk_bam_decode_device", "bk_bam_decode_device_part", "bk_bam_decode_device_ctx", "bk_bam_dev_free", "bk_feed_release_caches", "bk_debug_bgzf_inflate", "bk_debug_std_sort", "bk_debug_ahc", "bk_debug_points", "bk_debug_cigar", "bk_debug_vote", "bk_debug_region", "bk_shard_begin", "bk_shard_get_stats", "bk_shard_set_stats",
           "bk_shard_sd_local", "bk_shard_sd_finish", "bk_shard_buffer", "bk_shard_set_buffer", "bk_shard_group_sizes",
           "bk_shard_own_groups", "bk_shard_route_candidates", "bk_shard_group_keys", "bk_shard_route_pairs", "bk_shard_group_pairs", "bk_shard_bp_cov", "bk_shard_bp_vote", "bk_shard_bp_vote_slice", "bk_shard_bp_set_voted", "bk_shard_bp_depth", "bk_shard_bp_finish"]


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise BreakIDError(abi.BK_ERR_NO_DEVICE, "libbreakid_hip.so is not built (run __graft_entry__.build()); "
                                                     "there is no CPU fallback")
        try:
            # when PyTorch-ROCm lives in the same process it must load its HIP runtime first: two copies of
            # libamdhip64 (torch's bundled one and /opt/rocm's) cannot both own the device
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        vp, u64p, dp = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)
        L.bk_init.argtypes = [C.c_int, vp, C.POINTER(C.c_char_p), C.c_int, C.POINTER(vp)]
        L.bk_free.argtypes = [vp]
        L.bk_last_error.restype = C.c_char_p
        L.bk_last_error.argtypes = [vp]
        L.bk_set_stream.argtypes = [vp, vp]
        L.bk_sync.argtypes = [vp]
        L.bk_upload_records.argtypes = [vp, C.POINTER(abi.Soa), C.c_int]
        L.bk_isize_stats.argtypes = [vp, dp, dp]
        L.bk_discordant_pairs.argtypes = [vp, C.c_int, C.c_double, u64p, C.POINTER(C.c_uint32)]
        L.bk_mask_and_cluster.argtypes = [vp, C.c_double, C.c_int, u64p]
        L.bk_split_evidence.argtypes = [vp, u64p]
        L.bk_cluster_summary.argtypes = [vp, C.c_double, u64p]
        L.bk_split_breakpoints.argtypes = [vp, C.c_double, u64p]
        L.bk_run.argtypes = [vp, C.c_int, C.c_int, dp, u64p]
        L.bk_fetch.argtypes = [vp, C.c_int, C.POINTER(vp), u64p, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint32)]
        L.bk_timing.argtypes = [vp, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_float)),
                                C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_int)]
        L.bk_timing_enable.argtypes = [vp, C.c_int]
        L.bk_timing_touched.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_int)]
        L.bk_debug_std_sort.argtypes = [vp, vp, vp, C.c_uint32, vp]
        L.bk_debug_ahc.argtypes = [vp, vp, vp, C.c_uint32, C.c_double, vp, vp, C.POINTER(C.c_uint32)]
        L.bk_debug_points.argtypes = [vp, C.c_int, vp, vp, C.c_uint32, C.c_double, vp, vp, C.POINTER(C.c_uint32)]
        L.bk_debug_cigar.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp]
        L.bk_debug_vote.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_int32, C.c_int32, vp]
        L.bk_debug_region.argtypes = [vp, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint64, vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32)]
        L.bk_shard_begin.argtypes = [vp, C.c_uint64, C.c_int]
        L.bk_shard_get_stats.argtypes = [vp, C.POINTER(abi.ShardStats)]
        L.bk_shard_set_stats.argtypes = [vp, C.POINTER(abi.ShardStats)]
        L.bk_shard_sd_local.argtypes = [vp, u64p, C.POINTER(vp), u64p]
        L.bk_shard_sd_finish.argtypes = [vp, vp, C.c_uint64, C.c_uint64, dp, dp]
        L.bk_shard_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), u64p, C.POINTER(C.c_uint32)]
        L.bk_shard_set_buffer.argtypes = [vp, C.c_int, vp, C.c_uint64]
        L.bk_shard_group_sizes.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint32)]
        L.bk_shard_own_groups.argtypes = [vp, vp, C.c_uint32]
        L.bk_shard_route_candidates.argtypes = [vp, C.c_uint32, C.POINTER(vp), C.POINTER(C.POINTER(C.c_uint64))]
        L.bk_shard_group_keys.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32)]
        L.bk_shard_route_pairs.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp), C.POINTER(C.POINTER(C.c_uint64))]
        L.bk_shard_group_pairs.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32]
        L.bk_shard_bp_cov.argtypes = [vp, C.c_double, C.POINTER(vp), u64p]
        L.bk_shard_bp_vote.argtypes = [vp, C.c_double, vp]
        L.bk_shard_bp_vote_slice.argtypes = [vp, C.c_double, vp, C.c_uint64, C.c_uint64, C.POINTER(vp), C.POINTER(vp)]
        L.bk_shard_bp_set_voted.argtypes = [vp, vp]
        L.bk_shard_bp_depth.argtypes = [vp, C.POINTER(vp), u64p]
        L.bk_shard_bp_finish.argtypes = [vp, vp]
        L.bk_qname_hash.restype = C.c_uint64
        L.bk_qname_hash.argtypes = [C.c_char_p, C.c_size_t]
        L.bk_qname_check.restype = C.c_uint32
        L.bk_qname_check.argtypes = [C.c_char_p, C.c_size_t]
        L.bk_bam_open.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_size_t]
        L.bk_bam_header.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_uint32))]
        L.bk_bam_decode.argtypes = [vp, C.POINTER(abi.Soa), C.c_char_p, C.c_size_t]
        L.bk_bam_close.argtypes = [vp]
        L.bk_bam_decode_device.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(abi.Soa), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_char_p)),
                                           C.POINTER(C.POINTER(C.c_uint32)), C.c_char_p, C.c_size_t]
        L.bk_bam_decode_device_part.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(abi.Soa), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_char_p)),
                                           C.POINTER(C.POINTER(C.c_uint32)), C.c_char_p, C.c_size_t]
        L.bk_bam_dev_free.argtypes = [vp]
        L.bk_feed_release_caches.argtypes = []
        L.bk_feed_release_caches.restype = None
        L.bk_bam_decode_device_ctx.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_char_p)),
                                               C.POINTER(C.POINTER(C.c_uint32)), C.c_char_p, C.c_size_t]
        L.bk_debug_bgzf_inflate.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.c_char_p, C.c_size_t]
        _LIB = L
    return _LIB


def w_from(mean, sd):
    times = 2
    return times * math.sqrt(times) * (mean + 3 * sd)  # BreakID.cc:103


class Context:
    """One GPU context = the state the reference keeps in main() between BreakID.cc:93 and :167."""

    def __init__(self, contigs, device=0):
        self.L = lib()
        self.contigs = list(contigs)
        lens = np.asarray([l for _, l in contigs], dtype=np.uint32)
        names = (C.c_char_p * len(contigs))(*[n.encode() for n, _ in contigs])
        h = C.c_void_p()
        rc = self.L.bk_init(device, lens.ctypes.data, names, len(contigs), C.byref(h))
        if rc != 0:
            raise BreakIDError(rc, (self.L.bk_last_error(None) or b"").decode())
        self.h = h
        self._keep = None

    def close(self):
        if self.h:
            self.L.bk_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise BreakIDError(rc, (self.L.bk_last_error(self.h) or b"").decode())

    def set_stream(self, hip_stream_handle):
        self._check(self.L.bk_set_stream(self.h, C.c_void_p(hip_stream_handle)))

    def sync(self):
        self._check(self.L.bk_sync(self.h))

    def upload(self, cols):
        """cols: dict of numpy arrays (host) laid out as abi.SOA_COLS."""
        qc = cols.get("qcheck")
        cols = {k: np.ascontiguousarray(cols[k], dtype=dt) for k, dt in abi.SOA_COLS}
        if qc is not None:
            cols["qcheck"] = np.ascontiguousarray(qc, dtype=np.uint32)
        for k in ("cigar", "aux"):
            if cols[k].size == 0:
                cols[k] = np.zeros(1, cols[k].dtype)
        soa = abi.soa_from_numpy(cols)
        self._keep = (cols, soa)
        self._check(self.L.bk_upload_records(self.h, C.byref(soa), abi.BK_MEM_HOST))

    def attach_device_table(self, table):
        """table: DeviceBamTable; the columns are used in place (BK_MEM_DEVICE)"""
        self._keep = table
        self._check(self.L.bk_upload_records(self.h, C.byref(table.soa), abi.BK_MEM_DEVICE))

    def upload_soa(self, handle):
        """handle: BamTable from decode_bam(keep=True); the decoder's (pinned) columns go straight to bk_upload_records."""
        self._keep = handle
        self._check(self.L.bk_upload_records(self.h, C.byref(handle.soa), abi.BK_MEM_HOST))

    def attach_device(self, ptrs, n, n_cigar_words, n_aux_bytes):
        """ptrs: dict name -> device pointer (int) of columns already resident in HBM."""
        s = abi.Soa()
        s.n = n
        for name, _ in abi.SOA_COLS:
            setattr(s, name, ptrs[name])
        if ptrs.get("qcheck"):
            s.qcheck = ptrs["qcheck"]
        if ptrs.get("side"):
            s.side = ptrs["side"]
        s.n_cigar_words = n_cigar_words
        s.n_aux_bytes = n_aux_bytes
        self._keep = (ptrs, s)
        self._check(self.L.bk_upload_records(self.h, C.byref(s), abi.BK_MEM_DEVICE))

    def isize_stats(self):
        m, s = C.c_double(), C.c_double()
        self._check(self.L.bk_isize_stats(self.h, C.byref(m), C.byref(s)))
        return m.value, s.value

    def discordant_pairs(self, qual, w):
        n, g = C.c_uint64(), C.c_uint32()
        self._check(self.L.bk_discordant_pairs(self.h, qual, w, C.byref(n), C.byref(g)))
        return n.value, g.value

    def mask_and_cluster(self, w, fast):
        n = C.c_uint64()
        self._check(self.L.bk_mask_and_cluster(self.h, w, int(fast), C.byref(n)))
        return n.value

    def split_evidence(self):
        n = C.c_uint64()
        self._check(self.L.bk_split_evidence(self.h, C.byref(n)))
        return n.value

    def cluster_summary(self, w):
        n = C.c_uint64()
        self._check(self.L.bk_cluster_summary(self.h, w, C.byref(n)))
        return n.value

    def split_breakpoints(self, w, count=True):
        n = C.c_uint64()
        self._check(self.L.bk_split_breakpoints(self.h, w, C.byref(n) if count else None))
        return n.value

    def run(self, qual=20, fast=True):
        w, n = C.c_double(), C.c_uint64()
        self._check(self.L.bk_run(self.h, qual, int(fast), C.byref(w), C.byref(n)))
        return w.value, n.value

    def fetch(self, stage):
        def fn(h, st, d, c, g, ng):
            return self.L.bk_fetch(h, st, d, c, g, ng)
        try:
            return abi.fetch_array(self.L, self.h, fn, stage)
        except RuntimeError:
            self._check(-1 if not self.L.bk_last_error(self.h) else abi.BK_ERR_ARG)
            raise

    def debug_std_sort(self, key, group_off):
        key = np.ascontiguousarray(key, np.uint32)
        group_off = np.ascontiguousarray(group_off, np.uint64)
        perm = np.zeros(len(key), np.uint32)
        self._check(self.L.bk_debug_std_sort(self.h, key.ctypes.data, group_off.ctypes.data, len(group_off) - 1, perm.ctypes.data))
        return perm

    def debug_ahc(self, x, y, w):
        x = np.ascontiguousarray(x, np.uint32)
        y = np.ascontiguousarray(y, np.uint32)
        n = len(x)
        idx = np.zeros(max(n, 1), np.uint32)
        cl = np.zeros(max(n, 1), np.int32)
        m = C.c_uint32()
        self._check(self.L.bk_debug_ahc(self.h, x.ctypes.data, y.ctypes.data, n, float(w), idx.ctypes.data, cl.ctypes.data, C.byref(m)))
        return idx[:m.value].copy(), cl[:m.value].copy()

    def debug_points(self, mode, x, y, w):
        """mode: 'mask' | 'iso' | 'fast' (see include/breakid_hip.h: bk_debug_points)."""
        x = np.ascontiguousarray(x, np.uint32)
        y = np.ascontiguousarray(y, np.uint32)
        n = len(x)
        idx = np.zeros(max(n, 1), np.uint32)
        cl = np.zeros(max(n, 1), np.int32)
        m = C.c_uint32()
        self._check(self.L.bk_debug_points(self.h, {"mask": 0, "iso": 1, "fast": 2}[mode], x.ctypes.data, y.ctypes.data, n, float(w), idx.ctypes.data,
                                           cl.ctypes.data, C.byref(m)))
        return idx[:m.value].copy(), cl[:m.value].copy()

    def debug_cigar(self, rows):
        """rows: (kind 't'|'b', c1 text or list of BAM words, c2 text, e) -> int32 array (n, 6)."""
        kind = np.asarray([1 if r[0] == "b" else 0 for r in rows], np.uint8)
        c1 = bytearray()
        o1 = [0]
        for r in rows:
            while r[0] == "b" and len(c1) % 4:
                c1 += b"\0"            # word rows start 4-byte aligned (the padding belongs to no row)
                o1[-1] = len(c1)
            c1 += np.asarray(r[1], np.uint32).tobytes() if r[0] == "b" else r[1].encode()
            o1.append(len(c1))
        c2 = bytearray()
        o2 = [0]
        for r in rows:
            c2 += r[2].encode()
            o2.append(len(c2))
        # rows keep their own start: offsets are (start_i, end_i) pairs flattened as start of i = o[i], end of i = o[i+1] (padding shifted the start)
        o1a, o2a = np.asarray(o1, np.uint32), np.asarray(o2, np.uint32)
        b1 = np.frombuffer(bytes(c1) + b"\0" * 16, np.uint8)
        b2 = np.frombuffer(bytes(c2) + b"\0" * 16, np.uint8)
        e = np.asarray([r[3] for r in rows], np.int32)
        out = np.zeros((len(rows), 6), np.int32)
        self._check(self.L.bk_debug_cigar(self.h, len(rows), kind.ctypes.data, o1a.ctypes.data, b1.ctypes.data, o2a.ctypes.data, b2.ctypes.data, e.ctypes.data,
                                          out.ctypes.data))
        return out

    def debug_vote(self, s1, s2, p1_tid, p2_tid):
        s1 = np.ascontiguousarray(s1, abi.SPLIT)
        s2 = np.ascontiguousarray(s2, abi.SPLIT)
        out = np.zeros(3, np.int32)
        self._check(self.L.bk_debug_vote(self.h, s1.ctypes.data, len(s1), s2.ctypes.data, len(s2), p1_tid, p2_tid, out.ctypes.data))
        return tuple(int(v) for v in out)

    def debug_region(self, tid, start, end, depth_pos, cap=4096):
        out = np.zeros(cap, abi.SPLIT)
        n, cov, depth = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self.L.bk_debug_region(self.h, tid, start, end, depth_pos, out.ctypes.data, cap, C.byref(n), C.byref(cov), C.byref(depth)))
        return out[:min(n.value, cap)].copy(), n.value, cov.value, depth.value

    def timing_enable(self, on=True):
        self._check(self.L.bk_timing_enable(self.h, int(on)))

    def timing(self):
        names = C.POINTER(C.c_char_p)()
        ms = C.POINTER(C.c_float)()
        by = C.POINTER(C.c_uint64)()
        n = C.c_int()
        self._check(self.L.bk_timing(self.h, C.byref(names), C.byref(ms), C.byref(by), C.byref(n)))
        return [(names[i].decode(), float(ms[i]), int(by[i])) for i in range(n.value)]

    def timing_touched(self):
        """bytes the kernels of each timed stage load + store themselves (call after timing(); same order)"""
        t = C.POINTER(C.c_uint64)()
        n = C.c_int()
        self._check(self.L.bk_timing_touched(self.h, C.byref(t), C.byref(n)))
        return [int(t[i]) for i in range(n.value)]


class DeviceBamTable:
    """Record table decoded on the GPU (bk_bam_decode_device): device-resident columns owned by the library."""

    def __init__(self, L, h, soa, contigs):
        self.L, self.h, self.soa, self.contigs = L, h, soa, contigs

    def close(self):
        if self.h:
            self.L.bk_bam_dev_free(self.h)
            self.h = None


def decode_bam_device(path, device=0):
    """GPU BGZF inflate + BAM decode -> DeviceBamTable; raises BreakIDError(BK_ERR_IO) for files that are not block aligned."""
    L = lib()
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    s = abi.Soa()
    nt = C.c_int()
    names = C.POINTER(C.c_char_p)()
    lens = C.POINTER(C.c_uint32)()
    rc = L.bk_bam_decode_device(path.encode(), device, C.byref(h), C.byref(s), C.byref(nt), C.byref(names), C.byref(lens), err, 512)
    if rc != 0:
        raise BreakIDError(rc, err.value.decode())
    contigs = [(names[i].decode(), int(lens[i])) for i in range(nt.value)]
    return DeviceBamTable(L, h, s, contigs)


def decode_bam_device_part(path, part, parts, device=0):
    """The records of part `part` of `parts` of a block-aligned BAM (bk_bam_decode_device_part): one rank's record range of a
    sharded run, decoded on its own GPU."""
    L = lib()
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    s = abi.Soa()
    nt = C.c_int()
    names = C.POINTER(C.c_char_p)()
    lens = C.POINTER(C.c_uint32)()
    rc = L.bk_bam_decode_device_part(path.encode(), device, part, parts, C.byref(h), C.byref(s), C.byref(nt), C.byref(names), C.byref(lens), err, 512)
    if rc != 0:
        raise BreakIDError(rc, err.value.decode())
    contigs = [(names[i].decode(), int(lens[i])) for i in range(nt.value)]
    return DeviceBamTable(L, h, s, contigs)


def decode_bam_device_ctx(path, qual=20, device=0):
    """File -> device table with the stream pass of the hot path overlapped with the feed (bk_bam_decode_device_ctx):
    returns (Context with the table attached and the stream pass done, DeviceBamTable owning the columns)."""
    L = lib()
    hb, hc = C.c_void_p(), C.c_void_p()
    err = C.create_string_buffer(512)
    nt = C.c_int()
    names = C.POINTER(C.c_char_p)()
    lens = C.POINTER(C.c_uint32)()
    rc = L.bk_bam_decode_device_ctx(path.encode(), device, qual, C.byref(hb), C.byref(hc), C.byref(nt), C.byref(names), C.byref(lens), err, 512)
    if rc != 0:
        raise BreakIDError(rc, err.value.decode())
    contigs = [(names[i].decode(), int(lens[i])) for i in range(nt.value)]
    ctx = Context.__new__(Context)
    ctx.L, ctx.contigs, ctx.h = L, contigs, hc
    table = DeviceBamTable(L, hb, None, contigs)
    ctx._keep = table
    return ctx, table


class BamTable:
    """Decoded record table still owned by the C++ reader (pinned host columns); close() releases it."""

    def __init__(self, L, h, soa):
        self.L, self.h, self.soa = L, h, soa
        n = soa.n
        self.nbytes = n * (4 * 5 + 2 + 1 + 8) + (n + 1) * 8 + soa.n_cigar_words * 4 + soa.n_aux_bytes

    def close(self):
        if self.h:
            self.L.bk_bam_close(self.h)
            self.h = None


def decode_bam(path, keep=False):
    """C++ BGZF/BAM decoder -> (contigs, SoA dict of numpy copies); keep=True also returns the live BamTable."""
    L = lib()
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = L.bk_bam_open(path.encode(), C.byref(h), err, 512)
    if rc != 0:
        raise BreakIDError(rc, err.value.decode())
    try:
        nt = C.c_int()
        names = C.POINTER(C.c_char_p)()
        lens = C.POINTER(C.c_uint32)()
        L.bk_bam_header(h, C.byref(nt), C.byref(names), C.byref(lens))
        contigs = [(names[i].decode(), int(lens[i])) for i in range(nt.value)]
        s = abi.Soa()
        rc = L.bk_bam_decode(h, C.byref(s), err, 512)
        if rc != 0:
            raise BreakIDError(rc, err.value.decode())
        n = s.n
        sizes = {"cigar_off": n + 1, "aux_off": n + 1, "cigar": max(1, s.n_cigar_words), "aux": max(1, s.n_aux_bytes)}
        cols = {}
        for name, dt in abi.SOA_COLS_ALL:
            cnt = sizes.get(name, n)
            ptr = getattr(s, name)
            if cnt == 0 or not ptr:
                cols[name] = np.zeros(0, dt)
                continue
            buf = (C.c_char * (cnt * np.dtype(dt).itemsize)).from_address(ptr)
            cols[name] = np.frombuffer(buf, dtype=dt, count=cnt).copy()
        cols["cigar"] = cols["cigar"][: s.n_cigar_words]
        cols["aux"] = cols["aux"][: s.n_aux_bytes]
        if keep:
            t = BamTable(L, h, s)
            h = None
            return contigs, cols, t
        return contigs, cols
    finally:
        if h:
            L.bk_bam_close(h)
