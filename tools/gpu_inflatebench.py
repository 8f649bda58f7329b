"""GPU BGZF inflate vs zlib (GPU box): correctness on BAMs written at several zlib levels + stored blocks, and rate."""
import ctypes as C, os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("fb", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_feedbench.py"))
fb = importlib.util.module_from_spec(spec)
sys.argv = [sys.argv[0]] + sys.argv[1:]
spec.loader.exec_module(fb)
from breakid_amd import capi


def host_inflate(data):
    out, off = [], 0
    while off < len(data):
        xlen = struct.unpack_from("<H", data, off + 10)[0]
        bsize = struct.unpack_from("<H", data, off + 16)[0]
        out.append(zlib.decompress(data[off + 12 + xlen: off + bsize + 1 - 8], -15))
        off += bsize + 1
    return b"".join(out)


def rewrite(raw, level):
    """raw inflated stream -> BGZF at the given zlib level (0 = stored blocks)"""
    blocks = []
    for off in range(0, len(raw), 0xFF00):
        blk = raw[off:off + 0xFF00]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(blk) + c.flush()
        blocks.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(blk), len(blk)))
    blocks.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(blocks)


L = capi.lib()
L.bk_debug_bgzf_inflate.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.c_char_p, C.c_size_t]
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
path = "/tmp/inflate_%d.bam" % n_pairs
fb.write_bam(path, n_pairs)
base = open(path, "rb").read()
raw = host_inflate(base)
import torch  # noqa: F401  (device init order)
for level in [int(x) for x in os.environ.get("INFLATE_LEVELS", "1,6,9,0").split(",")]:
    data = base if level == 1 else rewrite(raw, level)
    t0 = time.perf_counter()
    ref = host_inflate(data) if level != 1 else raw
    t_host = time.perf_counter() - t0
    out = np.zeros(len(ref) + 16, np.uint8)
    olen, ms, err = C.c_uint64(), C.c_float(), C.create_string_buffer(256)
    src = np.frombuffer(data, np.uint8)
    for rep in range(2):
        rc = L.bk_debug_bgzf_inflate(src.ctypes.data, len(data), out.ctypes.data, len(out), C.byref(olen), C.byref(ms), err, 256)
    ok = rc == 0 and olen.value == len(ref) and out[:len(ref)].tobytes() == ref
    print("level %d: %s  file %.1f MB -> %.1f MB, GPU kernel %.3f ms = %.1f GB/s inflated (zlib one thread: %.2f s)" % (
        level, "OK" if ok else "MISMATCH rc=%d %s" % (rc, err.value), len(data) / 1e6, len(ref) / 1e6, ms.value, len(ref) / ms.value / 1e6, t_host), flush=True)
    if not ok and rc == 0:
        a = np.frombuffer(ref, np.uint8)
        d = np.nonzero(a != out[:len(ref)])[0]
        print("  first diffs at", d[:10], "of", len(d))
