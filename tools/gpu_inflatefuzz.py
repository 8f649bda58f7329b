"""GPU inflate against zlib on random deflate streams (GPU box): data shapes (uniform bytes, skewed alphabets, few letters, text,
runs, repeats at random distances, mixtures), zlib levels / strategies / window sizes / flush patterns, block sizes from 0 to
0xFF00.  Usage: python tools/gpu_inflatefuzz.py [cases] [seed]"""
import ctypes as C, os, struct, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakid_amd import capi

L = capi.lib()
L.bk_debug_bgzf_inflate.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.c_char_p, C.c_size_t]
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
import torch  # noqa: F401


def piece(n):
    k = int(rng.integers(0, 8))
    if k == 0:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if k == 1:
        p = 0.5 ** (np.arange(256) / float(rng.uniform(2, 12)))
        return rng.choice(256, n, p=p / p.sum()).astype(np.uint8).tobytes()
    if k == 2:
        return rng.integers(0, int(rng.integers(2, 6)), n, dtype=np.uint8).tobytes()
    if k == 3:
        return (b"read_%07d\tchr%d\t%d\t60\t151M\t=\t%d\t%d\n" % (int(rng.integers(0, 10**7)), int(rng.integers(1, 23)), int(rng.integers(1, 10**8)), int(rng.integers(1, 10**8)), int(rng.integers(100, 900)))) * (n // 40 + 1)
    if k == 4:
        return bytes([int(rng.integers(0, 256))]) * n
    if k == 5:
        unit = rng.integers(0, 256, int(rng.integers(1, 40000)), dtype=np.uint8).tobytes()
        return unit * (n // len(unit) + 1)
    if k == 6:
        q = rng.choice(np.asarray([2, 11, 25, 37, 37, 37, 37], np.uint8), n).tobytes()
        return q
    return (rng.integers(0, 16, n, dtype=np.uint8) * 17).astype(np.uint8).tobytes()


def block():
    n = int(rng.choice([0, 1, 2, 100, 5000, 30000, 0xFF00, int(rng.integers(1, 0xFF00))]))
    out = b""
    while len(out) < n:
        m = int(rng.integers(1, n - len(out) + 1))
        out += piece(m)[:m]
    return out[:n]


def deflate(b):
    level = int(rng.choice([0, 1, 3, 6, 9]))
    strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    wbits = -int(rng.choice([9, 12, 15, 15]))
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, int(rng.choice([1, 8, 9])), strat)
    if rng.random() < 0.3 and len(b) > 10:
        out, step = b"", max(1, len(b) // int(rng.integers(2, 7)))
        for i in range(0, len(b), step):
            out += c.compress(b[i:i + step]) + c.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH])))
        return out + c.flush()
    return c.compress(b) + c.flush()


bad = 0
for it in range(cases):
    blks = [block() for _ in range(int(rng.integers(1, 12)))]
    comp = []
    for b in blks:
        d = deflate(b)
        while len(d) + 26 > 65536:  # does not fit a BGZF block: store a shorter one
            b = b[:len(b) // 2]
            d = deflate(b)
        comp.append((b, d))
    raw = b"".join(b for b, _ in comp)
    data = b"".join(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(d) + 25) + d + struct.pack("<II", zlib.crc32(b), len(b)) for b, d in comp)
    data += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    src = np.frombuffer(data, np.uint8)
    out = np.zeros(len(raw) + 16, np.uint8)
    olen, ms, err = C.c_uint64(), C.c_float(), C.create_string_buffer(256)
    rc = L.bk_debug_bgzf_inflate(src.ctypes.data, len(data), out.ctypes.data, len(out), C.byref(olen), C.byref(ms), err, 256)
    if rc != 0 or olen.value != len(raw) or out[:len(raw)].tobytes() != raw:
        bad += 1
        print("case %d: rc=%d %s, %d blocks, %d bytes" % (it, rc, err.value, len(blks), len(raw)), flush=True)
    if it % 50 == 49:
        print("inflate fuzz: %d cases, %d bad" % (it + 1, bad), flush=True)
print("INFLATEFUZZ %d cases, %d bad" % (cases, bad))
