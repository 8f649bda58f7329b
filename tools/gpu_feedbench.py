"""Feed-path measurement (run on the GPU box): synthetic coordinate-sorted BAM -> bk_bam_open/decode (BGZF inflate +
record decode on the host cores, pinned columns) -> bk_upload_records(BK_MEM_HOST) -> bk_run.  Prints the rate of
every stage so DESIGN.md can quote the PCIe-inclusive number next to the HBM-resident one."""
import os, struct, sys, time, zlib
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # as bench.py and the command line: one hardware queue per feed slot (read when the runtime starts)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def write_bam(path, n_pairs, seed=7, read_len=150, aligned=True):
    rng = np.random.default_rng(seed)
    contigs = [("chr1", 249250621), ("chr2", 243199373), ("chr3", 198022430)]
    lens = np.asarray([l for _, l in contigs])
    n = 2 * n_pairs
    tid = rng.integers(0, 3, n_pairs)
    pos = (rng.random(n_pairs) * (lens[tid] - 2000)).astype(np.int64)
    ins = np.clip(np.round(rng.normal(350, 40, n_pairs)), read_len + 1, None).astype(np.int64)
    disc = rng.random(n_pairs) < 0.05
    mt = np.where(disc, rng.integers(0, 3, n_pairs), tid)
    mp = np.where(disc, (rng.random(n_pairs) * (lens[mt] - 2000)).astype(np.int64), pos + ins - read_len)
    l_name = 20
    rec_len = 32 + l_name + 4 + (read_len + 1) // 2 + read_len
    dt = np.dtype([("bs", "<u4"), ("tid", "<i4"), ("pos", "<i4"), ("l_name", "u1"), ("mapq", "u1"), ("bin", "<u2"), ("ncig", "<u2"), ("flag", "<u2"), ("lseq", "<u4"),
                   ("mtid", "<i4"), ("mpos", "<i4"), ("isize", "<i4"), ("name", "S%d" % l_name), ("cigar", "<u4"), ("seq", "u1", ((read_len + 1) // 2,)), ("qual", "u1", (read_len,))])
    assert dt.itemsize == rec_len + 4
    a = np.zeros(n, dt)
    names = np.char.add("read_", np.char.zfill(np.arange(n_pairs).astype("U14"), 14)).astype("S%d" % l_name)
    for half, (t, p, t2, p2, fl, sgn) in enumerate([(tid, pos, mt, mp, 0x63, 1), (mt, mp, tid, pos, 0x93, -1)]):
        v = a[half::2]
        v["tid"], v["pos"], v["mtid"], v["mpos"] = t, p, t2, p2
        v["flag"] = np.where(disc, (fl & ~0x2), fl)
        v["isize"] = np.where(disc, 0, sgn * ins)
        v["name"] = names
    a["bs"] = rec_len; a["l_name"] = l_name; a["mapq"] = 60; a["ncig"] = 1; a["lseq"] = read_len; a["cigar"] = read_len << 4
    a["seq"] = rng.integers(0, 256, (n, (read_len + 1) // 2), dtype=np.uint8) & 0x77 | 0x11
    a["qual"] = rng.choice(np.asarray([2, 11, 25, 37, 37, 37, 37], np.uint8), (n, read_len))
    order = np.lexsort((a["pos"], a["tid"]))
    a = a[order]
    text = b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % (nm.encode(), ln) for nm, ln in contigs)
    hdr = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(contigs))
    for nm, ln in contigs:
        hdr += struct.pack("<I", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<I", ln)
    body = a.tobytes()
    raw = hdr + body
    if aligned:   # blocks as htslib writes them: header flushed, whole records per block
        per = (0xFF00 // (rec_len + 4)) * (rec_len + 4)
        chunks = [hdr] + [body[o:o + per] for o in range(0, len(body), per)]
    else:
        chunks = [raw[o:o + 0xFF00] for o in range(0, len(raw), 0xFF00)]
    with open(path, "wb") as f:
        for blk in chunks:
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            comp = c.compress(blk) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(blk), len(blk)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
        f.flush()
        os.fsync(f.fileno())  # a file at rest in the page cache: the timed reads do not run beside the write-back of a GB of dirty pages
    return n, len(raw), os.path.getsize(path)


if __name__ == "__main__":
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    path = "/tmp/feed_%d.bam" % n_pairs
    t = time.time()
    n, raw, comp = write_bam(path, n_pairs)
    print("wrote %s: %d records, %.1f MB inflated, %.1f MB file (%.1f s)" % (path, n, raw / 1e6, comp / 1e6, time.time() - t), flush=True)
    import torch
    from breakid_amd import abi, capi
    for threads in ("1", ""):
        if threads:
            os.environ["BREAKID_THREADS"] = threads
        else:
            os.environ.pop("BREAKID_THREADS", None)
        os.environ["BK_DEBUG"] = "feed"
        t0 = time.perf_counter()
        contigs, cols, handle = capi.decode_bam(path, keep=True)
        t1 = time.perf_counter()
        print("threads=%s: open+decode %.3f s -> %.2f M records/s, %.1f MB/s of BAM" % (threads or "all", t1 - t0, n / (t1 - t0) / 1e6, comp / (t1 - t0) / 1e6), flush=True)
        if not threads:
            ctx = capi.Context(contigs)
            for rep in range(3):
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                ctx.upload_soa(handle)
                ctx.sync()
                t3 = time.perf_counter()
                w, nv = ctx.run(qual=20, fast=True)
                ctx.sync()
                t4 = time.perf_counter()
                print("  upload (pinned H2D) %.1f ms = %.1f GB/s, %.1f M records/s; run %.1f ms; upload+run %.1f M records/s" % (
                    (t3 - t2) * 1e3, handle.nbytes / (t3 - t2) / 1e9, n / (t3 - t2) / 1e6, (t4 - t3) * 1e3, n / (t4 - t2) / 1e6), flush=True)
            ctx.close()
        handle.close()
    # the same file through the GPU decoder (file image -> HBM, inflate + record decode on the device)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        table = capi.decode_bam_device(path)
        t1 = time.perf_counter()
        ctx = capi.Context(table.contigs)
        ctx.attach_device_table(table)
        w, nv = ctx.run(qual=20, fast=True)
        ctx.sync()
        t2 = time.perf_counter()
        print("GPU decoder: file -> device table %.3f s = %.1f M records/s, %.0f MB/s of BAM; run %.1f ms" % (t1 - t0, n / (t1 - t0) / 1e6, comp / (t1 - t0) / 1e6, (t2 - t1) * 1e3), flush=True)
        ctx.close()
        table.close()
    # feed and stream pass overlapped (bk_bam_decode_device_ctx): k_stream runs on the chunks already decoded
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx, table = capi.decode_bam_device_ctx(path, qual=20)
        ctx.sync()
        t1 = time.perf_counter()
        w, nv = ctx.run(qual=20, fast=True)
        ctx.sync()
        t2 = time.perf_counter()
        print("GPU decoder + stream pass overlapped: file -> device table AND candidates/sums %.3f s = %.1f M records/s, %.0f MB/s of BAM; rest of the hot path %.1f ms; "
              "file -> calls %.3f s = %.1f M records/s" % (t1 - t0, n / (t1 - t0) / 1e6, comp / (t1 - t0) / 1e6, (t2 - t1) * 1e3, t2 - t0, n / (t2 - t0) / 1e6), flush=True)
        ctx.close()
        table.close()
    # the same records in fixed-size blocks (records across BGZF blocks, as htsjdk / Picard write them): one-batch variant
    path2 = "/tmp/feed_%d_across.bam" % n_pairs
    n2, raw2, comp2 = write_bam(path2, n_pairs, aligned=False)
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        table = capi.decode_bam_device(path2)
        t1 = time.perf_counter()
        print("GPU decoder, records across blocks: file -> device table %.3f s = %.1f M records/s, %.0f MB/s of BAM" % (t1 - t0, n2 / (t1 - t0) / 1e6, comp2 / (t1 - t0) / 1e6), flush=True)
        table.close()
