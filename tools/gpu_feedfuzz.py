"""Randomised check of the GPU feed (GPU box): BAMs with random record shapes, block layouts and deflate settings through
bk_bam_decode_device against the generator's own table and the host decoder.  usage: gpu_feedfuzz.py [cases] [seed]"""
import os, struct, sys, tempfile, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from breakid_amd import abi, bamio, capi, synth
from breakid_amd.sharded import tensor_from_ptr


def device_cols(table):
    dev = torch.device("cuda", 0)
    s = table.soa
    n = s.n
    sizes = {"cigar_off": n + 1, "aux_off": n + 1, "cigar": s.n_cigar_words, "aux": s.n_aux_bytes}
    out = {}
    for name, dt in abi.SOA_COLS_ALL:
        cnt = sizes.get(name, n)
        nb = cnt * np.dtype(dt).itemsize
        out[name] = tensor_from_ptr(getattr(s, name), nb, dev).cpu().numpy().view(dt).copy() if nb else np.zeros(0, dt)
    return out


class Writer(bamio.BgzfWriter):
    """block sizes, deflate level and strategy drawn per block"""
    def __init__(self, path, rng, max_block):
        super().__init__(path)
        self.rng, self.max_block = rng, max_block

    def write(self, data):
        self.buf += data
        while len(self.buf) >= self.max_block:
            k = self.max_block if self.rng.random() < 0.7 else int(self.rng.integers(1, self.max_block + 1))
            self._flush_block(bytes(self.buf[:k]))
            del self.buf[:k]

    def write_record(self, rec):
        if len(self.buf) + len(rec) > self.max_block:
            self.flush()
        self.buf += rec   # (a record longer than a block goes out across blocks, as in htslib)
        while len(self.buf) >= self.max_block:
            self._flush_block(bytes(self.buf[:self.max_block]))
            del self.buf[:self.max_block]

    def _flush_block(self, data):
        r = self.rng
        level = int(r.choice([0, 1, 1, 4, 6, 6, 9]))
        strat = int(r.choice([zlib.Z_DEFAULT_STRATEGY] * 5 + [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
        c = zlib.compressobj(level, zlib.DEFLATED, -15, int(r.integers(1, 10)), strat)
        comp = b""
        if len(data) > 4 and r.random() < 0.2:   # several deflate blocks in one BGZF block
            cut = int(r.integers(1, len(data)))
            comp = c.compress(data[:cut]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(data[cut:]) + c.flush()
        else:
            comp = c.compress(data) + c.flush()
        if len(comp) + 26 > 65536:   # incompressible at this setting: store
            c = zlib.compressobj(0, zlib.DEFLATED, -15)
            comp = c.compress(data) + c.flush()
        hdr = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, len(comp) + 25)
        self.f.write(hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def one(case, seed):
    rng = np.random.default_rng(seed)
    ncontig = int(rng.integers(1, 6))
    contigs = [("c%d" % i, int(rng.integers(200_000, 3_000_000))) for i in range(ncontig)]
    n_rec = int(rng.choice([50, 400, 3000, 20000]))
    ds = synth.make_cfg(int(rng.integers(1 << 30)), contigs, n_rec, int(rng.integers(1, 12)), int(rng.integers(2, 30)), int(rng.integers(0, 40)),
                        jitter=int(rng.integers(10, 500)), read_len=int(rng.choice([36, 100, 150, 250])))
    for i in range(0, len(ds.recs), int(rng.integers(3, 400))):
        ds.recs[i].sa = "c0,%d,+,40S60M,60,0;" % (100 + i)
        if rng.random() < 0.5:
            ds.recs[i].oc = "60M40S"
    ref = ds.to_soa()
    aligned = bool(rng.random() < 0.5)
    long_every = int(rng.choice([0, 0, 97, 13]))
    seq_mode = int(rng.integers(0, 3))
    max_block = int(rng.choice([0xFF00, 0xFF00, 4096, 20000, 65280]))

    def gen():
        for i, r in enumerate(ds.recs):
            aux = ([("SA", r.sa)] if r.sa else []) + ([("OC", r.oc)] if r.oc else [])
            if long_every and i % long_every == 5:
                sl = int(rng.integers(30_000, 250_000))
            else:
                sl = [0, 150, int(rng.integers(0, 600))][seq_mode]
            rec = bamio.encode_record(r.qname, r.flag, r.tid, r.pos, r.mapq, bamio.parse_cigar(r.cigar), r.mtid, r.mpos, r.isize, aux, seq_len=sl)
            if sl and rng.random() < 0.5:   # random bases / qualities instead of the writer's constant fill
                body = bytearray(rec)
                at = len(rec) - sum(3 + len(v) + 1 for _, v in aux) - ((sl + 1) // 2 + sl)
                body[at:at + (sl + 1) // 2 + sl] = rng.integers(0, 64, (sl + 1) // 2 + sl, dtype=np.uint8).tobytes()
                rec = bytes(body)
            yield rec

    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "f.bam")
        w = Writer(p, rng, max_block)
        text = ("@HD\tVN:1.4\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs) + "@CO\t" + "x" * int(rng.choice([0, 10, 70000, 200000])) + "\n").encode()
        hdr = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(contigs))
        for nm, ln in contigs:
            nb = nm.encode() + b"\0"
            hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln)
        w.write(hdr)
        if aligned:
            w.flush()
            for r in gen():
                w.write_record(r)
        else:
            for r in gen():
                w.write(r)
        w.close()
        # records across blocks: in chunks (the default) or as one batch
        if not aligned and not os.environ.get("BREAKID_FEED_PACKED_CHUNKS") and rng.random() < 0.4:
            os.environ["BREAKID_FEED_PACKED_BATCH"] = "1"
        else:
            os.environ.pop("BREAKID_FEED_PACKED_BATCH", None)
        if rng.random() < 0.5:
            os.environ["BREAKID_FEED_CHUNK_MB"] = repr(float(rng.choice([0.07, 0.2, 1.0])))
        else:
            os.environ.pop("BREAKID_FEED_CHUNK_MB", None)
        hc, hcols = capi.decode_bam(p)
        try:
            table = capi.decode_bam_device(p)
        except capi.BreakIDError as e:
            keep = "/tmp/feedfuzz_fail_%d.bam" % case
            os.replace(p, keep)
            print("ERROR case %d seed %d (aligned=%s long_every=%d seq_mode=%d max_block=%d chunk=%s packed_chunks=%s): %s -> %s" % (
                case, seed, aligned, long_every, seq_mode, max_block, os.environ.get("BREAKID_FEED_CHUNK_MB"), os.environ.get("BREAKID_FEED_PACKED_CHUNKS"), e, keep), flush=True)
            # records across blocks: the decoder may refuse a layout whose boundary guesses do not chain (BK_ERR_IO: the caller
            # takes the host decoder) - a refusal is allowed, a wrong table is not
            if not aligned and e.code == abi.BK_ERR_IO and "could not be established" in str(e):
                global REFUSED
                REFUSED += 1
                return True
            return False
        got = device_cols(table)
        ok = table.contigs == contigs == hc
        for k, _ in abi.SOA_COLS_ALL:
            ok = ok and np.array_equal(got[k], ref[k]) and np.array_equal(got[k], hcols[k])
        table.close()
        if not ok:
            keep = "/tmp/feedfuzz_fail_%d.bam" % case
            os.replace(p, keep)
            print("MISMATCH case %d seed %d (aligned=%s long_every=%d seq_mode=%d max_block=%d) -> %s" % (case, seed, aligned, long_every, seq_mode, max_block, keep), flush=True)
        return ok


REFUSED = 0

if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = 0
    for c in range(cases):
        bad += not one(c, seed0 * 100003 + c)
        if c % 20 == 19:
            print("%d cases, %d mismatches" % (c + 1, bad), flush=True)
    print("FEEDFUZZ %d cases, %d mismatches, %d layouts refused (boundaries not established: host decoder)" % (cases, bad, REFUSED))
    sys.exit(1 if bad else 0)
