"""Decodes the same BAM R times on the GPU (block-aligned and with records across BGZF blocks, eight chunks in flight) and
compares every column with the first run's: the feed's chunks, staging buffers and column growth are timing dependent.
    python tools/gpu_feedsoak.py [pairs] [runs]"""
import hashlib, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from breakid_amd import abi, capi
from breakid_amd.sharded import tensor_from_ptr
from tools.gpu_feedbench import write_bam

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = torch.device("cuda", 0)


def digest(table):
    s = table.soa
    n = s.n
    sizes = {"cigar_off": n + 1, "aux_off": n + 1, "cigar": s.n_cigar_words, "aux": s.n_aux_bytes}
    h = hashlib.sha256()
    for name, dt in abi.SOA_COLS_ALL:
        nb = sizes.get(name, n) * np.dtype(dt).itemsize
        if nb:
            h.update(tensor_from_ptr(getattr(s, name), nb, dev).cpu().numpy().tobytes())
    return "%d:%s" % (n, h.hexdigest()[:16])


bad = 0
for aligned in (True, False):
    path = "/tmp/feedsoak_%d.bam" % aligned
    write_bam(path, pairs, aligned=aligned)
    host = None
    for chunk in (None, "3"):
        if chunk:
            os.environ["BREAKID_FEED_CHUNK_MB"] = chunk
        else:
            os.environ.pop("BREAKID_FEED_CHUNK_MB", None)
        first = None
        for r in range(runs):
            t = capi.decode_bam_device(path)
            d = digest(t)
            t.close()
            if first is None:
                first = d
            elif d != first:
                bad += 1
                print("aligned=%s chunk=%s run %d DIFFERS: %s vs %s" % (aligned, chunk, r, d, first), flush=True)
        if host is None:
            host = first
        elif first != host:
            bad += 1
            print("aligned=%s: chunk sizes disagree: %s vs %s" % (aligned, first, host), flush=True)
        print("aligned=%s chunk=%s MiB: %d runs, digest %s" % (aligned, chunk or "default", runs, first), flush=True)
print("FEEDSOAK %d bad" % bad, flush=True)
sys.exit(1 if bad else 0)
