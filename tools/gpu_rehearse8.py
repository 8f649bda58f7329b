"""Eight ranks on the ONE GPU that exists (in-process transport): a WGS-shape sample as a BAM file through `BreakID -gpus 8 -comm local`
against the single-context run - same txt files -, the bytes every rank exchanged per step and its mask + cluster time (BK_DEBUG=multi).
usage: gpu_rehearse8.py [records = 50_000_000] [ranks = 8]"""
import hashlib, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch
from breakid_amd import bamio, synth, synth_gpu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
BIN = os.path.join(ROOT, "breakid_amd", "bin", "BreakID")
t0 = time.time()
contigs, cols = synth_gpu.make_wgs(n, 777, torch.device("cuda", 0))
cols = synth_gpu.to_numpy_cols(cols)
torch.cuda.empty_cache()
cols, names = synth.name_records(cols)
cols["target_len"] = np.asarray([l for _, l in contigs], np.uint32)
print("table of %d records in %.1f s" % (len(cols["tid"]), time.time() - t0), flush=True)
with tempfile.TemporaryDirectory() as tmp:
    bam = os.path.join(tmp, "s.bam")
    t0 = time.time()
    bamio.write_bam_from_soa_fast(bam, contigs, cols, names)
    bamio.write_bai(bam)
    print("BAM of %.2f GB in %.1f s" % (os.path.getsize(bam) / 1e9, time.time() - t0), flush=True)
    del cols, names
    side = synth.write_side_files(contigs, tmp, refgene_lines=synth.random_refgene(contigs, 200, 5), max_nib_len=300_000_000)
    env = dict(os.environ, BREAKID_INSTALLDIR=side["install"], BK_DEBUG="multi")
    outs = {}
    for label, extra in (("one context", []), ("%d ranks" % W, ["-gpus", str(W), "-comm", "local"])):
        prefix = os.path.join(tmp, "out_" + label.replace(" ", "_"))
        t0 = time.time()
        r = subprocess.run([BIN, "-i", bam, "-o", prefix, "-n", side["nib"], "-all", "-fast"] + extra, env=env, capture_output=True, text=True)
        print("%s: rc %d, %.1f s" % (label, r.returncode, time.time() - t0), flush=True)
        if r.returncode != 0:
            print(r.stderr[-3000:])
            sys.exit(1)
        outs[label] = {s: hashlib.sha256(open(prefix + s, "rb").read()).hexdigest() for s in ("_fusion.txt", "_fusion_all.txt")}
        outs[label]["perf5"] = ["\t".join(l.split("\t")[:5]) for l in open(prefix + "_performance.txt").read().split("\n")[:2]]
        for line in r.stderr.split("\n"):
            if line.startswith("[multi] rank") and "mask + cluster" in line:
                print(line)
    a, b = outs["one context"], outs["%d ranks" % W]
    print("txt files equal:", a["_fusion.txt"] == b["_fusion.txt"] and a["_fusion_all.txt"] == b["_fusion_all.txt"], "| _performance.txt columns equal:", a["perf5"] == b["perf5"])
    sys.exit(0 if a == b else 2)
