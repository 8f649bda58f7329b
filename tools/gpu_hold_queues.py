"""A second process that holds hardware queues on the same GPU (what a pytest parent with its own HIP context is to a test's child
process): does the resident sort service of the child still work?  usage: gpu_hold_queues.py <streams the parent holds> <child runs>"""
import os, subprocess, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1])
streams = [torch.cuda.Stream() for _ in range(n)]
for s in streams:
    with torch.cuda.stream(s):
        torch.zeros(16, device="cuda").add_(1)
torch.cuda.synchronize()
print("parent holds", n, "streams", flush=True)
code = """
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from breakid_amd import abi, capi, synth_gpu
dev = torch.device("cuda", 0)
contigs, cols = synth_gpu.make_wgs(6_000_000, 4711, dev, disc_frac=0.3)
ctx = capi.Context(contigs)
ctx.attach_device(abi.device_ptrs(cols), cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])
import time
for rep in range(3):
    t0 = time.time()
    w, nv = ctx.run(qual=20, fast=True)
    print("run", rep, nv, "%%.1f ms" %% ((time.time() - t0) * 1e3))
""" % ROOT
for i in range(int(sys.argv[2])):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, BK_DEBUG=os.environ.get("CHILD_DEBUG", "lanes,svc"), BREAKID_GROUP_LANES="12", BREAKID_LANES_MIN_PAIRS="1000"))
    bad = r.returncode != 0
    print("child", i, "rc", r.returncode, "%.1f s" % (time.time() - t0), [l for l in r.stdout.split("\n") if l.startswith("run")], flush=True)
    if bad:
        print("\n".join(l[:600] for l in r.stderr.split("\n") if "svc]" in l and "slot" not in l or "Error" in l or "probe" in l or "shares" in l)[-6000:], flush=True)
