"""Feed experiment (GPU box): file -> device table for one block-aligned BAM under different chunk sizes and numbers of
hardware queues (the runtime reads GPU_MAX_HW_QUEUES when it starts: one child process per setting).
python tools/gpu_feed_sweep.py [pairs]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
import torch
from breakid_amd import capi
path, n = sys.argv[1], int(sys.argv[2])
best = 1e9
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    table = capi.decode_bam_device(path)
    t1 = time.perf_counter()
    table.close()
    best = min(best, t1 - t0)
print("queues=%%s chunk=%%s MiB: file -> device table %%.1f ms = %%.1f M records/s" %% (os.environ.get("GPU_MAX_HW_QUEUES", "default"), os.environ.get("BREAKID_FEED_CHUNK_MB", "default"), best * 1e3, n / best / 1e6), flush=True)
""" % ROOT

if __name__ == "__main__":
    from tools import gpu_feedbench
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    path = "/tmp/feed_sweep_%d.bam" % n_pairs
    n, raw, comp = gpu_feedbench.write_bam(path, n_pairs)
    print("wrote %s: %d records, %.0f MB file" % (path, n, comp / 1e6), flush=True)
    for q in ("", "16"):
        for chunk in ("", "32", "128", "256", "1024"):
            env = dict(os.environ)
            env.pop("GPU_MAX_HW_QUEUES", None)
            env.pop("BREAKID_FEED_CHUNK_MB", None)
            if q: env["GPU_MAX_HW_QUEUES"] = q
            if chunk: env["BREAKID_FEED_CHUNK_MB"] = chunk
            env["BREAKID_FEED_STATS"] = "0"
            subprocess.run([sys.executable, "-c", CHILD, path, str(n)], env=env, timeout=300)
