"""Kernel / copy timeline of the GPU feed (run on the GPU box under rocprofv3 --kernel-trace --memory-copy-trace):
    python tools/gpu_feedtrace.py write <pairs>     writes /tmp/feedtrace.bam
    python tools/gpu_feedtrace.py run [reps]        bk_bam_decode_device on it (the last repetition is the one to read)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PATH = "/tmp/feedtrace.bam"

if __name__ == "__main__":
    if sys.argv[1] == "write":
        from tools.gpu_feedbench import write_bam
        n, raw, comp = write_bam(PATH, int(sys.argv[2]))
        print("wrote %s: %d records, %.1f MB inflated, %.1f MB file" % (PATH, n, raw / 1e6, comp / 1e6), flush=True)
    else:
        import torch
        from breakid_amd import capi
        os.environ["BK_DEBUG"] = "feed"
        reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
        for rep in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            table = capi.decode_bam_device(PATH)
            t1 = time.perf_counter()
            print("rep %d: file -> device table %.3f s" % (rep, t1 - t0), flush=True)
            table.close()
