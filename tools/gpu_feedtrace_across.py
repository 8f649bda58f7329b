"""The GPU feed on a BAM whose records run across BGZF blocks (htsjdk / Picard / GATK writers), one batch and in chunks:
    python tools/gpu_feedtrace_across.py write <pairs>   writes /tmp/feedtrace_across.bam
    python tools/gpu_feedtrace_across.py run [reps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PATH = "/tmp/feedtrace_across.bam"

if __name__ == "__main__":
    if sys.argv[1] == "write":
        from tools.gpu_feedbench import write_bam
        n, raw, comp = write_bam(PATH, int(sys.argv[2]), aligned=False)
        print("wrote %s: %d records, %.1f MB inflated, %.1f MB file" % (PATH, n, raw / 1e6, comp / 1e6), flush=True)
    else:
        import torch
        from breakid_amd import capi
        os.environ["BK_DEBUG"] = "feed"
        reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
        for mode in ("batch", "chunks"):
            if mode == "chunks":
                os.environ["BREAKID_FEED_PACKED_CHUNKS"] = "1"
            for rep in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                table = capi.decode_bam_device(PATH)
                t1 = time.perf_counter()
                print("%s rep %d: file -> device table %.3f s (%d records)" % (mode, rep, t1 - t0, table.soa.n), flush=True)
                table.close()
