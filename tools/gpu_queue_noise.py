"""How much do extra ACTIVE hardware queues cost the lanes of bk_mask_and_cluster?  The step is timed alone and beside K background
streams that keep launching tiny kernels (nothing that competes for CUs or bandwidth): beyond four active queues every launch /
round trip of every queue gets slower on this part (tools/ubench/readback.hip).   python tools/gpu_queue_noise.py [records]"""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from breakid_amd import abi, capi, synth_gpu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 620_000_000
dev = torch.device("cuda", 0)
contigs, cols = synth_gpu.make_wgs(n, 12346, dev)
torch.cuda.empty_cache()
ctx = capi.Context(contigs)
ptrs = abi.device_ptrs(cols)


def step():
    ctx.attach_device(ptrs, cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])
    ctx.run(qual=20, fast=True)


for _ in range(2):
    step()
for k in (0, 1, 2, 4, 8):
    stop = threading.Event()

    def noise(i):
        st = torch.cuda.Stream(device=dev)
        x = torch.zeros(64, device=dev)
        with torch.cuda.stream(st):
            while not stop.is_set():
                for _ in range(20):
                    x.add_(1.0)
                st.synchronize()

    th = [threading.Thread(target=noise, args=(i,)) for i in range(k)]
    for t in th:
        t.start()
    time.sleep(0.05)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5
    stop.set()
    for t in th:
        t.join()
    print("%d background stream(s): %.2f ms per step" % (k, dt * 1e3), flush=True)
