#!/usr/bin/env python3
"""bench.py — BreakID hot path on MI355X: M records/s "clustered + split-scanned".

A step = one pass of the whole hot path (insert-size statistics, discordant-pair scan + mate join,
isolated-pair masking, -fast clustering, per-read split evidence, cluster summary, split-read
breakpoints) over one synthetic WGS-shape record table that is already resident in HBM when the timed
region starts.  Workload at N=1 = BASELINE.json configs[1] (30x WGS shape, hg19, 2x150 bp); for N>1 it is configs[2]:
the SAME sample sharded over the ranks - every rank holds a contiguous range of its coordinate-sorted records, candidates
and pairs travel to their owners by RCCL all-to-all (breakid_amd/sharded.py) - so the total work stays fixed (strong
scaling; `--scaling weak` shards one sample of N x that size instead: record indices of a sample are 64-bit, only one
rank's table stays below 2^32 records).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--records R]

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel k_stream, live HIP-event timing) and
`cpu_baseline` (the CPU oracle port timed on a bounded sample of the same workload, rank 0, N=1)."""
import argparse
import json
import os
import sys
import time

# the lanes of chromosome-pair groups in bk_mask_and_cluster (csrc/api.hip; the library's default) need more hardware queues than
# ROCm's default of 4 (every lane's heap kernels sit on side streams); the runtime reads this when it starts, i.e. before torch
# touches the GPU (importing breakid_amd.capi sets it as well)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def from_bam_side_line(n_pairs, device, fast, reps=5):
    """File -> calls, end to end (SURVEY 8(f2/f3)): the metric of the headline line is defined on the resident table; this is
    what a user of the command line sees.  Three timings of the same block-aligned BAM (page cache warm, best of 5): the GPU feed
    alone (file -> device table), the feed with the stream pass overlapped (file -> table + candidates + sums), and the rest
    of the hot path on the resident table."""
    import torch
    from breakid_amd import capi
    from tools import gpu_feedbench
    path = "/tmp/bench_from_bam_%d.bam" % n_pairs
    n, raw, comp = gpu_feedbench.write_bam(path, n_pairs)
    # what the link itself would take for the file's bytes: pinned host -> device copy rate, measured here
    pin = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
    dst = torch.empty(256 << 20, dtype=torch.uint8, device=device)
    dst.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        dst.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    h2d = 4 * (256 << 20) / (time.perf_counter() - t0)
    del pin, dst
    feed, over, rest, total = [], [], [], []
    for rep in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        table = capi.decode_bam_device(path, device)
        t1 = time.perf_counter()
        table.close()
        feed.append(t1 - t0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx, table = capi.decode_bam_device_ctx(path, 20, device)
        ctx.sync()
        t1 = time.perf_counter()
        ctx.run(qual=20, fast=fast)
        ctx.sync()
        t2 = time.perf_counter()
        over.append(t1 - t0)
        rest.append(t2 - t1)
        total.append(t2 - t0)
        ctx.close()
        table.close()
    os.remove(path)
    return {"records": n, "bam_MB": round(comp / 1e6, 1), "inflated_MB": round(raw / 1e6, 1), "reps": reps,
            "pinned_h2d_GBps": round(h2d / 1e9, 1), "h2d_bound_s": round(comp / h2d, 4),  # the file's bytes at the measured copy rate: the floor of any feed
            "file_to_calls_steady_s": round(sorted(total)[len(total) // 2], 4),  # median of the calls (first call included when reps is small)
            "feed_first_call_s": round(feed[0], 4),  # the process's first file: staging buffers and feed slots are allocated and page-locked
            "feed_only_s": round(min(feed), 4), "feed_only_M_records_per_s": round(n / min(feed) / 1e6, 1),
            "feed_plus_stream_pass_overlapped_s": round(min(over), 4), "feed_plus_stream_pass_M_records_per_s": round(n / min(over) / 1e6, 1),
            "rest_of_hot_path_s": round(min(rest), 4), "file_to_calls_s": round(min(total), 4), "file_to_calls_M_records_per_s": round(n / min(total) / 1e6, 1)}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: N fresh rank processes over RCCL on this node."""
    import socket
    import subprocess
    # The launcher does not touch the GPU runtime (it stays alive for the whole run beside the ranks it starts as fresh children):
    # the devices are counted from the kernel driver's topology, narrowed by the *_VISIBLE_DEVICES variables.  Every rank checks
    # again with its own runtime and refuses fewer devices than ranks.
    have = 0
    try:
        for node in os.listdir("/sys/class/kfd/kfd/topology/nodes"):
            try:
                have += 1 if int(open("/sys/class/kfd/kfd/topology/nodes/%s/gpu_id" % node).read().strip() or "0") != 0 else 0
            except (OSError, ValueError):
                pass
    except OSError:
        have = 0  # (no kernel driver: no GPU)
    try:
        usable = len([d for d in os.listdir("/dev/dri") if d.startswith("renderD") and os.access("/dev/dri/" + d, os.R_OK | os.W_OK)])
        have = min(have, usable)  # (a container sees the whole topology but only its own devices' nodes)
    except OSError:
        have = 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            have = min(have, len([x for x in v.split(",") if x.strip() != ""]))
    if have < n:
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this node: refusing to run (a line with fewer ranks "
                         "than asked for would be mislabelled)\n" % (n, have))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--records", type=int, default=620_000_000, help="records per GPU (configs[1]: ~620 M)")
    ap.add_argument("--cpu-sample", type=int, default=160_000_000, help="records in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--mode", default="fast", choices=["fast", "ahc"])
    ap.add_argument("--seed", type=int, default=12346)
    ap.add_argument("--workload", default="wgs", choices=["wgs", "panel"],
                    help="wgs = configs[1] (the headline line); panel = configs[3] targeted-panel shape (500 loci x 2000x, single GPU, side measurement)")
    ap.add_argument("--from-bam", type=int, default=4_000_000, metavar="PAIRS",
                    help="side measurement (N=1, 0 = skip): write a synthetic BAM of PAIRS read pairs and time file -> calls end to end: GPU feed alone, feed "
                         "with the stream pass overlapped (bk_bam_decode_device_ctx), rest of the hot path; reported as `from_bam` beside the headline line "
                         "(the metric itself is defined on the resident table)")
    ap.add_argument("--from-bam-big", type=int, default=0, metavar="PAIRS",
                    help="a second file -> calls leg on a file of this many pairs (32 000 000 = 64 M records = 8 GB: a file that does not fit a burst); 0 = skip")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = configs[2], the --records sample sharded over the ranks (default); weak = one sample of N x --records")
    ap.add_argument("--sharded", type=int, default=-1, help="1: one sample sharded over the ranks (default when --gpus > 1), 0: plain single-table run")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started by hand as `python bench.py --gpus N`: this process never touches a GPU, it starts the N ranks as fresh child
        # processes (one per GPU, torch.distributed.run = what the driver would have started), relays rank 0's line and ends
        # with their status
        sys.exit(spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    from breakid_amd import abi, capi, synth_gpu

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` (it starts its own ranks) or under "
                         "torch.distributed.run with --nproc-per-node equal to --gpus" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    if torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks asked for, %d GPU(s) visible: refusing to run (one rank per GPU, no rank shares a device)" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_shards = args.sharded == 1 or (args.sharded < 0 and world > 1)
    if world > 1 or use_shards:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # size the table to the card (configs[1] needs ~60 GB including generator temporaries)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    n_rec = args.records
    if use_shards and args.scaling == "strong":
        n_rec = max(1_000_000, n_rec // world)
    while n_rec * 110 > free_b and n_rec > 1_000_000:
        n_rec //= 2
    t0 = time.time()
    if use_shards:
        from breakid_amd import sharded
        contigs, cols = synth_gpu.make_wgs_shard(n_rec, args.seed, dev, rank, world)
    elif args.workload == "panel":
        contigs, cols = synth_gpu.make_panel(args.seed + rank, dev)
    else:
        contigs, cols = synth_gpu.make_wgs(n_rec, args.seed + rank, dev)
    torch.cuda.synchronize(dev)
    gen_s = time.time() - t0
    torch.cuda.empty_cache()  # the generator's temporaries go back to the driver: the library allocates with hipMalloc
    n = cols["n"]

    ctx = capi.Context(contigs, device=local_rank)
    ptrs = abi.device_ptrs(cols)
    ctx.attach_device(ptrs, n, cols["n_cigar_words"], cols["n_aux_bytes"])
    fast = args.mode == "fast"
    n_total = n * world
    if use_shards:
        comm = sharded.Comm(dev)
        counts = comm.all_gather_scalars([n])[:, 0].tolist()
        rec_base, n_total = int(sum(counts[:rank])), int(sum(counts))
        runner = sharded.ShardedRun(ctx, comm)

    def step():
        # bk_upload_records(BK_MEM_DEVICE) is zero copy; re-attaching invalidates every cached stage result
        ctx.attach_device(ptrs, n, cols["n_cigar_words"], cols["n_aux_bytes"])
        if use_shards:
            return runner.run(rec_base, qual=20, fast=fast), 0
        w, nv = ctx.run(qual=20, fast=fast)
        return w, nv

    for _ in range(args.warmup):
        step()
    ctx.sync()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    ctx.timing_enable(True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w, n_valid = step()
    ctx.sync()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timing = ctx.timing()
    touched = ctx.timing_touched()
    ctx.timing_enable(False)
    if use_shards and rank == 0:
        # every rank ends a sharded step with the whole cluster table; the count for the report is taken from the last step's
        # (bk_fetch assembles on the host: inspection path, outside the timed region)
        cl, _ = ctx.fetch(abi.STAGE_CLUSTERS)
        n_valid = int(((cl["flags"] & 2) != 0).sum())

    if rank == 0:
        PEAK = 8000.0  # GB/s, HBM3E (MI355X_MICROARCH.md)
        per = {}
        for (name, ms, by), tb in zip(timing, touched):
            a = per.setdefault(name, [0.0, 0, 0, 0])
            a[0] += ms
            a[1] += by
            a[2] += 1
            a[3] += tb
        ks = per.get("k_stream", [0.0, 0, 1, 0])
        launches = max(1, ks[2])
        avg_ms = ks[0] / launches
        # roofline of the dominant HBM kernel.  `achieved` counts the bytes k_stream itself has to move per launch (23 B of
        # fixed columns per record + CIGAR words + 48 B per candidate + 4 B per SA-bearing record: DESIGN.md 4); the SURVEY
        # 8(d) per-record formula (39 B/record: qhash/mtid/mpos credited for every record although only candidates load them)
        # is reported beside it as survey_formula_*, and the PMC traffic (separate rocprofv3 --pmc passes) as traffic.
        achieved = (ks[3] / launches / 1e9) / (avg_ms / 1e3) if avg_ms > 0 else 0.0
        survey = (ks[1] / launches / 1e9) / (avg_ms / 1e3) if avg_ms > 0 else 0.0
        # HBM traffic of the launch: NOT measured in this run - rocprofv3 --pmc passes cannot share a process with the timed region -
        # but read from the committed counter summary of the same table (matched by the SURVEY-formula bytes of the launch)
        traffic, traffic_src = None, None
        for pm_name in ("r04_pmc_k_stream.json", "r03_pmc_k_stream.json", "r02_pmc_k_stream.json", "r01_pmc_k_stream.json"):
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", pm_name)))
                if abs(pm["algorithmic_bytes_per_launch"] - ks[1] / launches) < 0.02 * pm["algorithmic_bytes_per_launch"]:
                    traffic = pm["traffic_bytes_per_launch"]
                    traffic_src = "profiles/" + pm_name + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same table, not this run)"
                    break
            except Exception:
                pass
        # path-level fraction (SURVEY 8(d) / BASELINE.md 3): sum of algorithmic bytes / sum of stage time / peak
        path_ms = sum(v[0] for v in per.values()) / args.steps
        path_bytes = sum(v[1] for v in per.values()) / args.steps
        path_gbs = (path_bytes / 1e9) / (path_ms / 1e3) if path_ms > 0 else 0.0
        stages = []
        for k, v in per.items():
            ms = v[0] / args.steps
            b_alg, b_own = v[1] / args.steps, v[3] / args.steps
            gbs = (b_own / 1e9) / (ms / 1e3) if ms > 0 and b_own else None
            stages.append({"stage": k, "ms": round(ms, 3), "algorithmic_bytes": int(b_alg), "own_bytes": int(b_own) if b_own else None,
                           "GBps": round(gbs, 1) if gbs else None, "frac": round(gbs / PEAK, 4) if gbs else None,
                           "share_of_step": round(ms / path_ms, 4) if path_ms else None})
        value = (n_total * args.steps) / dt / 1e6
        out = {
            "metric": "M reads/s clustered+split-scanned", "value": round(value, 3), "unit": "M records/s",
            "n_gpus": dist.get_world_size() if dist.is_initialized() else 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else None, "vs_baseline": None, "dtype": "int32/u8 (+f64 sd replay)",
            "data": "synthetic",
            "config": {"workload": (("configs[2]: the " if use_shards and args.scaling == "strong" and world > 1 else "configs[1]: ") + "30x WGS-shape synthetic table, hg19, 2x150bp, 5%% discordant, -%s clustering" if args.workload == "wgs" else
                                    "configs[3]: targeted-panel shape, 500 fusion loci x 2000x, 20%% split reads, 10%% discordant, -%s clustering") % args.mode,
                       "records_per_gpu": int(n), "bytes_per_record_algorithmic": round(path_bytes / max(1, n), 2),
                       "valid_clusters": int(n_valid), "w": w, "generator_s": round(gen_s, 2),
                       "sharding": ("one sample of %d records, contiguous record range per rank; RCCL all-to-all of candidates (to the owner of "
                                    "the read-name hash) and of pairs (to the owner of the chr-pair group, LPT), all-gather of tuples/cluster "
                                    "summaries, all-reduce of coverage/depth counts" % n_total) if use_shards else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "k_stream", "achieved": round(achieved, 1), "peak": PEAK, "unit": "GB/s",
                         "frac": round(achieved / PEAK, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_frac": round((traffic / 1e9) / (avg_ms / 1e3) / PEAK, 4) if traffic and avg_ms > 0 else None,
                         "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(ks[3] / launches),
                         "survey_formula_bytes_per_launch": int(ks[1] / launches), "survey_formula_frac": round(survey / PEAK, 4),
                         "path_frac": round(path_gbs / PEAK, 5), "path_achieved": round(path_gbs, 1), "path_ms": round(path_ms, 3),
                         "path_algorithmic_bytes": int(path_bytes),
                         "note": "frac = k_stream own bytes / HIP-event time / 8 TB/s; path_frac = SURVEY 8(d): sum of algorithmic bytes of the whole "
                                 "hot path / sum of stage times / 8 TB/s (the sort emulation and the joins add time, no algorithmic bytes)",
                         "stages": stages},
            "stage_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in per.items()},
            "ranks": {"world_size": dist.get_world_size() if dist.is_initialized() else 1, "backend": dist.get_backend() if dist.is_initialized() else None,
                      "devices_visible": torch.cuda.device_count()},
        }
        # CPU baseline: the oracle port on a bounded sample of the same workload (rank 0, N = 1 only)
        if world == 1 and not use_shards and args.cpu_sample > 0:
            from oracle import pyoracle
            ns = min(args.cpu_sample, n)
            if args.workload == "panel":
                c2, scols = synth_gpu.make_panel(args.seed + 1000, dev)
            else:
                c2, scols = synth_gpu.make_wgs(ns, args.seed + 1000, dev)
            host = synth_gpu.to_numpy_cols(scols)
            t1 = time.perf_counter()
            o = pyoracle.Oracle(c2, host)
            ow, rc = o.run(20, fast=fast)
            cpu_s = time.perf_counter() - t1
            # same sample through the GPU path: bit-exact check of the final calls
            sctx = capi.Context(c2, device=local_rank)
            sctx.attach_device(abi.device_ptrs(scols), scols["n"], scols["n_cigar_words"], scols["n_aux_bytes"])
            gw, _ = sctx.run(qual=20, fast=fast)
            a, _ = sctx.fetch(abi.STAGE_CLUSTERS)
            b, _ = o.fetch(abi.STAGE_CLUSTERS)
            exact = bool(gw == ow and np.array_equal(a, b))
            out["cpu_baseline"] = {"value": round(scols["n"] / cpu_s / 1e6, 3), "unit": "M records/s", "cores": 1, "kind": "port",
                                   "sample": "%d-record table from the same generator (seed+1000), oracle/liboracle.so single thread, %.1f s; "
                                             "GPU result on the sample bit-identical: %s" % (scols["n"], cpu_s, exact)}
            sctx.close()
            o.close()
        if world == 1 and not use_shards and args.from_bam > 0:
            out["from_bam"] = from_bam_side_line(args.from_bam, local_rank, fast)
        if world == 1 and not use_shards and args.from_bam_big > 0:
            out["from_bam_big"] = from_bam_side_line(args.from_bam_big, local_rank, fast, reps=3)
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
