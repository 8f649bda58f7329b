"""Worker of the 2-rank sharded-sample test: run under torch.distributed.run (gloo), every rank on cuda:0.
Rank 0 rebuilds the whole sample from the deterministic shard generator and checks the sharded result against
the CPU oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from breakid_amd import abi, capi, sharded, synth_gpu  # noqa: E402


def concat_shards(hosts):
    full = {k: np.concatenate([h[k] for h in hosts]) for k in ("tid", "pos", "mtid", "mpos", "isize", "flag", "mapq", "qhash", "qcheck")}
    cb = ab = 0
    co, ao = [], []
    for h in hosts:
        co.append(h["cigar_off"][:-1].astype(np.int64) + cb)
        cb += int(h["cigar_off"][-1])
        ao.append(h["aux_off"][:-1].astype(np.int64) + ab)
        ab += int(h["aux_off"][-1])
    full["cigar_off"] = np.concatenate(co + [np.array([cb])]).astype(np.uint32)
    full["aux_off"] = np.concatenate(ao + [np.array([ab])]).astype(np.uint32)
    full["cigar"] = np.concatenate([h["cigar"] for h in hosts])
    full["aux"] = np.concatenate([h["aux"] for h in hosts])
    return full


def main():
    n_per_rank, seed, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    routed = (sys.argv[4] if len(sys.argv) > 4 else "routed") == "routed"
    backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"  # "nccl" = RCCL: one rank per GPU (world size 1 on a one-GPU box)
    # every rank on cuda:0 (one-GPU box) unless the test asks for a device per rank (RCCL with more than one rank)
    one_each = os.environ.get("SHARD_WORKER_ONE_GPU_PER_RANK") == "1"
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if one_each else 0)
    if backend == "nccl":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    contigs, cols = synth_gpu.make_wgs_shard(n_per_rank, seed, dev, rank, world)
    comm = sharded.Comm(dev)
    counts = comm.all_gather_scalars([cols["n"]])[:, 0].tolist()
    rec_base = int(sum(counts[:rank]))
    # SHARD_WORKER_REC_GAP: the ranks behind rank 0 number their records from beyond 2^32 (a sample of more than 4 G records whose
    # first shard is short): discovery order only depends on the ORDER of the indices, so every call must stay the oracle's
    gap = int(os.environ.get("SHARD_WORKER_REC_GAP", "0"))
    if rank > 0:
        rec_base += gap
    ctx = capi.Context(contigs, device=dev.index)
    ctx.attach_device(abi.device_ptrs(cols), cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])
    run = sharded.ShardedRun(ctx, comm, routed=routed)
    w = run.run(rec_base, qual=20, fast=(mode == "fast"))
    got, _ = ctx.fetch(abi.STAGE_CLUSTERS)
    ok = True
    if rank == 0:
        from oracle import pyoracle
        hosts = [synth_gpu.to_numpy_cols(synth_gpu.make_wgs_shard(n_per_rank, seed, dev, r, world)[1]) for r in range(world)]
        o = pyoracle.Oracle(contigs, concat_shards(hosts))
        ow, rc = o.run(20, fast=(mode == "fast"))
        exp, _ = o.fetch(abi.STAGE_CLUSTERS)
        ok = rc == 0 and w == ow and np.array_equal(got, exp)
        if gap:
            # the gathered evidence tuples carry the sample-wide record index: rank 0's as they are, the others' beyond the gap
            gs, _ = ctx.fetch(abi.STAGE_SPLITS)
            es, _ = o.fetch(abi.STAGE_SPLITS)
            back = gs.copy()
            far = back["rec"] >= counts[0]
            back["rec"][far] -= gap
            ok = ok and len(gs) == len(es) and np.array_equal(back, es) and bool(far.any()) and int(gs["rec"].max()) >= (1 << 32)
            print("SHARD_REC64", "OK" if ok else "MISMATCH", "largest record index", int(gs["rec"].max()), flush=True)
        print("SHARD_CHECK", "OK" if ok else "MISMATCH", "w", w, ow, "clusters", len(got), len(exp), "valid", int(((got["flags"] & 2) != 0).sum()), flush=True)
        if not ok and len(got) == len(exp):
            bad = [i for i in range(len(got)) if got[i] != exp[i]][:5]
            print(got[bad], exp[bad], flush=True)
    # every rank must hold the same final table
    import zlib
    h = torch.tensor([zlib.crc32(got.tobytes())], dtype=torch.int64)
    hs = [torch.zeros_like(h) for _ in range(world)]
    if backend == "nccl":
        h = h.to(dev)
        hs = [torch.zeros_like(h) for _ in range(world)]
    dist.all_gather(hs, h)
    same = all(int(x) == int(hs[0]) for x in hs)
    if rank == 0:
        print("SHARD_REPLICAS", "OK" if same else "DIFFER", flush=True)
    ctx.close()
    dist.destroy_process_group()
    sys.exit(0 if (ok and same) else 1)


if __name__ == "__main__":
    main()
