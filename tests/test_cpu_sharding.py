"""CPU, world_size 2 over gloo: the exchange layer of the sharded pipeline (Comm), group ownership and the shard
generator's cross-rank consistency.  (The device work itself cannot run without an MI355X.)"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from breakid_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = sharded.Comm(torch.device("cpu"))
    assert comm.world == world and comm.host_staged
    # variable-length all-gather keeps rank order and lengths
    mine = torch.arange(3 + 4 * rank, dtype=torch.uint8) + 10 * rank
    allt = comm.all_gather_var(mine)
    exp = torch.cat([torch.arange(3 + 4 * r, dtype=torch.uint8) + 10 * r for r in range(world)])
    ok = torch.equal(allt, exp)
    # empty contribution from one rank
    e = comm.all_gather_var(torch.arange(5, dtype=torch.uint8) if rank == 1 else torch.empty(0, dtype=torch.uint8))
    ok &= torch.equal(e, torch.arange(5, dtype=torch.uint8))
    s = comm.all_reduce(torch.tensor([rank + 1, 10], dtype=torch.int64))
    ok &= s.tolist() == [3, 20]
    m = comm.all_reduce(torch.tensor([rank * 7], dtype=torch.int64), op="max")
    ok &= m.tolist() == [7]
    sc = comm.all_gather_scalars([rank, 5 - rank])
    ok &= sc.tolist() == [[0, 5], [1, 4]]
    # variable all-to-all: rank r sends (r + 1 + d) bytes valued 16*r + d to rank d (one of the blocks is empty)
    blocks = [torch.full((0 if (rank == 1 and d == 0) else rank + 1 + d,), 16 * rank + d, dtype=torch.uint8) for d in range(world)]
    got = comm.all_to_all_var(torch.cat(blocks), [b.numel() for b in blocks])
    want = torch.cat([torch.full((0 if (r == 1 and rank == 0) else r + 1 + rank,), 16 * r + rank, dtype=torch.uint8) for r in range(world)])
    ok &= torch.equal(got, want)
    owner = sharded.lpt_owner([9, 7, 7, 5, 1], world)
    q.put((rank, bool(ok), owner))
    dist.destroy_process_group()


def test_comm_layer_world2_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2] == [0, 1, 1, 0, 0]


def test_shard_generator_is_consistent_across_ranks():
    sys.path.insert(0, ROOT)
    from breakid_amd import synth_gpu
    W = 3
    shards = [synth_gpu.to_numpy_cols(synth_gpu.make_wgs_shard(200_000, 5, "cpu", r, W)[1]) for r in range(W)]
    tid = np.concatenate([s["tid"] for s in shards]).astype(np.int64)
    pos = np.concatenate([s["pos"] for s in shards]).astype(np.int64)
    key = (tid << 32) | pos
    assert (np.diff(key) >= 0).all()  # rank ranges are disjoint and ordered: the concatenation is coordinate sorted
    # every discordant read finds its mate (possibly on another shard) with mirrored coordinates
    flag = np.concatenate([s["flag"] for s in shards])
    qh = np.concatenate([s["qhash"] for s in shards])
    mtid = np.concatenate([s["mtid"] for s in shards]).astype(np.int64)
    mpos = np.concatenate([s["mpos"] for s in shards]).astype(np.int64)
    disc = ((flag & 1) != 0) & ((flag & 2) == 0) & ((flag & 0x100) == 0)
    order = np.argsort(qh[disc], kind="stable")
    q, t, p, mt, mp_ = (a[disc][order] for a in (qh, tid, pos, mtid, mpos))
    same = q[1:] == q[:-1]
    i = np.nonzero(same)[0]
    assert len(i) > 1000
    assert (t[i] == mt[i + 1]).all() and (p[i] == mp_[i + 1]).all() and (t[i + 1] == mt[i]).all() and (p[i + 1] == mp_[i]).all()
