"""One sample sharded over ranks (breakid_amd/sharded.py): world-size-1 path against the plain pipeline, and a
real 2-rank run (gloo, both ranks on cuda:0) against the CPU oracle on the concatenated sample."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from breakid_amd import abi, capi, sharded, synth_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world1_sharded_equals_plain_run():
    dev = torch.device("cuda", 0)
    contigs, cols = synth_gpu.make_wgs(1_500_000, 4242, dev)
    ptrs = abi.device_ptrs(cols)
    a = capi.Context(contigs)
    a.attach_device(ptrs, cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])
    w, _ = a.run(qual=20, fast=True)
    b = capi.Context(contigs)
    b.attach_device(ptrs, cols["n"], cols["n_cigar_words"], cols["n_aux_bytes"])
    w2 = sharded.ShardedRun(b, sharded.Comm(dev)).run(0, qual=20, fast=True)
    assert w == w2
    for st in (abi.STAGE_SCAN, abi.STAGE_ISO, abi.STAGE_CLUSTERED, abi.STAGE_SPLITS, abi.STAGE_CLUSTERS):
        x, xo = a.fetch(st)
        y, yo = b.fetch(st)
        assert np.array_equal(x, y), st
    a.close()
    b.close()


@pytest.mark.parametrize("mode,exchange", [("fast", "routed"), ("ahc", "routed"), ("fast", "replicated")])
def test_two_rank_sharded_sample_matches_oracle(mode, exchange):
    """routed = candidates / pairs travel to their owner ranks by all-to-all; replicated = every rank joins everything"""
    port = {"fastrouted": "29611", "ahcrouted": "29612", "fastreplicated": "29613"}[mode + exchange]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    n = "600000" if mode == "fast" else "150000"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "tests", "shard_worker.py"), n, "77", mode, exchange]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARD_CHECK OK" in r.stdout and "SHARD_REPLICAS OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.parametrize("exchange", ["routed", "replicated"])
def test_record_indices_beyond_2_32_across_ranks(exchange):
    """rank 1 numbers its records from 2^32 - 1000 + n_0 on: candidates, pairs and tuples carry 64-bit sample-wide indices (the
    reference counts in long / size_t, BreakID.cc:1379,1911-1913); join order, pair order inside the groups and tuple order must
    come out as the oracle's on the concatenated sample"""
    port = "29641" if exchange == "routed" else "29642"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, SHARD_WORKER_REC_GAP=str((1 << 32) - 1000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "tests", "shard_worker.py"), "400000", "83", "fast", exchange]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARD_CHECK OK" in r.stdout and "SHARD_REC64 OK" in r.stdout and "SHARD_REPLICAS OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


def test_two_rank_sharded_sample_with_lanes_of_groups_inside_every_rank():
    """BREAKID_GROUP_LANES=2 on a sharded sample: every rank deals the groups it OWNS to two lanes (the others belong to no lane)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29614", BREAKID_GROUP_LANES="2", BREAKID_LANES_MIN_PAIRS="1000", GPU_MAX_HW_QUEUES="16")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "tests", "shard_worker.py"), "600000", "79", "fast", "routed"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARD_CHECK OK" in r.stdout and "SHARD_REPLICAS OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.parametrize("exchange", ["routed", "replicated"])
def test_rccl_backend_world_size_1(exchange):
    """backend "nccl" (= RCCL) at world size 1: the device-tensor branch of the exchange layer - collectives on device buffers
    of the library, the library running on torch's current stream (device-side ordering, no host sync), unpadded all-gather by
    per-rank broadcasts, all_to_all_single - against the CPU oracle.  More RCCL ranks need one GPU each (driver's scaling run)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29621" if exchange == "routed" else "29622")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "tests", "shard_worker.py"), "600000", "78", "fast", exchange, "nccl"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARD_CHECK OK" in r.stdout and "SHARD_REPLICAS OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


def test_bench_sharded_path_runs_over_rccl():
    """bench.py --gpus 1 --sharded 1: the multi-GPU bench path (RCCL process group, sharded generator, ShardedRun) on one GPU"""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29623", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--sharded", "1", "--records", "3000000", "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().split("\n")[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["valid_clusters"] > 100 and "RCCL" in line["config"]["sharding"]
    assert line["roofline"]["path_frac"] > 0 and any(s["stage"] == "k_stream" for s in line["roofline"]["stages"])


# ---- more than one RCCL rank: one GPU per rank, so these need a node with >= 2 devices (skipped on the one-GPU box) ----------
_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: RCCL takes one device per rank")


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` starts its own N ranks (fresh processes, nothing re-executed after a GPU call); with fewer
    than N devices it must end non-zero and print no line at all - never a line that says n_gpus 1"""
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--records", "3000000", "--steps", "1", "--warmup", "0", "--cpu-sample", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to run" in r.stderr and "{" not in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    # and the flag must agree with the launcher's world size
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "{" not in r.stdout


@_two_gpus
@pytest.mark.parametrize("mode,exchange", [("fast", "routed"), ("ahc", "routed"), ("fast", "replicated")])
def test_two_rccl_ranks_on_two_gpus_match_oracle(mode, exchange):
    """ShardedRun over backend nccl (= RCCL over xGMI) with two ranks, one GPU each, against the CPU oracle on the whole sample"""
    port = {"fastrouted": "29631", "ahcrouted": "29632", "fastreplicated": "29633"}[mode + exchange]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, SHARD_WORKER_ONE_GPU_PER_RANK="1")
    n = "600000" if mode == "fast" else "150000"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "tests", "shard_worker.py"), n, "81", mode, exchange, "nccl"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SHARD_CHECK OK" in r.stdout and "SHARD_REPLICAS OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


@_two_gpus
def test_bench_two_gpus_starts_its_own_ranks():
    """the way the driver's scaling run may start it: `python bench.py --gpus 2` with no launcher around it"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--records", "3000000", "--steps", "2", "--warmup", "1", "--cpu-sample", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.strip().split("\n") if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"]["world_size"] == 2 and line["ranks"]["backend"] == "nccl" and line["value"] > 0
    assert line["config"]["valid_clusters"] > 100 and "RCCL" in line["config"]["sharding"]
