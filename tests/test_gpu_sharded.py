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
