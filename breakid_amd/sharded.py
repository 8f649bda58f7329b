"""One sample sharded over several GPUs: every rank holds a contiguous range of the
coordinate-sorted records in its own libbreakid_hip context; the small derived tables move between the ranks
with torch.distributed collectives (backend "nccl" = RCCL over xGMI on a node; "gloo" stages through the host
and is what the 2-rank tests use).  Exchange steps:

  all-reduce   insert-size sums / spans                      (4 scalars)
  all-gather   sd exceptions (16 B x ~2e-4 n)                 bit-exact `long += double` replay in record order
  all-gather   discordant candidates (40 B x ~5 % n)          mates live on other shards
  all-gather   split-evidence tuples (80 B x ~0.5 % n), cluster summaries of the owned chr-pair groups
  all-reduce   coverage and depth counts per cluster          range counts add over record shards

Chromosome-pair groups are independent in the reference (BreakID.cc:119-167), so after the replicated mate join
each group is clustered by exactly one rank (LPT on pair counts)."""
from __future__ import annotations

import ctypes as C
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from . import abi, capi

CAND_BYTES = 40  # sizeof(Cand), csrc/bk_common.h (bk_shard_buffer reports it as elem_bytes)


class _Raw:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def tensor_from_ptr(ptr, nbytes, device):
    """uint8 view of raw device memory owned by the library (no copy)."""
    if nbytes == 0 or not ptr:
        return torch.empty(0, dtype=torch.uint8, device=device)
    return torch.as_tensor(_Raw(ptr, nbytes), device=device)


class Comm:
    """torch.distributed wrapper that works with device tensors (nccl) or stages them through the host (gloo)."""

    def __init__(self, device, group=None):
        self.device = device
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1
        self.host_staged = self.on and dist.get_backend(group) == "gloo"
        self.before = None  # set by ShardedRun: waits for the library's own stream before a buffer of it is sent
        self.same_stream = False  # ShardedRun: the library runs on torch's current stream (device-side ordering)

    def _to_comm(self, t):
        if self.before is not None:
            self.before()
        return t.cpu() if self.host_staged else t

    def _done(self, t):
        """RCCL collectives are queued on the process group's own stream and torch makes the CURRENT stream wait for them
        (event, no host round trip).  ShardedRun puts the library on that same stream (bk_set_stream), so kernels and
        collectives are ordered on the device; only when the library keeps its own stream does the host wait here."""
        if not self.host_staged and t.is_cuda and not self.same_stream:
            torch.cuda.current_stream(t.device).synchronize()
        return t

    def all_reduce(self, t, op="sum"):
        if not self.on or self.world == 1:
            return t
        x = self._to_comm(t.contiguous())
        dist.all_reduce(x, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=self.group)
        return self._done(x.to(t.device))

    def all_gather_scalars(self, vals):
        """list of python ints -> int64 tensor [world, len(vals)] on the host."""
        t = torch.tensor(list(vals), dtype=torch.int64)
        if not self.on or self.world == 1:
            return t.unsqueeze(0)
        x = t.to(self.device) if not self.host_staged else t
        out = [torch.empty_like(x) for _ in range(self.world)]
        dist.all_gather(out, x, group=self.group)
        return torch.stack([o.cpu() for o in out])

    def all_gather_var(self, t):
        """uint8 tensor of rank-dependent length -> concatenation in rank order (device tensor)."""
        if not self.on or self.world == 1:
            return t
        sizes = self.all_gather_scalars([t.numel()])[:, 0].tolist()
        if max(sizes) == 0:
            return t
        # one broadcast per rank straight into its slice of the result: no padding to the largest rank, no concatenation
        x = self._to_comm(t)
        out = torch.empty(sum(sizes), dtype=torch.uint8, device=x.device)
        off, works = 0, []
        for r, s in enumerate(sizes):
            if s:
                piece = out[off: off + s]
                if r == self.rank:
                    piece.copy_(x)
                works.append(dist.broadcast(piece, src=dist.get_global_rank(self.group, r) if self.group is not None else r, group=self.group, async_op=True))
            off += s
        for wk in works:
            wk.wait()
        return self._done(out.to(t.device))


    def all_to_all_var(self, t, send_bytes):
        """uint8 tensor laid out as world contiguous blocks of send_bytes[d] bytes for rank d -> concatenation of the
        blocks every rank sent here, in rank order (device tensor).  One all-to-all: each GPU talks to its 7 xGMI peers
        directly, so the volume per rank stays O(total / world)."""
        send_bytes = [int(x) for x in send_bytes]
        if not self.on or self.world == 1:
            return t[: send_bytes[0]]
        per = self.all_gather_scalars(send_bytes)          # per[r][d] = bytes rank r sends to rank d
        recv_bytes = [int(per[r][self.rank]) for r in range(self.world)]
        x = self._to_comm(t[: sum(send_bytes)].contiguous())
        out = torch.empty(sum(recv_bytes), dtype=torch.uint8, device=x.device)
        dist.all_to_all_single(out, x, output_split_sizes=recv_bytes, input_split_sizes=send_bytes, group=self.group)
        return self._done(out.to(t.device))


def lpt_owner(sizes, world):
    """Longest-processing-time assignment of chr-pair groups to ranks (deterministic on every rank)."""
    order = sorted(range(len(sizes)), key=lambda g: (-int(sizes[g]), g))
    load = [0] * world
    owner = [0] * len(sizes)
    for g in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[g] = r
        load[r] += int(sizes[g])
    return owner


class ShardedRun:
    """Drives one rank's context through the sharded pipeline.  After run() every rank's context holds the whole
    cluster table (ctx.fetch(STAGE_CLUSTERS))."""

    def __init__(self, ctx: capi.Context, comm: Comm, routed=True):
        self.ctx, self.comm = ctx, comm
        self.routed = routed  # False: the simpler replicated join (all-gather of every candidate to every rank)
        self._keep = []
        self._trace = os.environ.get("BREAKID_SHARD_TRACE") is not None
        self._t_mark = 0.0
        if comm.on and not comm.host_staged:
            # RCCL: the library runs on torch's current stream, so a collective is ordered behind the kernel that produced its
            # input and the next kernel behind the collective - on the device, without host synchronisation
            ctx.set_stream(torch.cuda.current_stream(comm.device).cuda_stream)
            comm.same_stream = True
        else:
            comm.before = ctx.sync

    def _mark(self, label):
        """BREAKID_SHARD_TRACE=1: host time of every phase (device drained at each mark) on stderr, rank 0."""
        if not self._trace:
            return
        self.ctx.sync()
        if self.comm.device.type == "cuda":
            torch.cuda.synchronize(self.comm.device)
        now = time.perf_counter()
        if label is not None and self.comm.rank == 0:
            print("[sharded] %-24s %8.3f ms" % (label, (now - self._t_mark) * 1e3), file=sys.stderr, flush=True)
        self._t_mark = now

    def _buffer(self, which):
        L, h = self.ctx.L, self.ctx.h
        ptr, n, eb = C.c_void_p(), C.c_uint64(), C.c_uint32()
        self.ctx._check(L.bk_shard_buffer(h, which, C.byref(ptr), C.byref(n), C.byref(eb)))
        return tensor_from_ptr(ptr.value, n.value * eb.value, self.comm.device), eb.value

    def _gather_into(self, which):
        L, h = self.ctx.L, self.ctx.h
        local, eb = self._buffer(which)
        allt = self.comm.all_gather_var(local.clone() if self.comm.world > 1 else local)
        self._keep.append(allt)
        self.ctx._check(L.bk_shard_set_buffer(h, which, C.c_void_p(allt.data_ptr() if allt.numel() else 0), allt.numel() // eb))
        return allt.numel() // eb

    def _routed_join(self, qual, w):
        """candidates -> owner of the read-name hash (all-to-all) -> local mate join -> pairs -> owner of the
        chr-pair group (all-to-all) -> grouped table of exactly this rank's groups."""
        ctx, comm = self.ctx, self.comm
        L, h, dev = ctx.L, ctx.h, comm.device
        W = comm.world
        ptr, cnt = C.c_void_p(), C.POINTER(C.c_uint64)()
        ctx._check(L.bk_shard_route_candidates(h, W, C.byref(ptr), C.byref(cnt)))
        counts = [int(cnt[d]) for d in range(W)]
        send = tensor_from_ptr(ptr.value, sum(counts) * CAND_BYTES, dev)
        mine = comm.all_to_all_var(send, [c * CAND_BYTES for c in counts])
        self._keep.append(mine)
        ctx._check(L.bk_shard_set_buffer(h, abi.BUF_CANDIDATES, C.c_void_p(mine.data_ptr() if mine.numel() else 0), mine.numel() // CAND_BYTES))
        ctx.discordant_pairs(qual, w)  # joins the read names this rank owns
        starts, ng, keys = C.POINTER(C.c_uint64)(), C.c_uint32(), C.POINTER(C.c_uint32)()
        ctx._check(L.bk_shard_group_sizes(h, C.byref(starts), C.byref(ng)))
        ctx._check(L.bk_shard_group_keys(h, C.byref(keys), C.byref(ng)))
        local = np.zeros((ng.value, 2), np.int64)
        for g in range(ng.value):
            local[g, 0] = keys[g]
            local[g, 1] = starts[g + 1] - starts[g]
        allk = comm.all_gather_var(torch.from_numpy(local.reshape(-1)).view(torch.uint8).to(dev)).cpu().numpy().view(np.int64).reshape(-1, 2)
        tot = {}
        for k, sz in allk:
            tot[int(k)] = tot.get(int(k), 0) + int(sz)
        gkeys = sorted(tot)                                   # numeric key order, the same list on every rank
        owner = dict(zip(gkeys, lpt_owner([tot[k] for k in gkeys], W)))
        dest = np.asarray([owner[int(keys[g])] for g in range(ng.value)], dtype=np.uint32)
        ctx._check(L.bk_shard_route_pairs(h, dest.ctypes.data if len(dest) else None, ng.value, W, C.byref(ptr), C.byref(cnt)))
        counts = [int(cnt[d]) for d in range(W)]
        send = tensor_from_ptr(ptr.value, sum(counts) * abi.PAIR.itemsize, dev)
        pairs = comm.all_to_all_var(send, [c * abi.PAIR.itemsize for c in counts])
        self._keep.append(pairs)
        ak = np.asarray(gkeys, dtype=np.uint32)
        ctx._check(L.bk_shard_group_pairs(h, C.c_void_p(pairs.data_ptr() if pairs.numel() else 0), pairs.numel() // abi.PAIR.itemsize,
                                          ak.ctypes.data if len(ak) else None, len(ak)))

    def run(self, rec_base, qual=20, fast=True):
        ctx, comm = self.ctx, self.comm
        L, h, dev = ctx.L, ctx.h, comm.device
        self._keep = []
        self._mark(None)
        ctx._check(L.bk_shard_begin(h, rec_base, qual))
        self._mark("begin (stream pass)")
        st = abi.ShardStats()
        ctx._check(L.bk_shard_get_stats(h, C.byref(st)))
        sums = comm.all_reduce(torch.tensor([st.isize_sum, st.isize_n], dtype=torch.int64, device=dev))
        sq = comm.all_reduce(torch.tensor([st.sumsq], dtype=torch.float64, device=dev))
        mx = comm.all_reduce(torch.tensor([st.vmax, st.max_span], dtype=torch.int64, device=dev), op="max")
        st.isize_sum, st.isize_n = int(sums[0]), int(sums[1])
        st.sumsq = float(sq[0])
        st.vmax, st.max_span = int(mx[0]), int(mx[1])
        ctx._check(L.bk_shard_set_stats(h, C.byref(st)))
        # bit-exact sd: exceptions of all shards replayed in global record order
        lt, exp, nex = C.c_uint64(), C.c_void_p(), C.c_uint64()
        ctx._check(L.bk_shard_sd_local(h, C.byref(lt), C.byref(exp), C.byref(nex)))
        per = comm.all_gather_scalars([lt.value, nex.value])
        offset = int(per[: comm.rank, 0].sum())
        ex = tensor_from_ptr(exp.value, nex.value * 16, dev).clone()
        if nex.value and offset:
            v = ex.view(torch.int64).view(-1, 2)
            v[:, 0] += offset  # l_before becomes global (two's complement add on the u64 bits)
        all_ex = comm.all_gather_var(ex)
        self._keep.append(all_ex)
        mean, sd = C.c_double(), C.c_double()
        ctx._check(L.bk_shard_sd_finish(h, C.c_void_p(all_ex.data_ptr() if all_ex.numel() else 0), all_ex.numel() // 16, int(per[:, 0].sum()),
                                        C.byref(mean), C.byref(sd)))
        w = capi.w_from(mean.value, sd.value)
        self._mark("statistics + exact sd")
        if self.routed:
            self._routed_join(qual, w)
        else:
            # candidates -> replicated mate join, every rank then masks / clusters the groups it owns
            self._gather_into(abi.BUF_CANDIDATES)
            ctx.discordant_pairs(qual, w)
            starts, ng = C.POINTER(C.c_uint64)(), C.c_uint32()
            ctx._check(L.bk_shard_group_sizes(h, C.byref(starts), C.byref(ng)))
            sizes = [int(starts[g + 1] - starts[g]) for g in range(ng.value)]
            owner = lpt_owner(sizes, comm.world)
            own = np.asarray([1 if o == comm.rank else 0 for o in owner], dtype=np.uint8)
            ctx._check(L.bk_shard_own_groups(h, own.ctypes.data if len(own) else None, ng.value))
        self._mark("join + routing")
        ctx.mask_and_cluster(w, fast)
        self._mark("mask + cluster")
        ctx.cluster_summary(w)
        # tuples and cluster summaries to everybody
        self._gather_into(abi.BUF_TUPLES)
        ncl = self._gather_into(abi.BUF_CLUSTERS)
        self._mark("summary + gathers")
        # breakpoints: range counts add over the record shards
        p, n = C.c_void_p(), C.c_uint64()
        ctx._check(L.bk_shard_bp_cov(h, w, C.byref(p), C.byref(n)))
        cov = comm.all_reduce(tensor_from_ptr(p.value, n.value * 4, dev).view(torch.int32).clone())
        self._keep.append(cov)
        if self.routed and comm.world > 1:
            # every rank votes for its slice of the cluster table; rows and flags of the slices are gathered in rank order
            lo, hi = ncl * comm.rank // comm.world, ncl * (comm.rank + 1) // comm.world
            cp, vp_ = C.c_void_p(), C.c_void_p()
            ctx._check(L.bk_shard_bp_vote_slice(h, w, C.c_void_p(cov.data_ptr() if cov.numel() else 0), lo, hi, C.byref(cp), C.byref(vp_)))
            rows = comm.all_gather_var(tensor_from_ptr(cp.value, (hi - lo) * 72, dev).clone())
            flags = comm.all_gather_var(tensor_from_ptr(vp_.value, (hi - lo) * 4, dev).clone())
            self._keep += [rows, flags]
            ctx._check(L.bk_shard_set_buffer(h, abi.BUF_CLUSTERS, C.c_void_p(rows.data_ptr() if rows.numel() else 0), rows.numel() // 72))
            ctx._check(L.bk_shard_bp_set_voted(h, C.c_void_p(flags.data_ptr() if flags.numel() else 0)))
        else:
            ctx._check(L.bk_shard_bp_vote(h, w, C.c_void_p(cov.data_ptr() if cov.numel() else 0)))
        ctx._check(L.bk_shard_bp_depth(h, C.byref(p), C.byref(n)))
        dep = comm.all_reduce(tensor_from_ptr(p.value, n.value * 4, dev).view(torch.int32).clone())
        self._keep.append(dep)
        ctx._check(L.bk_shard_bp_finish(h, C.c_void_p(dep.data_ptr() if dep.numel() else 0)))
        self._mark("breakpoints")
        self.mean, self.sd, self.w, self.n_clusters = mean.value, sd.value, w, ncl
        return w
