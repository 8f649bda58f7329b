// libbreakid_rccl.so: one sample sharded over the GPUs of a node, orchestrated from C++ (include/breakid_multi.h).
// One host thread per rank drives its own bk_ctx through the bk_shard_* entry points of libbreakid_hip.so; the
// exchanges go through a Transport: RCCL (librccl called directly, every call queued on the context's HIP stream) or
// device-to-device copies between contexts of this process.  Sequence = SURVEY 8(e), the same as breakid_amd/sharded.py
// (routed variant): the reference's groups are independent (BreakID.cc:119-167), so only candidates, pairs and the small
// tuple / cluster tables travel.
#include <rccl/rccl.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>

#include "../../include/breakid_multi.h"
#include "bk_common.h"

namespace
{
#define NCCL_CHECK(expr)                                                                                                   \
  do                                                                                                                       \
  {                                                                                                                        \
    ncclResult_t _r = (expr);                                                                                              \
    if (_r != ncclSuccess) throw bk_error(BK_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(_r));                 \
  } while (0)

struct Transport
{
  int rank = 0, world = 1;
  virtual ~Transport() {}
  // what this rank sent to the other ranks / took in from them, per kind of exchange (BK_DEBUG=multi prints it against DESIGN section
  // 6's table: every per-record and per-pair stage is O(n / W) per rank)
  struct Tally
  {
    uint64_t calls = 0, sent = 0, received = 0;
  };
  Tally tally[4];  // 0 host metadata (all-gather), 1 all-gather of device tables, 2 all-to-all, 3 all-reduce
  std::map<std::string, Tally> by_step;
  const char *step = "";
  void count(int kind, uint64_t sent, uint64_t received)
  {
    tally[kind].calls += 1;
    tally[kind].sent += sent;
    tally[kind].received += received;
    Tally &t = by_step[step];
    t.calls += 1;
    t.sent += sent;
    t.received += received;
  }
  // small host metadata: `bytes` from every rank, concatenated in rank order
  void allgather_host(const void *in, size_t bytes, void *out, hipStream_t st)
  {
    count(0, bytes * (size_t) (world - 1), bytes * (size_t) (world - 1));
    do_allgather_host(in, bytes, out, st);
  }
  // device buffers of rank-dependent size -> concatenation in rank order at recv (sizes[r] bytes from rank r)
  void allgatherv(const void *send, void *recv, const std::vector<size_t> &sizes, hipStream_t st)
  {
    uint64_t total = 0;
    for (size_t v : sizes) total += v;
    count(1, (uint64_t) sizes[rank] * (uint64_t) (world - 1), total - sizes[rank]);
    do_allgatherv(send, recv, sizes, st);
  }
  // send = world blocks (sb[d] bytes for rank d, back to back); recv = what every rank sent here, in rank order
  void alltoallv(const void *send, const std::vector<size_t> &sb, void *recv, const std::vector<size_t> &rb, hipStream_t st)
  {
    uint64_t so = 0, ro = 0;
    for (int r = 0; r < world; ++r)
      if (r != rank)
      {
        so += sb[r];
        ro += rb[r];
      }
    count(2, so, ro);
    do_alltoallv(send, sb, recv, rb, st);
  }
  void allreduce_sum_u32(uint32_t *buf, size_t n, hipStream_t st)
  {
    const uint64_t ring = world > 1 ? 2ull * (uint64_t) (world - 1) * (n * 4ull) / (uint64_t) world : 0ull;  // what a ring moves per rank
    count(3, ring, ring);
    do_allreduce_sum_u32(buf, n, st);
  }
  virtual void do_allgather_host(const void *in, size_t bytes, void *out, hipStream_t st) = 0;
  virtual void do_allgatherv(const void *send, void *recv, const std::vector<size_t> &sizes, hipStream_t st) = 0;
  virtual void do_alltoallv(const void *send, const std::vector<size_t> &sb, void *recv, const std::vector<size_t> &rb, hipStream_t st) = 0;
  virtual void do_allreduce_sum_u32(uint32_t *buf, size_t count, hipStream_t st) = 0;
};

// ---- the rank threads of this process meet here: before every exchange (both transports), so that a rank that failed on its own
// (a corrupt block in its part of the file, out of memory on its GPU, an error raised in its shard) takes the others out with
// "another rank failed" instead of leaving them inside a collective nobody else will enter ---------------------------------------
struct LocalHub
{
  int world;
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0;
  uint64_t generation = 0;
  bool failed = false;
  std::atomic<bool> failed_flag{false};  // the same, for the owners of communicators to poll without the lock
  std::vector<const void *> ptr;
  std::vector<std::vector<size_t>> blocks;  // alltoallv: blocks[r][d] = bytes rank r sends to rank d
  std::vector<std::vector<uint8_t>> host;
  explicit LocalHub(int w) : world(w), ptr(w), blocks(w), host(w) {}
  void barrier()
  {
    std::unique_lock<std::mutex> l(m);
    if (failed) throw bk_error(BK_ERR_HIP, "another rank failed");
    const uint64_t g = generation;
    if (++waiting == world)
    {
      waiting = 0;
      ++generation;
      cv.notify_all();
    }
    else
      cv.wait(l, [&] { return generation != g || failed; });
    if (failed) throw bk_error(BK_ERR_HIP, "another rank failed");
  }
  void fail()
  {
    std::lock_guard<std::mutex> l(m);
    failed = true;
    failed_flag.store(true);
    cv.notify_all();
  }
};
// ---- RCCL: one rank per GPU --------------------------------------------------------------------------------------------------
struct RcclTransport : Transport
{
  ncclComm_t comm = nullptr;
  LocalHub *hub = nullptr;
  std::atomic<bool> aborted{false};
  DevBuf sin, sout;
  ~RcclTransport() override
  {
    if (comm && !aborted.exchange(true)) (void) ncclCommDestroy(comm);
  }
  // A collective is only queued when every rank thread has arrived in front of it in good health (LocalHub::barrier throws on all
  // of them otherwise).  What is left is a rank that fails AFTER that meeting (inside a librccl call, or between two meetings): it
  // raises the hub's flag and nothing else - a communicator is only ever touched by the thread that owns it.  Every owner waits
  // for the collective it has queued HERE, polling its stream and that flag, and aborts its own communicator when the flag goes up
  // (the collective would wait for the failed rank for ever otherwise).  A rank inside ncclCommInitRank cannot be reached that way
  // (blocking initialisation; the unique id is only handed out when every rank thread exists).
  void abort_own()
  {
    if (comm && !aborted.exchange(true)) (void) ncclCommAbort(comm);
  }
  void finish(hipStream_t st)
  {
    for (uint32_t spins = 0;; ++spins)
    {
      const hipError_t e = hipStreamQuery(st);
      if (e == hipSuccess) return;
      if (e != hipErrorNotReady) throw bk_error(BK_ERR_HIP, std::string("RCCL exchange: ") + hipGetErrorString(e));
      if (hub->failed_flag.load())
      {
        abort_own();
        throw bk_error(BK_ERR_HIP, "another rank failed");
      }
      if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  void do_allgather_host(const void *in, size_t bytes, void *out, hipStream_t st) override
  {
    hub->barrier();
    void *di = sin.ensure(bytes + 16), *dout = sout.ensure(bytes * world + 16);
    HIP_CHECK(hipMemcpyAsync(di, in, bytes, hipMemcpyHostToDevice, st));
    NCCL_CHECK(ncclAllGather(di, dout, bytes, ncclChar, comm, st));
    HIP_CHECK(hipMemcpyAsync(out, dout, bytes * world, hipMemcpyDeviceToHost, st));
    finish(st);
  }
  void do_allgatherv(const void *send, void *recv, const std::vector<size_t> &sizes, hipStream_t st) override
  {
    // one broadcast per root inside a group: no padding to the largest rank, no staging copy
    hub->barrier();
    size_t off = 0;
    NCCL_CHECK(ncclGroupStart());
    for (int r = 0; r < world; ++r)
    {
      if (sizes[r])
      {
        char *dst = (char *) recv + off;
        NCCL_CHECK(ncclBroadcast(r == rank ? send : (const void *) dst, dst, sizes[r], ncclChar, r, comm, st));
      }
      off += sizes[r];
    }
    NCCL_CHECK(ncclGroupEnd());
    finish(st);
  }
  void do_alltoallv(const void *send, const std::vector<size_t> &sb, void *recv, const std::vector<size_t> &rb, hipStream_t st) override
  {
    hub->barrier();
    size_t so = 0, ro = 0;
    NCCL_CHECK(ncclGroupStart());
    for (int r = 0; r < world; ++r)
    {
      if (sb[r]) NCCL_CHECK(ncclSend((const char *) send + so, sb[r], ncclChar, r, comm, st));
      if (rb[r]) NCCL_CHECK(ncclRecv((char *) recv + ro, rb[r], ncclChar, r, comm, st));
      so += sb[r];
      ro += rb[r];
    }
    NCCL_CHECK(ncclGroupEnd());
    finish(st);
  }
  void do_allreduce_sum_u32(uint32_t *buf, size_t count, hipStream_t st) override
  {
    hub->barrier();
    if (count) NCCL_CHECK(ncclAllReduce(buf, buf, count, ncclUint32, ncclSum, comm, st));
    finish(st);
  }
};

// ---- contexts of one process: device-to-device copies between the meetings ------------------------------------------------------
struct LocalTransport : Transport
{
  LocalHub *hub = nullptr;
  void do_allgather_host(const void *in, size_t bytes, void *out, hipStream_t) override
  {
    hub->host[rank].assign((const uint8_t *) in, (const uint8_t *) in + bytes);
    hub->barrier();
    for (int r = 0; r < world; ++r) memcpy((char *) out + (size_t) r * bytes, hub->host[r].data(), bytes);
    hub->barrier();
  }
  void do_allgatherv(const void *send, void *recv, const std::vector<size_t> &sizes, hipStream_t st) override
  {
    HIP_CHECK(hipStreamSynchronize(st));  // the peers read this rank's buffer with their own streams
    hub->ptr[rank] = send;
    hub->barrier();
    size_t off = 0;
    for (int r = 0; r < world; ++r)
    {
      if (sizes[r]) HIP_CHECK(hipMemcpyAsync((char *) recv + off, hub->ptr[r], sizes[r], hipMemcpyDefault, st));
      off += sizes[r];
    }
    HIP_CHECK(hipStreamSynchronize(st));
    hub->barrier();
  }
  void do_alltoallv(const void *send, const std::vector<size_t> &sb, void *recv, const std::vector<size_t> &rb, hipStream_t st) override
  {
    HIP_CHECK(hipStreamSynchronize(st));
    hub->ptr[rank] = send;
    hub->blocks[rank] = sb;
    hub->barrier();
    size_t ro = 0;
    for (int r = 0; r < world; ++r)
    {
      size_t so = 0;
      for (int d = 0; d < rank; ++d) so += hub->blocks[r][d];
      if (rb[r]) HIP_CHECK(hipMemcpyAsync((char *) recv + ro, (const char *) hub->ptr[r] + so, rb[r], hipMemcpyDefault, st));
      ro += rb[r];
    }
    HIP_CHECK(hipStreamSynchronize(st));
    hub->barrier();
  }
  void do_allreduce_sum_u32(uint32_t *buf, size_t count, hipStream_t st) override
  {
    hub->host[rank].resize(count * 4);
    if (count) HIP_CHECK(hipMemcpyAsync(hub->host[rank].data(), buf, count * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    hub->barrier();
    std::vector<uint32_t> sum(count, 0);
    for (int r = 0; r < world; ++r)
    {
      const uint32_t *p = (const uint32_t *) hub->host[r].data();
      for (size_t i = 0; i < count; ++i) sum[i] += p[i];
    }
    hub->barrier();
    if (count) HIP_CHECK(hipMemcpyAsync(buf, sum.data(), count * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
  }
};

__global__ void k_add_l_before(unsigned long long *ex, unsigned long long n, unsigned long long offset)
{
  unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ex[2 * i] += offset;  // SdException{l_before, d}: l_before becomes global
}

#define BK_CALL(expr)                                                                                  \
  do                                                                                                   \
  {                                                                                                    \
    int _rc = (expr);                                                                                  \
    if (_rc != BK_OK) throw bk_error(_rc, std::string(bk_last_error(ctx)));                            \
  } while (0)

// Longest-processing-time assignment of chr-pair groups to ranks (deterministic on every rank)
std::vector<uint32_t> lpt_owner(const std::vector<uint64_t> &sizes, int world)
{
  std::vector<uint32_t> order(sizes.size()), owner(sizes.size(), 0);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sizes[a] != sizes[b] ? sizes[a] > sizes[b] : a < b; });
  std::vector<uint64_t> load(world, 0);
  for (uint32_t g : order)
  {
    int best = 0;
    for (int r = 1; r < world; ++r)
      if (load[r] < load[best]) best = r;
    owner[g] = (uint32_t) best;
    load[best] += sizes[g];
  }
  return owner;
}

struct Keep
{
  std::vector<std::unique_ptr<DevBuf>> bufs;
  void *get(size_t bytes)
  {
    bufs.emplace_back(new DevBuf());
    return bufs.back()->ensure(bytes + 16);
  }
};

// gathers the rank-local table `which` of every rank and hands the concatenation back to the context
uint64_t gather_table(bk_ctx *ctx, Transport &T, int which, Keep &keep, hipStream_t st)
{
  void *dev = nullptr;
  uint64_t n = 0;
  uint32_t eb = 0;
  BK_CALL(bk_shard_buffer(ctx, which, &dev, &n, &eb));
  std::vector<uint64_t> counts(T.world);
  T.allgather_host(&n, 8, counts.data(), st);
  std::vector<size_t> sizes(T.world);
  size_t total = 0;
  for (int r = 0; r < T.world; ++r) total += sizes[r] = (size_t) counts[r] * eb;
  void *all = keep.get(total);
  T.allgatherv(dev, all, sizes, st);
  BK_CALL(bk_shard_set_buffer(ctx, which, all, total / eb));
  return total / eb;
}

struct RankSummary
{
  double mean = 0, sd = 0;
  std::vector<bk_group_stat> groups;  // every group of the sample in the reference's order, summed over the ranks (a group lives on one)
};
void run_rank(bk_ctx *ctx, Transport &T, uint64_t rec_base, int qual, int fast, double *w_out, uint64_t *n_clustered_total, Keep &keep, RankSummary &sum)
{
  void *stv = nullptr;
  BK_CALL(bk_get_stream(ctx, &stv));
  hipStream_t st = (hipStream_t) stv;
  const int W = T.world;
  T.step = "insert-size sums, spans";
  BK_CALL(bk_shard_begin(ctx, rec_base, qual));
  // insert-size sums / spans
  bk_shard_stats s;
  BK_CALL(bk_shard_get_stats(ctx, &s));
  {
    std::vector<bk_shard_stats> all(W);
    T.allgather_host(&s, sizeof s, all.data(), st);
    bk_shard_stats t = all[0];
    for (int r = 1; r < W; ++r)
    {
      t.isize_sum += all[r].isize_sum;
      t.isize_n += all[r].isize_n;
      t.sumsq += all[r].sumsq;  // rank order on every rank: the same double everywhere
      t.vmax = std::max(t.vmax, all[r].vmax);
      t.max_span = std::max(t.max_span, all[r].max_span);
    }
    BK_CALL(bk_shard_set_stats(ctx, &t));
  }
  // bit-exact sd: the exceptions of all shards, replayed in global record order
  double mean = 0, sd = 0;
  {
    uint64_t lt = 0, nex = 0;
    void *exd = nullptr;
  T.step = "bit-exact sd: exception lists";
    BK_CALL(bk_shard_sd_local(ctx, &lt, &exd, &nex));
    uint64_t mine[2] = {lt, nex};
    std::vector<uint64_t> per(2 * W);
    T.allgather_host(mine, 16, per.data(), st);
    uint64_t offset = 0, l_grand = 0;
    std::vector<size_t> sizes(W);
    size_t total = 0;
    for (int r = 0; r < W; ++r)
    {
      if (r < T.rank) offset += per[2 * r];
      l_grand += per[2 * r];
      total += sizes[r] = (size_t) per[2 * r + 1] * 16;
    }
    void *own = keep.get(nex * 16);
    if (nex)
    {
      HIP_CHECK(hipMemcpyAsync(own, exd, nex * 16, hipMemcpyDeviceToDevice, st));
      if (offset) hipLaunchKernelGGL(k_add_l_before, dim3((unsigned) ((nex + 255) / 256)), dim3(256), 0, st, (unsigned long long *) own, nex, offset);
    }
    void *all = keep.get(total);
    T.allgatherv(own, all, sizes, st);
    BK_CALL(bk_shard_sd_finish(ctx, all, total / 16, l_grand, &mean, &sd));
  }
  const int times = 2;
  const double w = times * std::sqrt((double) times) * (mean + 3 * sd);  // BreakID.cc:103
  if (w_out) *w_out = w;
  // candidates -> owner of the read-name hash -> local mate join
  {
    void *send = nullptr;
    const uint64_t *cnt = nullptr;
  T.step = "mate join: candidates to the owner of their read name";
    BK_CALL(bk_shard_route_candidates(ctx, (uint32_t) W, &send, &cnt));
    void *dummy = nullptr;
    uint64_t n0 = 0;
    uint32_t eb = 0;
    BK_CALL(bk_shard_buffer(ctx, BK_BUF_CANDIDATES, &dummy, &n0, &eb));
    std::vector<uint64_t> mine(cnt, cnt + W), per((size_t) W * W);
    T.allgather_host(mine.data(), 8 * W, per.data(), st);
    std::vector<size_t> sb(W), rb(W);
    size_t rtotal = 0;
    for (int r = 0; r < W; ++r)
    {
      sb[r] = (size_t) mine[r] * eb;
      rtotal += rb[r] = (size_t) per[(size_t) r * W + T.rank] * eb;
    }
    void *recv = keep.get(rtotal);
    T.alltoallv(send, sb, recv, rb, st);
    BK_CALL(bk_shard_set_buffer(ctx, BK_BUF_CANDIDATES, recv, rtotal / eb));
    BK_CALL(bk_discordant_pairs(ctx, qual, w, nullptr, nullptr));
  }
  // pairs -> owner of their chromosome-pair group (LPT on the group totals)
  {
    const uint64_t *starts = nullptr;
    const uint32_t *keys = nullptr;
    uint32_t ng = 0;
    BK_CALL(bk_shard_group_sizes(ctx, &starts, &ng));
    BK_CALL(bk_shard_group_keys(ctx, &keys, &ng));
    uint64_t ngl = ng;
    std::vector<uint64_t> ngs(W);
    T.allgather_host(&ngl, 8, ngs.data(), st);
    const uint64_t mx = *std::max_element(ngs.begin(), ngs.end());
    std::vector<uint64_t> mine(2 * mx + 2, 0), per((2 * mx + 2) * W);
    for (uint32_t g = 0; g < ng; ++g)
    {
      mine[2 * g] = keys[g];
      mine[2 * g + 1] = starts[g + 1] - starts[g];
    }
    T.allgather_host(mine.data(), mine.size() * 8, per.data(), st);
    std::map<uint32_t, uint64_t> tot;
    for (int r = 0; r < W; ++r)
      for (uint64_t g = 0; g < ngs[r]; ++g) tot[(uint32_t) per[r * mine.size() + 2 * g]] += per[r * mine.size() + 2 * g + 1];
    std::vector<uint32_t> gkeys;
    std::vector<uint64_t> gsizes;
    for (auto &kv : tot)
    {
      gkeys.push_back(kv.first);
      gsizes.push_back(kv.second);
    }
    const std::vector<uint32_t> own = lpt_owner(gsizes, W);
    std::map<uint32_t, uint32_t> owner_of;
    for (size_t i = 0; i < gkeys.size(); ++i) owner_of[gkeys[i]] = own[i];
    std::vector<uint32_t> dest(ng ? ng : 1, 0);
    for (uint32_t g = 0; g < ng; ++g) dest[g] = owner_of[keys[g]];
    void *send = nullptr;
    const uint64_t *cnt = nullptr;
  T.step = "grouping: pairs to the owner of their group";
    BK_CALL(bk_shard_route_pairs(ctx, dest.data(), ng, (uint32_t) W, &send, &cnt));
    std::vector<uint64_t> mc(cnt, cnt + W), pc((size_t) W * W);
    T.allgather_host(mc.data(), 8 * W, pc.data(), st);
    std::vector<size_t> sb(W), rb(W);
    size_t rtotal = 0;
    for (int r = 0; r < W; ++r)
    {
      sb[r] = (size_t) mc[r] * sizeof(bk_pair);
      rtotal += rb[r] = (size_t) pc[(size_t) r * W + T.rank] * sizeof(bk_pair);
    }
    void *recv = keep.get(rtotal);
    T.alltoallv(send, sb, recv, rb, st);
    BK_CALL(bk_shard_group_pairs(ctx, recv, rtotal / sizeof(bk_pair), gkeys.data(), (uint32_t) gkeys.size()));
  }
  uint64_t n_clustered = 0;
  T.step = "split evidence, cluster summaries";
  const auto t_mc0 = std::chrono::steady_clock::now();
  BK_CALL(bk_mask_and_cluster(ctx, w, fast, &n_clustered));
  HIP_CHECK(hipStreamSynchronize(st));
  const double t_mc = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_mc0).count();
  BK_CALL(bk_cluster_summary(ctx, w, nullptr));
  {
    std::vector<uint64_t> all(W);
    T.allgather_host(&n_clustered, 8, all.data(), st);
    if (n_clustered_total) *n_clustered_total = std::accumulate(all.begin(), all.end(), (uint64_t) 0);
  }
  {
    // per-group counters (the _performance.txt columns of BreakID.cc:175-191): every context lists all groups of the sample in the
    // same order and holds the pairs of the groups it owns, so the rows add up over the ranks
    const bk_group_stat *gs = nullptr;
    uint32_t ngs = 0;
    BK_CALL(bk_group_stats(ctx, &gs, &ngs));
    sum.mean = mean;
    sum.sd = sd;
    sum.groups.assign(gs, gs + ngs);
    if (bk_debug("multi"))
    {
      uint64_t a = 0, b = 0, c = 0;
      for (uint32_t g = 0; g < ngs; ++g)
      {
        a += gs[g].n_scan;
        b += gs[g].n_isolated_removed;
        c += gs[g].n_clustered;
      }
      fprintf(stderr, "[multi] rank %d: %u groups, scan %llu, after the masks %llu, clustered %llu\n", T.rank, ngs, (unsigned long long) a, (unsigned long long) b, (unsigned long long) c);
    }
    if (W > 1)
    {
      // a group lives on the rank that owns it; its row carries the group's ordinal in the reference's order among all groups of
      // the sample, so the whole list is every rank's rows (padded to the longest list for the exchange) sorted by ordinal
      uint64_t mine = ngs;
      std::vector<uint64_t> cnts(W);
      T.allgather_host(&mine, 8, cnts.data(), st);
      const uint64_t mx = *std::max_element(cnts.begin(), cnts.end());
      sum.groups.clear();
      if (mx)
      {
        std::vector<bk_group_stat> send((size_t) mx, bk_group_stat{}), all((size_t) W * mx);
        for (uint64_t g = 0; g < ngs; ++g) send[g] = gs[g];
        T.allgather_host(send.data(), (size_t) mx * sizeof(bk_group_stat), all.data(), st);
        for (int r = 0; r < W; ++r)
          for (uint64_t g = 0; g < cnts[r]; ++g) sum.groups.push_back(all[(size_t) r * mx + g]);
        std::stable_sort(sum.groups.begin(), sum.groups.end(), [](const bk_group_stat &a, const bk_group_stat &b) { return a.ordinal < b.ordinal; });
      }
    }
  }
  // evidence tuples and cluster summaries to everybody
  gather_table(ctx, T, BK_BUF_TUPLES, keep, st);
  const uint64_t ncl = gather_table(ctx, T, BK_BUF_CLUSTERS, keep, st);
  // breakpoints: coverage and depth are range counts over records, they add over the record shards
  void *p = nullptr;
  uint64_t n = 0;
  T.step = "breakpoints: coverage counts, voted rows, depth counts";
  BK_CALL(bk_shard_bp_cov(ctx, w, &p, &n));
  T.allreduce_sum_u32((uint32_t *) p, n, st);
  if (W > 1)
  {
    // every rank votes for its slice of the cluster table; rows and flags are gathered in rank order
    const uint64_t lo = ncl * T.rank / W, hi = ncl * (T.rank + 1) / W;
    void *rows = nullptr, *flags = nullptr;
    BK_CALL(bk_shard_bp_vote_slice(ctx, w, p, lo, hi, &rows, &flags));
    std::vector<size_t> rs(W), fs(W);
    for (int r = 0; r < W; ++r)
    {
      const uint64_t l = ncl * r / W, h = ncl * (r + 1) / W;
      rs[r] = (size_t) (h - l) * sizeof(bk_cluster);
      fs[r] = (size_t) (h - l) * 4;
    }
    void *all_rows = keep.get(ncl * sizeof(bk_cluster)), *all_flags = keep.get(ncl * 4);
    T.allgatherv(rows, all_rows, rs, st);
    T.allgatherv(flags, all_flags, fs, st);
    BK_CALL(bk_shard_set_buffer(ctx, BK_BUF_CLUSTERS, all_rows, ncl));
    BK_CALL(bk_shard_bp_set_voted(ctx, all_flags));
  }
  else
    BK_CALL(bk_shard_bp_vote(ctx, w, p));
  BK_CALL(bk_shard_bp_depth(ctx, &p, &n));
  T.allreduce_sum_u32((uint32_t *) p, n, st);
  BK_CALL(bk_shard_bp_finish(ctx, p));
  HIP_CHECK(hipStreamSynchronize(st));
  if (bk_debug("multi"))
  {
    // what crossed the links, per step (sent to the other ranks / taken in from them), and this rank's mask + cluster time: the O(n / W)
    // claim of DESIGN section 6 and the balance of the groups' ownership can be read off a one-GPU rehearsal (-comm local)
    std::string o = "[multi] rank " + std::to_string(T.rank) + " of " + std::to_string(W) + ": mask + cluster " + std::to_string(t_mc) + " ms, " + std::to_string(n_clustered) + " pairs clustered here;";
    for (const auto &kv : T.by_step) o += " [" + kv.first + ": " + std::to_string(kv.second.calls) + " exchanges, " + std::to_string(kv.second.sent) + " B sent, " + std::to_string(kv.second.received) + " B received]";
    fprintf(stderr, "%s\n", o.c_str());
  }
}
}  // namespace

// the orchestration both entry points share: W rank threads, each with its transport and its context; `load` gives rank r's
// context its records (and creates the context: the host-table entry knows the targets up front, the BAM entry learns them
// from the file) and returns the rank's record count; the counts give every rank its rec_base.
namespace
{
// what the context handed out by bk_multi_run / bk_multi_run_bam points at (gathered tables, rank 0's decoded records)
struct Owned
{
  Keep keep;
  bk_bam_dev *bam = nullptr;
  RankSummary sum;
};
std::map<bk_ctx *, Owned> &owned()
{
  static std::map<bk_ctx *, Owned> *m = new std::map<bk_ctx *, Owned>();
  return *m;
}
std::mutex &owned_m()
{
  static std::mutex *m = new std::mutex();
  return *m;
}
struct RankInput
{
  bk_ctx *ctx = nullptr;
  uint64_t n = 0;
};
template <class Load, class Release>
int multi_run_common(int n_gpus, int transport, int mapq_min, int fast, double *w_out, uint64_t *n_clustered_total, bk_ctx **ctx0_out, char *err, size_t errlen, const char *who,
                     Load &&load, Release &&release_rank)
{
  auto fail = [&](int code, const std::string &m) {
    if (err && errlen) snprintf(err, errlen, "%s", m.c_str());
    return code;
  };
  if (!ctx0_out || n_gpus < 1 || n_gpus > 64) return fail(BK_ERR_ARG, std::string(who) + ": bad arguments");
  *ctx0_out = nullptr;
  bk_prepare_process();  // (the process's environment is settled before the runtime starts and before there are rank threads)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(BK_ERR_NO_DEVICE, std::string(who) + ": no HIP device (this library has no CPU path)");
  if (transport == BK_TRANSPORT_AUTO) transport = ndev >= n_gpus ? BK_TRANSPORT_RCCL : BK_TRANSPORT_LOCAL;
  if (transport == BK_TRANSPORT_RCCL && ndev < n_gpus) return fail(BK_ERR_ARG, std::string(who) + ": RCCL needs one device per rank (use --comm local to share a GPU)");
  const int W = n_gpus;
  std::vector<bk_ctx *> ctxs(W, nullptr);
  std::vector<std::string> errs(W);
  std::vector<int> codes(W, BK_OK);
  std::vector<double> ws(W, 0);
  std::vector<uint64_t> ncl(W, 0);
  ncclUniqueId id;
  if (transport == BK_TRANSPORT_RCCL && ncclGetUniqueId(&id) != ncclSuccess) return fail(BK_ERR_HIP, "ncclGetUniqueId failed");
  LocalHub hub(W);
  std::vector<Keep> keeps(W);
  std::vector<RankSummary> sums(W);
  auto fail_all = [&] { hub.fail(); };  // (the owners of the communicators see the flag: RcclTransport::finish)
  std::vector<std::thread> th;
  for (int r = 0; r < W; ++r)
    th.emplace_back([&, r] {
      std::unique_ptr<Transport> T;
      try
      {
        const int dev = r % ndev;
        HIP_CHECK(hipSetDevice(dev));
        if (transport == BK_TRANSPORT_RCCL)
        {
          auto *t = new RcclTransport();
          T.reset(t);
          t->rank = r;
          t->world = W;
          t->hub = &hub;
          NCCL_CHECK(ncclCommInitRank(&t->comm, W, id, r));
        }
        else
        {
          auto *t = new LocalTransport();
          T.reset(t);
          t->rank = r;
          t->world = W;
          t->hub = &hub;
        }
        const RankInput in = load(r, W, dev);
        bk_ctx *ctx = in.ctx;
        ctxs[r] = ctx;
        // rec_base = records of the ranks in front of this one
        void *stv = nullptr;
        BK_CALL(bk_get_stream(ctx, &stv));
        std::vector<uint64_t> counts(W);
        T->allgather_host(&in.n, 8, counts.data(), (hipStream_t) stv);
        uint64_t base = 0;
        for (int k = 0; k < r; ++k) base += counts[k];
        run_rank(in.ctx, *T, base, mapq_min, fast, &ws[r], &ncl[r], keeps[r], sums[r]);
      }
      catch (const bk_error &e)
      {
        codes[r] = e.code;
        errs[r] = e.msg;
        fail_all();
        if (auto *t = dynamic_cast<RcclTransport *>(T.get())) t->abort_own();  // (its own communicator, from its own thread)
      }
      catch (const std::exception &e)
      {
        codes[r] = BK_ERR_HIP;
        errs[r] = e.what();
        fail_all();
        if (auto *t = dynamic_cast<RcclTransport *>(T.get())) t->abort_own();
      }
    });
  for (auto &t : th) t.join();
  int rc = BK_OK;
  std::string msg;
  for (int r = 0; r < W; ++r)
    if (codes[r] != BK_OK && (rc == BK_OK || errs[r] != "another rank failed"))
    {
      rc = codes[r];
      msg = "rank " + std::to_string(r) + ": " + errs[r];
    }
  // the gathered tables a context points at must outlive it: they are released after every context
  for (int r = 1; r < W; ++r)
  {
    if (ctxs[r]) bk_free(ctxs[r]);
    release_rank(r);
  }
  if (rc != BK_OK)
  {
    if (ctxs[0]) bk_free(ctxs[0]);
    release_rank(0);
    return fail(rc, msg);
  }
  // rank 0's context keeps pointing at its gathered tables (and at its records): they are parked under the context and released
  // by bk_multi_free
  {
    std::lock_guard<std::mutex> l(owned_m());
    owned()[ctxs[0]].keep = std::move(keeps[0]);
    owned()[ctxs[0]].sum = std::move(sums[0]);
  }
  if (w_out) *w_out = ws[0];
  if (n_clustered_total) *n_clustered_total = ncl[0];
  *ctx0_out = ctxs[0];
  return BK_OK;
}
}  // namespace

extern "C" int bk_multi_run(const bk_soa *tab, const uint32_t *target_len, const char *const *target_name, int n_targets, int n_gpus, int transport, int mapq_min, int fast,
                            double *w_out, uint64_t *n_clustered_total, bk_ctx **ctx0_out, char *err, size_t errlen)
{
  if (!tab)
  {
    if (err && errlen) snprintf(err, errlen, "bk_multi_run: bad arguments");
    return BK_ERR_ARG;
  }
  const uint64_t n = tab->n;
  auto load = [&](int r, int W, int dev) {
    RankInput in;
    if (bk_init(dev, target_len, target_name, n_targets, &in.ctx) != BK_OK) throw bk_error(BK_ERR_NO_DEVICE, bk_last_error(nullptr));
    // contiguous record range of this rank; the CIGAR / aux offsets of a slice are rebased to its own blobs
    const uint64_t lo = n * r / W, hi = n * (r + 1) / W, m = hi - lo;
    std::vector<uint32_t> coff(m + 1), aoff(m + 1);
    const uint32_t c0 = n ? tab->cigar_off[lo] : 0, a0 = n ? tab->aux_off[lo] : 0;
    for (uint64_t i = 0; i <= m; ++i)
    {
      coff[i] = (n ? tab->cigar_off[lo + i] : 0) - c0;
      aoff[i] = (n ? tab->aux_off[lo + i] : 0) - a0;
    }
    bk_soa s = *tab;
    s.n = m;
    s.tid += lo; s.pos += lo; s.mtid += lo; s.mpos += lo; s.isize += lo; s.flag += lo; s.mapq += lo; s.qhash += lo;
    if (s.qcheck) s.qcheck += lo;
    s.cigar_off = coff.data();
    s.aux_off = aoff.data();
    s.cigar = tab->cigar + c0;
    s.aux = tab->aux + a0;
    s.n_cigar_words = coff[m];
    s.n_aux_bytes = aoff[m];
    const int urc = bk_upload_records(in.ctx, &s, BK_MEM_HOST);
    if (urc != BK_OK)
    {
      const bk_error e(urc, bk_last_error(in.ctx));
      bk_free(in.ctx);
      throw e;
    }
    in.n = m;
    return in;
  };
  return multi_run_common(n_gpus, transport, mapq_min, fast, w_out, n_clustered_total, ctx0_out, err, errlen, "bk_multi_run", load, [](int) {});
}

// the same with every rank decoding its own part of the file on its own GPU (bk_bam_decode_device_part)
extern "C" int bk_multi_run_bam(const char *path, int n_gpus, int transport, int mapq_min, int fast, double *w_out, uint64_t *n_clustered_total, bk_ctx **ctx0_out,
                                int *n_targets, const char *const **names, const uint32_t **lens, char *err, size_t errlen)
{
  if (!path || n_gpus < 1 || n_gpus > 64)
  {
    if (err && errlen) snprintf(err, errlen, "bk_multi_run_bam: bad arguments");
    return BK_ERR_ARG;
  }
  std::vector<bk_bam_dev *> bams(n_gpus, nullptr);
  std::vector<int> nts(n_gpus, 0);
  std::vector<const char *const *> nms(n_gpus, nullptr);
  std::vector<const uint32_t *> lns(n_gpus, nullptr);
  auto load = [&](int r, int W, int dev) {
    RankInput in;
    bk_soa cols;
    char e[512] = {0};
    const int drc = bk_bam_decode_device_part(path, dev, r, W, &bams[r], &cols, &nts[r], &nms[r], &lns[r], e, sizeof e);
    if (drc != BK_OK) throw bk_error(drc, e);
    bk_feed_release_caches();  // (what is idle of the feed's staging buffers and slots: a rank decodes one part of one file)
    if (bk_init(dev, lns[r], nms[r], nts[r], &in.ctx) != BK_OK) throw bk_error(BK_ERR_NO_DEVICE, bk_last_error(nullptr));
    const int urc = bk_upload_records(in.ctx, &cols, BK_MEM_DEVICE);
    if (urc != BK_OK)
    {
      const bk_error ex(urc, bk_last_error(in.ctx));
      bk_free(in.ctx);
      throw ex;
    }
    in.n = cols.n;
    return in;
  };
  // the records of a rank live in its bk_bam_dev: released behind its context (multi_run_common does that for rank 0 only when
  // the run failed: after a good run rank 0's records stay with the context that is handed out, for the life of the process)
  auto release_rank = [&](int r) {
    if (bams[r]) bk_bam_dev_free(bams[r]);
    bams[r] = nullptr;
  };
  const int rc = multi_run_common(n_gpus, transport, mapq_min, fast, w_out, n_clustered_total, ctx0_out, err, errlen, "bk_multi_run_bam", load, release_rank);
  const bool ok = rc == BK_OK;
  if (ok)
  {
    {
      std::lock_guard<std::mutex> l(owned_m());
      owned()[*ctx0_out].bam = bams[0];
    }
    if (n_targets) *n_targets = nts[0];
    if (names) *names = nms[0];
    if (lens) *lens = lns[0];
  }
  return rc;
}

// releases the context of a finished bk_multi_run / bk_multi_run_bam together with the tables it points at (gathered tuples /
// clusters, rank 0's decoded records and reference names); a context of plain bk_init is simply freed
extern "C" void bk_multi_free(bk_ctx *ctx)
{
  if (!ctx) return;
  Owned o;
  bool found = false;
  {
    std::lock_guard<std::mutex> l(owned_m());
    auto it = owned().find(ctx);
    if (it != owned().end())
    {
      o = std::move(it->second);
      owned().erase(it);
      found = true;
    }
  }
  bk_free(ctx);  // first the context, then what it pointed at
  if (found && o.bam) bk_bam_dev_free(o.bam);
}

// insert-size statistics and per-group counters of the finished run behind `ctx` (what bk_isize_stats / bk_group_stats give for one GPU)
extern "C" int bk_multi_stats(bk_ctx *ctx, double *mean, double *sd, const bk_group_stat **groups, uint32_t *n_groups)
{
  std::lock_guard<std::mutex> l(owned_m());
  auto it = owned().find(ctx);
  if (it == owned().end()) return BK_ERR_ARG;
  if (mean) *mean = it->second.sum.mean;
  if (sd) *sd = it->second.sum.sd;
  if (groups) *groups = it->second.sum.groups.data();
  if (n_groups) *n_groups = (uint32_t) it->second.sum.groups.size();
  return BK_OK;
}
