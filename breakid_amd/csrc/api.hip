// C ABI of libbreakid_hip.so (include/breakid_hip.h): context, record upload, stage drivers, fetch.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <numeric>
#include <thread>
#include <chrono>

#include "bk_common.h"
#include <dirent.h>
#include "prims.h"
#include "stream.h"
#include "join.h"
#include "cluster.h"
#include "bp.h"
#include "ahc.h"
#include "bgzf_gpu.h"

namespace
{
thread_local std::string g_init_error;

uint64_t fnv64(const std::string &s)
{
  uint64_t h = 0xCBF29CE484222325ull;
  for (unsigned char c : s)
  {
    h ^= c;
    h *= 0x100000001B3ull;
  }
  return h;
}
std::string chrom_id_to_name(int tid)  // util_bam.cc:128-142
{
  if (tid == 23) return "chrY";
  if (tid == 22) return "chrX";
  if (tid >= 0 && tid < 22) return "chr" + std::to_string(tid + 1);
  return "";
}

struct StageTimer
{
  std::string name;
  hipEvent_t a = nullptr, b = nullptr;
  uint64_t bytes = 0;    // SURVEY 8(d) algorithmic bytes credited to this stage (every column once for the whole path)
  uint64_t touched = 0;  // bytes this stage's kernels themselves load + store (0 = not modelled)
};
}  // namespace

// the runtime's hardware-queue count as far as this library can know it: GPU_MAX_HW_QUEUES as the environment had it when the
// first context was made (bk_init sets 16 when it is unset - that only helps when no HIP call came before; ROCm's default is 4)
static int g_hw_queues_at_init = 0;
static bool g_hw_queues_set_here = false;

struct bk_ctx
{
  int device = 0;
  hipStream_t st = nullptr;
  bool own_stream = true;
  std::string err;
  int nt = 0;
  std::vector<uint32_t> tlen;
  std::vector<std::string> tname;
  std::vector<int32_t> hdr_id_host;
  DevBuf d_tprefix, d_nhash, d_nid, d_own, d_hdr;
  NameTableDev names{};

  // records
  bk_soa rec{};
  DevBuf col[13];
  DevBuf d_side;  // bk_side rows of an uploaded host table
  bool have_records = false;

  // stream pass
  DevBuf d_counters, d_sd, d_cand, d_split_raw, d_split, d_sa_list;
  StreamCounters hc{};
  SdState hsd{};
  bool stream_done = false, splits_sorted = false;
  int mapq_min = 20;
  int stream_mapq = -1;  // threshold the candidate list in d_cand was filtered with
  uint64_t cand_cap = 0, split_cap = 0, sa_cap = 0;
  // sharded sample: this table is records [rec_base, rec_base + n) of the sample; gathered tables replace the local ones
  uint64_t rec_base = 0;
  std::vector<uint64_t> route_counts;
  JoinBufs jb2;  // grouping of the pairs received from the other ranks (jb still backs the routed send buffer)
  const Cand *ext_cand = nullptr;
  const bk_split *ext_split = nullptr;
  bk_cluster *ext_clusters = nullptr;
  std::vector<uint8_t> own_groups;  // per group (numeric key order): 1 = this rank clusters it; empty = all
  DevBuf d_drop;
  const Cand *cand_ptr() const { return ext_cand ? ext_cand : d_cand.get<Cand>(); }
  const bk_split *split_raw_ptr() const { return ext_split ? ext_split : d_split_raw.get<bk_split>(); }
  bk_cluster *clusters_ptr() const { return ext_clusters ? ext_clusters : d_clusters.get<bk_cluster>(); }
  SdBufs sdb;
  double mean = 0, sd = 0;
  bool stats_done = false;

  // join
  JoinBufs jb;
  JoinResult jr;
  std::vector<uint32_t> gkey_host, glex_host, lex_to_num;
  std::vector<uint64_t> gstart_host;
  DevBuf d_glex;

  // mask + cluster
  SortService svc;  // resident sort service of the stage (sortsvc.inc)
  std::vector<hipEvent_t> svc_probe;
  int svc_late = 0;
  bool svc_refused = false;  // a stage of this context found the service out of reach once (shared hardware queue, crowded device): not tried again
  ClusterBufs cb;
  PairList list;
  DevBuf iso_idx, iso_goff, d_cluster;
  // second lane of chromosome-pair groups (bk_mask_and_cluster): its own buffers, stream and host thread
  struct Lane
  {
    ClusterBufs cb;
    PairList list, iso;
    DevBuf d_cluster;
    hipStream_t st = nullptr;
    ~Lane()
    {
      if (st) (void) hipStreamDestroy(st);
    }
  };
  std::vector<std::unique_ptr<Lane>> lanes;  // lanes 1 .. K-1 (lane 0 uses the context's own stream and buffers)
  PairList listA, isoA;  // first lane's lists before the merge
  PairList lane_mid, lane_iso_m, lane_acc[4];  // kept between calls: a list that is a local is allocated and freed (a device-wide wait) in every call
  DevBuf lane_cacc[4];
  DevBuf d_clusterA, d_dropA;
  uint64_t iso_n = 0;
  bool clustered = false;
  AhcBufs ab;

  // summary + breakpoints
  BpBufs bb;
  DevBuf d_clusters;
  uint64_t n_clusters = 0;

  // fetch staging
  std::vector<bk_pair> f_pairs[3];
  std::vector<uint64_t> f_off[3];
  std::vector<bk_split> f_splits;
  std::vector<bk_cluster> f_clusters;
  std::vector<int32_t> f_gkeys;
  std::vector<bk_group_stat> f_gstats;

  // timing
  bool timing = false;
  std::vector<StageTimer> timers;
  Timing tout;

  void tick(const char *name, uint64_t bytes, bool begin, uint64_t touched = 0)
  {
    if (!timing) return;
    if (begin)
    {
      StageTimer t;
      t.name = name;
      t.bytes = bytes;
      t.touched = touched;
      HIP_CHECK(hipEventCreate(&t.a));
      HIP_CHECK(hipEventCreate(&t.b));
      HIP_CHECK(hipEventRecord(t.a, st));
      timers.push_back(t);
    }
    else
      HIP_CHECK(hipEventRecord(timers.back().b, st));
  }
};

static RecView rec_view(const bk_ctx *ctx)
{
  RecView r;
  r.n = ctx->rec.n;
  r.tid = ctx->rec.tid; r.pos = ctx->rec.pos; r.flag = ctx->rec.flag; r.mapq = ctx->rec.mapq;
  r.cigar_off = ctx->rec.cigar_off; r.cigar = ctx->rec.cigar;
  return r;
}

namespace
{
struct Scope
{
  bk_ctx *c;
  Scope(bk_ctx *c, const char *name, uint64_t bytes = 0, uint64_t touched = 0) : c(c) { c->tick(name, bytes, true, touched); }
  ~Scope()
  {
    try
    {
      c->tick("", 0, false);
    }
    catch (...)
    {
    }
  }
};

template <class F> int guarded(bk_ctx *ctx, F &&f)
{
  if (!ctx) return BK_ERR_ARG;
  try
  {
    HIP_CHECK(hipSetDevice(ctx->device));
    f();
    return BK_OK;
  }
  catch (const bk_error &e)
  {
    ctx->err = e.msg;
    return e.code;
  }
  catch (const std::exception &e)
  {
    ctx->err = e.what();
    return BK_ERR_HIP;
  }
}

void build_name_tables(bk_ctx *c)
{
  const int nt = c->nt;
  std::map<std::string, int> ids;
  for (int i = 0; i < nt; ++i)
    if (!ids.count(c->tname[i])) ids[c->tname[i]] = i;
  if (!ids.count("")) ids[""] = nt;
  if (!ids.count("*")) ids["*"] = nt + 1;
  for (int t = 0; t < 24; ++t)
  {
    std::string s = chrom_id_to_name(t);
    if (!ids.count(s)) ids[s] = nt + 2 + t;
  }
  uint32_t cap = 64;
  while (cap < ids.size() * 4) cap <<= 1;
  std::vector<uint64_t> hash(cap, 0);
  std::vector<int32_t> idv(cap, -1);
  for (auto &kv : ids)
  {
    uint64_t h = fnv64(kv.first);
    if (h == 0) h = 1;
    uint32_t slot = (uint32_t) h & (cap - 1);
    while (hash[slot] != 0 && hash[slot] != h) slot = (slot + 1) & (cap - 1);
    if (hash[slot] == 0)
    {
      hash[slot] = h;
      idv[slot] = kv.second;
    }
  }
  std::vector<int32_t> own(std::max(nt, 1));
  for (int t = 0; t < nt; ++t) own[t] = ids[chrom_id_to_name(t)];
  c->hdr_id_host.assign(nt + 1, 0);
  c->hdr_id_host[0] = ids["*"];
  for (int t = 0; t < nt; ++t) c->hdr_id_host[t + 1] = ids[c->tname[t]];
  std::vector<uint32_t> prefix(nt + 1, 0);
  for (int t = 0; t < nt; ++t) prefix[t + 1] = prefix[t] + c->tlen[t];
  HIP_CHECK(hipMemcpy(c->d_nhash.as<uint64_t>(cap), hash.data(), cap * 8, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(c->d_nid.as<int32_t>(cap), idv.data(), cap * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(c->d_own.as<int32_t>(own.size()), own.data(), own.size() * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(c->d_hdr.as<int32_t>(nt + 1), c->hdr_id_host.data(), (nt + 1) * 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(c->d_tprefix.as<uint32_t>(nt + 1), prefix.data(), (nt + 1) * 4, hipMemcpyHostToDevice));
  c->names.hash = c->d_nhash.get<uint64_t>();
  c->names.id = c->d_nid.get<int32_t>();
  c->names.mask = cap - 1;
  c->names.own_id = c->d_own.get<int32_t>();
  c->names.n_targets = nt;
  c->names.empty_id = ids[""];
}

std::string rname(const bk_ctx *c, int tid) { return tid < 0 ? "*" : c->tname[tid]; }

// ---- stream pass (A1 sums, A2 filter, A12 gate) ------------------------------------------------------------------
// Three steps so that the table may arrive in pieces (bk_bam_decode_device_ctx: the pass runs on the records of a feed chunk
// while the next chunks are still being copied and inflated): prepare (outputs sized, counters cleared), any number of
// launches over consecutive record ranges (the counters accumulate, candidates / SA-bearing record indices append), finish
// (counters read back, rare-path kernel over the SA-bearing records).
StreamArgs stream_args(bk_ctx *c, uint64_t n)
{
  StreamArgs a{};
  a.n = n;
  a.q_begin = 0;
  a.rec_base = c->rec_base;
  a.tid = c->rec.tid; a.pos = c->rec.pos; a.mtid = c->rec.mtid; a.mpos = c->rec.mpos; a.isize = c->rec.isize;
  a.flag = c->rec.flag; a.mapq = c->rec.mapq; a.qhash = c->rec.qhash; a.qcheck = c->rec.qcheck;
  static const bool no_side = getenv("BREAKID_NO_SIDE") != nullptr;  // (comparison: the four columns even when the table has the rows)
  a.side = no_side && c->rec.qhash ? nullptr : c->rec.side;
  a.cigar_off = c->rec.cigar_off; a.cigar = c->rec.cigar; a.aux_off = c->rec.aux_off; a.aux = c->rec.aux;
  a.mapq_min = c->mapq_min;
  a.names = c->names;
  a.counters = c->d_counters.get<StreamCounters>();
  a.sd = c->d_sd.get<SdState>();
  a.cand = c->d_cand.get<Cand>();
  a.cand_cap = c->cand_cap;
  a.split = c->d_split_raw.get<bk_split>();
  a.split_cap = c->split_cap;
  a.sa_list = c->d_sa_list.get<uint32_t>();
  a.sa_cap = c->sa_cap;
  return a;
}
void stream_prepare(bk_ctx *c, uint64_t n_expected)
{
  c->cand_cap = std::max<uint64_t>(c->cand_cap, std::max<uint64_t>(1u << 16, n_expected / 8 + 1024));
  c->split_cap = std::max<uint64_t>(c->split_cap, std::max<uint64_t>(1u << 14, n_expected / 32 + 1024));
  c->sa_cap = std::max<uint64_t>(c->sa_cap, std::max<uint64_t>(1u << 14, n_expected / 16 + 1024));
  StreamCounters *dc = c->d_counters.as<StreamCounters>(1);
  SdState *dsd = c->d_sd.as<SdState>(1);
  HIP_CHECK(hipMemsetAsync(dc, 0, sizeof(StreamCounters), c->st));
  HIP_CHECK(hipMemsetAsync(dsd, 0, sizeof(SdState), c->st));
  (void) c->d_cand.as<Cand>(c->cand_cap);
  (void) c->d_split_raw.as<bk_split>(c->split_cap);
  (void) c->d_sa_list.as<uint32_t>(c->sa_cap);
}
// returns false when an output capacity was exceeded (the caller enlarges and repeats the pass)
bool stream_finish(bk_ctx *c)
{
  HIP_CHECK(hipMemcpyAsync(&c->hc, c->d_counters.get<StreamCounters>(), sizeof(StreamCounters), hipMemcpyDeviceToHost, c->st));
  HIP_CHECK(hipMemcpyAsync(&c->hsd, c->d_sd.get<SdState>(), sizeof(SdState), hipMemcpyDeviceToHost, c->st));
  HIP_CHECK(hipStreamSynchronize(c->st));
  if (c->hc.n_cand > c->cand_cap || c->hc.n_sa > c->sa_cap)
  {
    c->cand_cap = std::max<uint64_t>(c->cand_cap, c->hc.n_cand + 1024);
    c->sa_cap = std::max<uint64_t>(c->sa_cap, c->hc.n_sa + 1024);
    return false;
  }
  return true;
}
void stream_rare_path(bk_ctx *c)
{
  const uint64_t n = c->rec.n;
  if (c->timing && !c->timers.empty())
  {
    c->timers.back().bytes += 32ull * c->hc.n_cand + 4ull * c->hc.n_sa;
    // what k_stream itself moves: tid, pos, isize, flag, mapq, cigar_off, aux_off of every record (23 B), the CIGAR words (span bound),
    // qhash + mtid + mpos (+ qcheck) only of the candidates (16-20 B read; with the bk_side rows the kernel asks for the whole 32-byte row -
    // the model keeps the 16-20 bytes it needs, i.e. it does not credit the row's padding) + the 40-byte candidate written, 4 B per SA-bearing record index
    c->timers.back().touched = 23ull * n + 4ull * c->rec.n_cigar_words + (c->rec.qcheck ? 60ull : 56ull) * c->hc.n_cand + 4ull * c->hc.n_sa;
  }
  if (c->hc.unsorted) throw bk_error(BK_ERR_UNSORTED, "records are not coordinate sorted (the reference requires an indexed, sorted BAM)");
  // rare path: evidence tuples of the SA-bearing records (capacity = one tuple per listed record)
  if (c->hc.n_sa > c->split_cap)
  {
    c->split_cap = c->hc.n_sa + 1024;
    (void) c->d_split_raw.as<bk_split>(c->split_cap);
  }
  StreamArgs a = stream_args(c, n);
  {
    Scope s(c, "k_split_records", c->rec.n_aux_bytes + 4ull * c->hc.n_sa);
    launch_split_records(a, c->hc.n_sa, c->st);
  }
  HIP_CHECK(hipMemcpyAsync(&c->hc, c->d_counters.get<StreamCounters>(), sizeof(StreamCounters), hipMemcpyDeviceToHost, c->st));
  HIP_CHECK(hipStreamSynchronize(c->st));
  if (c->timing && !c->timers.empty())
  {
    c->timers.back().bytes += 48ull * c->hc.n_split;
    c->timers.back().touched = c->timers.back().bytes + 40ull * c->hc.n_sa + 32ull * c->hc.n_split;  // + the fixed columns of the listed records, 80-byte tuples
  }
  c->stream_done = true;
  c->stream_mapq = c->mapq_min;
  c->splits_sorted = false;
  c->stats_done = false;
}

void run_stream(bk_ctx *c)
{
  if (!c->have_records) throw bk_error(BK_ERR_ARG, "no records uploaded");
  const uint64_t n = c->rec.n;
  for (int attempt = 0; attempt < 3; ++attempt)
  {
    stream_prepare(c, n);
    {
      // algorithmic bytes of this pass: 39 B/record + 4 B per CIGAR op (+ 32 B per candidate, added below)
      Scope s(c, "k_stream", 39ull * n + 4ull * c->rec.n_cigar_words);
      launch_stream(stream_args(c, n), c->st);
    }
    if (stream_finish(c)) break;
    if (c->timing && !c->timers.empty()) c->timers.pop_back();  // overflowed attempt is not a measured pass
    if (attempt == 2) throw bk_error(BK_ERR_LIMIT, "stream pass: output capacity");
  }
  stream_rare_path(c);
}

void ensure_splits_sorted(bk_ctx *c)
{
  if (c->splits_sorted) return;
  Scope s(c, "split_sort");
  bk_split *sorted = c->d_split.as<bk_split>(c->hc.n_split + 1);
  // tuples of this table carry its own record indices; a sharded sample's carry rec_base + i of every rank: the passes of the sort
  // follow the largest index among the tuples then (bits = 0: sort_splits looks)
  int bits = 0;
  if (!c->ext_split && c->rec_base == 0)
  {
    bits = 1;
    while (bits < 32 && (1ull << bits) < c->rec.n) ++bits;
  }
  sort_splits(const_cast<bk_split *>(c->split_raw_ptr()), c->hc.n_split, sorted, c->bb, c->st, bits);
  c->splits_sorted = true;
}
}  // namespace

extern "C" {

// once per process, before its first HIP call if the caller allows: bk_multi_run* calls this before it starts its rank threads
// (setenv must not run beside threads that read the environment)
void bk_prepare_process()
{
  static std::once_flag once;
  std::call_once(once, [] {
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    g_hw_queues_at_init = q ? atoi(q) : 4;
    if (!q)
    {
      setenv("GPU_MAX_HW_QUEUES", "20", 0);
      g_hw_queues_set_here = true;
    }
  });
}

int bk_init(int device, const uint32_t *target_len, const char *const *target_name, int n_targets, bk_ctx **out)
{
  {
    // the lanes of bk_mask_and_cluster and the chunk streams of the GPU feed want more hardware queues than ROCm's default of 4;
    // the runtime reads the variable when it starts, so this helps a caller whose first HIP call is this one (a caller that has
    // initialised HIP already keeps what it had: lanes_apply tells it once on stderr)
    bk_prepare_process();
  }
  if (!out || n_targets < 0 || (n_targets && (!target_len || !target_name)))
  {
    g_init_error = "bk_init: bad arguments";
    return BK_ERR_ARG;
  }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
  {
    g_init_error = "bk_init: no HIP device " + std::to_string(device) + " (this library has no CPU path)";
    return BK_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
  {
    g_init_error = std::string("bk_init: device is not gfx950 (") + prop.gcnArchName + "); kernels are built for MI355X only";
    return BK_ERR_NO_DEVICE;
  }
  bk_ctx *c = new bk_ctx();
  c->device = device;
  c->nt = n_targets;
  for (int i = 0; i < n_targets; ++i)
  {
    c->tlen.push_back(target_len[i]);
    c->tname.push_back(target_name[i]);
  }
  int rc = guarded(c, [&] {
    HIP_CHECK(hipStreamCreate(&c->st));
    build_name_tables(c);
  });
  if (rc != BK_OK)
  {
    g_init_error = c->err;
    delete c;
    return rc;
  }
  *out = c;
  return BK_OK;
}

void bk_free(bk_ctx *ctx)
{
  if (!ctx) return;
  (void) hipSetDevice(ctx->device);
  (void) hipStreamSynchronize(ctx->st);
  for (auto &t : ctx->timers)
  {
    if (t.a) (void) hipEventDestroy(t.a);
    if (t.b) (void) hipEventDestroy(t.b);
  }
  if (ctx->own_stream && ctx->st) (void) hipStreamDestroy(ctx->st);
  delete ctx;
}

const char *bk_last_error(const bk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int bk_set_stream(bk_ctx *ctx, void *hip_stream)
{
  return guarded(ctx, [&] {
    if (ctx->own_stream && ctx->st) HIP_CHECK(hipStreamDestroy(ctx->st));
    ctx->st = (hipStream_t) hip_stream;
    ctx->own_stream = false;
  });
}
int bk_get_stream(bk_ctx *ctx, void **hip_stream)
{
  return guarded(ctx, [&] {
    if (!hip_stream) throw bk_error(BK_ERR_ARG, "bk_get_stream: null output");
    *hip_stream = (void *) ctx->st;
  });
}
int bk_sync(bk_ctx *ctx)
{
  return guarded(ctx, [&] { HIP_CHECK(hipStreamSynchronize(ctx->st)); });
}

int bk_upload_records(bk_ctx *ctx, const bk_soa *s, int mem_space)
{
  return guarded(ctx, [&] {
    if (!s) throw bk_error(BK_ERR_ARG, "bk_upload_records: null table");
    if (s->n > 0xFFFFFFF0ull) throw bk_error(BK_ERR_LIMIT, "more than 2^32 records in one context (shard across GPUs)");
    ctx->stream_done = ctx->stats_done = ctx->clustered = false;
    ctx->ext_cand = nullptr;
    ctx->ext_split = nullptr;
    ctx->ext_clusters = nullptr;
    ctx->own_groups.clear();
    ctx->rec_base = 0;
    if (mem_space == BK_MEM_DEVICE)
    {
      // the columns are used in place: they must live on this context's GPU (a table decoded on another device would be
      // read across xGMI, or fault without peer access)
      hipPointerAttribute_t at;
      if (s->n && s->tid && hipPointerGetAttributes(&at, s->tid) == hipSuccess && at.type == hipMemoryTypeDevice && at.device != ctx->device)
        throw bk_error(BK_ERR_ARG, "bk_upload_records: BK_MEM_DEVICE table lives on device " + std::to_string(at.device) + ", context on device " +
                                       std::to_string(ctx->device));
      (void) hipGetLastError();
      // the streaming kernels read the fixed columns as 16-byte vectors (four or eight records per lane)
      for (const void *p : {(const void *) s->tid, (const void *) s->pos, (const void *) s->isize, (const void *) s->flag, (const void *) s->mapq, (const void *) s->cigar_off,
                            (const void *) s->aux_off})
        if (s->n && ((uintptr_t) p & 15u)) throw bk_error(BK_ERR_ARG, "bk_upload_records: BK_MEM_DEVICE columns must be 16-byte aligned");
      if (s->n && s->side && ((uintptr_t) s->side & 15u)) throw bk_error(BK_ERR_ARG, "bk_upload_records: BK_MEM_DEVICE side rows must be 16-byte aligned");
      ctx->rec = *s;
    }
    else
    {
      const uint64_t n = s->n;
      const void *src[13] = {s->tid, s->pos, s->mtid, s->mpos, s->isize, s->flag, s->mapq, s->qhash, s->cigar_off, s->cigar, s->aux_off, s->aux, s->qcheck};
      const size_t bytes[13] = {n * 4, n * 4, n * 4, n * 4, n * 4, n * 2, n, n * 8, (n + 1) * 4, s->n_cigar_words * 4, (n + 1) * 4, s->n_aux_bytes,
                                s->qcheck ? n * 4 : 0};
      void *dst[13];
      for (int k = 0; k < 13; ++k)
      {
        dst[k] = ctx->col[k].ensure(bytes[k] + 16);
        if (bytes[k]) HIP_CHECK(hipMemcpyAsync(dst[k], src[k], bytes[k], hipMemcpyHostToDevice, ctx->st));
      }
      HIP_CHECK(hipStreamSynchronize(ctx->st));
      bk_soa d = *s;
      d.tid = (const int32_t *) dst[0]; d.pos = (const int32_t *) dst[1]; d.mtid = (const int32_t *) dst[2]; d.mpos = (const int32_t *) dst[3];
      d.isize = (const int32_t *) dst[4]; d.flag = (const uint16_t *) dst[5]; d.mapq = (const uint8_t *) dst[6]; d.qhash = (const uint64_t *) dst[7];
      d.cigar_off = (const uint32_t *) dst[8]; d.cigar = (const uint32_t *) dst[9]; d.aux_off = (const uint32_t *) dst[10]; d.aux = (const uint8_t *) dst[11];
      d.qcheck = s->qcheck ? (const uint32_t *) dst[12] : nullptr;
      // the side layout for the streaming pass (one 32-byte row per record, include/breakid_hip.h: bk_side), made on the device
      bk_side *side = ctx->d_side.as<bk_side>(n + 1);
      launch_make_side(d.qhash, d.mtid, d.mpos, d.qcheck, n, side, ctx->st);
      d.side = side;
      ctx->rec = d;
    }
    ctx->have_records = true;
  });
}

int bk_isize_stats(bk_ctx *ctx, double *mean, double *sd)
{
  return guarded(ctx, [&] {
    if (!ctx->stream_done) run_stream(ctx);
    if (!ctx->stats_done)
    {
      const double n = (double) ctx->hc.isize_n;
      const double m = (double) (long long) ctx->hc.isize_sum / n;  // (double) long / (double) size_t, BreakID.cc:1941
      ctx->mean = m;
      if (ctx->hc.isize_n == 0)
        ctx->sd = std::nan("");
      else
      {
        double sum_d = ctx->hsd.sumsq - 2.0 * m * (double) ctx->hc.isize_sum + n * m * m;
        if (!(sum_d > 0)) sum_d = 0;
        double da = (double) ctx->hsd.vmax - m, dmax = da * da + m * m;
        double bound = 2.0 * sum_d + 2.0 * n + 2.0 * dmax + 4.0;
        int k = std::ilogb(bound) + 1;
        double thr = k >= 51 ? 1.0e300 : std::ldexp(1.0, k - 53);
        {
          Scope s(ctx, "isize_sd", 0, 6ull * ctx->rec.n);  // flag + isize re-read: already credited to the path once (k_stream)
          launch_sd(ctx->rec.flag, ctx->rec.isize, ctx->rec.n, m, thr, ctx->d_sd.get<SdState>(), ctx->sdb, ctx->st);
        }
        HIP_CHECK(hipMemcpyAsync(&ctx->hsd, ctx->d_sd.get<SdState>(), sizeof(SdState), hipMemcpyDeviceToHost, ctx->st));
        HIP_CHECK(hipStreamSynchronize(ctx->st));
        ctx->sd = std::sqrt((double) ctx->hsd.t_final / n);  // sqrt(long / (double) size), :1946
      }
      ctx->stats_done = true;
    }
    if (mean) *mean = ctx->mean;
    if (sd) *sd = ctx->sd;
  });
}

// host-side group tables of ctx->jr and the reference's group order; `all_keys` (sharded sample: the chr-pair keys of
// every rank) makes the cluster/pair `group` ordinals global
static void finish_groups(bk_ctx *ctx, const std::vector<uint32_t> *all_keys)
{
  const uint32_t ng = ctx->jr.n_groups;
  ctx->gkey_host.assign(ng, 0);
  ctx->gstart_host.assign(ng + 1, 0);
  if (ng)
  {
    HIP_CHECK(hipMemcpyAsync(ctx->gkey_host.data(), ctx->jr.gkey, ng * 4, hipMemcpyDeviceToHost, ctx->st));
    HIP_CHECK(hipMemcpyAsync(ctx->gstart_host.data(), ctx->jr.gstart, (ng + 1) * 8, hipMemcpyDeviceToHost, ctx->st));
    HIP_CHECK(hipStreamSynchronize(ctx->st));
  }
  // the reference iterates groups in std::map<string> order of "chrA_chrB" (BreakID.cc:93,119)
  auto key_name = [&](uint32_t k) {
    int t1 = (int) (k / (uint32_t) (ctx->nt + 1)) - 1, t2 = (int) (k % (uint32_t) (ctx->nt + 1)) - 1;
    return rname(ctx, t1) + "_" + rname(ctx, t2);
  };
  std::vector<std::string> keys(ng);
  for (uint32_t g = 0; g < ng; ++g) keys[g] = key_name(ctx->gkey_host[g]);
  ctx->lex_to_num.resize(ng);
  std::iota(ctx->lex_to_num.begin(), ctx->lex_to_num.end(), 0u);
  std::sort(ctx->lex_to_num.begin(), ctx->lex_to_num.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
  ctx->glex_host.assign(ng, 0);
  if (!all_keys)
    for (uint32_t l = 0; l < ng; ++l) ctx->glex_host[ctx->lex_to_num[l]] = l;
  else
  {
    std::vector<std::string> names;
    names.reserve(all_keys->size());
    for (uint32_t k : *all_keys) names.push_back(key_name(k));
    std::sort(names.begin(), names.end());
    names.erase(std::unique(names.begin(), names.end()), names.end());
    for (uint32_t g = 0; g < ng; ++g)
    {
      auto it = std::lower_bound(names.begin(), names.end(), keys[g]);
      if (it == names.end() || *it != keys[g]) throw bk_error(BK_ERR_ARG, "bk_shard_group_pairs: a local group is missing from the global key list");
      ctx->glex_host[g] = (uint32_t) (it - names.begin());
    }
  }
  uint32_t *dg = ctx->d_glex.as<uint32_t>((uint64_t) ng + 1);
  if (ng) HIP_CHECK(hipMemcpyAsync(dg, ctx->glex_host.data(), ng * 4, hipMemcpyHostToDevice, ctx->st));
  join_assign_ids(ctx->jr, dg, ctx->st);
}

int bk_discordant_pairs(bk_ctx *ctx, int mapq_min, double w, uint64_t *n_pairs, uint32_t *n_groups)
{
  return guarded(ctx, [&] {
    if (!ctx->stream_done || mapq_min != ctx->stream_mapq)
    {
      ctx->mapq_min = mapq_min;
      run_stream(ctx);
    }
    {
      Scope s(ctx, "mate_join");
      // discovery indices are record indices: of this table when the candidates are its own, of the whole sample (up to 64 bits:
      // rec_base + i) when they came from other shards
      ctx->jb.rec_bits = -1;
      if (!ctx->ext_cand && ctx->rec_base == 0)
      {
        int bits = 1;
        while (bits < 32 && (1ull << bits) < ctx->rec.n) ++bits;
        ctx->jb.rec_bits = bits;
      }
      join_candidates(ctx->cand_ptr(), ctx->hc.n_cand, w, ctx->d_tprefix.get<uint32_t>(), ctx->nt, ctx->jb, ctx->st, ctx->jr);
    }
    finish_groups(ctx, nullptr);
    ctx->clustered = false;
    if (n_pairs) *n_pairs = ctx->jr.n_pairs;
    if (n_groups) *n_groups = ctx->jr.n_groups;
  });
}

// Two lanes.  The reference clusters its chromosome-pair groups one after the other and independently of each other
// (BreakID.cc:119-167).  All groups in one pass pay, in each of the five sorts, the longest heapsort segment of ANY group; two
// disjoint sets of groups on two streams, each driven by its own host thread, overlap the lone-wave heaps of one set with the
// bandwidth- and launch-bound partition levels of the other.  Needs more hardware queues than ROCm's default of 4 (the heap
// kernels of both lanes sit on side streams that must not share a queue): enabled by BREAKID_GROUP_LANES=2 (bench.py and the
// command line set it together with GPU_MAX_HW_QUEUES before the runtime starts).  Results are identical by construction:
// the groups never interact, the lanes' lists are merged back into group order.
static void run_lane(const bk_pair *pairs, const uint32_t *gof, const uint64_t *gstart, uint32_t ng, uint64_t n, double w, int fast, const uint32_t *drop, PairList &L, PairList &iso,
                     DevBuf &d_cluster, ClusterBufs &cb, AhcBufs &ab, hipStream_t st, const uint64_t *gstart_host, const uint8_t *keep_host)
{
  remove_isolated_begin(pairs, gof, gstart, ng, n, w, L, cb, st, drop, gstart_host, keep_host);
  remove_isolated_end(pairs, L, cb, st);
  iso.n = L.n;
  iso.ng = L.ng;
  uint32_t *ii = iso.idx.as<uint32_t>(L.n + 1), *ig = iso.gof.as<uint32_t>(L.n + 1);
  uint64_t *io = iso.goff.as<uint64_t>((uint64_t) L.ng + 1);
  if (L.n) HIP_CHECK(hipMemcpyAsync(ii, L.idx.get<uint32_t>(), L.n * 4, hipMemcpyDeviceToDevice, st));
  if (L.n) HIP_CHECK(hipMemcpyAsync(ig, L.gof.get<uint32_t>(), L.n * 4, hipMemcpyDeviceToDevice, st));
  HIP_CHECK(hipMemcpyAsync(io, L.goff.get<uint64_t>(), ((uint64_t) L.ng + 1) * 8, hipMemcpyDeviceToDevice, st));
  if (fast)
    fast_cluster_all(pairs, L, w, d_cluster, cb, st);
  else
    ahc_cluster_all(pairs, L, w, d_cluster, ab, cb, st);
}

static bool sort_service_on()
{
  // BREAKID_SORT_SERVICE=0: every std::sort replay as its own chain of launches (the earlier form)
  static const bool on = !(getenv("BREAKID_SORT_SERVICE") && atoi(getenv("BREAKID_SORT_SERVICE")) == 0);
  return on;
}
// The resident sort service runs while one of these lives: every sort through the context's (and its lanes') buffers is a job.
// Its two persistent kernels occupy a hardware queue each until the stage ends, so a stream of this stage that shares one of those
// queues (more streams in the process than the runtime has hardware queues: GPU_MAX_HW_QUEUES) would never get its turn.  That is
// why the stage (a) is the only one on its device (contexts of one process on one GPU - `-comm local` - take turns: the others sort
// by launches), and (b) probes every stream it is going to use after the kernels have started: an empty kernel that has not run
// after 50 ms sends the whole stage back to the launch path (the service stops, the stream drains).
__global__ void k_svc_probe() {}
// Compute queues that exist on the device right now, over ALL processes (the kernel driver's sysfs: /sys/class/kfd/kfd/proc/<pid>/
// queues/<n>/{gpuid,type}); -1 when that cannot be told.  Measured on MI355X: beyond 24 compute queues on a device - this process's
// 16-17 plus a second process holding 8 or more - the driver maps the queues in turns, and persistent kernels whose submitters wait
// for their turn leave jobs unfinished for seconds (tools/gpu_hold_queues.py: 4 streams held by a second process are fine, 8 are
// not).  The service runs only while the census stays at or below that.
static int kfd_compute_queues(int device)
{
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int) sizeof bus, device) != hipSuccess) return -1;
  unsigned dom = 0, b = 0, d = 0, f = 0;
  if (sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) != 4) return -1;
  const unsigned long want_loc = (b << 8) | (d << 3) | f;
  auto read_file = [](const std::string &path, std::string &out) {
    FILE *fp = fopen(path.c_str(), "r");
    if (!fp) return false;
    char buf[4096];
    const size_t n = fread(buf, 1, sizeof buf - 1, fp);
    fclose(fp);
    buf[n] = 0;
    out = buf;
    return true;
  };
  auto list_dir = [](const std::string &path, std::vector<std::string> &names) {
    DIR *dp = opendir(path.c_str());
    if (!dp) return false;
    while (dirent *e = readdir(dp))
      if (e->d_name[0] != '.') names.push_back(e->d_name);
    closedir(dp);
    return true;
  };
  // the device's gpu_id: the topology node with its PCI location
  unsigned long gpu_id = 0;
  {
    std::vector<std::string> nodes;
    if (!list_dir("/sys/class/kfd/kfd/topology/nodes", nodes)) return -1;
    for (const std::string &n : nodes)
    {
      std::string props, id;
      const std::string base = "/sys/class/kfd/kfd/topology/nodes/" + n;
      if (!read_file(base + "/properties", props) || !read_file(base + "/gpu_id", id)) continue;
      unsigned long loc = ~0ul, domain = ~0ul;
      size_t p = props.find("location_id ");
      if (p != std::string::npos) loc = strtoul(props.c_str() + p + 12, nullptr, 10);
      p = props.find("domain ");
      if (p != std::string::npos) domain = strtoul(props.c_str() + p + 7, nullptr, 10);
      if (loc == want_loc && (domain == ~0ul || domain == dom) && strtoul(id.c_str(), nullptr, 10) != 0) gpu_id = strtoul(id.c_str(), nullptr, 10);
    }
  }
  if (!gpu_id) return -1;
  std::vector<std::string> procs;
  if (!list_dir("/sys/class/kfd/kfd/proc", procs)) return -1;
  int total = 0;
  for (const std::string &pid : procs)
  {
    std::vector<std::string> qs;
    const std::string qdir = "/sys/class/kfd/kfd/proc/" + pid + "/queues";
    if (!list_dir(qdir, qs)) continue;  // (another user's process: not readable - and not on a GPU this process may use either)
    for (const std::string &q : qs)
    {
      std::string g, t;
      if (!read_file(qdir + "/" + q + "/gpuid", g) || !read_file(qdir + "/" + q + "/type", t)) continue;
      if (strtoul(g.c_str(), nullptr, 10) == gpu_id && strtoul(t.c_str(), nullptr, 10) == 0) ++total;
    }
  }
  return total;
}
constexpr int SVC_MAX_DEVICE_QUEUES = 24;
static std::mutex g_svc_device_m[64];
struct SvcStage
{
  bk_ctx *ctx;
  bool on;
  std::unique_lock<std::mutex> device_turn;
  SvcStage(bk_ctx *c, bool want, uint64_t n_bound, uint64_t max_group, const std::vector<hipStream_t> &streams) : ctx(c), on(want && sort_service_on() && !c->svc_refused)
  {
    if (on && ctx->device >= 0 && ctx->device < 64)
    {
      device_turn = std::unique_lock<std::mutex>(g_svc_device_m[ctx->device], std::try_to_lock);
      on = device_turn.owns_lock();
    }
    if (!on) return;
    const auto ts0 = std::chrono::steady_clock::now();
    ctx->svc.start(n_bound, max_group + 2, ctx->st);  // (+2: a mask may emit one element twice)
    const auto ts1 = std::chrono::steady_clock::now();
    // (the count is kept for a quarter of a second per device: reading it is ~0.5 ms of sysfs, and a sample's stages - or a bench's
    // steps - follow each other faster than processes come and go)
    int census;
    {
      static std::mutex cm;
      static std::chrono::steady_clock::time_point when[64];
      static int last[64];
      static bool have[64] = {};
      std::lock_guard<std::mutex> l(cm);
      const int d = ctx->device & 63;
      if (!have[d] || std::chrono::duration<double>(ts1 - when[d]).count() > 0.25)
      {
        last[d] = kfd_compute_queues(ctx->device);
        when[d] = std::chrono::steady_clock::now();
        have[d] = true;
      }
      census = last[d];
    }
    if (bk_debug("lanes"))
      fprintf(stderr, "[lanes] sort service started in %.3f ms; compute queues on the device (all processes): %d (counted in %.3f ms)\n", std::chrono::duration<double, std::milli>(ts1 - ts0).count(), census,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts1).count());
    const bool crowded = census > SVC_MAX_DEVICE_QUEUES && !bk_debug("nocensus");
    const bool behind = !crowded && !reachable(streams);
    const bool late = !crowded && !behind && !ctx->svc.narrow_running(0.03);
    if (crowded || behind || late)
    {
      ctx->svc.stop();
      on = false;
      // a stream behind a persistent kernel's queue or a crowded device stay that way: this context does not try again; a narrow
      // kernel that was merely late (a busy device) gets a second chance
      if (crowded || behind || ++ctx->svc_late >= 2) ctx->svc_refused = true;
      device_turn.unlock();
      static bool told = false;
      if (!told && !getenv("BREAKID_QUIET"))
      {
        told = true;
        fprintf(stderr, "[breakid] the resident sort service shares a hardware queue with a stream of its own stage (GPU_MAX_HW_QUEUES too low for the streams of this process), or other processes hold hardware queues on this device: sorting by launches instead\n");
      }
      return;
    }
    ctx->svc_late = 0;
    set(&ctx->svc);
  }
  bool reachable(const std::vector<hipStream_t> &streams)
  {
    while (ctx->svc_probe.size() < streams.size())
    {
      hipEvent_t e;
      HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ctx->svc_probe.push_back(e);
    }
    for (size_t k = 0; k < streams.size(); ++k)
    {
      hipLaunchKernelGGL(k_svc_probe, dim3(1), dim3(64), 0, streams[k]);
      HIP_CHECK(hipEventRecord(ctx->svc_probe[k], streams[k]));
    }
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    bool ok = true;
    for (size_t k = 0; k < streams.size() && ok; ++k)
      for (;;)
      {
        const hipError_t e = hipEventQuery(ctx->svc_probe[k]);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) throw bk_error(BK_ERR_HIP, std::string("sort service probe: ") + hipGetErrorString(e));
        if (since() > 0.05)
        {
          ok = false;
          break;
        }
      }
    // An empty kernel per stream comes back within ~0.1-0.3 ms.  Milliseconds mean that the device's hardware queues are
    // oversubscribed (other processes hold queues too: the scheduler then maps the queues in turns, and persistent kernels whose
    // submitters wait for their turn crawl - measured with a second process holding 16 queues: the stage 3-6 times slower or a job
    // that never finished) - no service then either.
    const double took = since();
    if (bk_debug("lanes")) fprintf(stderr, "[lanes] sort service probe: %zu streams in %.3f ms%s\n", streams.size(), took * 1e3, ok ? "" : " (gave up)");
    return ok && took < 0.003;
  }
  void set(SortService *s)
  {
    ctx->cb.se.svc = s;
    ctx->cb.se.svc_slot = 0xFFFFFFFFu;
    for (auto &l : ctx->lanes)
    {
      l->cb.se.svc = s;
      l->cb.se.svc_slot = 0xFFFFFFFFu;
    }
  }
  void finish()
  {
    if (!on) return;
    on = false;
    set(nullptr);
    // every job has been waited for by its caller: the streams must be through before the workgroups are told to leave
    hipError_t e = hipStreamSynchronize(ctx->st);
    for (auto &l : ctx->lanes)
      if (l->st && e == hipSuccess) e = hipStreamSynchronize(l->st);
    ctx->svc.stop();
    device_turn.unlock();
    if (e != hipSuccess) throw bk_error(BK_ERR_HIP, std::string("sort service: ") + hipGetErrorString(e));
  }
  ~SvcStage()
  {
    try
    {
      finish();
    }
    catch (const bk_error &)
    {
    }
  }
};
static int lanes_wanted(bool svc)
{
  // with the resident sort service a lane's sort is a submit and a wait of its thread, so there can be a lane for every one or two
  // of the groups that carry long heap segments: twelve by default (each with a stream of its own; measured 12 / 16 / 18 / 24 lanes
  // on 12 streams: 30.6 / 33.4 / 34.3 / 35.7 ms for the stage - every lane costs its ~160 other launches)
  static const int want_svc = getenv("BREAKID_GROUP_LANES") ? atoi(getenv("BREAKID_GROUP_LANES")) : 12;
  if (svc) return want_svc < 1 ? 1 : (want_svc > 26 ? 26 : want_svc);
  // four lanes unless the caller says otherwise (BREAKID_GROUP_LANES=1: one pass); lanes_apply decides from the data whether they
  // pay.  Measured on the 30x WGS shape with the segment-per-workgroup tail of the level loop: 2 lanes 42.0 ms, 3 lanes 42.6,
  // 4 lanes 39.6, 5 lanes 48.5 (more lanes shorten a lane's "longest heap of any of its groups" per sort, and cost a level loop,
  // a ranking and a finisher chain of their own, each ~100 launches that wait for each other across lanes)
  static const int want = getenv("BREAKID_GROUP_LANES") ? atoi(getenv("BREAKID_GROUP_LANES")) : 4;
  return want < 1 ? 1 : (want > 26 ? 26 : want);
}
static bool lanes_apply(const bk_ctx *ctx, int fast)
{
  static const uint64_t min_pairs = getenv("BREAKID_LANES_MIN_PAIRS") ? strtoull(getenv("BREAKID_LANES_MIN_PAIRS"), nullptr, 10) : (1ull << 20);  // below: launch-bound anyway
  // picked from the data: lanes pay when at least two groups are large enough to run into long sorts side by side
  uint32_t large = 0;
  const uint64_t big = std::max<uint64_t>(2, min_pairs >> 6);  // 16 K pairs with the default threshold
  for (uint32_t g = 0; g < ctx->jr.n_groups && g + 1 < ctx->gstart_host.size(); ++g) large += ctx->gstart_host[g + 1] - ctx->gstart_host[g] >= big ? 1u : 0u;
  const bool yes = lanes_wanted(sort_service_on() && fast) >= 2 && fast && ctx->jr.n_groups >= 4 && ctx->jr.n_pairs >= min_pairs && large >= 2;
  if (yes)
  {
    static bool told = false;
    if (!told && g_hw_queues_at_init < 8 && !g_hw_queues_set_here && !getenv("BREAKID_QUIET"))
    {
      told = true;
      fprintf(stderr, "[breakid] GPU_MAX_HW_QUEUES=%d: the lanes of chromosome-pair groups (bk_mask_and_cluster) and their heap kernels will share hardware queues; "
                        "set GPU_MAX_HW_QUEUES=16 in the environment before the process makes its first HIP call (or BREAKID_GROUP_LANES=1)\n", g_hw_queues_at_init);
    }
  }
  return yes;
}

// K lanes of groups.  A lane's time is (a) per sort the LONGEST heapsort segment of any of its groups - a serial chain of one wave
// - plus (b) partition levels and masks in proportion to its pairs plus (c) a fixed number of launch-bound late levels.  Which
// groups own long heap segments cannot be told from their sizes (all same-chromosome groups of a WGS sample are about equally
// large; two or three of them carry segments of 30-46 K elements, most carry a few thousand), but it can be OBSERVED: a group
// whose sort by x (by y) ran into the depth limit does so again in the next sort by the same coordinate.  So the stage runs in
// two parts:
//   part 1  sorts 1-3 (x, mask, y, mask, x: remove_isolated_pairs) in K lanes split blindly (longest-processing-time on size^e);
//           every lane records the longest heap segment of each of its groups in the sort by y and in the LAST sort by x;
//   part 2  sorts 4-5 (x-windows, y, y-windows, x) in K lanes split on what was observed: the groups are placed, heaviest
//           chain first, where the lane's longest segment by y + longest segment by x (+ a term for its pair count) grows least.
// BREAKID_LANE_ADAPT=0 keeps the blind split for the whole stage.  On the 30x WGS shape part 2 comes out balanced (18.9 / 21.4 ms
// with two lanes) where the blind split leaves one lane 9 ms behind the other (46.6 / 37 ms); part 1 stays as the blind split
// leaves it (24.1 / 21.3 ms); the barrier and the re-split cost ~1 ms.
// Results are identical whatever the split: the groups never
// interact, a lane's list keeps every group's order, and the lists are merged back into group order.
struct LanePlan
{
  std::vector<int> lane_of;
};
static LanePlan plan_blind(const bk_ctx *ctx, int K)
{
  const uint32_t ng = ctx->jr.n_groups;
  std::vector<uint32_t> order(ng);
  std::iota(order.begin(), order.end(), 0u);
  auto size_of = [&](uint32_t g) { return ctx->gstart_host[g + 1] - ctx->gstart_host[g]; };
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return size_of(a) > size_of(b); });
  constexpr double wexp = 2.0;  // longest-processing-time on size^2
  LanePlan p;
  p.lane_of.assign(ng, 0);
  std::vector<double> load(K, 0.0);
  // sharded sample: this rank only masks and clusters the groups it owns (bk_shard_own_groups); the others belong to no lane
  const bool owned_only = !ctx->own_groups.empty();
  if (owned_only && ctx->own_groups.size() != ng) throw bk_error(BK_ERR_ARG, "bk_shard_own_groups: group count changed");
  for (uint32_t i = 0; i < ng; ++i)
  {
    const uint32_t g = order[i];
    if (owned_only && !ctx->own_groups[g])
    {
      p.lane_of[g] = -1;
      continue;
    }
    int l = 0;
    for (int k = 1; k < K; ++k)
      if (load[k] < load[l]) l = k;
    load[l] += std::pow((double) size_of(g), wexp);
    p.lane_of[g] = l;
  }
  return p;
}
// sizes = pairs per group now; hx / hy = longest heap segment per group seen in a sort by x / by y (0: none)
static LanePlan plan_observed(const std::vector<uint64_t> &sizes, const std::vector<uint32_t> &hx, const std::vector<uint32_t> &hy, int K, double x_sorts = 1.0)
{
  const uint32_t ng = (uint32_t) sizes.size();
  // in units of one pop of a lone wave in LDS (~0.15 us): a pair costs a lane ~0.15 ns in the two sorts that are left (most of a
  // lane's time outside the heaps is a fixed number of launches), an element of a heap segment beyond what fits the LDS of a
  // CU costs twice as much (the hybrid loop: 0.30 us per pop while the heap's tail is in global memory)
  constexpr double per_pair = 0.001;
  auto heap_cost = [](uint32_t m) { return (double) m + (m > 40947u ? 1.0 * (double) (m - 40947u) : 0.0); };
  std::vector<uint32_t> order(ng);
  std::iota(order.begin(), order.end(), 0u);
  auto chain = [&](uint32_t g) { return heap_cost(hx[g]) + heap_cost(hy[g]); };
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    const double ca = chain(a), cb = chain(b);
    return ca != cb ? ca > cb : sizes[a] > sizes[b];
  });
  LanePlan p;
  p.lane_of.assign(ng, 0);
  std::vector<double> mx(K, 0.0), my(K, 0.0), pairs(K, 0.0);
  auto cost = [&](int l) { return x_sorts * mx[l] + my[l] + per_pair * pairs[l]; };  // x_sorts: how many sorts by x are left (one by y)
  for (uint32_t g : order)
  {
    const int l0 = 0;
    int best = l0;
    double best_cost = 0;
    for (int l = l0; l < K; ++l)
    {
      const double c = x_sorts * std::max(mx[l], heap_cost(hx[g])) + std::max(my[l], heap_cost(hy[g])) + per_pair * (pairs[l] + (double) sizes[g]);
      // the lane whose own cost ends lowest takes the group (ties: the emptier lane)
      if (l == l0 || c < best_cost || (c == best_cost && cost(l) < cost(best)))
      {
        best = l;
        best_cost = c;
      }
    }
    mx[best] = std::max(mx[best], heap_cost(hx[g]));
    my[best] = std::max(my[best], heap_cost(hy[g]));
    pairs[best] += (double) sizes[g];
    p.lane_of[g] = best;
  }
  if (bk_debug("lanes"))
    for (int l = 0; l < K; ++l)
    {
      fprintf(stderr, "[lanes] lane %d: max heap x %.0f y %.0f, %.0f pairs, cost %.0f; heavy groups:", l, mx[l], my[l], pairs[l], cost(l));
      for (uint32_t g = 0; g < ng; ++g)
        if (p.lane_of[g] == l && (hx[g] || hy[g])) fprintf(stderr, " %u(%u,%u)", g, hx[g], hy[g]);
      fprintf(stderr, "\n");
    }
  return p;
}

static void group_lanes(bk_ctx *ctx, double w, int fast)
{
  const uint32_t ng = ctx->jr.n_groups;
  // With the resident sort service a lane's stream is idle most of the time (its thread waits for the sort's job), so the twelve
  // lanes share FOUR streams: measured 12 lanes on 12 / 8 / 4 streams 44.5-44.8 / 44.4-44.7 / 44.5-45.2 ms per step (6 streams, two
  // heavy lanes per stream: 46.7-47.4; round-4 start, with waiting kernels on the streams: 12 / 4 / 3 / 2 streams 31.6 / 31.2 / 35.0 /
  // 40.2 ms for the stage).  Fewer streams = fewer hardware queues: the stage then needs 4 + 2 of them, and a second process on the
  // device (a test runner's parent, another sample) leaves the device's queues uncrowded (SVC_MAX_DEVICE_QUEUES).
  constexpr int lane_streams_max = 4;
  auto make_lanes = [&](int K, int S) {
    while ((int) ctx->lanes.size() < K - 1)
    {
      ctx->lanes.emplace_back(new bk_ctx::Lane());
      ctx->lanes.back()->cb.max_group_bound = ctx->cb.max_group_bound;
    }
    for (int k = 0; k < S - 1; ++k)
      if (!ctx->lanes[k]->st) HIP_CHECK(hipStreamCreateWithFlags(&ctx->lanes[k]->st, hipStreamNonBlocking));
  };
  bool use_svc = sort_service_on() && fast;
  int K = lanes_wanted(use_svc);
  int S = use_svc ? std::max(1, std::min(lane_streams_max, K)) : K;
  make_lanes(K, S);
  std::vector<hipStream_t> stage_streams{ctx->st};
  for (int k = 0; k < S - 1; ++k) stage_streams.push_back(ctx->lanes[k]->st);
  SvcStage svc_stage(ctx, use_svc, ctx->jr.n_pairs + 2ull * ng + 4096, ctx->cb.max_group_bound, stage_streams);  // the service runs from here to the end of the lanes (also when one of them throws)
  if (use_svc && !svc_stage.on)
  {
    // the service is not to be had (another context of this process has it on this device, or a hardware queue is shared): the lanes of the launch path
    use_svc = false;
    K = lanes_wanted(false);
    S = K;
    make_lanes(K, S);
  }
  const bool adapt = !use_svc;  // (the service does not report the groups' longest heap segments back to the host)
  auto lane_cb = [&](int l) -> ClusterBufs & { return l == 0 ? ctx->cb : ctx->lanes[l - 1]->cb; };
  auto lane_st = [&](int l) { const int k = l % S; return k == 0 ? ctx->st : ctx->lanes[k - 1]->st; };
  auto lane_list = [&](int l) -> PairList & { return l == 0 ? ctx->listA : ctx->lanes[l - 1]->list; };
  auto lane_iso = [&](int l) -> PairList & { return l == 0 ? ctx->isoA : ctx->lanes[l - 1]->iso; };
  auto lane_cl = [&](int l) -> DevBuf & { return l == 0 ? ctx->d_clusterA : ctx->lanes[l - 1]->d_cluster; };
  uint32_t *drop_base = nullptr;  // the lanes' plans, one row of ng + 1 words each, uploaded in one copy
  auto lane_drop = [&](int l) { return drop_base + (size_t) l * ((size_t) ng + 1); };
  std::vector<std::vector<uint8_t>> keep(K, std::vector<uint8_t>(ng, 0));  // keep[l][g]: lane l owns group g (host copy of the plan)
  auto upload_plan = [&](const LanePlan &p) {
    for (int l = 0; l < K; ++l)
      for (uint32_t g = 0; g < ng; ++g) keep[l][g] = p.lane_of[g] == l ? 1 : 0;
    std::vector<uint32_t> drop((size_t) K * ((size_t) ng + 1), 0u);
    for (int l = 0; l < K; ++l)
      for (uint32_t g = 0; g < ng; ++g) drop[(size_t) l * ((size_t) ng + 1) + g] = p.lane_of[g] == l ? 0u : 1u;  // a lane drops what the others own
    drop_base = ctx->d_dropA.as<uint32_t>(drop.size() + 1);
    HIP_CHECK(hipMemcpyAsync(drop_base, drop.data(), drop.size() * 4, hipMemcpyHostToDevice, ctx->st));
    HIP_CHECK(hipStreamSynchronize(ctx->st));  // (the pair table and the masks are ready for all lanes)
  };
  // runs body(l) for every lane, lane 0 on this thread; the lanes' streams are synchronised when this returns
  auto in_lanes = [&](auto body) {
    std::vector<std::string> err(K);
    std::vector<int> code(K, BK_OK);
    static const bool dbg_lanes = bk_debug("lanes");
    const auto t_start = std::chrono::steady_clock::now();
    auto guarded_body = [&](int l) {
      try
      {
        body(l);
        HIP_CHECK(hipStreamSynchronize(lane_st(l)));
        if (dbg_lanes)
          fprintf(stderr, "[lanes] lane %d done after %.2f ms\n", l, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
      }
      catch (const bk_error &e)
      {
        code[l] = e.code;
        err[l] = e.msg;
      }
    };
    std::vector<std::thread> th;
    for (int l = 1; l < K; ++l)
      th.emplace_back([&, l] {
        (void) hipSetDevice(ctx->device);
        guarded_body(l);
      });
    guarded_body(0);
    for (std::thread &t : th) t.join();
    for (int l = 0; l < K; ++l)
      if (code[l] != BK_OK) throw bk_error(code[l], err[l]);
  };
  // folds the lanes' lists (disjoint groups) into one list in group order; cl = the cluster numbers travel along
  auto merge_all = [&](auto list_of, auto cl_of, PairList &out, DevBuf *cl_out, PairList *, DevBuf *) {
    std::vector<const PairList *> ls(K);
    std::vector<const uint32_t *> cs(K);
    for (int l = 0; l < K; ++l)
    {
      ls[l] = &list_of(l);
      cs[l] = cl_of(l);
    }
    merge_lists_many(ls.data(), cl_out ? cs.data() : nullptr, K, out, cl_out, ctx->st);
  };
  const bk_pair *pairs = ctx->jr.pairs;
  static const bool dbg_phases = bk_debug("lanes");
  const auto tp0 = std::chrono::steady_clock::now();
  auto phase = [&](const char *what) {
    if (dbg_phases) fprintf(stderr, "[lanes] %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
  };
  upload_plan(plan_blind(ctx, K));
  phase("plan uploaded");
  phase("service started");
  if (!adapt)
  {
    in_lanes([&](int l) { run_lane(pairs, ctx->jr.gof, ctx->jr.gstart, ng, ctx->jr.n_pairs, w, fast, lane_drop(l), lane_list(l), lane_iso(l), lane_cl(l), lane_cb(l), ctx->ab, lane_st(l), ctx->gstart_host.data(), keep[l].data()); });
  }
  else
  {
    // part 1: sorts 1-3, observed.  (Dealing again after the second sort already - x from the first sort, y from the second, three
    // sorts left - was measured worse, 42.0 ms against 38.1: the heaps of the FIRST sort by x, short, on the unmasked list, say
    // little about the later ones.)
    constexpr bool split_early = false;
    in_lanes([&](int l) {
      ClusterBufs &cb = lane_cb(l);
      cb.heavy_x.assign(ng, 0u);
      cb.heavy_y.assign(ng, 0u);
      cb.observe = true;
      cb.se.heavy_all = false;
      remove_isolated_begin(pairs, ctx->jr.gof, ctx->jr.gstart, ng, ctx->jr.n_pairs, w, lane_list(l), cb, lane_st(l), lane_drop(l), ctx->gstart_host.data(), keep[l].data());
      if (!split_early)
      {
        cb.heavy_x.assign(ng, 0u);  // the third sort (by x, on the masked list) is the one that tells about the fifth
        remove_isolated_end(pairs, lane_list(l), cb, lane_st(l));
      }
      cb.observe = false;
    });
    PairList &mid = ctx->lane_mid;
    PairList *acc = ctx->lane_acc;
    DevBuf *cacc = ctx->lane_cacc;
    merge_all([&](int l) -> const PairList & { return lane_list(l); }, [&](int) -> const uint32_t * { return nullptr; }, mid, nullptr, acc, cacc);
    std::vector<uint64_t> goff_h((size_t) ng + 1);
    HIP_CHECK(hipMemcpyAsync(goff_h.data(), mid.goff.get<uint64_t>(), ((size_t) ng + 1) * 8, hipMemcpyDeviceToHost, ctx->st));
    HIP_CHECK(hipStreamSynchronize(ctx->st));
    std::vector<uint64_t> sizes(ng);
    std::vector<uint32_t> hx(ng, 0u), hy(ng, 0u);
    for (uint32_t g = 0; g < ng; ++g) sizes[g] = goff_h[g + 1] - goff_h[g];
    for (int l = 0; l < K; ++l)
      for (uint32_t g = 0; g < ng; ++g)
      {
        hx[g] = std::max(hx[g], lane_cb(l).heavy_x[g]);
        hy[g] = std::max(hy[g], lane_cb(l).heavy_y[g]);
      }
    upload_plan(plan_observed(sizes, hx, hy, K, split_early ? 2.0 : 1.0));
    // part 2: sorts (3,) 4, 5
    in_lanes([&](int l) {
      PairList &L = lane_list(l), &iso = lane_iso(l);
      hipStream_t st = lane_st(l);
      list_subset_ranges(mid, goff_h.data(), keep[l].data(), L, st);
      if (split_early) remove_isolated_end(pairs, L, lane_cb(l), st);
      iso.n = L.n;
      iso.ng = L.ng;
      uint32_t *ii = iso.idx.as<uint32_t>(L.n + 1), *ig = iso.gof.as<uint32_t>(L.n + 1);
      uint64_t *io = iso.goff.as<uint64_t>((uint64_t) L.ng + 1);
      if (L.n) HIP_CHECK(hipMemcpyAsync(ii, L.idx.get<uint32_t>(), L.n * 4, hipMemcpyDeviceToDevice, st));
      if (L.n) HIP_CHECK(hipMemcpyAsync(ig, L.gof.get<uint32_t>(), L.n * 4, hipMemcpyDeviceToDevice, st));
      HIP_CHECK(hipMemcpyAsync(io, L.goff.get<uint64_t>(), ((uint64_t) L.ng + 1) * 8, hipMemcpyDeviceToDevice, st));
      if (fast)
        fast_cluster_all(pairs, L, w, lane_cl(l), lane_cb(l), st);
      else
        ahc_cluster_all(pairs, L, w, lane_cl(l), ctx->ab, lane_cb(l), st);
    });
  }
  phase("lanes done");
  svc_stage.finish();  // (throws what a task reported)
  phase("service stopped");
  if (use_svc && bk_debug("lanes")) fprintf(stderr, "[svc] tasks: %u wide, %u narrow\n", ctx->svc.stats[0], ctx->svc.stats[1]);
  // one list in group order again
  PairList &iso_m = ctx->lane_iso_m;
  PairList *iacc = ctx->lane_acc, *lacc = ctx->lane_acc + 2;
  DevBuf *icacc = ctx->lane_cacc, *lcacc = ctx->lane_cacc + 2;
  merge_all([&](int l) -> const PairList & { return lane_iso(l); }, [&](int) -> const uint32_t * { return nullptr; }, iso_m, nullptr, iacc, icacc);
  merge_all([&](int l) -> const PairList & { return lane_list(l); }, [&](int l) -> const uint32_t * { return lane_cl(l).get<uint32_t>(); }, ctx->list, &ctx->d_cluster, lacc, lcacc);
  HIP_CHECK(hipStreamSynchronize(ctx->st));
  phase("lists merged");
  ctx->iso_n = iso_m.n;
  std::swap(ctx->iso_idx, iso_m.idx);
  std::swap(ctx->iso_goff, iso_m.goff);
}

int bk_mask_and_cluster(bk_ctx *ctx, double w, int fast, uint64_t *n_clustered)
{
  return guarded(ctx, [&] {
    {
      uint64_t mg = 0;
      for (uint32_t g = 0; g < ctx->jr.n_groups && g + 1 < ctx->gstart_host.size(); ++g) mg = std::max<uint64_t>(mg, ctx->gstart_host[g + 1] - ctx->gstart_host[g]);
      ctx->cb.max_group_bound = mg;
      for (auto &l : ctx->lanes) l->cb.max_group_bound = mg;
    }
    if (lanes_apply(ctx, fast))
    {
      {
        Scope s(ctx, "mask_and_cluster_lanes");
        group_lanes(ctx, w, fast);
      }
      ctx->clustered = true;
      if (n_clustered) *n_clustered = ctx->list.n;
      return;
    }
    SvcStage svc_stage(ctx, true, ctx->jr.n_pairs + 2ull * ctx->jr.n_groups + 4096, ctx->cb.max_group_bound, {ctx->st});
    {
      Scope s(ctx, "remove_isolated");
      const uint32_t *drop = nullptr;
      if (!ctx->own_groups.empty())
      {
        // sharded sample: this rank masks and clusters only the chromosome-pair groups it owns
        if (ctx->own_groups.size() != ctx->jr.n_groups) throw bk_error(BK_ERR_ARG, "bk_shard_own_groups: group count changed");
        std::vector<uint32_t> d(ctx->jr.n_groups);
        for (uint32_t g = 0; g < ctx->jr.n_groups; ++g) d[g] = ctx->own_groups[g] ? 0u : 1u;
        uint32_t *dd = ctx->d_drop.as<uint32_t>((uint64_t) ctx->jr.n_groups + 1);
        if (ctx->jr.n_groups) HIP_CHECK(hipMemcpyAsync(dd, d.data(), d.size() * 4, hipMemcpyHostToDevice, ctx->st));
        HIP_CHECK(hipStreamSynchronize(ctx->st));
        drop = dd;
      }
      remove_isolated_all(ctx->jr.pairs, ctx->jr.gof, ctx->jr.gstart, ctx->jr.n_groups, ctx->jr.n_pairs, w, ctx->list, ctx->cb, ctx->st, drop);
    }
    ctx->iso_n = ctx->list.n;
    uint32_t *ii = ctx->iso_idx.as<uint32_t>(ctx->list.n + 1);
    uint64_t *ig = ctx->iso_goff.as<uint64_t>((uint64_t) ctx->list.ng + 1);
    if (ctx->list.n) HIP_CHECK(hipMemcpyAsync(ii, ctx->list.idx.get<uint32_t>(), ctx->list.n * 4, hipMemcpyDeviceToDevice, ctx->st));
    if (ctx->list.ng) HIP_CHECK(hipMemcpyAsync(ig, ctx->list.goff.get<uint64_t>(), ((uint64_t) ctx->list.ng + 1) * 8, hipMemcpyDeviceToDevice, ctx->st));
    if (!fast) svc_stage.finish();  // (the exact UPGMA replay sorts nothing and may take long: the service's workgroups would hold their CUs, then leave on their own)
    {
      Scope s(ctx, fast ? "fast_cluster" : "ahc_cluster");
      if (fast)
        fast_cluster_all(ctx->jr.pairs, ctx->list, w, ctx->d_cluster, ctx->cb, ctx->st);
      else
        ahc_cluster_all(ctx->jr.pairs, ctx->list, w, ctx->d_cluster, ctx->ab, ctx->cb, ctx->st);
    }
    svc_stage.finish();
    ctx->clustered = true;
    if (n_clustered) *n_clustered = ctx->list.n;
  });
}

int bk_split_evidence(bk_ctx *ctx, uint64_t *n_tuples)
{
  return guarded(ctx, [&] {
    if (!ctx->stream_done) run_stream(ctx);
    ensure_splits_sorted(ctx);
    if (n_tuples) *n_tuples = ctx->hc.n_split;
  });
}

int bk_cluster_summary(bk_ctx *ctx, double w, uint64_t *n_clusters)
{
  return guarded(ctx, [&] {
    if (!ctx->clustered) throw bk_error(BK_ERR_ARG, "bk_cluster_summary: call bk_mask_and_cluster first");
    Scope s(ctx, "cluster_summary");
    ctx->n_clusters = cluster_summary(ctx->jr.pairs, ctx->list.idx.get<uint32_t>(), ctx->list.gof.get<uint32_t>(), ctx->d_cluster.get<uint32_t>(), ctx->list.n,
                                      ctx->list.ng, ctx->jr.gkey, ctx->d_glex.get<uint32_t>(), ctx->nt, w, ctx->d_clusters, ctx->bb, ctx->st);
    if (n_clusters) *n_clusters = ctx->n_clusters;
  });
}

int bk_split_breakpoints(bk_ctx *ctx, double w, uint64_t *n_valid)
{
  return guarded(ctx, [&] {
    ensure_splits_sorted(ctx);
    RecView r;
    r.n = ctx->rec.n;
    r.tid = ctx->rec.tid; r.pos = ctx->rec.pos; r.flag = ctx->rec.flag; r.mapq = ctx->rec.mapq;
    r.cigar_off = ctx->rec.cigar_off; r.cigar = ctx->rec.cigar;
    {
      Scope s(ctx, "split_breakpoints");
      split_breakpoints(r, ctx->d_split.get<bk_split>(), ctx->hc.n_split, ctx->clusters_ptr(), ctx->n_clusters, w, (int) ctx->hc.max_span,
                        ctx->d_hdr.get<int32_t>(), ctx->bb, ctx->st);
    }
    if (n_valid) *n_valid = count_valid_clusters(ctx->clusters_ptr(), ctx->n_clusters, ctx->bb, ctx->st);
  });
}

int bk_run(bk_ctx *ctx, int mapq_min, int fast, double *w_out, uint64_t *n_valid)
{
  if (!ctx) return BK_ERR_ARG;
  if (ctx->stream_done && mapq_min != ctx->stream_mapq) ctx->stream_done = false;  // candidates were filtered with another threshold
  ctx->mapq_min = mapq_min;
  double mean, sd;
  int rc = bk_isize_stats(ctx, &mean, &sd);
  if (rc) return rc;
  const int times = 2;
  const double w = times * std::sqrt(times) * (mean + 3 * sd);  // BreakID.cc:103
  if (w_out) *w_out = w;
  if ((rc = bk_discordant_pairs(ctx, mapq_min, w, nullptr, nullptr))) return rc;
  if ((rc = bk_mask_and_cluster(ctx, w, fast, nullptr))) return rc;
  if ((rc = bk_split_evidence(ctx, nullptr))) return rc;
  if ((rc = bk_cluster_summary(ctx, w, nullptr))) return rc;
  if ((rc = bk_split_breakpoints(ctx, w, n_valid))) return rc;
  return BK_OK;
}

int bk_fetch(bk_ctx *ctx, int stage, const void **data, uint64_t *count, const uint64_t **group_off, uint32_t *n_groups)
{
  return guarded(ctx, [&] {
    if (!data || !count) throw bk_error(BK_ERR_ARG, "bk_fetch: null output");
    const uint32_t ng = ctx->jr.n_groups;
    if (group_off) *group_off = nullptr;
    if (n_groups) *n_groups = 0;
    auto fetch_list = [&](int slot, const uint32_t *d_idx, const uint64_t *d_goff, const uint32_t *d_cl, uint64_t n) {
      std::vector<bk_pair> all(ctx->jr.n_pairs);
      if (!all.empty()) HIP_CHECK(hipMemcpyAsync(all.data(), ctx->jr.pairs, all.size() * sizeof(bk_pair), hipMemcpyDeviceToHost, ctx->st));
      std::vector<uint32_t> idx(n), cl(d_cl ? n : 0);
      std::vector<uint64_t> goff(ng + 1, 0);
      if (n && d_idx) HIP_CHECK(hipMemcpyAsync(idx.data(), d_idx, n * 4, hipMemcpyDeviceToHost, ctx->st));
      if (n && d_cl) HIP_CHECK(hipMemcpyAsync(cl.data(), d_cl, n * 4, hipMemcpyDeviceToHost, ctx->st));
      if (ng && d_goff) HIP_CHECK(hipMemcpyAsync(goff.data(), d_goff, (ng + 1) * 8, hipMemcpyDeviceToHost, ctx->st));
      HIP_CHECK(hipStreamSynchronize(ctx->st));
      auto &out = ctx->f_pairs[slot];
      auto &off = ctx->f_off[slot];
      out.clear();
      off.assign(1, 0);
      for (uint32_t l = 0; l < ng; ++l)
      {
        uint32_t g = ctx->lex_to_num[l];
        for (uint64_t p = goff[g]; p < goff[g + 1]; ++p)
        {
          bk_pair pr = all[d_idx ? idx[p] : p];
          if (d_cl) pr.cluster = (int32_t) cl[p];
          out.push_back(pr);
        }
        off.push_back(out.size());
      }
      *data = out.data();
      *count = out.size();
      if (group_off) *group_off = off.data();
      if (n_groups) *n_groups = ng;
    };
    switch (stage)
    {
    case BK_STAGE_SCAN:
      fetch_list(0, nullptr, ctx->jr.gstart, nullptr, ctx->jr.n_pairs);
      break;
    case BK_STAGE_ISO:
      fetch_list(1, ctx->iso_idx.get<uint32_t>(), ctx->iso_goff.get<uint64_t>(), nullptr, ctx->iso_n);
      break;
    case BK_STAGE_CLUSTERED:
      if (!ctx->clustered) throw bk_error(BK_ERR_ARG, "bk_fetch: not clustered yet");
      fetch_list(2, ctx->list.idx.get<uint32_t>(), ctx->list.goff.get<uint64_t>(), ctx->d_cluster.get<uint32_t>(), ctx->list.n);
      break;
    case BK_STAGE_SPLITS:
      ensure_splits_sorted(ctx);
      ctx->f_splits.resize(ctx->hc.n_split);
      if (ctx->hc.n_split)
        HIP_CHECK(hipMemcpyAsync(ctx->f_splits.data(), ctx->d_split.get<bk_split>(), ctx->hc.n_split * sizeof(bk_split), hipMemcpyDeviceToHost, ctx->st));
      HIP_CHECK(hipStreamSynchronize(ctx->st));
      *data = ctx->f_splits.data();
      *count = ctx->f_splits.size();
      break;
    case BK_STAGE_CLUSTERS:
      ctx->f_clusters.resize(ctx->n_clusters);
      if (ctx->n_clusters)
        HIP_CHECK(hipMemcpyAsync(ctx->f_clusters.data(), ctx->clusters_ptr(), ctx->n_clusters * sizeof(bk_cluster), hipMemcpyDeviceToHost, ctx->st));
      HIP_CHECK(hipStreamSynchronize(ctx->st));
      // device order is (numeric chr-pair key, id); the reference appends groups in std::map<string> order
      std::stable_sort(ctx->f_clusters.begin(), ctx->f_clusters.end(), [](const bk_cluster &a, const bk_cluster &b) { return a.group < b.group; });
      *data = ctx->f_clusters.data();
      *count = ctx->f_clusters.size();
      break;
    case BK_STAGE_GROUP_KEYS:
      ctx->f_gkeys.clear();
      for (uint32_t l = 0; l < ng; ++l)
      {
        uint32_t k = ctx->gkey_host[ctx->lex_to_num[l]];
        ctx->f_gkeys.push_back((int32_t) (k / (uint32_t) (ctx->nt + 1)) - 1);
        ctx->f_gkeys.push_back((int32_t) (k % (uint32_t) (ctx->nt + 1)) - 1);
      }
      *data = ctx->f_gkeys.data();
      *count = ng;
      break;
    default:
      throw bk_error(BK_ERR_ARG, "bk_fetch: unknown stage");
    }
  });
}

// ---- single-sample sharding: one context per GPU holds a contiguous range of the sample's records ----------
int bk_shard_begin(bk_ctx *ctx, uint64_t rec_base, int mapq_min)
{
  return guarded(ctx, [&] {
    ctx->rec_base = rec_base;
    ctx->mapq_min = mapq_min;
    ctx->ext_cand = nullptr;
    ctx->ext_split = nullptr;
    ctx->ext_clusters = nullptr;
    run_stream(ctx);
  });
}

int bk_shard_get_stats(bk_ctx *ctx, bk_shard_stats *out)
{
  return guarded(ctx, [&] {
    if (!ctx->stream_done || !out) throw bk_error(BK_ERR_ARG, "bk_shard_get_stats: call bk_shard_begin first");
    out->isize_sum = ctx->hc.isize_sum;
    out->isize_n = ctx->hc.isize_n;
    out->sumsq = ctx->hsd.sumsq;
    out->vmax = ctx->hsd.vmax;
    out->max_span = ctx->hc.max_span;
    out->n_cand = ctx->hc.n_cand;
    out->n_split = ctx->hc.n_split;
  });
}

int bk_shard_set_stats(bk_ctx *ctx, const bk_shard_stats *total)
{
  return guarded(ctx, [&] {
    if (!ctx->stream_done || !total) throw bk_error(BK_ERR_ARG, "bk_shard_set_stats: call bk_shard_begin first");
    ctx->hc.isize_sum = total->isize_sum;
    ctx->hc.isize_n = total->isize_n;
    ctx->hsd.sumsq = total->sumsq;
    ctx->hsd.vmax = total->vmax;
    ctx->hc.max_span = total->max_span;
    ctx->stats_done = false;
  });
}

static void sd_mean_thr(bk_ctx *ctx, double &m, double &thr)
{
  const double n = (double) ctx->hc.isize_n;
  m = (double) (long long) ctx->hc.isize_sum / n;  // (double) long / (double) size_t, BreakID.cc:1941
  double sum_d = ctx->hsd.sumsq - 2.0 * m * (double) ctx->hc.isize_sum + n * m * m;
  if (!(sum_d > 0)) sum_d = 0;
  double da = (double) ctx->hsd.vmax - m, dmax = da * da + m * m;
  double bound = 2.0 * sum_d + 2.0 * n + 2.0 * dmax + 4.0;
  int k = std::ilogb(bound) + 1;
  thr = k >= 51 ? 1.0e300 : std::ldexp(1.0, k - 53);
}

int bk_shard_sd_local(bk_ctx *ctx, uint64_t *l_total, void **ex_dev, uint64_t *n_ex)
{
  return guarded(ctx, [&] {
    if (!ctx->stream_done) throw bk_error(BK_ERR_ARG, "bk_shard_sd_local: call bk_shard_begin / bk_shard_set_stats first");
    double m, thr;
    sd_mean_thr(ctx, m, thr);
    unsigned long long lt = 0, ne = 0;
    launch_sd_local(ctx->rec.flag, ctx->rec.isize, ctx->rec.n, m, thr, ctx->sdb, ctx->st, &lt, &ne);
    HIP_CHECK(hipStreamSynchronize(ctx->st));
    if (l_total) *l_total = lt;
    if (n_ex) *n_ex = ne;
    if (ex_dev) *ex_dev = ctx->sdb.exceptions.p;
  });
}

int bk_shard_sd_finish(bk_ctx *ctx, const void *all_ex_dev, uint64_t n_all, uint64_t l_grand, double *mean, double *sd)
{
  return guarded(ctx, [&] {
    double m, thr;
    sd_mean_thr(ctx, m, thr);
    launch_sd_walk((const SdException *) all_ex_dev, n_all, l_grand, ctx->d_sd.get<SdState>(), ctx->st);
    HIP_CHECK(hipMemcpyAsync(&ctx->hsd.t_final, &ctx->d_sd.get<SdState>()->t_final, 8, hipMemcpyDeviceToHost, ctx->st));
    HIP_CHECK(hipStreamSynchronize(ctx->st));
    ctx->mean = m;
    ctx->sd = std::sqrt((double) ctx->hsd.t_final / (double) ctx->hc.isize_n);
    ctx->stats_done = true;
    if (mean) *mean = ctx->mean;
    if (sd) *sd = ctx->sd;
  });
}

int bk_shard_buffer(bk_ctx *ctx, int which, void **dev, uint64_t *count, uint32_t *elem_bytes)
{
  return guarded(ctx, [&] {
    if (!dev || !count || !elem_bytes) throw bk_error(BK_ERR_ARG, "bk_shard_buffer: null output");
    switch (which)
    {
    case BK_BUF_CANDIDATES: *dev = ctx->d_cand.p; *count = ctx->ext_cand ? 0 : ctx->hc.n_cand; *elem_bytes = sizeof(Cand); break;
    case BK_BUF_TUPLES: *dev = ctx->d_split_raw.p; *count = ctx->ext_split ? 0 : ctx->hc.n_split; *elem_bytes = sizeof(bk_split); break;
    case BK_BUF_CLUSTERS: *dev = ctx->d_clusters.p; *count = ctx->ext_clusters ? 0 : ctx->n_clusters; *elem_bytes = sizeof(bk_cluster); break;
    default: throw bk_error(BK_ERR_ARG, "bk_shard_buffer: unknown buffer");
    }
  });
}

int bk_shard_set_buffer(bk_ctx *ctx, int which, const void *dev, uint64_t count)
{
  return guarded(ctx, [&] {
    switch (which)
    {
    case BK_BUF_CANDIDATES: ctx->ext_cand = (const Cand *) dev; ctx->hc.n_cand = count; break;
    case BK_BUF_TUPLES: ctx->ext_split = (const bk_split *) dev; ctx->hc.n_split = count; ctx->splits_sorted = false; break;
    case BK_BUF_CLUSTERS: ctx->ext_clusters = (bk_cluster *) dev; ctx->n_clusters = count; break;
    default: throw bk_error(BK_ERR_ARG, "bk_shard_set_buffer: unknown buffer");
    }
  });
}

int bk_shard_group_sizes(bk_ctx *ctx, const uint64_t **starts, uint32_t *n_groups)
{
  return guarded(ctx, [&] {
    if (starts) *starts = ctx->gstart_host.data();
    if (n_groups) *n_groups = ctx->jr.n_groups;
  });
}

int bk_shard_own_groups(bk_ctx *ctx, const uint8_t *own, uint32_t n_groups)
{
  return guarded(ctx, [&] {
    if (n_groups != ctx->jr.n_groups) throw bk_error(BK_ERR_ARG, "bk_shard_own_groups: wrong group count");
    ctx->own_groups.assign(own, own + n_groups);
  });
}

int bk_shard_route_candidates(bk_ctx *ctx, uint32_t world, void **dev, const uint64_t **counts)
{
  return guarded(ctx, [&] {
    if (!ctx->stream_done || ctx->ext_cand) throw bk_error(BK_ERR_ARG, "bk_shard_route_candidates: needs this shard's own candidates (after bk_shard_begin)");
    if (!world || world > 4096 || !dev || !counts) throw bk_error(BK_ERR_ARG, "bk_shard_route_candidates: bad arguments");
    *dev = route_candidates(ctx->d_cand.get<Cand>(), ctx->hc.n_cand, world, ctx->jb, ctx->st, ctx->route_counts);
    *counts = ctx->route_counts.data();
  });
}

int bk_shard_group_keys(bk_ctx *ctx, const uint32_t **keys, uint32_t *n_groups)
{
  return guarded(ctx, [&] {
    if (keys) *keys = ctx->gkey_host.data();
    if (n_groups) *n_groups = ctx->jr.n_groups;
  });
}

int bk_shard_route_pairs(bk_ctx *ctx, const uint32_t *dest_of_group, uint32_t n_groups, uint32_t world, void **dev, const uint64_t **counts)
{
  return guarded(ctx, [&] {
    if (n_groups != ctx->jr.n_groups || !world || !dev || !counts || (n_groups && !dest_of_group)) throw bk_error(BK_ERR_ARG, "bk_shard_route_pairs: bad arguments");
    ctx->route_counts.assign(world, 0);
    for (uint32_t g = 0; g < n_groups; ++g)
    {
      if (dest_of_group[g] >= world) throw bk_error(BK_ERR_ARG, "bk_shard_route_pairs: destination out of range");
      ctx->route_counts[dest_of_group[g]] += ctx->gstart_host[g + 1] - ctx->gstart_host[g];
    }
    std::vector<uint64_t> base(world, 0), off(n_groups, 0);
    for (uint32_t d = 1; d < world; ++d) base[d] = base[d - 1] + ctx->route_counts[d - 1];
    for (uint32_t g = 0; g < n_groups; ++g)
    {
      off[g] = base[dest_of_group[g]];
      base[dest_of_group[g]] += ctx->gstart_host[g + 1] - ctx->gstart_host[g];
    }
    *dev = route_pairs(ctx->jr, off, ctx->jb, ctx->st);
    *counts = ctx->route_counts.data();
  });
}

int bk_shard_group_pairs(bk_ctx *ctx, const void *pairs_dev, uint64_t n, const uint32_t *all_keys, uint32_t n_all_keys)
{
  return guarded(ctx, [&] {
    if ((n && !pairs_dev) || (n_all_keys && !all_keys)) throw bk_error(BK_ERR_ARG, "bk_shard_group_pairs: null input");
    {
      Scope s(ctx, "group_pairs");
      ctx->jb2.rec_bits = -1;  // the pairs carry the discovery indices of the whole sample
      group_pairs((const bk_pair *) pairs_dev, n, ctx->nt, ctx->jb2, ctx->st, ctx->jr);
    }
    std::vector<uint32_t> keys(all_keys, all_keys + n_all_keys);
    finish_groups(ctx, &keys);
    ctx->own_groups.clear();  // the table holds exactly the groups this rank owns
    ctx->clustered = false;
  });
}

int bk_shard_bp_cov(bk_ctx *ctx, double w, void **cov_dev, uint64_t *n)
{
  return guarded(ctx, [&] {
    uint32_t *cov = bp_cov_partial(rec_view(ctx), ctx->clusters_ptr(), ctx->n_clusters, w, (int) ctx->hc.max_span, ctx->bb, ctx->st);
    HIP_CHECK(hipStreamSynchronize(ctx->st));
    if (cov_dev) *cov_dev = cov;
    if (n) *n = 2 * ctx->n_clusters;
  });
}

int bk_shard_bp_vote(bk_ctx *ctx, double w, const void *cov_total_dev)
{
  return guarded(ctx, [&] {
    ensure_splits_sorted(ctx);
    bp_vote(ctx->d_split.get<bk_split>(), ctx->hc.n_split, ctx->clusters_ptr(), ctx->n_clusters, w, (int) ctx->hc.max_span, (const uint32_t *) cov_total_dev,
            ctx->d_hdr.get<int32_t>(), ctx->bb, ctx->st);
    HIP_CHECK(hipStreamSynchronize(ctx->st));
  });
}

int bk_shard_bp_vote_slice(bk_ctx *ctx, double w, const void *cov_total_dev, uint64_t lo, uint64_t hi, void **clusters_dev, void **voted_dev)
{
  return guarded(ctx, [&] {
    if (lo > hi || hi > ctx->n_clusters || !clusters_dev || !voted_dev) throw bk_error(BK_ERR_ARG, "bk_shard_bp_vote_slice: bad range");
    ensure_splits_sorted(ctx);
    bp_vote(ctx->d_split.get<bk_split>(), ctx->hc.n_split, ctx->clusters_ptr() + lo, hi - lo, w, (int) ctx->hc.max_span,
            cov_total_dev ? (const uint32_t *) cov_total_dev + 2 * lo : nullptr, ctx->d_hdr.get<int32_t>(), ctx->bb, ctx->st);
    HIP_CHECK(hipStreamSynchronize(ctx->st));
    *clusters_dev = ctx->clusters_ptr() + lo;
    *voted_dev = ctx->bb.voted.p;
  });
}

int bk_shard_bp_set_voted(bk_ctx *ctx, const void *voted_all_dev)
{
  return guarded(ctx, [&] {
    uint32_t *v = ctx->bb.voted.as<uint32_t>(ctx->n_clusters + 1);
    if (ctx->n_clusters && voted_all_dev != v)
      HIP_CHECK(hipMemcpyAsync(v, voted_all_dev, ctx->n_clusters * 4, hipMemcpyDeviceToDevice, ctx->st));
    HIP_CHECK(hipStreamSynchronize(ctx->st));
  });
}

int bk_shard_bp_depth(bk_ctx *ctx, void **depth_dev, uint64_t *n)
{
  return guarded(ctx, [&] {
    uint32_t *d = bp_depth_partial(rec_view(ctx), ctx->clusters_ptr(), ctx->n_clusters, (int) ctx->hc.max_span, ctx->bb, ctx->st);
    HIP_CHECK(hipStreamSynchronize(ctx->st));
    if (depth_dev) *depth_dev = d;
    if (n) *n = 2 * ctx->n_clusters;
  });
}

int bk_shard_bp_finish(bk_ctx *ctx, const void *depth_total_dev)
{
  return guarded(ctx, [&] {
    bp_finish(ctx->clusters_ptr(), ctx->n_clusters, (const uint32_t *) depth_total_dev, ctx->bb, ctx->st);
    HIP_CHECK(hipStreamSynchronize(ctx->st));
  });
}

int bk_debug_std_sort(bk_ctx *ctx, const uint32_t *key, const uint64_t *group_off, uint32_t n_groups, uint32_t *perm_out)
{
  return guarded(ctx, [&] {
    if (!key || !group_off || !perm_out) throw bk_error(BK_ERR_ARG, "bk_debug_std_sort: null argument");
    const uint64_t n = group_off[n_groups];
    DevBuf dk, dp, dgof, dgoff;
    std::vector<uint32_t> gof(n), iota(n);
    for (uint32_t g = 0; g < n_groups; ++g)
      for (uint64_t p = group_off[g]; p < group_off[g + 1]; ++p) gof[p] = g;
    std::iota(iota.begin(), iota.end(), 0u);
    HIP_CHECK(hipMemcpy(dk.as<uint32_t>(n + 1), key, n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dp.as<uint32_t>(n + 1), iota.data(), n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dgof.as<uint32_t>(n + 1), gof.data(), n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dgoff.as<uint64_t>((uint64_t) n_groups + 1), group_off, ((uint64_t) n_groups + 1) * 8, hipMemcpyHostToDevice));
    uint64_t max_group = 0;
    for (uint32_t g = 0; g < n_groups; ++g) max_group = std::max<uint64_t>(max_group, group_off[g + 1] - group_off[g]);
    SvcStage svc_stage(ctx, true, n + 2ull * n_groups + 4096, max_group, {ctx->st});
    std_sort_groups(dk.get<uint32_t>(), dp.get<uint32_t>(), dgof.get<uint32_t>(), dgoff.get<uint64_t>(), n_groups, n, ctx->cb.se, ctx->st);
    svc_stage.finish();
    HIP_CHECK(hipMemcpyAsync(perm_out, dp.get<uint32_t>(), n * 4, hipMemcpyDeviceToHost, ctx->st));
    HIP_CHECK(hipStreamSynchronize(ctx->st));
  });
}

int bk_debug_ahc(bk_ctx *ctx, const uint32_t *x, const uint32_t *y, uint32_t n, double w, uint32_t *idx_out, int32_t *cluster_out, uint32_t *n_out)
{
  return guarded(ctx, [&] {
    if (!x || !y || !idx_out || !cluster_out || !n_out) throw bk_error(BK_ERR_ARG, "bk_debug_ahc: null argument");
    std::vector<bk_pair> hp(n);
    for (uint32_t i = 0; i < n; ++i)
    {
      memset(&hp[i], 0, sizeof(bk_pair));
      hp[i].x = x[i];
      hp[i].y = y[i];
    }
    DevBuf dp, dcl;
    PairList L;
    L.n = n;
    L.ng = 1;
    std::vector<uint32_t> iota(n), gof(n, 0);
    std::iota(iota.begin(), iota.end(), 0u);
    uint64_t goff[2] = {0, n};
    HIP_CHECK(hipMemcpy(dp.as<bk_pair>(n + 1), hp.data(), n * sizeof(bk_pair), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(L.idx.as<uint32_t>(n + 1), iota.data(), n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(L.gof.as<uint32_t>(n + 1), gof.data(), n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(L.goff.as<uint64_t>(2), goff, 16, hipMemcpyHostToDevice));
    ahc_cluster_all(dp.get<bk_pair>(), L, w, dcl, ctx->ab, ctx->cb, ctx->st);
    *n_out = (uint32_t) L.n;
    if (L.n)
    {
      HIP_CHECK(hipMemcpyAsync(idx_out, L.idx.get<uint32_t>(), L.n * 4, hipMemcpyDeviceToHost, ctx->st));
      HIP_CHECK(hipMemcpyAsync(cluster_out, dcl.get<uint32_t>(), L.n * 4, hipMemcpyDeviceToHost, ctx->st));
    }
    HIP_CHECK(hipStreamSynchronize(ctx->st));
  });
}

}  // extern "C"

// ---- feed and stream pass overlapped: the reference's two sequential BAM passes (BreakID.cc:1929, :1414)
// become one read of the file, and the record-level kernel of the hot path runs while the file is still arriving -------------
namespace
{
struct FeedLink
{
  int device = 0, mapq_min = 20;
  bk_ctx *ctx = nullptr;
  int rc = BK_OK;
  std::string err;
  uint64_t done = 0;      // records [0, done) have been through k_stream (a multiple of 4)
  bool prepared = false, overflow = false;
  hipEvent_t t0 = nullptr, t1 = nullptr;  // timing: first and last k_stream piece
  template <class F> void guard(F &&f)
  {
    if (rc != BK_OK) return;
    try
    {
      f();
    }
    catch (const bk_error &e)
    {
      rc = e.code;
      err = e.msg;
    }
  }
};
void link_header(void *u, int nt, const char *const *names, const uint32_t *lens)
{
  FeedLink *L = (FeedLink *) u;
  if (L->ctx || L->rc != BK_OK) return;
  L->rc = bk_init(L->device, lens, names, nt, &L->ctx);
  if (L->rc != BK_OK) L->err = bk_last_error(nullptr);
}
void link_reset(void *u)
{
  FeedLink *L = (FeedLink *) u;
  L->guard([&] {
    if (L->ctx) HIP_CHECK(hipStreamSynchronize(L->ctx->st));
    L->done = 0;
    L->prepared = false;
    L->overflow = false;
  });
}
void link_before_move(void *u)
{
  FeedLink *L = (FeedLink *) u;
  L->guard([&] {
    if (L->ctx) HIP_CHECK(hipStreamSynchronize(L->ctx->st));
  });
}
void link_chunk(void *u, const bk_soa *cols, uint64_t n_ready, uint64_t n_est, hipEvent_t ready)
{
  FeedLink *L = (FeedLink *) u;
  L->guard([&] {
    bk_ctx *c = L->ctx;
    if (!c || n_ready < (uint64_t) STREAM_V + 1) return;
    c->rec = *cols;  // device pointers of the columns as they stand now
    c->have_records = true;
    c->mapq_min = L->mapq_min;
    if (!L->prepared)
    {
      stream_prepare(c, n_est + n_est / 4);
      L->prepared = true;
    }
    // the last ready record stays for the next piece: its quad reads the offset entry that follows it
    const uint64_t lim = (n_ready - 1) / STREAM_V * STREAM_V;
    if (lim <= L->done) return;
    HIP_CHECK(hipStreamWaitEvent(c->st, ready, 0));
    StreamArgs a = stream_args(c, lim);
    a.q_begin = L->done / STREAM_V;
    launch_stream(a, c->st);
    L->done = lim;
  });
}
}  // namespace


extern "C" {

int bk_bam_decode_device_ctx(const char *path, int device, int mapq_min, bk_bam_dev **bam_out, bk_ctx **ctx_out, int *n_targets, const char *const **names,
                             const uint32_t **lens, char *err, size_t errlen)
{
  auto fail = [&](int code, const std::string &m) {
    if (err && errlen) snprintf(err, errlen, "%s", m.c_str());
    return code;
  };
  if (!path || !bam_out || !ctx_out) return fail(BK_ERR_ARG, "bk_bam_decode_device_ctx: null argument");
  *bam_out = nullptr;
  *ctx_out = nullptr;
  FeedLink L;
  L.device = device;
  L.mapq_min = mapq_min;
  FeedConsumer fc;
  fc.user = &L;
  fc.on_header = link_header;
  fc.on_chunk = link_chunk;
  fc.before_move = link_before_move;
  fc.on_reset = link_reset;
  bk_soa cols;
  bk_bam_dev *bam = nullptr;
  int rc = bam_decode_device_impl(path, device, &bam, &cols, n_targets, names, lens, err, errlen, &fc);
  if (rc == BK_OK && L.rc != BK_OK) rc = fail(L.rc, L.err);
  if (rc == BK_OK && !L.ctx) rc = fail(BK_ERR_IO, "bk_bam_decode_device_ctx: no header");
  if (rc != BK_OK)
  {
    if (L.ctx) bk_free(L.ctx);
    if (bam) bk_bam_dev_free(bam);
    return rc;
  }
  bk_ctx *c = L.ctx;
  rc = guarded(c, [&] {
    // the complete table (final pointers, end entries of the offset columns written), then the records the pieces left
    const uint64_t done = L.prepared ? L.done : 0;
    c->stream_done = c->stats_done = c->clustered = false;
    c->rec = cols;
    c->have_records = true;
    c->mapq_min = mapq_min;
    c->rec_base = 0;
    bool ok = false;
    if (L.prepared)
    {
      Scope s(c, "k_stream", 39ull * cols.n + 4ull * cols.n_cigar_words);
      StreamArgs a = stream_args(c, cols.n);
      a.q_begin = done / STREAM_V;
      launch_stream(a, c->st);
      ok = stream_finish(c);
    }
    if (ok)
      stream_rare_path(c);
    else
      run_stream(c);  // file without chunked feed (records across BGZF blocks), or an output capacity estimated too small
    if (bk_debug("feed"))
      fprintf(stderr, "[feed/stream] stream pass %s: %llu of %llu records went through k_stream while the file was still arriving\n", ok ? "overlapped" : "after the feed",
              (unsigned long long) (ok ? done : 0), (unsigned long long) cols.n);
  });
  if (rc != BK_OK)
  {
    fail(rc, c->err);
    bk_free(c);
    bk_bam_dev_free(bam);
    return rc;
  }
  *bam_out = bam;
  *ctx_out = c;
  return BK_OK;
}

int bk_group_stats(bk_ctx *ctx, const bk_group_stat **out, uint32_t *n_groups)
{
  return guarded(ctx, [&] {
    if (!out || !n_groups) throw bk_error(BK_ERR_ARG, "bk_group_stats: null output");
    if (!ctx->clustered) throw bk_error(BK_ERR_ARG, "bk_group_stats: call bk_mask_and_cluster (and bk_cluster_summary) first");
    const uint32_t ng = ctx->jr.n_groups;
    std::vector<uint64_t> iso(ng + 1, 0), clu(ng + 1, 0);
    std::vector<uint32_t> kmax(ng + 1, 0);
    if (ng)
    {
      HIP_CHECK(hipMemcpyAsync(iso.data(), ctx->iso_goff.get<uint64_t>(), ((uint64_t) ng + 1) * 8, hipMemcpyDeviceToHost, ctx->st));
      HIP_CHECK(hipMemcpyAsync(clu.data(), ctx->list.goff.get<uint64_t>(), ((uint64_t) ng + 1) * 8, hipMemcpyDeviceToHost, ctx->st));
      if (ctx->list.n && ctx->bb.kmax.p) HIP_CHECK(hipMemcpyAsync(kmax.data(), ctx->bb.kmax.get<uint32_t>(), (uint64_t) ng * 4, hipMemcpyDeviceToHost, ctx->st));
      HIP_CHECK(hipStreamSynchronize(ctx->st));
    }
    ctx->f_gstats.assign(ng, bk_group_stat{});
    for (uint32_t l = 0; l < ng; ++l)
    {
      const uint32_t g = ctx->lex_to_num[l];
      bk_group_stat &o = ctx->f_gstats[l];
      const uint32_t k = ctx->gkey_host[g];
      o.p1_tid = (int32_t) (k / (uint32_t) (ctx->nt + 1)) - 1;
      o.p2_tid = (int32_t) (k % (uint32_t) (ctx->nt + 1)) - 1;
      o.n_scan = ctx->gstart_host[g + 1] - ctx->gstart_host[g];
      o.n_isolated_removed = iso[g + 1] - iso[g];
      o.n_clustered = clu[g + 1] - clu[g];
      o.cluster_id_end = o.n_clustered ? kmax[g] : 0;
      o.ordinal = ctx->glex_host[g];
    }
    *out = ctx->f_gstats.data();
    *n_groups = ng;
  });
}

int bk_debug_points(bk_ctx *ctx, int mode, const uint32_t *x, const uint32_t *y, uint32_t n, double w, uint32_t *idx_out, int32_t *cluster_out, uint32_t *n_out)
{
  return guarded(ctx, [&] {
    if ((n && (!x || !y)) || !idx_out || !n_out || mode < 0 || mode > 2) throw bk_error(BK_ERR_ARG, "bk_debug_points: bad argument");
    std::vector<bk_pair> hp(n);
    for (uint32_t i = 0; i < n; ++i)
    {
      memset(&hp[i], 0, sizeof(bk_pair));
      hp[i].x = x[i];
      hp[i].y = y[i];
    }
    DevBuf dp, dcl, dgof, dgoff;
    PairList L;
    std::vector<uint32_t> gof(n, 0);
    uint64_t goff[2] = {0, n};
    if (n) HIP_CHECK(hipMemcpy(dp.as<bk_pair>(n + 1), hp.data(), n * sizeof(bk_pair), hipMemcpyHostToDevice));
    if (n) HIP_CHECK(hipMemcpy(dgof.as<uint32_t>(n + 1), gof.data(), n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dgoff.as<uint64_t>(2), goff, 16, hipMemcpyHostToDevice));
    const bk_pair *pairs = dp.as<bk_pair>(n + 1);
    // one group holding the points in the given order
    remove_isolated_all(pairs, dgof.get<uint32_t>(), dgoff.get<uint64_t>(), 1, mode == 1 ? n : 0, w, L, ctx->cb, ctx->st);
    if (mode != 1)
    {
      // the list as given (remove_isolated_all sized the buffers; fill identity order)
      L.n = n;
      L.ng = 1;
      std::vector<uint32_t> iota(n);
      std::iota(iota.begin(), iota.end(), 0u);
      if (n) HIP_CHECK(hipMemcpy(L.idx.as<uint32_t>(n + 1), iota.data(), n * 4, hipMemcpyHostToDevice));
      if (n) HIP_CHECK(hipMemcpy(L.gof.as<uint32_t>(n + 1), gof.data(), n * 4, hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(L.goff.as<uint64_t>(2), goff, 16, hipMemcpyHostToDevice));
      if (mode == 0)
        debug_mask_list(pairs, L, (long) w, ctx->cb, ctx->st);
      else
        fast_cluster_all(pairs, L, w, dcl, ctx->cb, ctx->st);
    }
    *n_out = (uint32_t) L.n;
    if (L.n)
    {
      HIP_CHECK(hipMemcpyAsync(idx_out, L.idx.get<uint32_t>(), L.n * 4, hipMemcpyDeviceToHost, ctx->st));
      if (mode == 2 && cluster_out) HIP_CHECK(hipMemcpyAsync(cluster_out, dcl.get<uint32_t>(), L.n * 4, hipMemcpyDeviceToHost, ctx->st));
    }
    HIP_CHECK(hipStreamSynchronize(ctx->st));
  });
}

int bk_debug_cigar(bk_ctx *ctx, uint32_t n, const uint8_t *kind, const uint32_t *c1_off, const uint8_t *c1, const uint32_t *c2_off, const uint8_t *c2, const int32_t *e,
                   int32_t *out6)
{
  return guarded(ctx, [&] {
    if (!n) return;
    if (!kind || !c1_off || !c1 || !c2_off || !c2 || !e || !out6) throw bk_error(BK_ERR_ARG, "bk_debug_cigar: null argument");
    DevBuf dk, d1o, d1, d2o, d2, de, dout;
    HIP_CHECK(hipMemcpy(dk.as<uint8_t>(n), kind, n, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d1o.as<uint32_t>(n + 1), c1_off, (n + 1) * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d1.as<uint8_t>(c1_off[n] + 16), c1, c1_off[n], hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d2o.as<uint32_t>(n + 1), c2_off, (n + 1) * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d2.as<uint8_t>(c2_off[n] + 16), c2, c2_off[n], hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(de.as<int32_t>(n), e, n * 4, hipMemcpyHostToDevice));
    int32_t *o = dout.as<int32_t>(6ull * n);
    debug_cigar(dk.get<uint8_t>(), d1o.get<uint32_t>(), d1.get<uint8_t>(), d2o.get<uint32_t>(), d2.get<uint8_t>(), de.get<int32_t>(), n, o, ctx->st);
    HIP_CHECK(hipMemcpyAsync(out6, o, 24ull * n, hipMemcpyDeviceToHost, ctx->st));
    HIP_CHECK(hipStreamSynchronize(ctx->st));
  });
}

int bk_debug_vote(bk_ctx *ctx, const bk_split *side1, uint32_t n1, const bk_split *side2, uint32_t n2, int32_t p1_tid, int32_t p2_tid, int32_t *out3)
{
  return guarded(ctx, [&] {
    if ((n1 && !side1) || (n2 && !side2) || !out3) throw bk_error(BK_ERR_ARG, "bk_debug_vote: null argument");
    // one cluster; side-1 tuples lie in its first region, side-2 tuples in its second one; two extra tuples per side that
    // match nothing keep find_sa_reads' "at least 2 evidence alignments" verdict out of the way (the vectors were taken
    // from find_bp_pair directly, BreakID.cc:577-857)
    const uint32_t pos1 = 100000, pos2 = 300000;
    std::vector<bk_split> t;
    auto put = [&](const bk_split *src, uint32_t n, int32_t tid, uint32_t pos, uint64_t salt) {
      for (uint32_t i = 0; i < n + 2; ++i)
      {
        bk_split s;
        if (i < n)
          s = src[i];
        else
        {
          memset(&s, 0, sizeof s);
          s.qhash = 0xD00D000000000000ull + salt + i;
          s.prim_chr = s.sec_chr = -7;
        }
        s.tid = tid;
        s.pos = (int32_t) pos;
        s.endpos = (int32_t) pos + 50;
        t.push_back(s);
      }
    };
    put(side1, n1, p1_tid, pos1, 0);
    put(side2, n2, p2_tid, pos2, 1000);
    // tuples must be in coordinate order
    std::stable_sort(t.begin(), t.end(), [](const bk_split &a, const bk_split &b) { return (uint32_t) a.tid != (uint32_t) b.tid ? (uint32_t) a.tid < (uint32_t) b.tid : a.pos < b.pos; });
    for (size_t i = 0; i < t.size(); ++i) t[i].rec = i;
    bk_cluster c;
    memset(&c, 0, sizeof c);
    c.p1_tid = p1_tid;
    c.p2_tid = p2_tid;
    c.p1_mean = pos1 + 10;
    c.p2_mean = pos2 + 10;
    c.p1_exact = 0xFFFFFFFFu;
    c.p2_exact = -1;
    DevBuf dt, dc, dcov;
    HIP_CHECK(hipMemcpy(dt.as<bk_split>(t.size() + 1), t.data(), t.size() * sizeof(bk_split), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dc.as<bk_cluster>(2), &c, sizeof c, hipMemcpyHostToDevice));
    const uint32_t cov[2] = {5, 5};
    HIP_CHECK(hipMemcpy(dcov.as<uint32_t>(4), cov, 8, hipMemcpyHostToDevice));
    bp_vote(dt.get<bk_split>(), t.size(), dc.get<bk_cluster>(), 1, 1000.0, 200, dcov.get<uint32_t>(), ctx->d_hdr.get<int32_t>(), ctx->bb, ctx->st);
    uint32_t voted = 0;
    HIP_CHECK(hipMemcpyAsync(&voted, ctx->bb.voted.get<uint32_t>(), 4, hipMemcpyDeviceToHost, ctx->st));
    HIP_CHECK(hipMemcpyAsync(&c, dc.get<bk_cluster>(), sizeof c, hipMemcpyDeviceToHost, ctx->st));
    HIP_CHECK(hipStreamSynchronize(ctx->st));
    out3[0] = voted ? (int32_t) c.p1_exact : -1;
    out3[1] = voted ? c.p2_exact : -1;
    out3[2] = voted ? (int32_t) c.n_sr : 0;
  });
}

int bk_debug_region(bk_ctx *ctx, int32_t tid, uint32_t start, uint32_t end, uint64_t depth_pos, bk_split *out, uint32_t cap, uint32_t *n_out, uint32_t *cov_out,
                    uint32_t *depth_out)
{
  return guarded(ctx, [&] {
    if (!ctx->stream_done) run_stream(ctx);
    ensure_splits_sorted(ctx);
    DevBuf dout, dres;
    bk_split *o = dout.as<bk_split>((uint64_t) cap + 1);
    uint32_t *res = dres.as<uint32_t>(4);
    debug_region(rec_view(ctx), ctx->d_split.get<bk_split>(), ctx->hc.n_split, tid, start, end, (int) ctx->hc.max_span, depth_pos, o, cap, res, ctx->st);
    uint32_t h[4] = {0, 0, 0, 0};
    HIP_CHECK(hipMemcpyAsync(h, res, 16, hipMemcpyDeviceToHost, ctx->st));
    HIP_CHECK(hipStreamSynchronize(ctx->st));
    const uint32_t n = h[0] < cap ? h[0] : cap;
    if (n && out) HIP_CHECK(hipMemcpy(out, o, n * sizeof(bk_split), hipMemcpyDeviceToHost));
    if (n_out) *n_out = h[0];
    if (cov_out) *cov_out = h[1];
    if (depth_out) *depth_out = h[2];
  });
}

int bk_timing_enable(bk_ctx *ctx, int on)
{
  return guarded(ctx, [&] {
    ctx->timing = on != 0;
    for (auto &t : ctx->timers)
    {
      if (t.a) (void) hipEventDestroy(t.a);
      if (t.b) (void) hipEventDestroy(t.b);
    }
    ctx->timers.clear();
  });
}

int bk_timing(bk_ctx *ctx, const char *const **names, const float **ms, const uint64_t **bytes, int *n)
{
  return guarded(ctx, [&] {
    HIP_CHECK(hipStreamSynchronize(ctx->st));
    Timing &t = ctx->tout;
    t.names.clear(); t.ms.clear(); t.bytes.clear(); t.touched.clear(); t.cnames.clear();
    for (auto &s : ctx->timers)
    {
      float v = 0;
      HIP_CHECK(hipEventElapsedTime(&v, s.a, s.b));
      t.names.push_back(s.name);
      t.ms.push_back(v);
      t.bytes.push_back(s.bytes);
      t.touched.push_back(s.touched);
    }
    for (auto &s : t.names) t.cnames.push_back(s.c_str());
    if (names) *names = t.cnames.data();
    if (ms) *ms = t.ms.data();
    if (bytes) *bytes = t.bytes.data();
    if (n) *n = (int) t.names.size();
  });
}

int bk_timing_touched(bk_ctx *ctx, const uint64_t **touched, int *n)
{
  return guarded(ctx, [&] {
    if (touched) *touched = ctx->tout.touched.data();
    if (n) *n = (int) ctx->tout.touched.size();
  });
}

}  // extern "C"
