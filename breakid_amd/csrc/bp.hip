// Cluster summary (findClusterBreakPointInfoSaTag, BreakID.cc:225-352) and the split-read breakpoint
// stage (findEncompassingReadsAndBreakPointInfo :390-490, find_sa_reads region rule :892-894,:1032,
// find_bp_pair :577-857, cal_single_base_depth util_bed.cc:154-192).  The reference pulls records per
// cluster through BAI region queries; here the per-read evidence tuples already exist (stream.hip) and
// a region query is a binary search on the coordinate-sorted record table with htslib's overlap
// predicate (hts.c:1963-1965: tid == T && pos < end && bam_endpos > beg).  One wavefront per cluster.
#include "bk_common.h"
#include "prims.h"
#include "bp.h"
#include <algorithm>
#include <vector>

namespace
{
// ---- summary -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_group_kmax(const uint32_t *__restrict__ gof, const uint32_t *__restrict__ cl, uint64_t n, uint32_t *__restrict__ kmax)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p < n;
  uint32_t g = live ? gof[p] : 0xFFFFFFFFu;
  uint32_t v = live ? cl[p] + 1 : 0u;
  // elements arrive grouped: a workgroup almost always holds one group -> one atomic per workgroup (the big groups own tens of
  // thousands of waves: one atomic per wave on their address was 0.6 ms), else one per wave, else one per lane
  __shared__ uint32_t s_g[4], s_v[4];
  const uint32_t g0 = __shfl(g, 0, 64);
  const bool wave_uniform = __ballot(live && g != g0) == 0ull;
  if (wave_uniform)
    for (int d = 32; d; d >>= 1) v = max(v, (uint32_t) __shfl_xor((int) v, d, 64));
  if ((threadIdx.x & 63) == 0)
  {
    s_g[threadIdx.x >> 6] = wave_uniform ? g0 : 0xFFFFFFFEu;
    s_v[threadIdx.x >> 6] = v;
  }
  __syncthreads();
  const bool block_uniform = s_g[0] == s_g[1] && s_g[1] == s_g[2] && s_g[2] == s_g[3] && s_g[0] < 0xFFFFFFFEu;
  if (block_uniform)
  {
    if (threadIdx.x == 0) atomicMax(&kmax[s_g[0]], max(max(s_v[0], s_v[1]), max(s_v[2], s_v[3])));
  }
  else if (wave_uniform)
  {
    if ((threadIdx.x & 63) == 0 && g0 != 0xFFFFFFFFu) atomicMax(&kmax[g0], v);
  }
  else if (live)
    atomicMax(&kmax[g], v);
}
// Members of a cluster are mostly neighbours in the list (fast clustering: always), so every wave first reduces its
// runs of equal slots with a segmented scan (head flags) and only the last lane of a run touches the accumulators:
// ~2 instead of 64 atomics per field and wave.
__global__ __launch_bounds__(256) void k_accumulate(const bk_pair *__restrict__ pairs, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ gof,
                                                    const uint32_t *__restrict__ cl, const uint32_t *__restrict__ slotbase, uint64_t n, ClusterAcc acc)
{
  const uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = p < n;
  uint32_t s = 0xFFFFFFFFu, cnt = 0, mn1 = 0xFFFFFFFFu, mx1 = 0, mn2 = 0xFFFFFFFFu, mx2 = 0, type = 0;
  unsigned long long s1 = 0, s2 = 0;
  if (live)
  {
    const bk_pair pr = pairs[idx[p]];
    s = slotbase[gof[p]] + cl[p];
    cnt = 1;
    s1 = pr.p1_pos;
    s2 = pr.p2_pos;
    mn1 = mx1 = pr.p1_pos;
    mn2 = mx2 = pr.p2_pos;
    if (pr.p1_tid != pr.p2_tid)
      type = BK_TYPE_DIFF_CHR;
    else
    {
      if (pr.p1_rev && !pr.p2_rev) type |= BK_TYPE_ABS_REVERSE;
      if (pr.p1_rev == pr.p2_rev) type |= BK_TYPE_SAME_ORIENT;
      if (!pr.p1_rev && pr.p2_rev) type |= BK_TYPE_DEFAULT_ORIENT;
    }
  }
  const uint32_t s_prev = __shfl_up(s, 1, 64), s_next = __shfl_down(s, 1, 64);
  bool head = lane == 0 || s_prev != s;
  const bool tail = lane == 63 || s_next != s;
  for (int d = 1; d < 64; d <<= 1)
  {
    const uint32_t o_cnt = __shfl_up(cnt, d, 64), o_mn1 = __shfl_up(mn1, d, 64), o_mx1 = __shfl_up(mx1, d, 64), o_mn2 = __shfl_up(mn2, d, 64),
                   o_mx2 = __shfl_up(mx2, d, 64), o_type = __shfl_up(type, d, 64);
    const unsigned long long o_s1 = __shfl_up(s1, d, 64), o_s2 = __shfl_up(s2, d, 64);
    const int o_head = __shfl_up((int) head, d, 64);
    if (lane >= d && !head)
    {
      cnt += o_cnt;
      s1 += o_s1;
      s2 += o_s2;
      mn1 = min(mn1, o_mn1);
      mx1 = max(mx1, o_mx1);
      mn2 = min(mn2, o_mn2);
      mx2 = max(mx2, o_mx2);
      type |= o_type;
      head = o_head != 0;
    }
  }
  if (!live || !tail) return;
  atomicAdd(&acc.n[s], cnt);
  atomicAdd(&acc.sum1[s], s1);
  atomicAdd(&acc.sum2[s], s2);
  atomicMin(&acc.min1[s], mn1);
  atomicMax(&acc.max1[s], mx1);
  atomicMin(&acc.min2[s], mn2);
  atomicMax(&acc.max2[s], mx2);
  atomicOr(&acc.type[s], type);
}
// slot -> (group, id): group found by binary search on slotbase
__global__ __launch_bounds__(256) void k_finalize(ClusterAcc acc, const uint32_t *__restrict__ slotbase, uint32_t ng, uint32_t nslots, const uint32_t *__restrict__ gkey,
                                                  const uint32_t *__restrict__ glex, int32_t nt, double w, uint32_t *__restrict__ keep, bk_cluster *__restrict__ tmp)
{
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  uint32_t k = 0;
  uint32_t n = acc.n[s];
  if (n)
  {
    uint32_t lo = 0, hi = ng;  // largest g with slotbase[g] <= s
    while (lo < hi)
    {
      uint32_t m = (lo + hi) >> 1;
      if (slotbase[m] <= s) lo = m + 1; else hi = m;
    }
    uint32_t g = lo - 1;
    bk_cluster c;
    c.group = glex[g];
    c.id = (int32_t) (s - slotbase[g]);
    c.p1_tid = (int32_t) (gkey[g] / (uint32_t) (nt + 1)) - 1;
    c.p2_tid = (int32_t) (gkey[g] % (uint32_t) (nt + 1)) - 1;
    unsigned long long m1 = (uint32_t) ((double) acc.sum1[s] / (double) n);  // :342-343
    unsigned long long m2 = (uint32_t) ((double) acc.sum2[s] / (double) n);
    c.p1_mean = (uint32_t) m1;
    c.p2_mean = (uint32_t) m2;
    c.p1_min = acc.min1[s];
    c.p1_max = acc.max1[s];
    c.p2_min = acc.min2[s];
    c.p2_max = acc.max2[s];
    c.p1_exact = 0xFFFFFFFFu;
    c.p2_exact = -1;
    c.n_drp = n;
    c.n_sr = 0;
    c.depth1 = c.depth2 = 0;
    c.type_mask = acc.type[s];
    long long dist = (long long) (m1 - m2);  // :345
    bool same = c.p1_tid == c.p2_tid;
    bool near = same && (double) dist <= 2 * w && (double) dist >= -2 * w;  // :348
    c.flags = near ? 0u : 1u;
    k = near ? 0u : 1u;
    tmp[s] = c;
  }
  keep[s] = k;
}
__global__ __launch_bounds__(256) void k_compact_clusters(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ off, uint32_t nslots, const bk_cluster *__restrict__ tmp,
                                                          bk_cluster *__restrict__ out)
{
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nslots && keep[s]) out[off[s]] = tmp[s];
}

// ---- region machinery ----------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t rec_endpos(const RecView &r, uint64_t i)
{
  uint32_t c0 = r.cigar_off[i], c1 = r.cigar_off[i + 1];
  int32_t pos = r.pos[i];
  if (!(r.flag[i] & 4) && c1 > c0)
  {
    int l = 0;
    for (uint32_t k = c0; k < c1; ++k)
    {
      uint32_t v = r.cigar[k], op = v & 15u;
      if ((0x3C1A7u >> (op << 1)) & 2u) l += (int) (v >> 4);
    }
    return pos + l;
  }
  return pos + 1;
}
// first record index with (tid,pos) >= (T,P) in coordinate order (unmapped tid=-1 sorts last).  A search over the whole table
// is ~30 dependent HBM round trips per region; with the sampled keys of every REC_SAMPLE-th record (5 MB for 620 M records:
// cache resident) the first ~20 steps stay in the cache and only the last 10 touch the record columns.
constexpr uint32_t REC_SAMPLE_SHIFT = 10;
__device__ __forceinline__ unsigned long long rec_key(int32_t tid, long long pos) { return ((unsigned long long) (uint32_t) tid << 32) | (uint32_t) (pos + 0x80000000ll); }
// Lower bound by a whole wavefront: every lookup of this file is made by all 64 lanes of a wave with the same arguments (one
// wave per cluster), and a binary search is a chain of dependent memory round trips - 30 of them for a record lookup, ~22 for
// a tuple lookup, the better part of k_bp_cov / k_bp_regions / k_bp_depth.  64 probes per round trip cut the range 65-fold:
// `less(i)` = "element i orders before the target" (monotone over [lo, hi)); returns the first index for which it is false.
template <class Less> __device__ __forceinline__ uint64_t wave_lower(uint64_t lo, uint64_t hi, Less less)
{
  const uint32_t lane = threadIdx.x & 63;
  while (hi - lo > 64)
  {
    const uint64_t width = hi - lo;
    const uint64_t p = lo + width * (lane + 1) / 65;  // lo < p < hi
    const uint32_t c = (uint32_t) __popcll(__ballot(less(p)));  // the probes that order before the target are a prefix of the lanes
    const uint64_t nlo = c ? lo + width * c / 65 + 1 : lo;
    const uint64_t nhi = c < 64 ? lo + width * (c + 1) / 65 : hi;
    lo = nlo;
    hi = nhi;
  }
  const uint64_t i = lo + lane;
  return lo + (uint64_t) __popcll(__ballot(i < hi && less(i)));
}
__device__ uint64_t rec_lower(const RecView &r, int32_t T, long long P)
{
  uint64_t lo = 0, hi = r.n;
  if (r.samp)
  {
    // samp[j] = key of record j << REC_SAMPLE_SHIFT; first sample >= target bounds the answer to one stride
    const unsigned long long want = rec_key(T, P < -0x80000000ll ? -0x80000000ll : P);
    const unsigned long long *__restrict__ samp = r.samp;
    const uint64_t a = wave_lower(0, r.n_samp, [&](uint64_t m) { return samp[m] < want; });
    lo = a ? ((a - 1) << REC_SAMPLE_SHIFT) + 1 : 0;  // record (a-1)<<shift is < target, record a<<shift is >= target
    hi = a < r.n_samp ? (a << REC_SAMPLE_SHIFT) : r.n;
  }
  const uint32_t Tu = (uint32_t) T;
  return wave_lower(lo, hi, [&](uint64_t m) {
    const uint32_t t = (uint32_t) r.tid[m];
    return t != Tu ? (t < Tu) : ((long long) r.pos[m] < P);
  });
}
__global__ __launch_bounds__(256) void k_rec_sample(const int32_t *__restrict__ tid, const int32_t *__restrict__ pos, uint64_t n_samp, unsigned long long *__restrict__ samp)
{
  const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_samp) samp[j] = rec_key(tid[j << REC_SAMPLE_SHIFT], (long long) pos[j << REC_SAMPLE_SHIFT]);
}
__device__ uint64_t split_lower(const bk_split *__restrict__ sp, uint64_t ns, uint64_t rec)
{
  uint64_t lo = 0, hi = ns;
  while (lo < hi)
  {
    uint64_t m = (lo + hi) >> 1;
    if ((uint64_t) sp[m].rec < rec) lo = m + 1; else hi = m;
  }
  return lo;
}
__device__ __forceinline__ long long wave_sum(long long v)
{
  for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

struct Region
{
  int32_t tid;
  int beg, end;
  bool valid;
};
__device__ __forceinline__ Region make_region(int32_t tid, uint32_t mean, int wi)
{
  Region r;
  uint32_t rs = (uint32_t) ((unsigned long long) mean - (unsigned long long) (long long) wi);  // :430-433 (uint64 arithmetic, uint32 result)
  uint32_t re = (uint32_t) ((unsigned long long) mean + (unsigned long long) (long long) wi);
  r.tid = tid;
  r.beg = (int) rs;  // uint32 -> int at bam_iter_query (:881)
  r.end = (int) re;
  if (r.beg < 0) r.beg = 0;  // hts.c:1776
  r.valid = !(r.end < r.beg || tid < 0);
  return r;
}
__device__ __forceinline__ bool in_region(const Region &rg, int32_t tid, int32_t pos, int32_t endpos) { return tid == rg.tid && pos < rg.end && endpos > rg.beg; }

__device__ uint64_t split_lower_pos(const bk_split *__restrict__ sp, uint64_t ns, int32_t T, long long P)
{
  // tuples are ordered by record index = coordinate order: first tuple with (tid,pos) >= (T,P)
  const uint32_t Tu = (uint32_t) T;
  return wave_lower(0, ns, [&](uint64_t m) {
    const uint32_t t = (uint32_t) sp[m].tid;
    return t != Tu ? (t < Tu) : ((long long) sp[m].pos < P);
  });
}

// number of records of THIS record table (one shard or the whole file) that overlap the region, counted up to
// COV_ENOUGH: total_coverage of find_sa_reads (:894) only enters the verdict `coverage < 5` (:1032), and
// sum_i min(c_i, 5) >= 5 <=> sum_i c_i >= 5 over the shards, so the scan stops at the first 64-record batch that
// reaches 5 (a 30x sample has ~500 records per region).
constexpr uint32_t COV_ENOUGH = 5;
__device__ uint32_t region_cov(const RecView &r, const Region &rg, int maxspan)
{
  if (!rg.valid) return 0;
  const int lane = threadIdx.x & 63;
  const uint64_t lo = rec_lower(r, rg.tid, (long long) rg.beg - maxspan);
  uint32_t cov = 0;
  for (uint64_t base = lo; base < r.n && cov < COV_ENOUGH; base += 64)
  {
    const uint64_t i = base + lane;
    bool hit = false, past = true;  // past: beyond the last record that can start inside the region
    if (i < r.n)
    {
      const int32_t t = r.tid[i], p = r.pos[i];
      past = t != rg.tid || p >= rg.end;
      hit = !past && in_region(rg, t, p, rec_endpos(r, i));
    }
    cov += (uint32_t) __popcll(__ballot(hit));
    if (__ballot(past)) break;
  }
  return cov < COV_ENOUGH ? cov : COV_ENOUGH;
}

// phase 1 (per shard): coverage counts of both regions of every cluster
__global__ __launch_bounds__(256) void k_bp_cov(RecView r, const bk_cluster *__restrict__ cl, uint32_t ncl, int wi, int maxspan, uint32_t *__restrict__ cov)
{
  const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= ncl) return;
  const bk_cluster k = cl[c];
  uint32_t c1 = region_cov(r, make_region(k.p1_tid, k.p1_mean, wi), maxspan);
  uint32_t c2 = region_cov(r, make_region(k.p2_tid, k.p2_mean, wi), maxspan);
  if ((threadIdx.x & 63) == 0)
  {
    cov[2 * c] = c1;
    cov[2 * c + 1] = c2;
  }
}

// find_sa_reads' region verdict from the (summed) coverage and the evidence tuples: coverage >= 5 and >= 2
// evidence alignments, else the map is cleared (:1032)
__device__ bool side_verdict(const bk_split *__restrict__ sp, uint64_t nsp, const Region &rg, int maxspan, uint32_t cov, uint64_t &tlo, uint64_t &thi, bool &poison)
{
  tlo = thi = 0;
  if (!rg.valid) return false;
  const int lane = threadIdx.x & 63;
  tlo = split_lower_pos(sp, nsp, rg.tid, (long long) rg.beg - maxspan);
  thi = split_lower_pos(sp, nsp, rg.tid, (long long) rg.end);
  long long ev = 0, bad = 0;
  for (uint64_t t = tlo + lane; t < thi; t += 64)
  {
    const bk_split s = sp[t];
    if (in_region(rg, s.tid, s.pos, s.endpos))
    {
      ++ev;
      if (s.flags & 2u) ++bad;
    }
  }
  ev = wave_sum(ev);
  bad = wave_sum(bad);
  if (bad) poison = true;
  return !(cov < 5 || ev < 2);
}

// the same over tuples in memory: the read-name hash decides nearly every comparison, so it is looked at first and the
// other 72 bytes of a tuple are only loaded for the pairs that share it
// bit 1 of *err: two tuples share the 64-bit read-name hash but not the second hash (BK_ERR_COLLISION)
__device__ __forceinline__ bool tuples_match_at(const bk_split *__restrict__ pa, const bk_split *__restrict__ pb, uint32_t *__restrict__ err)
{
  if (pa->qhash != pb->qhash) return false;
  const bk_split a = *pa, b = *pb;
  if (a.qcheck != b.qcheck)
  {
    if (err) atomicOr(err, 2u);
    return false;
  }
  return ((a.flags ^ b.flags) & 1u) && a.prim_chr == b.prim_chr && a.sec_chr == b.sec_chr && a.reserved == b.reserved && a.prim_start == b.prim_start && a.sec_start == b.sec_start &&
         a.prim_end == b.prim_end && a.sec_end == b.sec_end && a.prim_cigar == b.prim_cigar && a.sec_cigar == b.sec_cigar && a.prim_bp == b.prim_bp && a.sec_bp == b.sec_bp;
}

__device__ __forceinline__ bool tuples_match(const bk_split &a, const bk_split &b)
{
  return a.qhash == b.qhash && ((a.flags ^ b.flags) & 1u) && a.prim_chr == b.prim_chr && a.sec_chr == b.sec_chr && a.reserved == b.reserved && a.prim_start == b.prim_start &&
         a.sec_start == b.sec_start && a.prim_end == b.prim_end && a.sec_end == b.sec_end && a.prim_cigar == b.prim_cigar && a.sec_cigar == b.sec_cigar &&
         a.prim_bp == b.prim_bp && a.sec_bp == b.sec_bp;  // new_condition, :627-637
}

// phase 2: region verdicts + number of (i,j) matches per cluster
__global__ __launch_bounds__(256) void k_bp_regions(const bk_split *__restrict__ sp, uint64_t nsp, const bk_cluster *__restrict__ cl, uint32_t ncl, int wi, int maxspan,
                                                    const uint32_t *__restrict__ cov, BpWork *__restrict__ work, uint32_t *__restrict__ nmatch, uint32_t *__restrict__ err)
{
  const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= ncl) return;
  const bk_cluster k = cl[c];
  BpWork wk;
  wk.ok = 0;
  wk.pad = 0;
  wk.t1lo = wk.t1hi = wk.t2lo = wk.t2hi = 0;
  bool poison = false;
  Region r1 = make_region(k.p1_tid, k.p1_mean, wi), r2 = make_region(k.p2_tid, k.p2_mean, wi);
  bool ok1 = side_verdict(sp, nsp, r1, maxspan, cov[2 * c], wk.t1lo, wk.t1hi, poison);
  bool ok2 = false;
  if (ok1) ok2 = side_verdict(sp, nsp, r2, maxspan, cov[2 * c + 1], wk.t2lo, wk.t2hi, poison);  // side 2 only if side 1 is non-empty (:438)
  long long m = 0;
  if (ok1 && ok2)
  {
    wk.ok = 1;
    // all (a, b) combinations of the two tuple ranges, spread over the lanes
    const uint64_t n2 = wk.t2hi - wk.t2lo, total = (wk.t1hi - wk.t1lo) * n2;
    for (uint64_t q = lane; q < total; q += 64)
    {
      const uint64_t qi = total <= 0xFFFFFFFFull ? (uint64_t) ((uint32_t) q / (uint32_t) n2) : q / n2;
      const bk_split *pa = sp + wk.t1lo + qi, *pb = sp + wk.t2lo + (q - qi * n2);
      if (!tuples_match_at(pa, pb, err)) continue;
      if (in_region(r1, pa->tid, pa->pos, pa->endpos) && in_region(r2, pb->tid, pb->pos, pb->endpos)) ++m;
    }
    m = wave_sum(m);
  }
  if (lane == 0)
  {
    work[c] = wk;
    nmatch[c] = (uint32_t) m;
    if (poison) atomicOr(err, 1u);
  }
}

// decimal text of "p1,p2" as the reference builds it with to_string(int32)
__device__ int key_text(int32_t a, int32_t b, char *out)
{
  int n = 0;
  for (int part = 0; part < 2; ++part)
  {
    long long v = part ? b : a;
    char tmp[12];
    int t = 0;
    bool neg = v < 0;
    if (neg) v = -v;
    do
    {
      tmp[t++] = (char) ('0' + v % 10);
      v /= 10;
    } while (v);
    if (neg) out[n++] = '-';
    while (t) out[n++] = tmp[--t];
    if (!part) out[n++] = ',';
  }
  return n;
}
// std::string operator< on the two key texts
__device__ bool key_less(int32_t a1, int32_t a2, int32_t b1, int32_t b2)
{
  char sa[28], sb[28];
  int la = key_text(a1, a2, sa), lb = key_text(b1, b2, sb);
  int l = la < lb ? la : lb;
  for (int i = 0; i < l; ++i)
  {
    unsigned char x = (unsigned char) sa[i], y = (unsigned char) sb[i];
    if (x != y) return x < y;
  }
  return la < lb;
}

// The same order without building the texts: the decimal text of one int32 ("-" and up to 10 digits) as 11 four-bit codes,
// first character in the highest nibble ('-' = 2, digits = 3..12), the unused low nibbles filled with `pad`.  With pad = 1
// for the first number (the ',' that follows it: below '-' and the digits, as in ASCII) and pad = 0 for the second (end of
// text), "p1,p2" < "q1,q2" as std::string  <=>  (code(p1), code(p2)) < (code(q1), code(q2)) as integers.
__device__ __forceinline__ unsigned long long key_code(int32_t v, unsigned long long pad)
{
  const bool neg = v < 0;
  uint32_t u = neg ? (uint32_t) (-(long long) v) : (uint32_t) v;
  unsigned long long code = 0;
  int len = 0;
  do
  {
    code |= (unsigned long long) (3u + u % 10u) << (4 * len);
    u /= 10u;
    ++len;
  } while (u);
  if (neg) code |= 2ull << (4 * len++);
  const int sh = 4 * (11 - len);
  return (code << sh) | ((0x11111111111ull * pad) & ((1ull << sh) - 1ull));
}

constexpr uint32_t VOTE_LDS = 256;
// phase 3: emit (p1_bp, p2_bp) for every match and vote (find_bp_pair); voted[c] = 1 when encompass_num >= 2 (:446)
__global__ __launch_bounds__(256) void k_bp_vote(const bk_split *__restrict__ sp, bk_cluster *__restrict__ cl, uint32_t ncl, int wi, const BpWork *__restrict__ work,
                                                 const uint32_t *__restrict__ moff, int2 *__restrict__ emit, uint32_t *__restrict__ ecount,
                                                 const int32_t *__restrict__ hdr_id, uint32_t *__restrict__ voted)
{
  const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // the emitted list of a cluster with up to VOTE_LDS matches lives in LDS (no global atomics, no fence before it is
  // read back); longer lists use the global buffer
  __shared__ int2 s_emit[4][VOTE_LDS];
  __shared__ unsigned int s_ecount[4];
  if (c >= ncl) return;
  if (lane == 0) voted[c] = 0;
  // (one round trip for everything the wave needs to know about its cluster: this kernel is a chain of dependent loads
  // per wave - a cluster has ~100 tuple combinations and ~8 matches - so the chain's length is its run time)
  const BpWork wk = work[c];
  bk_cluster k = cl[c];
  const uint32_t m0 = moff[c], m1 = moff[c + 1];
  if (!wk.ok) return;
  const uint32_t K = m1 - m0;
  if (K == 0) return;  // k_bp_regions counted no match: nothing to emit, nothing to vote on
  Region r1 = make_region(k.p1_tid, k.p1_mean, wi), r2 = make_region(k.p2_tid, k.p2_mean, wi);
  const int32_t p1_chr = hdr_id[k.p1_tid + 1];
  const bool in_lds = K <= VOTE_LDS;
  const int wv = threadIdx.x >> 6;
  int2 *E = in_lds ? s_emit[wv] : emit + m0;
  if (in_lds && lane == 0) s_ecount[wv] = 0;
  __builtin_amdgcn_wave_barrier();
  // all (a, b) combinations of the two tuple ranges, spread over the lanes (a lane per `a` left most of the wave idle:
  // a cluster has a handful of tuples on either side)
  const uint64_t n1 = wk.t1hi - wk.t1lo, n2 = wk.t2hi - wk.t2lo, total = n1 * n2;
  for (uint64_t q = lane; q < total; q += 64)
  {
    const uint64_t qi = total <= 0xFFFFFFFFull ? (uint64_t) ((uint32_t) q / (uint32_t) n2) : q / n2;
    const uint64_t i = wk.t1lo + qi, j = wk.t2lo + (q - qi * n2);
    const bk_split *pa = sp + i, *pb = sp + j;
    if (!tuples_match_at(pa, pb, nullptr)) continue;
    {
      const bk_split a = *pa, b = *pb;
      if (in_region(r1, a.tid, a.pos, a.endpos) && in_region(r2, b.tid, b.pos, b.endpos))
      {
        uint32_t slot = in_lds ? atomicAdd(&s_ecount[wv], 1u) : atomicAdd(&ecount[c], 1u);
        int2 e;
        if (a.prim_chr == p1_chr)  // :647
        {
          e.x = (int32_t) a.prim_bp;
          e.y = (int32_t) a.sec_bp;
        }
        else
        {
          e.x = (int32_t) a.sec_bp;
          e.y = (int32_t) a.prim_bp;
        }
        if (slot < K) E[slot] = e;
      }
    }
  }
  // the list was written by other lanes of this wave: make it visible before it is read back
  if (in_lds)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  else
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  __builtin_amdgcn_wave_barrier();
  // vote (:804-855): for each distinct key, count emitted pairs within +-2 on both coordinates (unsigned
  // arithmetic, :820-821); first strict maximum in std::map<string> order wins
  int best_cnt = 0;
  int32_t best1 = 0, best2 = 0;
  unsigned long long bk1 = 0, bk2 = 0;  // key_code of (best1, best2): the text order of the keys as an integer order
  for (uint32_t q = lane; q < K; q += 64)
  {
    const int2 e = E[q];
    const uint32_t t1 = (uint32_t) e.x, t2 = (uint32_t) e.y;
    int cnt = 0;
    for (uint32_t m = 0; m < K; ++m)
    {
      const int2 u = E[m];
      if (((uint32_t) u.x <= t1 + 2u && (uint32_t) u.x >= t1 - 2u) && ((uint32_t) u.y <= t2 + 2u && (uint32_t) u.y >= t2 - 2u)) ++cnt;
    }
    if (cnt >= best_cnt && cnt > 0)
    {
      const unsigned long long k1 = key_code(e.x, 1ull), k2 = key_code(e.y, 0ull);
      if (cnt > best_cnt || k1 < bk1 || (k1 == bk1 && k2 < bk2))
      {
        best_cnt = cnt;
        best1 = e.x;
        best2 = e.y;
        bk1 = k1;
        bk2 = k2;
      }
    }
  }
  for (int d = 32; d; d >>= 1)
  {
    const int oc = __shfl_xor(best_cnt, d, 64);
    const int32_t o1 = __shfl_xor(best1, d, 64), o2 = __shfl_xor(best2, d, 64);
    const unsigned long long ok1 = __shfl_xor(bk1, d, 64), ok2 = __shfl_xor(bk2, d, 64);
    if (oc > best_cnt || (oc == best_cnt && oc > 0 && (ok1 < bk1 || (ok1 == bk1 && ok2 < bk2))))
    {
      best_cnt = oc;
      best1 = o1;
      best2 = o2;
      bk1 = ok1;
      bk2 = ok2;
    }
  }
  if (best_cnt >= 2 && lane == 0)  // :446
  {
    k.p1_exact = (uint32_t) best1;
    k.p2_exact = best2;
    k.n_sr = (uint32_t) best_cnt;
    cl[c] = k;
    voted[c] = 1;
  }
}

// phase 4 (per shard): cal_single_base_depth partial counts for the voted clusters
// cal_single_base_depth (util_bed.cc:154-192): records overlapping [pos-1, pos) with mapq > 0 && !DUP && PAIRED, counted by one wave;
// bam_iter_query(idx, tid, pos - 1, pos) with the reference's uint64 -> int conversions
__device__ uint32_t base_depth_wave(const RecView &r, int32_t tid, unsigned long long pos, int maxspan)
{
  int beg = (int) (pos - 1ull), end = (int) pos;
  if (beg < 0) beg = 0;
  if (end < beg || tid < 0) return 0;
  Region rg;
  rg.tid = tid;
  rg.beg = beg;
  rg.end = end;
  rg.valid = true;
  const int lane = threadIdx.x & 63;
  uint64_t lo = rec_lower(r, tid, (long long) beg - maxspan), hi = rec_lower(r, tid, (long long) end);
  long long acc = 0;
  for (uint64_t i = lo + lane; i < hi; i += 64)
  {
    if (!in_region(rg, r.tid[i], r.pos[i], rec_endpos(r, i))) continue;
    uint16_t f = r.flag[i];
    if (r.mapq[i] > 0 && !(f & 0x400) && (f & 1)) ++acc;  // util_bed.cc:183
  }
  return (uint32_t) wave_sum(acc);
}

__global__ __launch_bounds__(256) void k_bp_depth(RecView r, const bk_cluster *__restrict__ cl, uint32_t ncl, int maxspan, const uint32_t *__restrict__ voted,
                                                  uint32_t *__restrict__ depth)
{
  const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= ncl) return;
  uint32_t d1 = 0, d2 = 0;
  if (voted[c])
  {
    const bk_cluster k = cl[c];
    d1 = base_depth_wave(r, k.p1_tid, (unsigned long long) k.p1_exact, maxspan);
    d2 = base_depth_wave(r, k.p2_tid, (unsigned long long) (long long) k.p2_exact, maxspan);
  }
  if ((threadIdx.x & 63) == 0)
  {
    depth[2 * c] = d1;
    depth[2 * c + 1] = d2;
  }
}

// test hook (bk_debug_region): find_sa_reads (BreakID.cc:868-1037) on one raw region with the product's own device code:
// coverage (capped at 5), the region verdict, and the evidence tuples that survive it; out[0] = n, out[1] = cov, out[2] = depth
__global__ __launch_bounds__(64) void k_debug_region(RecView r, const bk_split *__restrict__ sp, uint64_t nsp, int32_t tid, uint32_t start, uint32_t end, int maxspan,
                                                     unsigned long long depth_pos, bk_split *__restrict__ out, uint32_t cap, uint32_t *__restrict__ res)
{
  Region rg;
  rg.tid = tid;
  rg.beg = (int) start;  // uint32 -> int at bam_iter_query (:881)
  rg.end = (int) end;
  if (rg.beg < 0) rg.beg = 0;
  rg.valid = !(rg.end < rg.beg || tid < 0);
  const int lane = threadIdx.x & 63;
  const uint32_t cov = region_cov(r, rg, maxspan);
  uint64_t tlo, thi;
  bool poison = false;
  const bool ok = side_verdict(sp, nsp, rg, maxspan, cov, tlo, thi, poison);
  const uint32_t depth = base_depth_wave(r, tid, depth_pos, maxspan);
  if (lane == 0)
  {
    uint32_t n = 0;
    if (ok)
      for (uint64_t t = tlo; t < thi; ++t)
        if (in_region(rg, sp[t].tid, sp[t].pos, sp[t].endpos))
        {
          if (n < cap) out[n] = sp[t];
          ++n;
        }
    res[0] = n;
    res[1] = cov;
    res[2] = depth;
    res[3] = poison ? 1u : 0u;
  }
}
__global__ __launch_bounds__(256) void k_bp_finish(bk_cluster *__restrict__ cl, uint32_t ncl, const uint32_t *__restrict__ voted, const uint32_t *__restrict__ depth)
{
  uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncl || !voted[c]) return;
  cl[c].depth1 = depth[2 * c];
  cl[c].depth2 = depth[2 * c + 1];
  cl[c].flags |= 2u;
}

__global__ __launch_bounds__(256) void k_split_keys(const bk_split *__restrict__ sp, uint64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ val, unsigned long long *__restrict__ max_rec)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long r = 0;
  if (i < n)
  {
    r = sp[i].rec;
    key[i] = r;
    val[i] = (uint32_t) i;
  }
  if (max_rec)  // (a sharded sample: the record indices are the whole sample's, the sort takes its passes from the largest one)
  {
    for (int d = 32; d >= 1; d >>= 1)
    {
      const unsigned long long o = __shfl_xor(r, d, 64);
      r = o > r ? o : r;
    }
    if ((threadIdx.x & 63) == 0 && r) atomicMax(max_rec, r);
  }
}
__global__ __launch_bounds__(256) void k_split_gather(const bk_split *__restrict__ in, const uint32_t *__restrict__ perm, uint64_t n, bk_split *__restrict__ out)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[perm[i]];
}
}  // namespace

static inline unsigned nb(uint64_t n) { return cdiv(n ? n : 1, 256); }

void sort_splits(bk_split *unsorted, uint64_t n, bk_split *sorted, BpBufs &b, hipStream_t st, int rec_bits)
{
  if (n == 0) return;
  uint64_t *key = b.key.as<uint64_t>(n);
  uint32_t *val = b.val.as<uint32_t>(n);
  unsigned long long *mx = nullptr;
  if (rec_bits <= 0)
  {
    mx = b.maxrec.as<unsigned long long>(2);
    HIP_CHECK(hipMemsetAsync(mx, 0, 8, st));
  }
  hipLaunchKernelGGL(k_split_keys, dim3(nb(n)), dim3(256), 0, st, unsorted, n, key, val, mx);
  if (mx)
  {
    unsigned long long h = 0;
    HIP_CHECK(hipMemcpyAsync(&h, mx, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    rec_bits = 1;
    while (rec_bits < 64 && (h >> rec_bits) != 0ull) ++rec_bits;
  }
  uint64_t *ks;
  uint32_t *vs;
  prims::radix_sort_pairs(key, val, n, 0, rec_bits, b.radix, st, &ks, &vs);
  hipLaunchKernelGGL(k_split_gather, dim3(nb(n)), dim3(256), 0, st, unsorted, vs, n, sorted);
}

uint64_t cluster_summary(const bk_pair *pairs, const uint32_t *idx, const uint32_t *gof, const uint32_t *cl, uint64_t n, uint32_t ng, const uint32_t *gkey,
                         const uint32_t *glex, int32_t nt, double w, DevBuf &clusters_out, BpBufs &b, hipStream_t st)
{
  if (n == 0 || ng == 0) return 0;
  uint32_t *kmax = b.kmax.as<uint32_t>((uint64_t) ng + 1);
  HIP_CHECK(hipMemsetAsync(kmax, 0, ((uint64_t) ng + 1) * 4, st));
  hipLaunchKernelGGL(k_group_kmax, dim3(nb(n)), dim3(256), 0, st, gof, cl, n, kmax);
  uint32_t *slotbase = b.slotbase.as<uint32_t>((uint64_t) ng + 1);
  prims::exclusive_scan<uint32_t>(kmax, slotbase, ng, b.scan_tmp, st);
  uint32_t nslots = 0;
  HIP_CHECK(hipMemcpyAsync(&nslots, slotbase + ng, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  if (nslots == 0) return 0;
  ClusterAcc acc;
  acc.n = b.an.as<uint32_t>(nslots);
  acc.sum1 = b.as1.as<unsigned long long>(nslots);
  acc.sum2 = b.as2.as<unsigned long long>(nslots);
  acc.min1 = b.amin1.as<uint32_t>(nslots);
  acc.max1 = b.amax1.as<uint32_t>(nslots);
  acc.min2 = b.amin2.as<uint32_t>(nslots);
  acc.max2 = b.amax2.as<uint32_t>(nslots);
  acc.type = b.atype.as<uint32_t>(nslots);
  HIP_CHECK(hipMemsetAsync(acc.n, 0, (size_t) nslots * 4, st));
  HIP_CHECK(hipMemsetAsync(acc.sum1, 0, (size_t) nslots * 8, st));
  HIP_CHECK(hipMemsetAsync(acc.sum2, 0, (size_t) nslots * 8, st));
  HIP_CHECK(hipMemsetAsync(acc.min1, 0xFF, (size_t) nslots * 4, st));
  HIP_CHECK(hipMemsetAsync(acc.max1, 0, (size_t) nslots * 4, st));
  HIP_CHECK(hipMemsetAsync(acc.min2, 0xFF, (size_t) nslots * 4, st));
  HIP_CHECK(hipMemsetAsync(acc.max2, 0, (size_t) nslots * 4, st));
  HIP_CHECK(hipMemsetAsync(acc.type, 0, (size_t) nslots * 4, st));
  hipLaunchKernelGGL(k_accumulate, dim3(nb(n)), dim3(256), 0, st, pairs, idx, gof, cl, slotbase, n, acc);
  uint32_t *keep = b.keep.as<uint32_t>((uint64_t) nslots + 1), *off = b.off.as<uint32_t>((uint64_t) nslots + 1);
  bk_cluster *tmp = b.tmpc.as<bk_cluster>(nslots);
  hipLaunchKernelGGL(k_finalize, dim3(nb(nslots)), dim3(256), 0, st, acc, slotbase, ng, nslots, gkey, glex, nt, w, keep, tmp);
  prims::exclusive_scan<uint32_t>(keep, off, nslots, b.scan_tmp, st);
  uint32_t nk = 0;
  HIP_CHECK(hipMemcpyAsync(&nk, off + nslots, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  bk_cluster *out = clusters_out.as<bk_cluster>((uint64_t) nk + 1);
  if (nk) hipLaunchKernelGGL(k_compact_clusters, dim3(nb(nslots)), dim3(256), 0, st, keep, off, nslots, tmp, out);
  return nk;
}

// the record view with the sampled search keys behind it (rebuilt per call: the table may have been replaced)
static RecView sampled(const RecView &r, BpBufs &b, hipStream_t st)
{
  RecView v = r;
  v.n_samp = (r.n + (1ull << REC_SAMPLE_SHIFT) - 1) >> REC_SAMPLE_SHIFT;
  v.samp = nullptr;
  if (r.n >= (64ull << REC_SAMPLE_SHIFT))
  {
    unsigned long long *sp = b.samp.as<unsigned long long>(v.n_samp + 1);
    hipLaunchKernelGGL(k_rec_sample, dim3(cdiv(v.n_samp, 256)), dim3(256), 0, st, r.tid, r.pos, v.n_samp, sp);
    v.samp = sp;
  }
  return v;
}

uint32_t *bp_cov_partial(const RecView &r0, const bk_cluster *cl, uint64_t ncl, double w, int maxspan, BpBufs &b, hipStream_t st)
{
  const RecView r = sampled(r0, b, st);
  uint32_t *cov = b.cov.as<uint32_t>(2 * ncl + 2);
  if (ncl) hipLaunchKernelGGL(k_bp_cov, dim3(cdiv(ncl, 4)), dim3(256), 0, st, r, cl, (uint32_t) ncl, (int) w, maxspan, cov);
  return cov;
}

void bp_vote(const bk_split *sp, uint64_t nsp, bk_cluster *cl, uint64_t ncl, double w, int maxspan, const uint32_t *cov, const int32_t *hdr_id, BpBufs &b, hipStream_t st)
{
  uint32_t *voted = b.voted.as<uint32_t>(ncl + 1);
  if (ncl == 0) return;
  if (ncl > 0x7FFFFFFFull) throw bk_error(BK_ERR_LIMIT, "too many clusters");
  const int wi = (int) w;  // `const int w` parameter, BreakID.cc:390
  BpWork *work = b.work.as<BpWork>(ncl);
  uint32_t *nmatch = b.nmatch.as<uint32_t>(ncl + 1), *moff = b.moff.as<uint32_t>(ncl + 1);
  uint32_t *err = b.err.as<uint32_t>(4);
  HIP_CHECK(hipMemsetAsync(err, 0, 16, st));
  hipLaunchKernelGGL(k_bp_regions, dim3(cdiv(ncl, 4)), dim3(256), 0, st, sp, nsp, cl, (uint32_t) ncl, wi, maxspan, cov, work, nmatch, err);
  prims::exclusive_scan<uint32_t>(nmatch, moff, ncl, b.scan_tmp, st);
  uint32_t host[2] = {0, 0};
  HIP_CHECK(hipMemcpyAsync(&host[0], moff + ncl, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&host[1], err, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  if (host[1] & 2u)
    throw bk_error(BK_ERR_COLLISION, "two different read names share one 64-bit name hash (their second hashes differ): the breakpoint vote would not be the reference's");
  if (host[1] & 1u) throw bk_error(BK_ERR_CIGAR, "error cigar: ");  // the reference's exit(-1), BreakID.cc:954-968
  if (bk_debug("bp"))
  {
    std::vector<uint32_t> nm(ncl);
    std::vector<BpWork> wk(ncl);
    HIP_CHECK(hipMemcpy(nm.data(), nmatch, ncl * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(wk.data(), work, ncl * sizeof(BpWork), hipMemcpyDeviceToHost));
    uint64_t okc = 0, kmax = 0, ksum = 0, pmax = 0, psum = 0, k0 = 0;
    for (uint64_t i = 0; i < ncl; ++i)
    {
      if (!wk[i].ok) continue;
      ++okc;
      const uint64_t pr = (wk[i].t1hi - wk[i].t1lo) * (wk[i].t2hi - wk[i].t2lo);
      kmax = std::max<uint64_t>(kmax, nm[i]);
      ksum += nm[i];
      pmax = std::max(pmax, pr);
      psum += pr;
      k0 += nm[i] == 0;
    }
    fprintf(stderr, "[bp] %llu clusters, %llu with both regions ok (%llu of them without a match): matches total %llu max %llu; tuple combinations total %llu max %llu\n",
            (unsigned long long) ncl, (unsigned long long) okc, (unsigned long long) k0, (unsigned long long) ksum, (unsigned long long) kmax, (unsigned long long) psum, (unsigned long long) pmax);
  }
  int2 *emit = b.emit.as<int2>((uint64_t) host[0] + 1);
  uint32_t *ecount = b.ecount.as<uint32_t>(ncl + 1);
  HIP_CHECK(hipMemsetAsync(ecount, 0, (ncl + 1) * 4, st));
  hipLaunchKernelGGL(k_bp_vote, dim3(cdiv(ncl, 4)), dim3(256), 0, st, sp, cl, (uint32_t) ncl, wi, work, moff, emit, ecount, hdr_id, voted);
}

uint32_t *bp_depth_partial(const RecView &r0, const bk_cluster *cl, uint64_t ncl, int maxspan, BpBufs &b, hipStream_t st)
{
  const RecView r = sampled(r0, b, st);
  uint32_t *depth = b.depth.as<uint32_t>(2 * ncl + 2);
  if (ncl) hipLaunchKernelGGL(k_bp_depth, dim3(cdiv(ncl, 4)), dim3(256), 0, st, r, cl, (uint32_t) ncl, maxspan, b.voted.get<uint32_t>(), depth);
  return depth;
}

void bp_finish(bk_cluster *cl, uint64_t ncl, const uint32_t *depth, BpBufs &b, hipStream_t st)
{
  if (ncl) hipLaunchKernelGGL(k_bp_finish, dim3(cdiv(ncl, 256)), dim3(256), 0, st, cl, (uint32_t) ncl, b.voted.get<uint32_t>(), depth);
}

// number of clusters that carry a voted breakpoint pair (flags bit 1), counted on the device
__global__ __launch_bounds__(256) void k_count_valid(const bk_cluster *__restrict__ cl, uint32_t ncl, unsigned long long *__restrict__ out)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const bool v = c < ncl && (cl[c].flags & 2u);
  const unsigned long long b = __ballot(v);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, (unsigned long long) __popcll(b));
}
uint64_t count_valid_clusters(const bk_cluster *cl, uint64_t ncl, BpBufs &b, hipStream_t st)
{
  if (ncl == 0) return 0;
  unsigned long long *d = b.nvalid.as<unsigned long long>(1), h = 0;
  HIP_CHECK(hipMemsetAsync(d, 0, 8, st));
  hipLaunchKernelGGL(k_count_valid, dim3(cdiv(ncl, 256)), dim3(256), 0, st, cl, (uint32_t) ncl, d);
  HIP_CHECK(hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return h;
}

// single table: all four phases back to back (the counts need no exchange)
void split_breakpoints(const RecView &r, const bk_split *sp, uint64_t nsp, bk_cluster *cl, uint64_t ncl, double w, int maxspan, const int32_t *hdr_id, BpBufs &b,
                       hipStream_t st)
{
  if (ncl == 0) return;
  const uint32_t *cov = bp_cov_partial(r, cl, ncl, w, maxspan, b, st);
  bp_vote(sp, nsp, cl, ncl, w, maxspan, cov, hdr_id, b, st);
  const uint32_t *depth = bp_depth_partial(r, cl, ncl, maxspan, b, st);
  bp_finish(cl, ncl, depth, b, st);
}

void debug_region(const RecView &r, const bk_split *sp, uint64_t nsp, int32_t tid, uint32_t start, uint32_t end, int maxspan, unsigned long long depth_pos, bk_split *out,
                  uint32_t cap, uint32_t *res, hipStream_t st)
{
  hipLaunchKernelGGL(k_debug_region, dim3(1), dim3(64), 0, st, r, sp, nsp, tid, start, end, maxspan, depth_pos, out, cap, res);
}
