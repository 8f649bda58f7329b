// Mate join by read-name hash and grouping by chromosome pair (scan_discordant_pairs,
// BreakID.cc:1424-1512).  The reference walks the BAM once, keeps the first record of a qname in a
// std::map, pairs the second with it and erases the entry, so a third record of that name is buffered
// again.  Here: radix sort the candidates by qhash, order each equal-hash run by record
// index, pair run elements (0,1), (2,3), ...; then sort the pairs by (chr-pair, discovery index).
#include "bk_common.h"
#include "prims.h"
#include "join.h"

namespace
{
__global__ __launch_bounds__(256) void k_join_keys(const Cand *__restrict__ c, uint64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
  {
    key[i] = c[i].qhash;
    val[i] = (uint32_t) i;
  }
}

__global__ __launch_bounds__(256) void k_join_rec_keys(const Cand *__restrict__ c, uint64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
  {
    key[i] = c[i].rec;
    val[i] = (uint32_t) i;
  }
}
__global__ __launch_bounds__(256) void k_join_hash_of(const Cand *__restrict__ c, const uint32_t *__restrict__ order, uint64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
  {
    key[i] = c[order[i]].qhash;
    val[i] = order[i];
  }
}

__device__ __forceinline__ uint32_t gpos(const uint32_t *__restrict__ tprefix, int32_t nt, int32_t tid, int32_t pos)
{
  // combine_genome_chr_pos, util_bam.cc:57-68 (uint32 wrap; loop does not run for tid <= 0)
  uint32_t base = tid <= 0 ? 0u : tprefix[tid < nt ? tid : nt];
  return base + (uint32_t) pos;
}

// the mate join of the run of equal read-name hashes that starts at sorted position i (nothing when i is not a run
// start); append(pair) takes every discordant pair, in the reference's arrival order
// ordered: the candidates of a run already stand in record order (the fallback for runs of any length: sorted by record index
// first, then stably by the name hash), so the run is walked as it lies
template <class F> __device__ __forceinline__ void join_run(const Cand *__restrict__ cand, const uint64_t *__restrict__ key, const uint32_t *__restrict__ val, uint64_t n, double w,
                                                            const uint32_t *__restrict__ tprefix, int32_t nt, uint32_t *__restrict__ err, uint64_t i, F &&append, bool ordered = false)
{
  if (i >= n) return;
  const uint64_t h = key[i];
  if (i > 0 && key[i - 1] == h) return;  // not a run start
  uint64_t e = i + 1;
  while (e < n && key[e] == h) ++e;
  const uint32_t len = (uint32_t) (e - i);
  if (len < 2) return;
  if (len > 4096 && !ordered)
  {
    atomicOr(err, 1u);  // the selection below is quadratic in the run: the caller sorts the runs and comes back
    return;
  }
  // visit the run in record order: repeatedly take the smallest record index above the last one
  // (a run of two, the usual case, is one comparison of the two records already in registers)
  Cand r0 = {}, r1 = {};
  if (len == 2)
  {
    r0 = cand[val[i]];
    r1 = cand[val[i + 1]];
    // same 64-bit read-name hash, different second hash: two different names collided (the reference compares the strings,
    // BreakID.cc:1424) - the run ends with BK_ERR_COLLISION rather than with a pair the reference would not form
    if (r0.qcheck != r1.qcheck) atomicOr(err, 2u);
    if (r1.rec < r0.rec)
    {
      const Cand t = r0;
      r0 = r1;
      r1 = t;
    }
  }
  long long last_rec = -1;
  uint32_t buffered = 0xFFFFFFFFu;  // candidate index of the buffered (first-arrived) mate
  for (uint32_t step = 0; step < len; ++step)
  {
    uint32_t best = 0xFFFFFFFFu;
    if (len == 2)
      best = step;  // marker only: r0 / r1 carry the data
    else if (ordered)
    {
      best = val[i + step];
      if (cand[best].qcheck != cand[val[i]].qcheck) atomicOr(err, 2u);
    }
    else
    {
      long long best_rec = 0x7fffffffffffffffLL;
      for (uint32_t k = 0; k < len; ++k)
      {
        uint32_t ci = val[i + k];
        if (step == 0 && cand[ci].qcheck != cand[val[i]].qcheck) atomicOr(err, 2u);
        long long r = cand[ci].rec;
        if (r > last_rec && r < best_rec)
        {
          best_rec = r;
          best = ci;
        }
      }
      last_rec = best_rec;
    }
    if (buffered == 0xFFFFFFFFu)
    {
      buffered = best;
      continue;
    }
    const Cand b = len == 2 ? r0 : cand[buffered], c = len == 2 ? r1 : cand[best];
    buffered = 0xFFFFFFFFu;  // readname_2_alignment.erase(it_mpr)
    // :1428  rname differs || abs(pos_cur - pos_buf) >= w   (positions are 1-based there; the difference is the same)
    int32_t bt = b.tid < 0 ? -1 : b.tid, ct = c.tid < 0 ? -1 : c.tid;
    long long dp = (long long) c.pos - (long long) b.pos;
    if (dp < 0) dp = -dp;
    if (!(bt != ct || (double) dp >= w)) continue;
    uint32_t c1 = gpos(tprefix, nt, c.tid, c.pos);
    uint32_t c2 = gpos(tprefix, nt, c.mtid, c.mpos);
    bk_pair p;
    if (c1 <= c2)
    {
      p.p1_flag = c.flag; p.p1_tid = ct; p.p1_pos = (uint32_t) ((long long) c.pos + 1); p.p1_mapq = c.mapq;
      p.x = c1; p.y = c2;
      p.p2_flag = b.flag; p.p2_tid = bt; p.p2_pos = (uint32_t) ((long long) b.pos + 1); p.p2_mapq = b.mapq;
    }
    else
    {
      p.p2_flag = c.flag; p.p2_tid = ct; p.p2_pos = (uint32_t) ((long long) c.pos + 1); p.p2_mapq = c.mapq;
      p.x = c2; p.y = c1;
      p.p1_flag = b.flag; p.p1_tid = bt; p.p1_pos = (uint32_t) ((long long) b.pos + 1); p.p1_mapq = b.mapq;
    }
    p.p1_rev = (p.p1_flag & 0x10) ? 1 : 0;
    p.p2_rev = (p.p2_flag & 0x10) ? 1 : 0;
    p.rec = c.rec;
    p.id = 0;
    p.cluster = -1;
    p.group = 0;
    p.reserved = 0;
    append(p);
  }
}


// one lane per sorted position; only run starts work.  Runs are a handful of records.
__global__ __launch_bounds__(256) void k_join_pairs(const Cand *__restrict__ cand, const uint64_t *__restrict__ key, const uint32_t *__restrict__ val, uint64_t n, double w,
                                                    const uint32_t *__restrict__ tprefix, int32_t nt, bk_pair *__restrict__ out, uint64_t *__restrict__ okey,
                                                    uint32_t *__restrict__ oval, unsigned long long cap, unsigned long long *__restrict__ counter,
                                                    uint32_t *__restrict__ err, int rbits, int ordered)
{
  // Pairs are appended through ONE counter: a returning atomic per wave tops out near 90 per microsecond on this part
  // (15 M pairs from 480 K waves: 5.3 ms, the whole kernel).  The first pair of every lane is therefore counted in LDS
  // and the workgroup takes its slots with a single global atomic; further pairs of a long run (rare) append directly.
  __shared__ unsigned int s_cnt;
  __shared__ unsigned long long s_base;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  bool have = false;
  bk_pair first_pair;
  auto append = [&](const bk_pair &p) {
    if (!have)
    {
      have = true;
      first_pair = p;
      return;
    }
    unsigned long long slot = atomicAdd(counter, 1ull);
    if (slot < cap)
    {
      out[slot] = p;
      uint64_t gk = (uint64_t) (uint32_t) (p.p1_tid + 1) * (uint64_t) (nt + 1) + (uint64_t) (uint32_t) (p.p2_tid + 1);
      okey[slot] = (gk << rbits) | p.rec;
      oval[slot] = (uint32_t) slot;
    }
  };
  const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  join_run(cand, key, val, n, w, tprefix, nt, err, i, append, ordered != 0);
  unsigned int my = 0;
  if (have) my = atomicAdd(&s_cnt, 1u);
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) s_base = atomicAdd(counter, (unsigned long long) s_cnt);
  __syncthreads();
  if (have)
  {
    const unsigned long long slot = s_base + my;
    if (slot < cap)
    {
      const bk_pair &p = first_pair;
      out[slot] = p;
      uint64_t gk = (uint64_t) (uint32_t) (p.p1_tid + 1) * (uint64_t) (nt + 1) + (uint64_t) (uint32_t) (p.p2_tid + 1);
      okey[slot] = (gk << rbits) | p.rec;
      oval[slot] = (uint32_t) slot;
    }
  }
}

__global__ __launch_bounds__(256) void k_gather_pairs(const bk_pair *__restrict__ in, const uint32_t *__restrict__ perm, const uint64_t *__restrict__ key, uint64_t n,
                                                      bk_pair *__restrict__ out, uint32_t *__restrict__ gflag, int rbits)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = in[perm[i]];
  gflag[i] = (i == 0 || (key[i] >> rbits) != (key[i - 1] >> rbits)) ? 1u : 0u;
}

// gscan = exclusive scan of gflag: group index (numeric key order) of element i is gscan[i] + gflag[i] - 1
__global__ __launch_bounds__(256) void k_group_starts(const uint32_t *__restrict__ gflag, const uint32_t *__restrict__ gscan, const uint64_t *__restrict__ key, uint64_t n,
                                                      uint64_t *__restrict__ gstart, uint32_t *__restrict__ gkey, uint32_t *__restrict__ gof, int rbits)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t g = gscan[i] + gflag[i] - 1;
  gof[i] = g;
  if (gflag[i])
  {
    gstart[g] = i;
    gkey[g] = (uint32_t) (key[i] >> rbits);
  }
  if (i == n - 1) gstart[g + 1] = n;
}

__global__ __launch_bounds__(256) void k_assign_ids(bk_pair *__restrict__ pairs, const uint32_t *__restrict__ gof, const uint64_t *__restrict__ gstart,
                                                    const uint32_t *__restrict__ glex, uint64_t n)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t g = gof[i];
  pairs[i].group = glex[g];
  pairs[i].id = (uint32_t) (i - gstart[g]);
}
// the largest discovery index among candidates / pairs (a sharded sample: the indices are the whole sample's, the table at hand does
// not tell how many bits they need)
__global__ __launch_bounds__(256) void k_max_rec_cand(const Cand *__restrict__ c, uint64_t n, unsigned long long *__restrict__ out)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long v = i < n ? (unsigned long long) c[i].rec : 0ull;
  for (int d = 32; d >= 1; d >>= 1)
  {
    const unsigned long long o = __shfl_xor(v, d, 64);
    v = o > v ? o : v;
  }
  if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}
__global__ __launch_bounds__(256) void k_max_rec_pair(const bk_pair *__restrict__ p, uint64_t n, unsigned long long *__restrict__ out)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long v = i < n ? (unsigned long long) p[i].rec : 0ull;
  for (int d = 32; d >= 1; d >>= 1)
  {
    const unsigned long long o = __shfl_xor(v, d, 64);
    v = o > v ? o : v;
  }
  if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}
// sort key of a pair: (numeric chr-pair key, discovery index)
__global__ __launch_bounds__(256) void k_pair_keys(const bk_pair *__restrict__ p, uint64_t n, int32_t nt, uint64_t *__restrict__ okey, uint32_t *__restrict__ oval, int rbits)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t gk = (uint64_t) (uint32_t) (p[i].p1_tid + 1) * (uint64_t) (nt + 1) + (uint64_t) (uint32_t) (p[i].p2_tid + 1);
  okey[i] = (gk << rbits) | p[i].rec;
  oval[i] = (uint32_t) i;
}
// dst[off[g] + rank inside group g] = src: groups leave for their destination rank as contiguous blocks
__global__ __launch_bounds__(256) void k_route_pairs(const bk_pair *__restrict__ src, const uint32_t *__restrict__ gof, const uint64_t *__restrict__ gstart,
                                                     const uint64_t *__restrict__ off, uint64_t n, bk_pair *__restrict__ dst)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = gof[i];
  dst[off[g] + (i - gstart[g])] = src[i];
}
__global__ __launch_bounds__(256) void k_cand_dest(const Cand *__restrict__ c, uint64_t n, uint32_t world, uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  key[i] = (uint32_t) ((c[i].qhash >> 17) % world);
  val[i] = (uint32_t) i;
}
// first position of every destination in the sorted key array (destinations that receive nothing stay unset)
__global__ __launch_bounds__(256) void k_dest_starts(const uint64_t *__restrict__ key, uint64_t n, unsigned long long *__restrict__ starts)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == 0 || key[i] != key[i - 1]) starts[key[i]] = i;
}
__global__ __launch_bounds__(256) void k_gather_cand(const Cand *__restrict__ in, const uint32_t *__restrict__ perm, uint64_t n, Cand *__restrict__ out)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[perm[i]];
}
}  // namespace

// mate join proper: candidates -> pairs in no particular order (b.unsorted, keys in b.okey / b.oval)
// The mate join only needs candidates of one read name NEXT to each other (join_run visits a run in record order whatever its
// order in memory), so the candidates are sorted by the upper half of the name hash only (four radix passes instead of eight)
// and the few runs of equal upper halves that hold more than one name (~n^2 / 2^33 of them) are put in order by the full hash
// here, one thread per run.  A mixed run of more than JOIN_FIX_MAX candidates sets bit 2 of err: the host sorts by all 64 bits.
constexpr uint32_t JOIN_FIX_MAX = 64;
__global__ __launch_bounds__(256) void k_join_fix_runs(uint64_t *__restrict__ key, uint32_t *__restrict__ val, uint64_t n, uint32_t *__restrict__ err)
{
  const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k0 = key[i];
  const uint32_t hi = (uint32_t) (k0 >> 32);
  if (i > 0 && (uint32_t) (key[i - 1] >> 32) == hi) return;  // not the start of a run of equal upper halves
  uint64_t e = i + 1;
  bool mixed = false;
  while (e < n)
  {
    const uint64_t k = key[e];
    if ((uint32_t) (k >> 32) != hi) break;
    mixed |= k != k0;
    ++e;
  }
  if (!mixed) return;
  const uint64_t len = e - i;
  if (len > JOIN_FIX_MAX)
  {
    atomicOr(err, 4u);
    return;
  }
  for (uint64_t a = 1; a < len; ++a)
  {
    const uint64_t kk = key[i + a];
    const uint32_t vv = val[i + a];
    uint64_t b = a;
    while (b > 0 && key[i + b - 1] > kk)
    {
      key[i + b] = key[i + b - 1];
      val[i + b] = val[i + b - 1];
      --b;
    }
    key[i + b] = kk;
    val[i + b] = vv;
  }
}

// bits of the discovery index in the pair sort key: what the caller knows (its own table), or - rec_bits < 0, a sharded sample -
// what the largest index at hand needs; together with the chr-pair key it must fit the 64-bit sort key
template <class T, class K> static int effective_rec_bits(const T *items, uint64_t n, K kernel, int32_t nt, JoinBufs &b, hipStream_t st)
{
  int bits = b.rec_bits;
  if (bits < 0)
  {
    unsigned long long *m = b.maxrec.as<unsigned long long>(1), h = 0;
    HIP_CHECK(hipMemsetAsync(m, 0, 8, st));
    if (n) hipLaunchKernelGGL(kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, items, n, m);
    HIP_CHECK(hipMemcpyAsync(&h, m, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    bits = 1;
    while (bits < 63 && (1ull << bits) <= h) ++bits;
  }
  int gbits = 1;
  while ((1ull << gbits) < (uint64_t) (nt + 1) * (uint64_t) (nt + 1) && gbits < 32) ++gbits;
  if (bits + gbits > 64) throw bk_error(BK_ERR_LIMIT, "pair sort key: chromosome-pair key and record index need more than 64 bits");
  b.rec_bits_eff = bits;
  return bits;
}

static uint64_t join_raw_pairs(const Cand *cand, uint64_t n_cand, double w, const uint32_t *tprefix, int32_t nt, JoinBufs &b, hipStream_t st)
{
  if (n_cand > 0xFFFFFFF0ull) throw bk_error(BK_ERR_LIMIT, "more than 2^32 discordant candidates");
  const int rec_bits = effective_rec_bits(cand, n_cand, k_max_rec_cand, nt, b, st);
  unsigned long long *counter = b.counter.as<unsigned long long>(2);
  uint32_t *err = (uint32_t *) (counter + 1);
  HIP_CHECK(hipMemsetAsync(counter, 0, 16, st));
  uint64_t cap = n_cand / 2 + 1;
  bk_pair *unsorted = b.unsorted.as<bk_pair>(cap);
  uint64_t *okey = b.okey.as<uint64_t>(cap);
  uint32_t *oval = b.oval.as<uint32_t>(cap);
  unsigned long long host[2] = {0, 0};
  // BK_JOIN_ATTEMPT=1 / 2 starts at the second / third of the attempts below (tests: the fallbacks are otherwise only reached by
  // read names that hash alike in their upper halves, or that more than 4096 candidates share)
  static const int first_attempt = getenv("BK_JOIN_ATTEMPT") ? std::max(0, std::min(2, atoi(getenv("BK_JOIN_ATTEMPT")))) : 0;
  // attempt 0: candidates sorted by the upper half of the name hash, mixed runs put right; 1: by the whole hash; 2: by record index
  // and then stably by the whole hash, so that a run of ANY length is walked in record order as it lies (a read name that more than
  // 4096 candidates share: the selection of the next record inside a run is quadratic and gives up there)
  for (int attempt = first_attempt; attempt < 3; ++attempt)
  {
    if (n_cand)
    {
      uint64_t *key = b.key.as<uint64_t>(n_cand);
      uint32_t *val = b.val.as<uint32_t>(n_cand);
      uint64_t *ks;
      uint32_t *vs;
      if (attempt < 2)
      {
        hipLaunchKernelGGL(k_join_keys, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, cand, n_cand, key, val);
        prims::radix_sort_pairs(key, val, n_cand, attempt == 0 ? 32 : 0, 64, b.radix, st, &ks, &vs);
        if (attempt == 0) hipLaunchKernelGGL(k_join_fix_runs, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, ks, vs, n_cand, err);
      }
      else
      {
        hipLaunchKernelGGL(k_join_rec_keys, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, cand, n_cand, key, val);
        prims::radix_sort_pairs(key, val, n_cand, 0, rec_bits, b.radix, st, &ks, &vs);
        uint64_t *key2 = b.key2.as<uint64_t>(n_cand);
        uint32_t *val2 = b.val2.as<uint32_t>(n_cand);
        hipLaunchKernelGGL(k_join_hash_of, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, cand, (const uint32_t *) vs, n_cand, key2, val2);
        prims::radix_sort_pairs(key2, val2, n_cand, 0, 64, b.radix, st, &ks, &vs);
      }
      hipLaunchKernelGGL(k_join_pairs, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, cand, ks, vs, n_cand, w, tprefix, nt, unsorted, okey, oval,
                         (unsigned long long) cap, counter, err, rec_bits, attempt == 2 ? 1 : 0);
    }
    HIP_CHECK(hipMemcpyAsync(host, counter, 16, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (attempt == 0 && ((uint32_t) host[1] & 4u))
    {
      HIP_CHECK(hipMemsetAsync(counter, 0, 16, st));  // a long run of equal upper halves with several names in it: once more, on all 64 bits
      continue;
    }
    if (attempt < 2 && ((uint32_t) host[1] & 1u) && !((uint32_t) host[1] & 2u))
    {
      HIP_CHECK(hipMemsetAsync(counter, 0, 16, st));  // a run of more than 4096 candidates: once more, every run in record order
      attempt = 1;
      continue;
    }
    break;
  }
  if ((uint32_t) host[1] & 2u)
    throw bk_error(BK_ERR_COLLISION, "two different read names share one 64-bit name hash (their second hashes differ): the mate join would not be the reference's");
  if ((uint32_t) host[1] & 1u) throw bk_error(BK_ERR_HIP, "mate join: a run was refused although its candidates were sorted (internal error)");
  if (host[0] > cap) throw bk_error(BK_ERR_LIMIT, "pair capacity exceeded");  // cannot happen: one pair per two candidates
  return host[0];
}

// pairs in any order (keys already in okey / oval) -> table sorted by (chr-pair key, discovery index) + group index
static void group_sorted(const bk_pair *unsorted, uint64_t *okey, uint32_t *oval, uint64_t np, int32_t nt, JoinBufs &b, hipStream_t st, JoinResult &res)
{
  res = JoinResult();
  res.n_pairs = np;
  bk_pair *pairs = b.pairs.as<bk_pair>(np + 1);
  uint32_t *gof = b.gof.as<uint32_t>(np + 1);
  res.pairs = pairs;
  res.gof = gof;
  if (np == 0) return;
  int gbits = 1;
  while ((1ull << gbits) < (uint64_t) (nt + 1) * (uint64_t) (nt + 1) && gbits < 32) ++gbits;
  uint64_t *ks;
  uint32_t *vs;
  const int rbits = b.rec_bits_eff;
  prims::radix_sort_pairs(okey, oval, np, 0, rbits + gbits, b.radix, st, &ks, &vs);
  uint32_t *gflag = b.gflag.as<uint32_t>(np + 1);
  uint32_t *gscan = b.gscan.as<uint32_t>(np + 1);
  hipLaunchKernelGGL(k_gather_pairs, dim3(cdiv(np, 256)), dim3(256), 0, st, unsorted, vs, ks, np, pairs, gflag, rbits);
  prims::exclusive_scan<uint32_t>(gflag, gscan, np, b.scan_tmp, st);
  uint32_t ng = 0;
  HIP_CHECK(hipMemcpyAsync(&ng, gscan + np, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  res.n_groups = ng;
  uint64_t *gstart = b.gstart.as<uint64_t>((uint64_t) ng + 1);
  uint32_t *gkey = b.gkey.as<uint32_t>((uint64_t) ng + 1);
  hipLaunchKernelGGL(k_group_starts, dim3(cdiv(np, 256)), dim3(256), 0, st, gflag, gscan, ks, np, gstart, gkey, gof, rbits);
  res.gstart = gstart;
  res.gkey = gkey;
}

void join_candidates(const Cand *cand, uint64_t n_cand, double w, const uint32_t *tprefix, int32_t nt, JoinBufs &b, hipStream_t st, JoinResult &res)
{
  const uint64_t np = join_raw_pairs(cand, n_cand, w, tprefix, nt, b, st);
  group_sorted(b.unsorted.get<bk_pair>(), b.okey.get<uint64_t>(), b.oval.get<uint32_t>(), np, nt, b, st, res);
}

void group_pairs(const bk_pair *raw, uint64_t np, int32_t nt, JoinBufs &b, hipStream_t st, JoinResult &res)
{
  if (np > 0xFFFFFFF0ull) throw bk_error(BK_ERR_LIMIT, "more than 2^32 pairs");
  uint64_t *okey = b.okey.as<uint64_t>(np + 1);
  uint32_t *oval = b.oval.as<uint32_t>(np + 1);
  const int rec_bits = effective_rec_bits(raw, np, k_max_rec_pair, nt, b, st);
  if (np) hipLaunchKernelGGL(k_pair_keys, dim3(cdiv(np, 256)), dim3(256), 0, st, raw, np, nt, okey, oval, rec_bits);
  group_sorted(raw, okey, oval, np, nt, b, st, res);
}

Cand *route_candidates(const Cand *cand, uint64_t n, uint32_t world, JoinBufs &b, hipStream_t st, std::vector<uint64_t> &counts)
{
  counts.assign(world, 0);
  Cand *out = b.route_cand.as<Cand>(n + 1);
  if (n == 0) return out;
  if (n > 0xFFFFFFF0ull) throw bk_error(BK_ERR_LIMIT, "more than 2^32 discordant candidates");
  unsigned long long *dc = b.route_cnt.as<unsigned long long>(world);
  HIP_CHECK(hipMemsetAsync(dc, 0xFF, (size_t) world * 8, st));
  uint64_t *key = b.key.as<uint64_t>(n);
  uint32_t *val = b.val.as<uint32_t>(n);
  hipLaunchKernelGGL(k_cand_dest, dim3(cdiv(n, 256)), dim3(256), 0, st, cand, n, world, key, val);
  int bits = 1;
  while ((1u << bits) < world && bits < 31) ++bits;
  uint64_t *ks;
  uint32_t *vs;
  prims::radix_sort_pairs(key, val, n, 0, bits, b.radix, st, &ks, &vs);
  hipLaunchKernelGGL(k_gather_cand, dim3(cdiv(n, 256)), dim3(256), 0, st, cand, vs, n, out);
  hipLaunchKernelGGL(k_dest_starts, dim3(cdiv(n, 256)), dim3(256), 0, st, ks, n, dc);
  std::vector<uint64_t> starts(world + 1, n);
  HIP_CHECK(hipMemcpyAsync(starts.data(), dc, (size_t) world * 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  for (uint32_t d = world; d-- > 0;)
    if (starts[d] == ~0ull) starts[d] = starts[d + 1];  // nothing for rank d
  for (uint32_t d = 0; d < world; ++d) counts[d] = starts[d + 1] - starts[d];
  return out;
}

bk_pair *route_pairs(const JoinResult &jr, const std::vector<uint64_t> &off_of_group, JoinBufs &b, hipStream_t st)
{
  bk_pair *out = b.route_pairs.as<bk_pair>(jr.n_pairs + 1);
  if (jr.n_pairs == 0) return out;
  uint64_t *doff = b.route_off.as<uint64_t>((uint64_t) jr.n_groups + 1);
  HIP_CHECK(hipMemcpyAsync(doff, off_of_group.data(), (size_t) jr.n_groups * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_route_pairs, dim3(cdiv(jr.n_pairs, 256)), dim3(256), 0, st, jr.pairs, jr.gof, jr.gstart, doff, jr.n_pairs, out);
  HIP_CHECK(hipStreamSynchronize(st));
  return out;
}

void join_assign_ids(JoinResult &res, const uint32_t *glex_dev, hipStream_t st)
{
  if (res.n_pairs == 0) return;
  hipLaunchKernelGGL(k_assign_ids, dim3(cdiv(res.n_pairs, 256)), dim3(256), 0, st, res.pairs, res.gof, res.gstart, glex_dev, res.n_pairs);
}
