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

__device__ __forceinline__ uint32_t gpos(const uint32_t *__restrict__ tprefix, int32_t nt, int32_t tid, int32_t pos)
{
  // combine_genome_chr_pos, util_bam.cc:57-68 (uint32 wrap; loop does not run for tid <= 0)
  uint32_t base = tid <= 0 ? 0u : tprefix[tid < nt ? tid : nt];
  return base + (uint32_t) pos;
}

// one lane per sorted position; only run starts work.  Runs are a handful of records.
__global__ __launch_bounds__(256) void k_join_pairs(const Cand *__restrict__ cand, const uint64_t *__restrict__ key, const uint32_t *__restrict__ val, uint64_t n, double w,
                                                    const uint32_t *__restrict__ tprefix, int32_t nt, bk_pair *__restrict__ out, uint64_t *__restrict__ okey,
                                                    uint32_t *__restrict__ oval, unsigned long long cap, unsigned long long *__restrict__ counter,
                                                    uint32_t *__restrict__ err)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t h = key[i];
  if (i > 0 && key[i - 1] == h) return;  // not a run start
  uint64_t e = i + 1;
  while (e < n && key[e] == h) ++e;
  const uint32_t len = (uint32_t) (e - i);
  if (len < 2) return;
  if (len > 4096)
  {
    atomicOr(err, 1u);
    return;
  }
  // visit the run in record order: repeatedly take the smallest record index above the last one
  long long last_rec = -1;
  uint32_t buffered = 0xFFFFFFFFu;  // candidate index of the buffered (first-arrived) mate
  for (uint32_t step = 0; step < len; ++step)
  {
    uint32_t best = 0xFFFFFFFFu;
    long long best_rec = 0x7fffffffffffffffLL;
    for (uint32_t k = 0; k < len; ++k)
    {
      uint32_t ci = val[i + k];
      long long r = cand[ci].rec;
      if (r > last_rec && r < best_rec)
      {
        best_rec = r;
        best = ci;
      }
    }
    last_rec = best_rec;
    if (buffered == 0xFFFFFFFFu)
    {
      buffered = best;
      continue;
    }
    const Cand b = cand[buffered], c = cand[best];
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
    unsigned long long slot = atomicAdd(counter, 1ull);
    if (slot < cap)
    {
      out[slot] = p;
      uint64_t gk = (uint64_t) (uint32_t) (p.p1_tid + 1) * (uint64_t) (nt + 1) + (uint64_t) (uint32_t) (p.p2_tid + 1);
      okey[slot] = (gk << 32) | p.rec;
      oval[slot] = (uint32_t) slot;
    }
  }
}

__global__ __launch_bounds__(256) void k_gather_pairs(const bk_pair *__restrict__ in, const uint32_t *__restrict__ perm, const uint64_t *__restrict__ key, uint64_t n,
                                                      bk_pair *__restrict__ out, uint32_t *__restrict__ gflag)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = in[perm[i]];
  gflag[i] = (i == 0 || (key[i] >> 32) != (key[i - 1] >> 32)) ? 1u : 0u;
}

// gscan = exclusive scan of gflag: group index (numeric key order) of element i is gscan[i] + gflag[i] - 1
__global__ __launch_bounds__(256) void k_group_starts(const uint32_t *__restrict__ gflag, const uint32_t *__restrict__ gscan, const uint64_t *__restrict__ key, uint64_t n,
                                                      uint64_t *__restrict__ gstart, uint32_t *__restrict__ gkey, uint32_t *__restrict__ gof)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t g = gscan[i] + gflag[i] - 1;
  gof[i] = g;
  if (gflag[i])
  {
    gstart[g] = i;
    gkey[g] = (uint32_t) (key[i] >> 32);
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
}  // namespace

void join_candidates(const Cand *cand, uint64_t n_cand, double w, const uint32_t *tprefix, int32_t nt, JoinBufs &b, hipStream_t st, JoinResult &res)
{
  res = JoinResult();
  if (n_cand > 0xFFFFFFF0ull) throw bk_error(BK_ERR_LIMIT, "more than 2^32 discordant candidates");
  unsigned long long *counter = b.counter.as<unsigned long long>(2);
  uint32_t *err = (uint32_t *) (counter + 1);
  HIP_CHECK(hipMemsetAsync(counter, 0, 16, st));
  uint64_t cap = n_cand / 2 + 1;
  bk_pair *unsorted = b.unsorted.as<bk_pair>(cap);
  uint64_t *okey = b.okey.as<uint64_t>(cap);
  uint32_t *oval = b.oval.as<uint32_t>(cap);
  if (n_cand)
  {
    uint64_t *key = b.key.as<uint64_t>(n_cand);
    uint32_t *val = b.val.as<uint32_t>(n_cand);
    hipLaunchKernelGGL(k_join_keys, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, cand, n_cand, key, val);
    uint64_t *ks;
    uint32_t *vs;
    prims::radix_sort_pairs(key, val, n_cand, 0, 64, b.radix, st, &ks, &vs);
    hipLaunchKernelGGL(k_join_pairs, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, cand, ks, vs, n_cand, w, tprefix, nt, unsorted, okey, oval,
                       (unsigned long long) cap, counter, err);
  }
  unsigned long long host[2] = {0, 0};
  HIP_CHECK(hipMemcpyAsync(host, counter, 16, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  if ((uint32_t) host[1]) throw bk_error(BK_ERR_LIMIT, "more than 4096 candidate records share one read-name hash");
  uint64_t np = host[0];
  if (np > cap) throw bk_error(BK_ERR_LIMIT, "pair capacity exceeded");  // cannot happen: one pair per two candidates
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
  prims::radix_sort_pairs(okey, oval, np, 0, 32 + gbits, b.radix, st, &ks, &vs);
  uint32_t *gflag = b.gflag.as<uint32_t>(np + 1);
  uint32_t *gscan = b.gscan.as<uint32_t>(np + 1);
  hipLaunchKernelGGL(k_gather_pairs, dim3(cdiv(np, 256)), dim3(256), 0, st, unsorted, vs, ks, np, pairs, gflag);
  prims::exclusive_scan<uint32_t>(gflag, gscan, np, b.scan_tmp, st);
  uint32_t ng = 0;
  HIP_CHECK(hipMemcpyAsync(&ng, gscan + np, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  res.n_groups = ng;
  uint64_t *gstart = b.gstart.as<uint64_t>((uint64_t) ng + 1);
  uint32_t *gkey = b.gkey.as<uint32_t>((uint64_t) ng + 1);
  hipLaunchKernelGGL(k_group_starts, dim3(cdiv(np, 256)), dim3(256), 0, st, gflag, gscan, ks, np, gstart, gkey, gof);
  res.gstart = gstart;
  res.gkey = gkey;
}

void join_assign_ids(JoinResult &res, const uint32_t *glex_dev, hipStream_t st)
{
  if (res.n_pairs == 0) return;
  hipLaunchKernelGGL(k_assign_ids, dim3(cdiv(res.n_pairs, 256)), dim3(256), 0, st, res.pairs, res.gof, res.gstart, glex_dev, res.n_pairs);
}
