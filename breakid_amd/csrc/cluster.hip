// Isolated-pair masking (remove_isolated_pairs, BreakID.cc:1271-1285 + mask_pairs_chr_pos :1813-1877)
// and the -fast window clustering (find_cluster_pairs_enspan_fast, :1046-1160) for all groups at once.
// Element lists are index lists into the pair table; every std::sort of the reference goes through the
// exact introsort emulation of sortemu.hip.
#include "bk_common.h"
#include "prims.h"
#include "sortemu.h"
#include "cluster.h"

namespace
{
constexpr uint32_t END = 0xFFFFFFFFu;

__global__ __launch_bounds__(256) void k_iota(uint32_t *__restrict__ a, uint64_t n)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = (uint32_t) i;
}
// key[p] = x or y of the pair behind element p
__global__ __launch_bounds__(256) void k_gather_key(const bk_pair *__restrict__ pairs, const uint32_t *__restrict__ idx, uint64_t n, int use_y, uint32_t *__restrict__ key)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) key[p] = use_y ? pairs[idx[p]].y : pairs[idx[p]].x;
}
__global__ __launch_bounds__(256) void k_gather_u32(const uint32_t *__restrict__ src, const uint32_t *__restrict__ perm, uint64_t n, uint32_t *__restrict__ dst)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) dst[p] = src[perm[p]];
}

__device__ __forceinline__ long absdiff(uint32_t a, uint32_t b)
{
  long d = (long) (int32_t) (a - b);  // abs((int32_t)(a - b)) on uint32 operands (:1830)
  return d < 0 ? -d : d;
}

// mask_pairs_chr_pos: how many copies of element p survive (0, 1, or 2 for element 1)
__global__ __launch_bounds__(256) void k_mask_count(const bk_pair *__restrict__ pairs, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ gof,
                                                    const uint64_t *__restrict__ goff, uint64_t n, long dist, uint32_t *__restrict__ cnt)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t g = gof[p];
  uint64_t gs = goff[g], ge = goff[g + 1];
  uint64_t np = ge - gs, i = p - gs;
  uint32_t c = 0;
  if (np >= 3 && i >= 1 && i + 1 < np)
  {
    const bk_pair a = pairs[idx[p]], l = pairs[idx[p - 1]], r = pairs[idx[p + 1]];
    long ll = absdiff(l.x, a.x), lr = absdiff(r.x, a.x);
    long Lx = ll < lr ? ll : lr;
    ll = absdiff(l.y, a.y);
    lr = absdiff(r.y, a.y);
    long Ly = ll < lr ? ll : lr;
    if (!(Lx > dist || Ly > dist)) c += 1;
    if (i == 1)
    {
      // "first read pair" block (:1830-1835): element 1 against element 2 only, emitted before the loop's copy
      long fx = absdiff(a.x, r.x), fy = absdiff(a.y, r.y);
      if (!(fx > dist || fy > dist)) c += 1;
    }
  }
  cnt[p] = c;
}

__global__ __launch_bounds__(256) void k_expand(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ off, const uint32_t *__restrict__ idx,
                                                const uint32_t *__restrict__ gof, uint64_t n, uint32_t *__restrict__ oidx, uint32_t *__restrict__ ogof)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t c = cnt[p], o = off[p];
  for (uint32_t k = 0; k < c; ++k)
  {
    oidx[o + k] = idx[p];
    ogof[o + k] = gof[p];
  }
}
__global__ void k_new_goff(const uint32_t *__restrict__ off, const uint64_t *__restrict__ goff, uint32_t ng, uint64_t n, uint64_t *__restrict__ ogoff)
{
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > ng) return;
  uint64_t s = g < ng ? goff[g] : n;
  ogoff[g] = off[s];  // off has n+1 entries
}

// ---- fast clustering ------------------------------------------------------------------------------------
// nxt[p]: first position after p that is not a member of the window anchored at p (:1064), or END for
// the last element of a group (its one-element cluster is never flushed, :1083-1085)
__global__ __launch_bounds__(256) void k_fast_next(const uint32_t *__restrict__ key, const uint32_t *__restrict__ gof, const uint64_t *__restrict__ goff, uint64_t n, double w,
                                                   uint32_t *__restrict__ nxt)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t g = gof[p];
  uint64_t ge = goff[g + 1];
  if (p + 1 >= ge)
  {
    nxt[p] = END;
    return;
  }
  double lim = (double) (long) key[p] + w;  // long pre_pos + double w
  uint64_t lo = p + 1, hi = ge;             // first j in (p, ge) with key[j] > lim
  while (lo < hi)
  {
    uint64_t m = (lo + hi) >> 1;
    if ((double) key[m] <= lim) lo = m + 1; else hi = m;
  }
  if (lo > ge - 1) lo = ge - 1;             // `i != n - 1`: the last element always breaks the window
  nxt[p] = (uint32_t) lo;
}
__global__ void k_mark_starts(const uint64_t *__restrict__ goff, uint32_t ng, uint32_t *__restrict__ mark)
{
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  if (goff[g + 1] > goff[g]) mark[goff[g]] = 1;
}
// ---- anchors of all windows: the positions a group's chain start -> nxt -> nxt ... visits ------------------------------
// (Pointer jumping over the whole list - log2(largest group) launches of doubling, as many of marking, every one a pass over n
// elements - was the first form.)  The chain is strictly increasing and never leaves its group, so it is followed tile by tile:
//   k_fw_exit   per tile of FW_TILE positions, in LDS: exit[p] = the first position of p's chain behind the tile (END if it ends
//               inside)
//   k_fw_chain  one thread per group: from the group's start, exit -> exit -> ...: the first anchor of every tile the chain
//               visits is marked (at most one hop per tile)
//   k_fw_mark   per tile, in LDS: exact 2^k-hop tables of the tile's part of nxt, then the marks spread from the seeds (group
//               starts and chain entries) highest level first
constexpr uint32_t FW_TILE = 1024, FW_LEVELS = 10;
__global__ __launch_bounds__(FW_TILE) void k_fw_exit(const uint32_t *__restrict__ nxt, uint64_t n, uint32_t *__restrict__ exitp)
{
  __shared__ uint32_t e[FW_TILE];
  const uint64_t t0 = (uint64_t) blockIdx.x * FW_TILE, p = t0 + threadIdx.x;
  const uint64_t tend = t0 + FW_TILE;
  e[threadIdx.x] = p < n ? nxt[p] : END;
  for (uint32_t r = 0; r < FW_LEVELS; ++r)
  {
    __syncthreads();
    const uint32_t v = e[threadIdx.x];
    // a value read while its owner moves it on is still a position of the chain, only further along
    if (v != END && v < tend) e[threadIdx.x] = e[v - t0];
  }
  __syncthreads();
  if (p < n) exitp[p] = e[threadIdx.x];
}
__global__ __launch_bounds__(64) void k_fw_chain(const uint64_t *__restrict__ goff, uint32_t ng, const uint32_t *__restrict__ exitp, uint32_t *__restrict__ mark)
{
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng || goff[g + 1] <= goff[g]) return;
  uint32_t p = (uint32_t) goff[g];  // marked by k_mark_starts
  for (;;)
  {
    const uint32_t q = exitp[p];
    if (q == END) break;
    mark[q] = 1;
    p = q;
  }
}
__global__ __launch_bounds__(FW_TILE) void k_fw_mark(const uint32_t *__restrict__ nxt, uint64_t n, uint32_t *__restrict__ mark)
{
  __shared__ uint16_t j[FW_LEVELS][FW_TILE];
  __shared__ uint32_t m[FW_TILE];
  constexpr uint16_t OUT = 0xFFFFu;
  const uint64_t t0 = (uint64_t) blockIdx.x * FW_TILE, p = t0 + threadIdx.x;
  const uint32_t i = threadIdx.x;
  uint32_t v = p < n ? nxt[p] : END;
  j[0][i] = (v != END && v < t0 + FW_TILE) ? (uint16_t) (v - t0) : OUT;
  m[i] = p < n ? mark[p] : 0u;
  for (uint32_t k = 1; k < FW_LEVELS; ++k)
  {
    __syncthreads();
    const uint16_t a = j[k - 1][i];
    j[k][i] = a == OUT ? OUT : j[k - 1][a];
  }
  for (int k = FW_LEVELS - 1; k >= 0; --k)
  {
    __syncthreads();
    if (m[i])
    {
      const uint16_t t = j[k][i];
      if (t != OUT) m[t] = 1u;
    }
  }
  __syncthreads();
  if (p < n) mark[p] = m[i];
}

// ascan = exclusive scan of mark.  Anchor ordinal of p = ascan[p] + mark[p] - 1.
__global__ __launch_bounds__(256) void k_anchor_pos(const uint32_t *__restrict__ mark, const uint32_t *__restrict__ ascan, uint64_t n, uint32_t *__restrict__ apos)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n && mark[p]) apos[ascan[p]] = (uint32_t) p;
}
__global__ __launch_bounds__(256) void k_fast_keep(const uint32_t *__restrict__ mark, const uint32_t *__restrict__ ascan, const uint32_t *__restrict__ apos,
                                                   const uint32_t *__restrict__ nxt, uint64_t n, uint32_t *__restrict__ keep, uint32_t *__restrict__ kid)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t ord = ascan[p] + mark[p] - 1;
  uint32_t a = apos[ord];
  uint32_t e = nxt[a];
  keep[p] = (e != END && e - a >= 2) ? 1u : 0u;  // cl_index.size() >= min_reads (2)
  kid[p] = ord;
}
__global__ __launch_bounds__(256) void k_compact3(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ off, uint64_t n, const uint32_t *__restrict__ a0,
                                                  const uint32_t *__restrict__ a1, const uint32_t *__restrict__ a2, const uint32_t *__restrict__ a3, uint32_t *__restrict__ o0,
                                                  uint32_t *__restrict__ o1, uint32_t *__restrict__ o2, uint32_t *__restrict__ o3)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || !keep[p]) return;
  uint32_t o = off[p];
  o0[o] = a0[p];
  if (a1) o1[o] = a1[p];
  if (a2) o2[o] = a2[p];
  if (a3) o3[o] = a3[p];
}
__global__ __launch_bounds__(256) void k_pack_k(const uint32_t *__restrict__ k1, const uint32_t *__restrict__ k2, uint64_t n, int kbits, uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  key[p] = ((uint64_t) k2[p] << kbits) | k1[p];
  val[p] = (uint32_t) p;
}
// sorted (k2,k1) keys: run flags
__global__ __launch_bounds__(256) void k_run_flag(const uint64_t *__restrict__ key, uint64_t n, uint32_t *__restrict__ flag)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) flag[p] = (p == 0 || key[p] != key[p - 1]) ? 1u : 0u;
}
// uid of sorted position q = fscan[q] + flag[q] - 1.  The members of a uid are a run of the (stable) sorted order, so the
// run's head knows everything: its sorted position (ustart; the next head's position ends the run: count = difference) and
// the smallest original position (the sort is stable and the values went in ascending).  No atomics: 12 M of them on
// ~250 K addresses, neighbouring lanes on the same one, were 1.3 ms.
__global__ __launch_bounds__(256) void k_uid_stats(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ fscan, const uint32_t *__restrict__ val, uint64_t n,
                                                   uint32_t *__restrict__ uid_of_elem, uint32_t *__restrict__ ustart, uint32_t *__restrict__ uminpos)
{
  uint64_t q = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const uint32_t f = flag[q];
  uint32_t u = fscan[q] + f - 1;
  uint32_t e = val[q];
  uid_of_elem[e] = u;
  if (f)
  {
    ustart[u] = (uint32_t) q;
    uminpos[u] = e;
  }
  if (q == n - 1) ustart[u + 1] = (uint32_t) n;
}
__global__ __launch_bounds__(256) void k_first_flag(const uint32_t *__restrict__ uid, const uint32_t *__restrict__ ustart, const uint32_t *__restrict__ uminpos, uint64_t n,
                                                    uint32_t *__restrict__ keep, uint32_t *__restrict__ first)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t u = uid[p];
  uint32_t k = ustart[u + 1] - ustart[u] >= 2 ? 1u : 0u;  // it->second >= min_reads (:1142)
  keep[p] = k;
  first[p] = (k && uminpos[u] == (uint32_t) p) ? 1u : 0u;
}
// cluster number = order of first appearance inside the group, starting at 1 (:1147-1149)
__global__ __launch_bounds__(256) void k_knum(const uint32_t *__restrict__ first, const uint32_t *__restrict__ fscan, const uint32_t *__restrict__ uid,
                                              const uint32_t *__restrict__ gof, const uint64_t *__restrict__ goff, uint64_t n, uint32_t *__restrict__ knum)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || !first[p]) return;
  uint64_t gs = goff[gof[p]];
  knum[uid[p]] = fscan[p] - fscan[gs] + 1;
}
__global__ __launch_bounds__(256) void k_cluster_of(const uint32_t *__restrict__ uid, const uint32_t *__restrict__ knum, const uint32_t *__restrict__ keep, uint64_t n,
                                                    uint32_t *__restrict__ cl)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) cl[p] = keep[p] ? knum[uid[p]] : 0u;
}
__global__ void k_group_small(const uint64_t *__restrict__ goff, uint32_t ng, uint32_t *__restrict__ small)
{
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < ng) small[g] = (goff[g + 1] - goff[g]) < 2 ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_drop_small(const uint32_t *__restrict__ gof, const uint32_t *__restrict__ small, uint64_t n, uint32_t *__restrict__ keep)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) keep[p] = small[gof[p]] ? 0u : 1u;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------
uint64_t PairList::total(hipStream_t st) const
{
  uint64_t t = 0;
  if (ng == 0) return 0;
  HIP_CHECK(hipMemcpyAsync(&t, goff.get<uint64_t>() + ng, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return t;
}

static inline unsigned nb(uint64_t n) { return cdiv(n ? n : 1, 256); }
static void filter_groups(PairList &L, const uint32_t *drop, ClusterBufs &b, hipStream_t st);

// sort list by x or y exactly like std::sort; all element attributes follow
static void sort_list(const bk_pair *pairs, PairList &L, int use_y, uint32_t **extra, int n_extra, ClusterBufs &b, hipStream_t st)
{
  if (L.n == 0) return;
  uint32_t *key = b.key.as<uint32_t>(L.n), *perm = b.perm.as<uint32_t>(L.n);
  hipLaunchKernelGGL(k_gather_key, dim3(nb(L.n)), dim3(256), 0, st, pairs, L.idx.get<uint32_t>(), L.n, use_y, key);
  hipLaunchKernelGGL(k_iota, dim3(nb(L.n)), dim3(256), 0, st, perm, L.n);
  b.se.heavy = b.observe ? (use_y ? &b.heavy_y : &b.heavy_x) : nullptr;
  std_sort_groups(key, perm, L.gof.get<uint32_t>(), L.goff.get<uint64_t>(), L.ng, L.n, b.se, st);
  b.se.heavy = nullptr;
  uint32_t *tmp = b.tmp.as<uint32_t>(L.n);
  auto apply = [&](uint32_t *arr) {
    hipLaunchKernelGGL(k_gather_u32, dim3(nb(L.n)), dim3(256), 0, st, arr, perm, L.n, tmp);
    HIP_CHECK(hipMemcpyAsync(arr, tmp, L.n * 4, hipMemcpyDeviceToDevice, st));
  };
  apply(L.idx.get<uint32_t>());
  for (int k = 0; k < n_extra; ++k) apply(extra[k]);
}

static void mask_list(const bk_pair *pairs, PairList &L, long dist, ClusterBufs &b, hipStream_t st)
{
  if (L.n == 0) return;
  uint32_t *cnt = b.cnt.as<uint32_t>(L.n + 1);
  hipLaunchKernelGGL(k_mask_count, dim3(nb(L.n)), dim3(256), 0, st, pairs, L.idx.get<uint32_t>(), L.gof.get<uint32_t>(), L.goff.get<uint64_t>(), L.n, dist, cnt);
  uint32_t *off = b.off.as<uint32_t>(L.n + 1);
  prims::exclusive_scan<uint32_t>(cnt, off, L.n, b.scan_tmp, st);
  uint32_t total = 0;
  HIP_CHECK(hipMemcpyAsync(&total, off + L.n, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  uint32_t *oidx = b.idx2.as<uint32_t>((uint64_t) total + 1), *ogof = b.gof2.as<uint32_t>((uint64_t) total + 1);
  uint64_t *ogoff = b.goff2.as<uint64_t>((uint64_t) L.ng + 1);
  hipLaunchKernelGGL(k_expand, dim3(nb(L.n)), dim3(256), 0, st, cnt, off, L.idx.get<uint32_t>(), L.gof.get<uint32_t>(), L.n, oidx, ogof);
  hipLaunchKernelGGL(k_new_goff, dim3(cdiv(L.ng + 1, 256)), dim3(256), 0, st, off, L.goff.get<uint64_t>(), L.ng, L.n, ogoff);
  std::swap(L.idx, b.idx2);
  std::swap(L.gof, b.gof2);
  std::swap(L.goff, b.goff2);
  L.n = total;
}

void remove_isolated_all(const bk_pair *pairs, const uint32_t *gof0, const uint64_t *gstart, uint32_t ng, uint64_t n, double w, PairList &L, ClusterBufs &b, hipStream_t st,
                         const uint32_t *drop_group)
{
  remove_isolated_begin(pairs, gof0, gstart, ng, n, w, L, b, st, drop_group, nullptr, nullptr);
  remove_isolated_end(pairs, L, b, st);
}

void remove_isolated_end(const bk_pair *pairs, PairList &L, ClusterBufs &b, hipStream_t st)
{
  if (L.n == 0 || L.ng == 0) return;
  sort_list(pairs, L, 0, nullptr, 0, b, st);
}

void list_subset(const PairList &src, const uint32_t *drop, PairList &dst, ClusterBufs &b, hipStream_t st)
{
  dst.n = src.n;
  dst.ng = src.ng;
  uint32_t *idx = dst.idx.as<uint32_t>(src.n + 1), *gof = dst.gof.as<uint32_t>(src.n + 1);
  uint64_t *goff = dst.goff.as<uint64_t>((uint64_t) src.ng + 1);
  if (src.n)
  {
    HIP_CHECK(hipMemcpyAsync(idx, src.idx.get<uint32_t>(), src.n * 4, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipMemcpyAsync(gof, src.gof.get<uint32_t>(), src.n * 4, hipMemcpyDeviceToDevice, st));
  }
  HIP_CHECK(hipMemcpyAsync(goff, src.goff.get<uint64_t>(), ((uint64_t) src.ng + 1) * 8, hipMemcpyDeviceToDevice, st));
  filter_groups(dst, drop, b, st);
}

// list of the pairs of the groups with keep[g] != 0, straight from the group ranges of the pair table (the pairs of a group are
// contiguous there): what iota + copy + filter_groups over all n pairs would leave, in one launch over the list itself
namespace
{
__global__ __launch_bounds__(256) void k_subset_list(const uint64_t *__restrict__ gstart, const uint64_t *__restrict__ goff, uint32_t ng, uint64_t n_list,
                                                     uint32_t *__restrict__ idx, uint32_t *__restrict__ gof)
{
  const uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_list) return;
  uint32_t lo = 0, hi = ng;  // largest g with goff[g] <= p (a group that is empty in the list never wins: the next one starts at the same offset)
  while (hi - lo > 1)
  {
    const uint32_t m = (lo + hi) >> 1;
    if (goff[m] <= p) lo = m; else hi = m;
  }
  idx[p] = (uint32_t) (gstart[lo] + (p - goff[lo]));
  gof[p] = lo;
}
}  // namespace
namespace
{
__global__ __launch_bounds__(256) void k_subset_of_list(const uint32_t *__restrict__ src_idx, const uint64_t *__restrict__ src_goff, const uint64_t *__restrict__ goff, uint32_t ng,
                                                        uint64_t n_list, uint32_t *__restrict__ idx, uint32_t *__restrict__ gof)
{
  const uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_list) return;
  uint32_t lo = 0, hi = ng;
  while (hi - lo > 1)
  {
    const uint32_t m = (lo + hi) >> 1;
    if (goff[m] <= p) lo = m; else hi = m;
  }
  idx[p] = src_idx[src_goff[lo] + (p - goff[lo])];
  gof[p] = lo;
}
}  // namespace
void list_subset_ranges(const PairList &src, const uint64_t *src_goff_host, const uint8_t *keep_host, PairList &dst, hipStream_t st)
{
  const uint32_t ng = src.ng;
  std::vector<uint64_t> goff((size_t) ng + 1, 0);
  for (uint32_t g = 0; g < ng; ++g) goff[g + 1] = goff[g] + (keep_host[g] ? src_goff_host[g + 1] - src_goff_host[g] : 0);
  dst.n = goff[ng];
  dst.ng = ng;
  uint32_t *idx = dst.idx.as<uint32_t>(dst.n + 1), *gof = dst.gof.as<uint32_t>(dst.n + 1);
  uint64_t *dgoff = dst.goff.as<uint64_t>((uint64_t) ng + 1);
  HIP_CHECK(hipMemcpyAsync(dgoff, goff.data(), ((size_t) ng + 1) * 8, hipMemcpyHostToDevice, st));
  HIP_CHECK(hipStreamSynchronize(st));  // (goff is a local)
  if (dst.n) hipLaunchKernelGGL(k_subset_of_list, dim3(nb(dst.n)), dim3(256), 0, st, src.idx.get<uint32_t>(), src.goff.get<uint64_t>(), dgoff, ng, dst.n, idx, gof);
}

void list_of_groups(const uint64_t *gstart_dev, const uint64_t *gstart_host, const uint8_t *keep_host, uint32_t ng, PairList &L, hipStream_t st)
{
  std::vector<uint64_t> goff((size_t) ng + 1, 0);
  for (uint32_t g = 0; g < ng; ++g) goff[g + 1] = goff[g] + (keep_host[g] ? gstart_host[g + 1] - gstart_host[g] : 0);
  L.n = goff[ng];
  L.ng = ng;
  uint32_t *idx = L.idx.as<uint32_t>(L.n + 1), *gof = L.gof.as<uint32_t>(L.n + 1);
  uint64_t *dgoff = L.goff.as<uint64_t>((uint64_t) ng + 1);
  HIP_CHECK(hipMemcpyAsync(dgoff, goff.data(), ((size_t) ng + 1) * 8, hipMemcpyHostToDevice, st));
  HIP_CHECK(hipStreamSynchronize(st));  // (goff is a local)
  if (L.n) hipLaunchKernelGGL(k_subset_list, dim3(nb(L.n)), dim3(256), 0, st, gstart_dev, dgoff, ng, L.n, idx, gof);
}

void remove_isolated_begin(const bk_pair *pairs, const uint32_t *gof0, const uint64_t *gstart, uint32_t ng, uint64_t n, double w, PairList &L, ClusterBufs &b, hipStream_t st,
                           const uint32_t *drop_group, const uint64_t *gstart_host, const uint8_t *keep_host)
{
  if (gstart_host && keep_host)
  {
    if (n == 0 || ng == 0)
    {
      L.n = 0;
      L.ng = ng;
      (void) L.idx.as<uint32_t>(1);
      (void) L.gof.as<uint32_t>(1);
      (void) L.goff.as<uint64_t>((uint64_t) ng + 1);
      return;
    }
    list_of_groups(gstart, gstart_host, keep_host, ng, L, st);
  }
  else
  {
    L.n = n;
    L.ng = ng;
    uint32_t *idx = L.idx.as<uint32_t>(n + 1);
    uint32_t *gof = L.gof.as<uint32_t>(n + 1);
    uint64_t *goff = L.goff.as<uint64_t>((uint64_t) ng + 1);
    if (n == 0 || ng == 0) return;
    hipLaunchKernelGGL(k_iota, dim3(nb(n)), dim3(256), 0, st, idx, n);
    HIP_CHECK(hipMemcpyAsync(gof, gof0, n * 4, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipMemcpyAsync(goff, gstart, ((uint64_t) ng + 1) * 8, hipMemcpyDeviceToDevice, st));
    if (drop_group) filter_groups(L, drop_group, b, st);
  }
  if (L.n == 0) return;
  const long dist = (long) w;  // remove_isolated_pairs passes double w to a `long distance` parameter
  sort_list(pairs, L, 0, nullptr, 0, b, st);
  mask_list(pairs, L, dist, b, st);
  sort_list(pairs, L, 1, nullptr, 0, b, st);
  mask_list(pairs, L, dist, b, st);
}

// one anchored-window pass (on the list's current order, keyed by x or y): keeps members of windows with
// >= 2 elements, records the window ordinal in kout.  Extra attribute k_prev (may be null) follows.
static void fast_pass(const bk_pair *pairs, PairList &L, int use_y, double w, DevBuf &k_prev, DevBuf &k_new, ClusterBufs &b, hipStream_t st)
{
  if (L.n == 0) return;
  const uint64_t n = L.n;
  uint32_t *key = b.key.as<uint32_t>(n);
  hipLaunchKernelGGL(k_gather_key, dim3(nb(n)), dim3(256), 0, st, pairs, L.idx.get<uint32_t>(), n, use_y, key);
  uint32_t *jump = b.jump.as<uint32_t>(n * 2ull);
  hipLaunchKernelGGL(k_fast_next, dim3(nb(n)), dim3(256), 0, st, key, L.gof.get<uint32_t>(), L.goff.get<uint64_t>(), n, w, jump);
  uint32_t *mark = b.mark.as<uint32_t>(n + 1);
  HIP_CHECK(hipMemsetAsync(mark, 0, (n + 1) * 4, st));
  hipLaunchKernelGGL(k_mark_starts, dim3(cdiv(L.ng, 256)), dim3(256), 0, st, L.goff.get<uint64_t>(), L.ng, mark);
  {
    uint32_t *exitp = jump + n;
    const unsigned tiles = (unsigned) cdiv(n, FW_TILE);
    hipLaunchKernelGGL(k_fw_exit, dim3(tiles), dim3(FW_TILE), 0, st, jump, n, exitp);
    hipLaunchKernelGGL(k_fw_chain, dim3(cdiv(L.ng, 64)), dim3(64), 0, st, L.goff.get<uint64_t>(), L.ng, exitp, mark);
    hipLaunchKernelGGL(k_fw_mark, dim3(tiles), dim3(FW_TILE), 0, st, jump, n, mark);
  }
  uint32_t *ascan = b.off.as<uint32_t>(n + 1);
  prims::exclusive_scan<uint32_t>(mark, ascan, n, b.scan_tmp, st);
  uint32_t *apos = b.apos.as<uint32_t>(n + 1);
  hipLaunchKernelGGL(k_anchor_pos, dim3(nb(n)), dim3(256), 0, st, mark, ascan, n, apos);
  uint32_t *keep = b.cnt.as<uint32_t>(n + 1), *kid = b.kid.as<uint32_t>(n + 1);
  hipLaunchKernelGGL(k_fast_keep, dim3(nb(n)), dim3(256), 0, st, mark, ascan, apos, jump, n, keep, kid);
  uint32_t *off = b.off2.as<uint32_t>(n + 1);
  prims::exclusive_scan<uint32_t>(keep, off, n, b.scan_tmp, st);
  uint32_t total = 0;
  HIP_CHECK(hipMemcpyAsync(&total, off + n, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  uint32_t *oidx = b.idx2.as<uint32_t>((uint64_t) total + 1), *ogof = b.gof2.as<uint32_t>((uint64_t) total + 1);
  uint32_t *okn = k_new.as<uint32_t>((uint64_t) total + 1);
  uint32_t *okp = k_prev.p ? b.kprev2.as<uint32_t>((uint64_t) total + 1) : nullptr;
  uint64_t *ogoff = b.goff2.as<uint64_t>((uint64_t) L.ng + 1);
  hipLaunchKernelGGL(k_compact3, dim3(nb(n)), dim3(256), 0, st, keep, off, n, L.idx.get<uint32_t>(), L.gof.get<uint32_t>(), kid, k_prev.get<uint32_t>(), oidx, ogof, okn, okp);
  hipLaunchKernelGGL(k_new_goff, dim3(cdiv(L.ng + 1, 256)), dim3(256), 0, st, off, L.goff.get<uint64_t>(), L.ng, n, ogoff);
  std::swap(L.idx, b.idx2);
  std::swap(L.gof, b.gof2);
  std::swap(L.goff, b.goff2);
  if (k_prev.p) std::swap(k_prev, b.kprev2);
  L.n = total;
}

// removes every element whose group is flagged in drop[] (device, one u32 per group)
static void filter_groups(PairList &L, const uint32_t *drop, ClusterBufs &b, hipStream_t st)
{
  if (L.n == 0) return;
  uint32_t *keep = b.cnt.as<uint32_t>(L.n + 1), *off = b.off.as<uint32_t>(L.n + 1);
  hipLaunchKernelGGL(k_drop_small, dim3(nb(L.n)), dim3(256), 0, st, L.gof.get<uint32_t>(), drop, L.n, keep);
  prims::exclusive_scan<uint32_t>(keep, off, L.n, b.scan_tmp, st);
  uint32_t total = 0;
  HIP_CHECK(hipMemcpyAsync(&total, off + L.n, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  if (total != L.n)
  {
    uint32_t *oidx = b.idx2.as<uint32_t>((uint64_t) total + 1), *ogof = b.gof2.as<uint32_t>((uint64_t) total + 1);
    uint64_t *ogoff = b.goff2.as<uint64_t>((uint64_t) L.ng + 1);
    hipLaunchKernelGGL(k_compact3, dim3(nb(L.n)), dim3(256), 0, st, keep, off, L.n, L.idx.get<uint32_t>(), L.gof.get<uint32_t>(), (const uint32_t *) nullptr,
                       (const uint32_t *) nullptr, oidx, ogof, (uint32_t *) nullptr, (uint32_t *) nullptr);
    hipLaunchKernelGGL(k_new_goff, dim3(cdiv(L.ng + 1, 256)), dim3(256), 0, st, off, L.goff.get<uint64_t>(), L.ng, L.n, ogoff);
    std::swap(L.idx, b.idx2);
    std::swap(L.gof, b.gof2);
    std::swap(L.goff, b.goff2);
    L.n = total;
  }
}

void drop_small_groups(PairList &L, ClusterBufs &b, hipStream_t st)
{
  // groups with fewer than 2 pairs after masking are not clustered at all (BreakID.cc:125)
  if (L.n == 0) return;
  uint32_t *small = b.small.as<uint32_t>((uint64_t) L.ng + 1);
  hipLaunchKernelGGL(k_group_small, dim3(cdiv(L.ng, 256)), dim3(256), 0, st, L.goff.get<uint64_t>(), L.ng, small);
  filter_groups(L, small, b, st);
}

void fast_cluster_all(const bk_pair *pairs, PairList &L, double w, DevBuf &cluster_out, ClusterBufs &b, hipStream_t st)
{
  drop_small_groups(L, b, st);
  DevBuf none;
  const uint64_t n_first = L.n;  // (the window numbers of both passes stay below the length of the list the first pass sees)
  // pass 1 on x (list arrives x-sorted from remove_isolated_pairs), pass 2 on y after std::sort by y
  fast_pass(pairs, L, 0, w, none, b.k1, b, st);
  {
    uint32_t *extra[1] = {b.k1.get<uint32_t>()};
    sort_list(pairs, L, 1, extra, L.n ? 1 : 0, b, st);
  }
  fast_pass(pairs, L, 1, w, b.k1, b.k2, b, st);
  {
    uint32_t *extra[2] = {b.k1.get<uint32_t>(), b.k2.get<uint32_t>()};
    sort_list(pairs, L, 0, extra, L.n ? 2 : 0, b, st);
  }
  const uint64_t n = L.n;
  uint32_t *cl = cluster_out.as<uint32_t>(n + 1);
  if (n == 0) return;
  // ids "k1:k2" with >= 2 members survive; numeric cluster = order of first appearance in x order
  uint64_t *pk = b.pk.as<uint64_t>(n);
  uint32_t *pv = b.perm.as<uint32_t>(n);
  // (k1 and k2 number windows of the list as the passes saw it: both are below n_first, so the pair needs 2 * bits(n_first) bits - the
  // sort takes the passes over those)
  int kbits = 1;
  while (kbits < 32 && (n_first >> kbits) != 0) ++kbits;
  hipLaunchKernelGGL(k_pack_k, dim3(nb(n)), dim3(256), 0, st, b.k1.get<uint32_t>(), b.k2.get<uint32_t>(), n, kbits, pk, pv);
  uint64_t *ks;
  uint32_t *vs;
  prims::radix_sort_pairs(pk, pv, n, 0, std::min(64, 2 * kbits), b.radix, st, &ks, &vs);
  uint32_t *flag = b.cnt.as<uint32_t>(n + 1), *fscan = b.off.as<uint32_t>(n + 1);
  hipLaunchKernelGGL(k_run_flag, dim3(nb(n)), dim3(256), 0, st, ks, n, flag);
  prims::exclusive_scan<uint32_t>(flag, fscan, n, b.scan_tmp, st);
  uint32_t *uid = b.kid.as<uint32_t>(n + 1), *ucount = b.mark.as<uint32_t>(n + 2), *uminpos = b.apos.as<uint32_t>(n + 1);
  hipLaunchKernelGGL(k_uid_stats, dim3(nb(n)), dim3(256), 0, st, flag, fscan, vs, n, uid, ucount, uminpos);
  uint32_t *keep = b.off2.as<uint32_t>(n + 1), *first = b.key.as<uint32_t>(n + 1);
  hipLaunchKernelGGL(k_first_flag, dim3(nb(n)), dim3(256), 0, st, uid, ucount, uminpos, n, keep, first);
  uint32_t *fs2 = b.tmp.as<uint32_t>(n + 1);
  prims::exclusive_scan<uint32_t>(first, fs2, n, b.scan_tmp, st);
  uint32_t *knum = b.knum.as<uint32_t>(n + 1);
  hipLaunchKernelGGL(k_knum, dim3(nb(n)), dim3(256), 0, st, first, fs2, uid, L.gof.get<uint32_t>(), L.goff.get<uint64_t>(), n, knum);
  uint32_t *clfull = b.clfull.as<uint32_t>(n + 1);
  hipLaunchKernelGGL(k_cluster_of, dim3(nb(n)), dim3(256), 0, st, uid, knum, keep, n, clfull);
  uint32_t *off = b.off.as<uint32_t>(n + 1);
  prims::exclusive_scan<uint32_t>(keep, off, n, b.scan_tmp, st);
  uint32_t total = 0;
  HIP_CHECK(hipMemcpyAsync(&total, off + n, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  uint32_t *oidx = b.idx2.as<uint32_t>((uint64_t) total + 1), *ogof = b.gof2.as<uint32_t>((uint64_t) total + 1);
  uint64_t *ogoff = b.goff2.as<uint64_t>((uint64_t) L.ng + 1);
  cl = cluster_out.as<uint32_t>((uint64_t) total + 1);
  hipLaunchKernelGGL(k_compact3, dim3(nb(n)), dim3(256), 0, st, keep, off, n, L.idx.get<uint32_t>(), L.gof.get<uint32_t>(), clfull, (const uint32_t *) nullptr, oidx, ogof, cl,
                     (uint32_t *) nullptr);
  hipLaunchKernelGGL(k_new_goff, dim3(cdiv(L.ng + 1, 256)), dim3(256), 0, st, off, L.goff.get<uint64_t>(), L.ng, n, ogoff);
  std::swap(L.idx, b.idx2);
  std::swap(L.gof, b.gof2);
  std::swap(L.goff, b.goff2);
  L.n = total;
}

// test hook: mask_pairs_chr_pos (BreakID.cc:1813-1877) on the list in its current order
void debug_mask_list(const bk_pair *pairs, PairList &L, long dist, ClusterBufs &b, hipStream_t st) { mask_list(pairs, L, dist, b, st); }

// ---- two lanes of chromosome-pair groups (api.hip: bk_mask_and_cluster) -> one list in group order --------------------
namespace
{
__global__ __launch_bounds__(256) void k_merge_goff(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b, uint32_t ng, uint64_t *__restrict__ out)
{
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g <= ng) out[g] = a[g] + b[g];  // the lanes own disjoint groups: the offsets add
}
__global__ __launch_bounds__(256) void k_merge_copy(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ gof, const uint32_t *__restrict__ cl, const uint64_t *__restrict__ goff,
                                                    uint64_t n, const uint64_t *__restrict__ mgoff, uint32_t *__restrict__ oidx, uint32_t *__restrict__ ogof, uint32_t *__restrict__ ocl)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t g = gof[p];
  // a group belongs to one lane, so inside the merged list it starts where the merged offsets say and keeps its order
  const uint64_t d = mgoff[g] + (p - goff[g]);
  oidx[d] = idx[p];
  ogof[d] = g;
  if (cl) ocl[d] = cl[p];
}
}  // namespace

// K lists of disjoint groups in one go: the merged offsets are the sum of all of them, then every list scatters itself
void merge_lists_many(const PairList *const *lists, const uint32_t *const *cls, int K, PairList &out, DevBuf *cl_out, hipStream_t st)
{
  const uint32_t ng = lists[0]->ng;
  out.ng = ng;
  out.n = 0;
  for (int l = 0; l < K; ++l) out.n += lists[l]->n;
  uint32_t *oidx = out.idx.as<uint32_t>(out.n + 1), *ogof = out.gof.as<uint32_t>(out.n + 1);
  uint64_t *ogoff = out.goff.as<uint64_t>((uint64_t) ng + 1);
  uint32_t *ocl = cl_out ? cl_out->as<uint32_t>(out.n + 1) : nullptr;
  HIP_CHECK(hipMemsetAsync(ogoff, 0, ((size_t) ng + 1) * 8, st));
  for (int l = 0; l < K; ++l) hipLaunchKernelGGL(k_merge_goff, dim3(cdiv(ng + 1, 256)), dim3(256), 0, st, ogoff, lists[l]->goff.get<uint64_t>(), ng, ogoff);
  for (int l = 0; l < K; ++l)
  {
    const PairList &A = *lists[l];
    if (A.n) hipLaunchKernelGGL(k_merge_copy, dim3(nb(A.n)), dim3(256), 0, st, A.idx.get<uint32_t>(), A.gof.get<uint32_t>(), cls ? cls[l] : nullptr, A.goff.get<uint64_t>(), A.n, ogoff, oidx, ogof, ocl);
  }
}

void merge_lists(const PairList &A, const uint32_t *clA, const PairList &B, const uint32_t *clB, PairList &out, DevBuf *cl_out, hipStream_t st)
{
  const uint32_t ng = A.ng;
  out.ng = ng;
  out.n = A.n + B.n;
  uint32_t *oidx = out.idx.as<uint32_t>(out.n + 1), *ogof = out.gof.as<uint32_t>(out.n + 1);
  uint64_t *ogoff = out.goff.as<uint64_t>((uint64_t) ng + 1);
  uint32_t *ocl = cl_out ? cl_out->as<uint32_t>(out.n + 1) : nullptr;
  hipLaunchKernelGGL(k_merge_goff, dim3(cdiv(ng + 1, 256)), dim3(256), 0, st, A.goff.get<uint64_t>(), B.goff.get<uint64_t>(), ng, ogoff);
  if (A.n) hipLaunchKernelGGL(k_merge_copy, dim3(nb(A.n)), dim3(256), 0, st, A.idx.get<uint32_t>(), A.gof.get<uint32_t>(), clA, A.goff.get<uint64_t>(), A.n, ogoff, oidx, ogof, ocl);
  if (B.n) hipLaunchKernelGGL(k_merge_copy, dim3(nb(B.n)), dim3(256), 0, st, B.idx.get<uint32_t>(), B.gof.get<uint32_t>(), clB, B.goff.get<uint64_t>(), B.n, ogoff, oidx, ogof, ocl);
}
