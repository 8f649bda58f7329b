// Exact agglomerative clustering of find_cluster_pairs_enspan_ahc (BreakID.cc:1304-1352) /
// util_cluster.cc (average linkage, distance_type 1) without the N x N matrix.
//
// What the reference does per group: N x N double matrix of Euclidean distances, per-node neighbour lists
// kept sorted by insert_sorted (util_cluster.cc:249-275), then greedily merges the globally closest pair of
// roots (roots scanned from the highest node index, strict `<`: :320-355) while best <= (long) w, average
// linkage recomputed as a sequential sum over a.points x b.points (:201-215).  Roots with >= 2 points get
// cluster numbers in node-index order (BreakID.cc:1334-1348).
//
// What happens here (one workgroup per chromosome-pair group, groups in parallel):
//   * leaves whose distance is <= T form connected components; clusters of different components are always
//     farther apart than T, so their list entries only matter as "something farther was inserted before"
//     (the tail rule of insert_sorted) — exact values are needed inside a component only;
//   * the merge sequence of the whole group is replayed literally (global best over all components each
//     step), so node indices, the "highest root wins" tie rule and the tail rule see the same roots as the
//     reference; distances are recomputed on the fly with the reference's summation order in FP64
//     (__dmul_rn/__dadd_rn/__dsqrt_rn: no contraction);
//   * a node's sorted neighbour list is represented by (distance, ord) keys where ord reproduces the order
//     insert_sorted leaves among equal distances: ascending target index, except that the second arrival of
//     a run sits after the first when the first was the list tail at that moment.
#include "ahc.h"
#include "prims.h"

namespace
{
constexpr int AT = 256;  // threads per group

struct Entry
{
  double d;
  double ord;
  int32_t t;
  int32_t pad;
};

struct GroupMem
{
  // leaves (N)
  const uint32_t *x, *y;
  uint32_t *comp;       // component id = smallest leaf index of the component
  // nodes (2N)
  int32_t *rootcomp;    // component id while the node is a root, -1 afterwards
  uint32_t *npts;
  uint64_t *pts_off;
  uint64_t *ent_off;
  uint32_t *ent_cnt;
  int32_t *cand_t;
  double *cand_d, *cand_o;
  // components (indexed by component id, N slots)
  uint32_t *csize;
  uint64_t *cnodes_off;  // into cnodes pool (capacity 2 * csize)
  uint32_t *cnodes_cnt;
  uint64_t *cent_off;    // merged-node entry region of the component (capacity csize^2)
  uint64_t *cent_used;
  uint64_t *cpts_off;    // merged-node point region
  uint64_t *cpts_cap;
  uint64_t *cpts_used;
  double *cbest_d;
  int32_t *cbest_j;
  int32_t *cmaxroot;    // per component: its root with the largest node index (the newest merged node, else the last leaf)
  uint32_t *act;         // active component ids (size >= 2)
};

__device__ __forceinline__ double leaf_dist(const GroupMem &m, int a, int b)
{
  // euclidean_distance, util_cluster.cc:79-84, on doubles converted from uint32 (BreakID.cc:1800-1801)
  double dx = __dsub_rn((double) m.x[a], (double) m.x[b]);
  double dy = __dsub_rn((double) m.y[a], (double) m.y[b]);
  return __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
}

// the same distance from coordinates already in hand (first argument = the point of the merged node, as in leaf_dist(q_i, t_j))
__device__ __forceinline__ double xy_dist(double xa, double ya, uint2 b)
{
  double dx = __dsub_rn(xa, (double) b.x);
  double dy = __dsub_rn(ya, (double) b.y);
  return __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
}

// ---- kernel A: connected components of the "distance <= T" graph on the leaves of every group ------------------
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t i)
{
  while (true)
  {
    uint32_t p = parent[i];
    if (p == i) return i;
    uint32_t gp = parent[p];
    if (gp != p) atomicCAS(&parent[i], p, gp);  // path halving
    i = p;
  }
}
__device__ void uf_union(uint32_t *parent, uint32_t a, uint32_t b)
{
  while (true)
  {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b)
    {
      uint32_t t = a;
      a = b;
      b = t;
    }
    // link the larger root under the smaller so that the component id is its smallest leaf
    if (atomicCAS(&parent[a], a, b) == a) return;
  }
}

__global__ __launch_bounds__(256) void k_ahc_edges(const uint32_t *__restrict__ x, const uint32_t *__restrict__ y, const uint32_t *__restrict__ gof,
                                                   const uint64_t *__restrict__ goff, uint64_t n, double T, uint32_t *__restrict__ parent)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint64_t ge = goff[gof[p] + 1];
  const double xi = (double) x[p], yi = (double) y[p];
  for (uint64_t q = p + 1; q < ge; ++q)
  {
    double dx = __dsub_rn((double) x[q], xi);  // list is x-sorted (remove_isolated_pairs ends with the x sort)
    if (dx > T) break;
    double dy = __dsub_rn((double) y[q], yi);
    double d = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
    if (d <= T) uf_union(parent, (uint32_t) p, (uint32_t) q);
  }
}
__global__ __launch_bounds__(256) void k_ahc_labels(uint32_t *__restrict__ parent, uint64_t n, uint32_t *__restrict__ csize, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t r = uf_find(parent, (uint32_t) p);
  atomicAdd(&csize[r], 1u);
  keys[p] = ((uint64_t) r << 32) | (uint32_t) p;
  vals[p] = (uint32_t) p;
}
// after sorting by (component, position): rank inside the component, per-component capacities
__global__ __launch_bounds__(256) void k_ahc_rank(const uint64_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ rank, uint32_t *__restrict__ comp_first_sorted)
{
  uint64_t s = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  uint32_t c = (uint32_t) (keys[s] >> 32);
  if (s == 0 || (uint32_t) (keys[s - 1] >> 32) != c) comp_first_sorted[c] = (uint32_t) s;
}
__global__ __launch_bounds__(256) void k_ahc_rank2(const uint64_t *__restrict__ keys, uint64_t n, const uint32_t *__restrict__ comp_first_sorted, uint32_t *__restrict__ rank,
                                                   unsigned long long *__restrict__ ent_need, unsigned long long *__restrict__ cent_need,
                                                   unsigned long long *__restrict__ cpts_need, const uint32_t *__restrict__ csize)
{
  uint64_t s = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  uint32_t c = (uint32_t) (keys[s] >> 32), p = (uint32_t) keys[s];
  uint32_t r = (uint32_t) s - comp_first_sorted[c];
  rank[p] = r;
  ent_need[p] = r;  // a leaf's list holds the lower leaves of its component
  if (p == c)
  {
    unsigned long long sz = csize[c];
    cent_need[p] = sz > 1 ? sz * sz : 0ull;
    unsigned long long per = sz < 512 ? sz : 512ull;
    cpts_need[p] = sz > 1 ? sz * per + sz : 0ull;
  }
  else
  {
    cent_need[p] = 0;
    cpts_need[p] = 0;
  }
}

__global__ __launch_bounds__(256) void k_ahc_set_labels(const uint64_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ label)
{
  uint64_t s = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) label[(uint32_t) keys[s]] = (uint32_t) (keys[s] >> 32);
}

// ---- kernel B: literal replay of add_leaves + merge_clusters for one group per workgroup -------------------------
struct Best
{
  double d;
  int32_t j;
};
__device__ __forceinline__ bool best_less(double d1, int j1, double d2, int j2)
{
  // roots are scanned from the highest index with a strict `<` (util_cluster.cc:320-355)
  if (j1 < 0) return false;
  if (j2 < 0) return true;
  return d1 < d2 || (d1 == d2 && j1 > j2);
}
__device__ __forceinline__ bool cand_less(double d1, double o1, double d2, double o2) { return d1 < d2 || (d1 == d2 && o1 < o2); }

// block-wide minimum under best_less: butterfly inside the waves, four partial results through LDS (two barriers)
__device__ __forceinline__ void block_best(double &d, int &j, double *s_d, int32_t *s_j)
{
  for (int off = 32; off; off >>= 1)
  {
    const double od = __shfl_xor(d, off, 64);
    const int oj = __shfl_xor(j, off, 64);
    if (best_less(od, oj, d, j))
    {
      d = od;
      j = oj;
    }
  }
  if ((threadIdx.x & 63) == 0)
  {
    s_d[threadIdx.x >> 6] = d;
    s_j[threadIdx.x >> 6] = j;
  }
  __syncthreads();
  d = s_d[0];
  j = s_j[0];
  for (int w = 1; w < AT / 64; ++w)
    if (best_less(s_d[w], s_j[w], d, j))
    {
      d = s_d[w];
      j = s_j[w];
    }
  __syncthreads();
}
__device__ __forceinline__ int block_max(int v, int32_t *s_j)
{
  for (int off = 32; off; off >>= 1)
  {
    const int o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  if ((threadIdx.x & 63) == 0) s_j[threadIdx.x >> 6] = v;
  __syncthreads();
  v = s_j[0];
  for (int w = 1; w < AT / 64; ++w) v = s_j[w] > v ? s_j[w] : v;
  __syncthreads();
  return v;
}

struct AhcArgs
{
  // per element (leaf position p in the clustered list)
  const uint32_t *x, *y, *gof;
  const uint64_t *goff;
  uint32_t *comp, *rank, *csize;
  const uint32_t *sorted_pos;          // leaf positions sorted by (component, position)
  const uint32_t *comp_first_sorted;
  const unsigned long long *ent_off_leaf, *cent_off, *cpts_off;  // exclusive scans
  // node arrays (2n)
  int32_t *rootcomp, *cand_t;
  uint32_t *npts, *ent_cnt;
  uint64_t *pts_off, *ent_off;
  double *cand_d, *cand_o;
  // component arrays (n, indexed by leaf position of the component's smallest leaf)
  uint32_t *cnodes_cnt;
  unsigned long long *cent_used, *cpts_used;
  double *cbest_d;
  int32_t *cbest_j;
  int32_t *cmaxroot;    // per component: its root with the largest node index (the newest merged node, else the last leaf)
  uint32_t *act, *cnodes;              // act: n, cnodes: 2n
  Entry *entries;
  uint32_t *pts;
  uint2 *ptsxy;  // coordinates of the same points, same offsets: the linkage loops stream them without index chasing
  unsigned long long ent_leaf_total;
  double T;
  // outputs
  uint32_t *out_cnt;                   // per group: clustered elements
  uint32_t *out_idx_local, *out_cl;    // n each, group-local regions starting at goff[g]
  uint32_t *err;
  uint32_t ng;
  // (BK_DEBUG=ahc) per group: [0] merges, [1] sum over the merges of the longest chain of ordered additions (the largest m * n of
  // a (new node, root) pair: those additions depend on each other), [2] sum of all additions, [3] 10 ns ticks in the kernel
  unsigned long long *dbg;
};

// order keys insert_sorted leaves among the entries of one node.  Entries must be visited in DESCENDING target
// index (update_neighbours walks target-- from the new node, util_cluster.cc:112-134).  cross_top = largest index
// below the node that is a root of another component (or -1): such an entry is always farther than anything in
// the component that can still merge, so it is never the run in question but it does end "first is the tail".
__device__ void assign_ord_and_candidate(Entry *e, uint32_t cnt, int cross_top, const int32_t *rootcomp_base, int32_t &ct, double &cd, double &co)
{
  double maxd = -1.0;
  uint32_t cnt_at_max = 0;
  int t_first_at_max = -1;
  ct = -1;
  cd = 0;
  co = 0;
  for (uint32_t k = cnt; k-- > 0;)  // entries are stored in ascending target index
  {
    Entry &en = e[k];
    double ord = (double) en.t;
    if (en.d > maxd)
    {
      maxd = en.d;
      cnt_at_max = 1;
      t_first_at_max = en.t;
    }
    else if (en.d == maxd)
    {
      // second arrival of the run at the running maximum: the first one is the tail unless something farther
      // (a root of another component between the two indices) was inserted in between
      if (cnt_at_max == 1 && !(cross_top > en.t)) ord = (double) t_first_at_max + 0.5;
      ++cnt_at_max;
    }
    en.ord = ord;
  }
  for (uint32_t k = 0; k < cnt; ++k)
  {
    const Entry &en = e[k];
    if (rootcomp_base[en.t] < 0) continue;
    if (ct < 0 || cand_less(en.d, en.ord, cd, co))
    {
      ct = en.t;
      cd = en.d;
      co = en.ord;
    }
  }
}

// The same for the list of a freshly merged node, by the whole workgroup: the right-to-left walk is a suffix scan
// under the monoid (running maximum, how many entries equal it so far, target of the first of them), the candidate a
// minimum under cand_less.  s_m / s_c / s_t hold one partial result per wave.
struct RunState
{
  double m;
  uint32_t c;
  int32_t t;
};
__device__ __forceinline__ RunState run_join(const RunState &before, const RunState &cur)  // `before` was visited earlier (higher k)
{
  if (cur.m > before.m) return cur;
  if (cur.m == before.m)
  {
    RunState r = before;
    r.c += cur.c;
    return r;
  }
  return before;
}
__device__ __forceinline__ RunState run_shfl_up(const RunState &v, int d)
{
  RunState o;
  o.m = __shfl_up(v.m, d, 64);
  o.c = __shfl_up(v.c, d, 64);
  o.t = __shfl_up(v.t, d, 64);
  return o;
}
__device__ void block_assign_ord_and_candidate(Entry *e, uint32_t cnt, int cross_top, const int32_t *rootcomp_base, int32_t &ct, double &cd, double &co, double *s_m,
                                               uint32_t *s_c, int32_t *s_t)
{
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  RunState carry;
  carry.m = -1.0;
  carry.c = 0;
  carry.t = -1;
  for (uint32_t done = 0; done < cnt; done += AT)
  {
    const bool live = done + tid < cnt;
    const uint32_t k = live ? cnt - 1 - done - tid : 0;  // visiting order = descending k
    RunState v;
    v.m = -1.0;
    v.c = 0;
    v.t = -1;
    double dk = 0;
    int32_t tk = 0;
    if (live)
    {
      dk = e[k].d;
      tk = e[k].t;
      v.m = dk;
      v.c = 1;
      v.t = tk;
    }
    RunState inc = v;
    for (int d = 1; d < 64; d <<= 1)
    {
      const RunState o = run_shfl_up(inc, d);
      if (lane >= d) inc = run_join(o, inc);
    }
    if (lane == 63)
    {
      s_m[w] = inc.m;
      s_c[w] = inc.c;
      s_t[w] = inc.t;
    }
    __syncthreads();
    RunState ex = run_shfl_up(inc, 1);  // exclusive inside the wave
    if (lane == 0)
    {
      ex.m = -1.0;
      ex.c = 0;
      ex.t = -1;
    }
    RunState pre = carry, tot = carry;
    for (int i = 0; i < AT / 64; ++i)
    {
      RunState ws;
      ws.m = s_m[i];
      ws.c = s_c[i];
      ws.t = s_t[i];
      if (i < w) pre = run_join(pre, ws);
      tot = run_join(tot, ws);
    }
    const RunState st = run_join(pre, ex);  // everything visited before entry k
    if (live)
    {
      double ord = (double) tk;
      if (dk == st.m && st.c == 1 && !(cross_top > tk)) ord = (double) st.t + 0.5;
      e[k].ord = ord;
    }
    carry = tot;
    __syncthreads();
  }
  // candidate = minimum (d, ord) over the entries whose target is still a root
  int32_t bt = -1;
  double bd = 0, bo = 0;
  for (uint32_t k = tid; k < cnt; k += AT)
  {
    const Entry en = e[k];
    if (rootcomp_base[en.t] < 0) continue;
    if (bt < 0 || cand_less(en.d, en.ord, bd, bo))
    {
      bt = en.t;
      bd = en.d;
      bo = en.ord;
    }
  }
  for (int off = 32; off; off >>= 1)
  {
    const int32_t ot = __shfl_xor(bt, off, 64);
    const double od = __shfl_xor(bd, off, 64), oo = __shfl_xor(bo, off, 64);
    if (ot >= 0 && (bt < 0 || cand_less(od, oo, bd, bo)))
    {
      bt = ot;
      bd = od;
      bo = oo;
    }
  }
  __shared__ double s_o[AT / 64];
  if (lane == 0)
  {
    s_m[w] = bd;
    s_o[w] = bo;
    s_t[w] = bt;
  }
  __syncthreads();
  ct = s_t[0];
  cd = s_m[0];
  co = s_o[0];
  for (int i = 1; i < AT / 64; ++i)
    if (s_t[i] >= 0 && (ct < 0 || cand_less(s_m[i], s_o[i], cd, co)))
    {
      ct = s_t[i];
      cd = s_m[i];
      co = s_o[i];
    }
  __syncthreads();
}

__global__ __launch_bounds__(AT) void k_ahc_group(AhcArgs a)
{
  const uint32_t g = blockIdx.x;
  if (g >= a.ng) return;
  const uint64_t gs = a.goff[g], ge = a.goff[g + 1];
  const uint32_t N = (uint32_t) (ge - gs);
  __shared__ double s_d[AT];
  __shared__ int32_t s_j[AT];
  __shared__ uint32_t s_nact, s_nnodes, s_stop, s_cross;
  __shared__ int32_t s_smax;  // largest leaf that is a component of its own (a root of 'another component' for ever)
  __shared__ uint32_t s_scan4[AT / 64];
  __shared__ int32_t s_first, s_second;
  __shared__ uint32_t s_scan[AT];
  __shared__ uint32_t s_chain_max;
  __shared__ unsigned long long s_dbg[3];
  const unsigned long long t_in = wall_clock64();
  if (threadIdx.x == 0)
  {
    s_chain_max = 0;
    s_dbg[0] = s_dbg[1] = s_dbg[2] = 0ull;
  }
  if (N < 2)
  {
    if (threadIdx.x == 0) a.out_cnt[g] = 0;
    return;
  }
  const uint32_t tid = threadIdx.x;
  // group-local views (node index i lives at 2*gs + i, leaf i at gs + i, component id c at gs + c)
  const uint32_t *x = a.x + gs, *y = a.y + gs;
  uint32_t *comp = a.comp + gs, *rank = a.rank + gs, *csize = a.csize + gs;
  int32_t *rootcomp = a.rootcomp + 2 * gs, *cand_t = a.cand_t + 2 * gs;
  uint32_t *npts = a.npts + 2 * gs, *ent_cnt = a.ent_cnt + 2 * gs;
  uint64_t *pts_off = a.pts_off + 2 * gs, *ent_off = a.ent_off + 2 * gs;
  double *cand_d = a.cand_d + 2 * gs, *cand_o = a.cand_o + 2 * gs;
  uint32_t *cnodes_cnt = a.cnodes_cnt + gs, *act = a.act + gs;
  unsigned long long *cent_used = a.cent_used + gs, *cpts_used = a.cpts_used + gs;
  double *cbest_d = a.cbest_d + gs;
  int32_t *cbest_j = a.cbest_j + gs;
  int32_t *cmaxroot = a.cmaxroot + gs;
  GroupMem gm{};
  gm.x = x;
  gm.y = y;
  if (tid == 0)
  {
    s_nact = 0;
    s_nnodes = N;
    s_stop = 0;
    s_smax = -1;
  }
  __syncthreads();
  // ---- leaves: add_leaf + update_neighbours for every point (util_cluster.cc:86-110) ----
  for (uint32_t i = tid; i < N; i += AT)
  {
    const uint32_t c = comp[i] - (uint32_t) gs;  // component id local to the group
    comp[i] = c;
    rootcomp[i] = (int32_t) c;
    rootcomp[N + i] = -1;
    npts[i] = 1;
    // a leaf's point list is itself: stored in the shared pool right behind the merged-node regions
    pts_off[i] = ~0ull;
    ent_off[i] = a.ent_off_leaf[gs + i];
    ent_cnt[i] = rank[i];
  }
  __syncthreads();
  for (uint32_t c = tid; c < N; c += AT)
  {
    if (comp[c] == c)
    {
      cnodes_cnt[c] = 0;
      cent_used[c] = 0;
      cpts_used[c] = 0;
      cbest_j[c] = -1;
      cbest_d[c] = 0;
      if (csize[c] >= 2)
        act[atomicAdd(&s_nact, 1u)] = c;
      else
        atomicMax(&s_smax, (int32_t) c);
    }
  }
  __syncthreads();
  // component node lists: leaves in ascending index (sorted_pos is ordered by (component, position))
  for (uint32_t i = tid; i < N; i += AT)
  {
    const uint32_t c = comp[i];
    const uint64_t base = 2ull * (a.comp_first_sorted[gs + c]);  // capacity 2*csize per component, laid out by sorted start
    a.cnodes[base + rank[i]] = i;
    if (rank[i] + 1 == csize[c])
    {
      cnodes_cnt[c] = csize[c];
      cmaxroot[c] = (int32_t) i;
    }
  }
  __syncthreads();
  // leaf neighbour lists + first candidates (one lane per leaf; lists hold the lower leaves of the component)
  for (uint32_t j = tid; j < N; j += AT)
  {
    const uint32_t c = comp[j];
    const uint32_t r = rank[j];
    Entry *e = a.entries + ent_off[j];
    const uint32_t *members = a.cnodes + 2ull * a.comp_first_sorted[gs + c];
    for (uint32_t k = 0; k < r; ++k)
    {
      const int t = (int) members[k];
      e[k].t = t;
      e[k].d = leaf_dist(gm, (int) j, t);
      e[k].pad = 0;
    }
    // largest lower leaf of another component (every lower leaf is a root while leaves are added)
    int cross_top = -1;
    for (int q = (int) j - 1; q >= 0; --q)
      if (comp[q] != c)
      {
        cross_top = q;
        break;
      }
    int32_t ct;
    double cd, co;
    assign_ord_and_candidate(e, r, cross_top, rootcomp, ct, cd, co);
    cand_t[j] = ct;
    cand_d[j] = cd;
    cand_o[j] = co;
  }
  __syncthreads();
  // component bests
  for (uint32_t k = tid; k < s_nact; k += AT)
  {
    const uint32_t c = act[k];
    const uint32_t *members = a.cnodes + 2ull * a.comp_first_sorted[gs + c];
    double bd = 0;
    int bj = -1;
    for (uint32_t q = 0; q < cnodes_cnt[c]; ++q)
    {
      int j = (int) members[q];
      if (cand_t[j] >= 0 && best_less(cand_d[j], j, bd, bj))
      {
        bd = cand_d[j];
        bj = j;
      }
    }
    cbest_d[c] = bd;
    cbest_j[c] = bj;
  }
  __syncthreads();
  // ---- merge_clusters (util_cluster.cc:299-318) ----
  uint32_t n_roots = N;
  while (n_roots > 1)
  {
    // find_cluster_to_merge: global best over all roots = best over the component bests
    double bd = 0;
    int bj = -1;
    for (uint32_t k = tid; k < s_nact; k += AT)
    {
      const uint32_t c = act[k];
      if (best_less(cbest_d[c], cbest_j[c], bd, bj))
      {
        bd = cbest_d[c];
        bj = cbest_j[c];
      }
    }
    block_best(bd, bj, s_d, s_j);
    if (tid == 0)
    {
      const int j = bj;
      if (j < 0 || !(bd <= a.T))
        s_stop = 1;
      else
      {
        s_first = j;
        s_second = cand_t[j];
      }
    }
    __syncthreads();
    if (s_stop) break;
    const int first = s_first, second = s_second;
    const uint32_t c = (uint32_t) rootcomp[first];
    const int q = (int) s_nnodes;  // index of the merged node
    const uint32_t mf = npts[first], ms = npts[second], mq = mf + ms;
    uint32_t *members = a.cnodes + 2ull * a.comp_first_sorted[gs + c];
    const uint32_t ncn = cnodes_cnt[c];
    // merged point list = first's points then second's (:379-382)
    uint32_t *qpts = a.pts + a.cpts_off[gs + c] + cpts_used[c];
    uint2 *qxy = a.ptsxy + a.cpts_off[gs + c] + cpts_used[c];
    __syncthreads();
    if (tid == 0)
    {
      const unsigned long long cap = (unsigned long long) csize[c] * (csize[c] < 512 ? csize[c] : 512u) + csize[c];
      if (cpts_used[c] + mq > cap || cent_used[c] + ncn > (unsigned long long) csize[c] * csize[c])
      {
        atomicOr(a.err, 1u);
        s_stop = 1;
      }
    }
    __syncthreads();
    if (s_stop) break;
    for (uint32_t k = tid; k < mq; k += AT)
    {
      const int src = k < mf ? first : second;
      const uint32_t kk = k < mf ? k : k - mf;
      const uint32_t leaf = src < (int) N ? (uint32_t) src : (a.pts + pts_off[src])[kk];
      qpts[k] = leaf;
      qxy[k] = make_uint2(x[leaf], y[leaf]);
    }
    Entry *qe = a.entries + a.ent_leaf_total + a.cent_off[gs + c] + cent_used[c];
    __syncthreads();
    if (tid == 0)
    {
      rootcomp[first] = -1;
      rootcomp[second] = -1;
      rootcomp[q] = (int32_t) c;
      npts[q] = mq;
      pts_off[q] = (uint64_t) (qpts - a.pts);
      ent_off[q] = (uint64_t) (qe - a.entries);
      cpts_used[c] += mq;
      s_cross = 0;
      s_scan[0] = 0;
    }
    __syncthreads();
    // update_neighbours for the merged node: distance to every current root of the component (others are farther
    // than T by construction).  Entries in ascending index.  The sequential average_linkage sum (:201-215) fixes the
    // order of the additions, not of the distances: pairs with little work take one lane each, the others one wave
    // each - 64 distances at a time in parallel, then added one after the other in list order.
    uint32_t nent = 0;
    for (uint32_t base = 0; base < ncn; base += AT)
    {
      const uint32_t k = base + tid;
      const int t = k < ncn ? (int) members[k] : -1;
      const uint32_t isr = (t >= 0 && rootcomp[t] >= 0) ? 1u : 0u;
      uint32_t tot;
      const uint32_t pos = nent + prims::block_exclusive_scan(isr, s_scan4, tot);
      if (isr)
      {
        Entry en;
        en.t = t;
        en.d = 0;
        en.ord = 0;
        en.pad = 0;
        qe[pos] = en;
      }
      nent += tot;
    }
    __syncthreads();
    constexpr uint32_t WAVE_WORK = 192;  // pairs with at least this many distances go to a whole wave
    for (uint32_t idx = tid; idx < nent; idx += AT)
    {
      const int t = qe[idx].t;
      const uint32_t mt = npts[t];
      if (a.dbg)
      {
        atomicMax(&s_chain_max, mq * mt);
        atomicAdd(&s_dbg[2], (unsigned long long) mq * mt);
      }
      if (mq * mt >= WAVE_WORK) continue;
      double total = 0.0;
      const uint2 leaf_xy = t < (int) N ? make_uint2(x[t], y[t]) : make_uint2(0u, 0u);
      const uint2 *txy = t < (int) N ? &leaf_xy : a.ptsxy + pts_off[t];
      for (uint32_t i = 0; i < mq; ++i)
      {
        const uint2 pi = qxy[i];
        const double xi = (double) pi.x, yi = (double) pi.y;
        for (uint32_t jj = 0; jj < mt; ++jj) total = __dadd_rn(total, xy_dist(xi, yi, txy[jj]));
      }
      qe[idx].d = __ddiv_rn(total, (double) (int) (mq * mt));  // total / (m * n) with an int product
    }
    {
      const uint32_t lane = tid & 63, wv = tid >> 6;
      for (uint32_t idx = wv; idx < nent; idx += AT / 64)
      {
        const int t = __builtin_amdgcn_readfirstlane(qe[idx].t);  // the same entry for the whole wave
        const uint32_t mt = (uint32_t) __builtin_amdgcn_readfirstlane((int) npts[t]);
        const uint32_t work = (uint32_t) __builtin_amdgcn_readfirstlane((int) mq) * mt;
        if (work < WAVE_WORK) continue;
        const uint2 *txy = t < (int) N ? nullptr : a.ptsxy + pts_off[t];
        const uint2 leaf_xy = t < (int) N ? make_uint2(x[t], y[t]) : make_uint2(0u, 0u);
        double total = 0.0;
        for (uint32_t e0 = 0; e0 < work; e0 += 64)
        {
          const uint32_t el = e0 + lane;
          double d = 0.0;
          if (el < work)
          {
            const uint32_t i = el / mt, jj = el - i * mt;
            const uint2 pi = qxy[i];
            d = xy_dist((double) pi.x, (double) pi.y, txy ? txy[jj] : leaf_xy);
          }
          const uint32_t cnt = work - e0 < 64 ? work - e0 : 64;
          const int dlo = __double2loint(d), dhi = __double2hiint(d);
          // ordered additions, one v_readlane pair + one add each (full chunks unrolled: constant lane numbers)
          if (cnt == 64)
          {
#pragma unroll
            for (int l = 0; l < 64; ++l) total = __dadd_rn(total, __hiloint2double(__builtin_amdgcn_readlane(dhi, l), __builtin_amdgcn_readlane(dlo, l)));
          }
          else
            for (uint32_t l = 0; l < cnt; ++l)
              total = __dadd_rn(total, __hiloint2double(__builtin_amdgcn_readlane(dhi, (int) l), __builtin_amdgcn_readlane(dlo, (int) l)));
        }
        if (lane == 0) qe[idx].d = __ddiv_rn(total, (double) (int) work);
      }
    }
    __syncthreads();
    if (a.dbg && tid == 0)
    {
      s_dbg[0] += 1ull;
      s_dbg[1] += s_chain_max;
      s_chain_max = 0;
    }
    // largest root index below q that belongs to another component: the newest root of every other component
    // (its latest merged node, or its last leaf before the first merge), or a leaf that is a component of its own
    {
      int mine = s_smax;
      for (uint32_t k = tid; k < s_nact; k += AT)
      {
        const uint32_t c2 = act[k];
        if (c2 != c && cmaxroot[c2] > mine) mine = cmaxroot[c2];
      }
      const int found = block_max(mine, s_j);
      if (tid == 0) s_first = found;  // reuse as cross_top
    }
    __syncthreads();
    {
      int32_t ct;
      double cd, co;
      block_assign_ord_and_candidate(qe, nent, s_first, rootcomp, ct, cd, co, s_d, s_scan, s_j);
      if (tid == 0)
      {
        ent_cnt[q] = nent;
        cent_used[c] += nent;
        cand_t[q] = ct;
        cand_d[q] = cd;
        cand_o[q] = co;
        members[ncn] = (uint32_t) q;
        cnodes_cnt[c] = ncn + 1;
        cmaxroot[c] = q;
        s_nnodes = (uint32_t) q + 1;
      }
    }
    __syncthreads();
    // roots whose candidate was one of the merged nodes look further down their list (:341-354): the affected
    // roots are collected, then every wave takes one of them and scans its list with all lanes
    {
      if (tid == 0) s_cross = 0;
      __syncthreads();
      for (uint32_t k = tid; k < ncn; k += AT)
      {
        const int r = (int) members[k];
        if (rootcomp[r] < 0) continue;
        if (cand_t[r] == first || cand_t[r] == second)
        {
          const uint32_t slot = atomicAdd(&s_cross, 1u);
          if (slot < AT)
            s_scan[slot] = (uint32_t) r;
          else
          {
            // more affected roots than list slots (never seen; kept exact): this lane scans its own list
            const Entry *e = a.entries + ent_off[r];
            int32_t ct = -1;
            double cd = 0, co = 0;
            for (uint32_t z = 0; z < ent_cnt[r]; ++z)
            {
              if (rootcomp[e[z].t] < 0) continue;
              if (ct < 0 || cand_less(e[z].d, e[z].ord, cd, co))
              {
                ct = e[z].t;
                cd = e[z].d;
                co = e[z].ord;
              }
            }
            cand_t[r] = ct;
            cand_d[r] = cd;
            cand_o[r] = co;
          }
        }
      }
      __syncthreads();
      const uint32_t naff = s_cross < AT ? s_cross : AT;
      const uint32_t lane = tid & 63;
      for (uint32_t w = tid >> 6; w < naff; w += AT / 64)
      {
        const int r = (int) s_scan[w];
        const Entry *e = a.entries + ent_off[r];
        const uint32_t ne = ent_cnt[r];
        int32_t ct = -1;
        double cd = 0, co = 0;
        for (uint32_t z = lane; z < ne; z += 64)
        {
          const Entry en = e[z];
          if (rootcomp[en.t] < 0) continue;
          if (ct < 0 || cand_less(en.d, en.ord, cd, co))
          {
            ct = en.t;
            cd = en.d;
            co = en.ord;
          }
        }
        for (int off = 32; off; off >>= 1)
        {
          const int32_t ot = __shfl_xor(ct, off, 64);
          const double od = __shfl_xor(cd, off, 64), oo = __shfl_xor(co, off, 64);
          if (ot >= 0 && (ct < 0 || cand_less(od, oo, cd, co)))
          {
            ct = ot;
            cd = od;
            co = oo;
          }
        }
        if (lane == 0)
        {
          cand_t[r] = ct;
          cand_d[r] = cd;
          cand_o[r] = co;
        }
      }
    }
    __syncthreads();
    // component best
    {
      double d0 = 0;
      int j0 = -1;
      for (uint32_t k = tid; k < ncn + 1; k += AT)
      {
        const int r = (int) members[k];
        if (rootcomp[r] >= 0 && cand_t[r] >= 0 && best_less(cand_d[r], r, d0, j0))
        {
          d0 = cand_d[r];
          j0 = r;
        }
      }
      block_best(d0, j0, s_d, s_j);
      if (tid == 0)
      {
        cbest_d[c] = d0;
        cbest_j[c] = j0;
      }
    }
    __syncthreads();
    --n_roots;
  }
  __syncthreads();
  // ---- add_cluster_id_for_enspan_vec (BreakID.cc:1328-1352): roots with >= 2 points, node-index order ----
  const uint32_t nn = s_nnodes;
  uint32_t kbase = 0, obase = 0;
  for (uint32_t base = 0; base < nn; base += AT)
  {
    const uint32_t i = base + tid;
    const bool keep = i < nn && rootcomp[i] >= 0 && npts[i] >= 2;
    // scan of flags (cluster number) and of sizes (output offset)
    s_scan[tid] = keep ? 1u : 0u;
    __syncthreads();
    for (int off = 1; off < AT; off <<= 1)
    {
      uint32_t v = (int) tid >= off ? s_scan[tid - off] : 0u;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    const uint32_t kex = s_scan[tid] - (keep ? 1u : 0u), ktot = s_scan[AT - 1];
    __syncthreads();
    s_scan[tid] = keep ? npts[i] : 0u;
    __syncthreads();
    for (int off = 1; off < AT; off <<= 1)
    {
      uint32_t v = (int) tid >= off ? s_scan[tid - off] : 0u;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    const uint32_t oex = s_scan[tid] - (keep ? npts[i] : 0u), otot = s_scan[AT - 1];
    __syncthreads();
    if (keep)
    {
      const uint32_t *pp = a.pts + pts_off[i];
      for (uint32_t z = 0; z < npts[i]; ++z)
      {
        a.out_idx_local[gs + obase + oex + z] = pp[z];
        a.out_cl[gs + obase + oex + z] = kbase + kex;
      }
    }
    kbase += ktot;
    obase += otot;
  }
  if (tid == 0) a.out_cnt[g] = obase;
  if (a.dbg && tid == 0)
  {
    a.dbg[4 * g] = s_dbg[0];
    a.dbg[4 * g + 1] = s_dbg[1];
    a.dbg[4 * g + 2] = s_dbg[2];
    a.dbg[4 * g + 3] = wall_clock64() - t_in;
  }
}

__global__ __launch_bounds__(256) void k_gather_xy(const bk_pair *__restrict__ pairs, const uint32_t *__restrict__ idx, uint64_t n, uint32_t *__restrict__ x, uint32_t *__restrict__ y,
                                                   uint32_t *__restrict__ parent)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const bk_pair pr = pairs[idx[p]];
  x[p] = pr.x;
  y[p] = pr.y;
  parent[p] = (uint32_t) p;
}
// out position of element z of group g = new_goff[g] + z
__global__ __launch_bounds__(256) void k_ahc_emit(const uint32_t *__restrict__ out_idx_local, const uint32_t *__restrict__ out_cl, const uint32_t *__restrict__ out_cnt,
                                                  const uint64_t *__restrict__ goff, const uint32_t *__restrict__ newoff, const uint32_t *__restrict__ old_idx, uint32_t ng,
                                                  uint32_t *__restrict__ idx_out, uint32_t *__restrict__ gof_out, uint32_t *__restrict__ cl_out, uint64_t *__restrict__ goff_out)
{
  const uint32_t g = blockIdx.x;
  if (g >= ng) return;
  const uint64_t gs = goff[g];
  const uint32_t cnt = out_cnt[g], o = newoff[g];
  for (uint32_t z = threadIdx.x; z < cnt; z += blockDim.x)
  {
    idx_out[o + z] = old_idx[gs + out_idx_local[gs + z]];
    gof_out[o + z] = g;
    cl_out[o + z] = out_cl[gs + z];
  }
  if (threadIdx.x == 0)
  {
    goff_out[g] = o;
    if (g + 1 == ng) goff_out[ng] = o + cnt;
  }
}
}  // namespace

static inline unsigned nbk(uint64_t n) { return cdiv(n ? n : 1, 256); }

void ahc_cluster_all(const bk_pair *pairs, PairList &L, double w, DevBuf &cluster_out, AhcBufs &ab, ClusterBufs &cb, hipStream_t st)
{
  drop_small_groups(L, cb, st);  // groups with fewer than 2 pairs are not clustered (BreakID.cc:125)
  const uint64_t n = L.n;
  const uint32_t ng = L.ng;
  (void) cluster_out.as<uint32_t>(n + 1);
  if (n == 0 || ng == 0) return;
  if (n > 0x3FFFFFFFull) throw bk_error(BK_ERR_LIMIT, "AHC: too many points");
  const double T = (double) (long) w;  // init_cluster takes `long distance_threshold` (util_cluster.cc:7)
  uint32_t *x = ab.x.as<uint32_t>(n), *y = ab.y.as<uint32_t>(n), *parent = ab.comp.as<uint32_t>(n);
  hipLaunchKernelGGL(k_gather_xy, dim3(nbk(n)), dim3(256), 0, st, pairs, L.idx.get<uint32_t>(), n, x, y, parent);
  hipLaunchKernelGGL(k_ahc_edges, dim3(nbk(n)), dim3(256), 0, st, x, y, L.gof.get<uint32_t>(), L.goff.get<uint64_t>(), n, T, parent);
  uint32_t *csize = ab.csize.as<uint32_t>(n);
  uint64_t *keys = ab.keys.as<uint64_t>(n);
  uint32_t *vals = ab.vals.as<uint32_t>(n);
  HIP_CHECK(hipMemsetAsync(csize, 0, n * 4, st));
  hipLaunchKernelGGL(k_ahc_labels, dim3(nbk(n)), dim3(256), 0, st, parent, n, csize, keys, vals);
  // after k_ahc_labels parent[] is not fully compressed: store the final label per leaf
  uint64_t *ks;
  uint32_t *vs;
  prims::radix_sort_pairs(keys, vals, n, 0, 64, ab.radix, st, &ks, &vs);
  uint32_t *rank = ab.rank.as<uint32_t>(n), *cfs = ab.cfs.as<uint32_t>(n);
  unsigned long long *ent_need = ab.need0.as<unsigned long long>(n + 1), *cent_need = ab.need1.as<unsigned long long>(n + 1),
                     *cpts_need = ab.need2.as<unsigned long long>(n + 1);
  hipLaunchKernelGGL(k_ahc_rank, dim3(nbk(n)), dim3(256), 0, st, ks, n, rank, cfs);
  hipLaunchKernelGGL(k_ahc_rank2, dim3(nbk(n)), dim3(256), 0, st, ks, n, cfs, rank, ent_need, cent_need, cpts_need, csize);
  prims::exclusive_scan<unsigned long long>(ent_need, ent_need, n, ab.scan_tmp, st);
  prims::exclusive_scan<unsigned long long>(cent_need, cent_need, n, ab.scan_tmp, st);
  prims::exclusive_scan<unsigned long long>(cpts_need, cpts_need, n, ab.scan_tmp, st);
  unsigned long long tot[3] = {0, 0, 0};
  HIP_CHECK(hipMemcpyAsync(&tot[0], ent_need + n, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&tot[1], cent_need + n, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&tot[2], cpts_need + n, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  const unsigned long long n_entries = tot[0] + tot[1];
  if (n_entries * sizeof(Entry) > (64ull << 30) || tot[2] * 4 > (32ull << 30))
    throw bk_error(BK_ERR_LIMIT, "AHC: a connected component is too large for the exact replay (the reference would need an N x N matrix of doubles)");
  AhcArgs a{};
  a.x = x;
  a.y = y;
  a.gof = L.gof.get<uint32_t>();
  a.goff = L.goff.get<uint64_t>();
  a.comp = ab.label.as<uint32_t>(n);
  // component label per leaf = high half of the sorted key, scattered back by position
  a.rank = rank;
  a.csize = csize;
  a.sorted_pos = vs;
  a.comp_first_sorted = cfs;
  a.ent_off_leaf = ent_need;
  a.cent_off = cent_need;
  a.cpts_off = cpts_need;
  a.rootcomp = ab.rootcomp.as<int32_t>(2 * n);
  a.cand_t = ab.cand_t.as<int32_t>(2 * n);
  a.npts = ab.npts.as<uint32_t>(2 * n);
  a.ent_cnt = ab.ent_cnt.as<uint32_t>(2 * n);
  a.pts_off = ab.pts_off.as<uint64_t>(2 * n);
  a.ent_off = ab.ent_off.as<uint64_t>(2 * n);
  a.cand_d = ab.cand_d.as<double>(2 * n);
  a.cand_o = ab.cand_o.as<double>(2 * n);
  a.cnodes_cnt = ab.cnodes_cnt.as<uint32_t>(n);
  a.cent_used = ab.cent_used.as<unsigned long long>(n);
  a.cpts_used = ab.cpts_used.as<unsigned long long>(n);
  a.cbest_d = ab.cbest_d.as<double>(n);
  a.cbest_j = ab.cbest_j.as<int32_t>(n);
  a.cmaxroot = ab.cmaxroot.as<int32_t>(n);
  a.act = ab.act.as<uint32_t>(n);
  a.cnodes = ab.cnodes.as<uint32_t>(2 * n);
  a.entries = ab.entries.as<Entry>(n_entries + 1);
  a.pts = ab.pts.as<uint32_t>(tot[2] + 1);
  a.ptsxy = ab.ptsxy.as<uint2>(tot[2] + 1);
  a.ent_leaf_total = tot[0];
  a.T = T;
  a.out_cnt = ab.out_cnt.as<uint32_t>((uint64_t) ng + 1);
  a.out_idx_local = ab.out_idx.as<uint32_t>(n);
  a.out_cl = ab.out_cl.as<uint32_t>(n);
  a.err = ab.err.as<uint32_t>(4);
  a.ng = ng;
  static const bool dbg_ahc = bk_debug("ahc");
  DevBuf dbg_buf;
  a.dbg = nullptr;
  if (dbg_ahc)
  {
    a.dbg = dbg_buf.as<unsigned long long>(4ull * ng + 4);
    HIP_CHECK(hipMemsetAsync(a.dbg, 0, (4ull * ng + 4) * 8, st));
  }
  HIP_CHECK(hipMemsetAsync(a.err, 0, 16, st));
  hipLaunchKernelGGL(k_ahc_set_labels, dim3(nbk(n)), dim3(256), 0, st, ks, n, a.comp);
  hipLaunchKernelGGL(k_ahc_group, dim3(ng), dim3(AT), 0, st, a);
  uint32_t *newoff = ab.newoff.as<uint32_t>((uint64_t) ng + 1);
  prims::exclusive_scan<uint32_t>(a.out_cnt, newoff, ng, ab.scan_tmp, st);
  uint32_t total = 0, err = 0;
  HIP_CHECK(hipMemcpyAsync(&total, newoff + ng, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&err, a.err, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  if (dbg_ahc)
  {
    // the group that took longest: its merges, the ordered additions on its critical path and what one of them cost
    std::vector<unsigned long long> h(4ull * ng);
    HIP_CHECK(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
    uint32_t gm = 0;
    unsigned long long all_adds = 0, all_chain = 0;
    for (uint32_t g = 0; g < ng; ++g)
    {
      if (h[4 * g + 3] > h[4 * gm + 3]) gm = g;
      all_adds += h[4 * g + 2];
      all_chain += h[4 * g + 1];
    }
    fprintf(stderr, "[ahc] %u groups, %llu ordered additions in all (%llu on the groups' critical paths); slowest group %u: %.3f ms, %llu merges, %llu additions of which %llu depend on each other: %.1f ns of kernel time per dependent addition, %.2f us per merge\n",
            ng, all_adds, all_chain, gm, (double) h[4 * gm + 3] * 1e-5, h[4 * gm], h[4 * gm + 2], h[4 * gm + 1], h[4 * gm + 1] ? (double) h[4 * gm + 3] * 10.0 / (double) h[4 * gm + 1] : 0.0,
            h[4 * gm] ? (double) h[4 * gm + 3] * 1e-2 / (double) h[4 * gm] : 0.0);
  }
  if (err) throw bk_error(BK_ERR_LIMIT, "AHC: point/entry pool of a component overflowed (very deep merge chain)");
  uint32_t *oidx = cb.idx2.as<uint32_t>((uint64_t) total + 1), *ogof = cb.gof2.as<uint32_t>((uint64_t) total + 1);
  uint64_t *ogoff = cb.goff2.as<uint64_t>((uint64_t) ng + 1);
  uint32_t *cl = cluster_out.as<uint32_t>((uint64_t) total + 1);
  hipLaunchKernelGGL(k_ahc_emit, dim3(ng), dim3(256), 0, st, a.out_idx_local, a.out_cl, a.out_cnt, L.goff.get<uint64_t>(), newoff, L.idx.get<uint32_t>(), ng, oidx, ogof, cl, ogoff);
  std::swap(L.idx, cb.idx2);
  std::swap(L.gof, cb.gof2);
  std::swap(L.goff, cb.goff2);
  L.n = total;
}
