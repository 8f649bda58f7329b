// Exact, data-parallel emulation of libstdc++'s std::sort (introsort) on every group at once.
//
// The reference sorts its pair vectors with *unstable* std::sort (BreakID.cc:1091,1127,1274,1278,1282)
// and then reads neighbours of the sorted vector (mask_pairs_chr_pos :1847-1858, fast clustering
// :1064,:1100), so the order std::sort leaves equal keys in is observable.  std::sort is
// deterministic given the comparison outcomes:
//   __introsort_loop: while (size > 16) { median-of-3 of (first+1, mid, last-1) swapped to first;
//                     Hoare partition of [first+1,last) around *first; recurse right, loop left }
//   __final_insertion_sort: stable for equal keys.
// so the result = a stable sort by key of the array as the introsort loop leaves it.  One partition
// level is a data-parallel step: with l_j the j-th position (from the left) whose key >= pivot and r_j
// the j-th position (from the right) whose key <= pivot, the loop swaps (l_j, r_j) for every j < J,
// J = #{j : l_j < r_j}, and returns cut = min(l_J, r_{J-1}) (l_0 when J = 0).  All segments of all groups
// advance one level per pass (prefix sums give the ranks).
//
// Layout of this file, in the order a sort goes through it:
//   device-wide level loop (k_se_pivot / flags / lists / swap / child_*)  segments of more than FIN_MAX elements, compact index space
//   k_se_finish                                                          segments of at most FIN_MAX elements: the same loop in LDS
//   k_hr_* + k_se_heapsort<0,1,2>                                        segments that hit the depth limit: make_heap + pipelined sort_heap
//   k_se_window_sort                                                     __final_insertion_sort as two tilings of stable window sorts
#include "bk_common.h"
#include "prims.h"
#include "sortemu.h"
#include <vector>
#include <mutex>
#include <algorithm>
#include <cstdio>
#include <chrono>
#include <thread>
#include <cstdlib>
#include <ctime>

namespace
{
inline double now_ms()
{
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
// A store that goes through to memory (sc1): what another workgroup of a running kernel reads after its acquire without the
// storing workgroup having to write back its whole L2 (resident sort service, sortsvc.inc).  p is a global address.
typedef __attribute__((address_space(1))) uint32_t gu32_t;
__device__ __forceinline__ void st_through(uint32_t *p, uint32_t v) { __hip_atomic_store((gu32_t *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Live segments of one level are laid out back to back in a compact index space (cbase = first compact index of
// the segment): the per-level passes touch only elements that are still being partitioned, not all n positions.
struct Seg
{
  uint32_t first, last;
  uint32_t pivot;
  int32_t depth;
  uint32_t cut;
  uint32_t cbase;
};

// depth_limit = 2 * floor(log2(n))  (std::__lg(n) * 2)
// Segments of up to FIN_MAX elements leave the device-wide level loop: one workgroup finishes each of them in LDS
// (k_se_finish).  fin[0] = number of entries of the finisher list.
constexpr uint32_t FIN_MAX = 2048;
constexpr uint32_t HEAP_BIG_MIN = 4096;  // = HEAP_RANKED_MIN: segments above it are heapsorted one per CU (k_se_heapsort<2>)
// The partition passes work on tiles of LV_TILE compact indices (k_lv_*).  Whoever writes a level's segment list also leaves,
// for every tile, the segment that owns the tile's first index: a tile then starts with one load instead of a binary search
// over the list (seven dependent round trips that were most of a late level's kernel time).
constexpr uint32_t LV_TILE = 2048;
__device__ __forceinline__ void lv_mark_tiles(uint32_t *__restrict__ tile_seg, uint32_t s, uint32_t cbase, uint32_t size)
{
  if (!tile_seg) return;
  for (uint32_t t = (cbase + LV_TILE - 1) / LV_TILE; (unsigned long long) t * LV_TILE < (unsigned long long) cbase + size; ++t) tile_seg[t] = s;
}
struct FinSeg
{
  uint32_t first, last;
  int32_t depth;
};
// two lists: segments of at most FIN_SMALL elements (one wavefront each) from the front, the others from the back
constexpr uint32_t FIN_SMALL = 256;
__device__ __forceinline__ void fin_append(FinSeg *__restrict__ fl, uint32_t *__restrict__ fin, uint32_t first, uint32_t last, int32_t depth)
{
  FinSeg f;
  f.first = first;
  f.last = last;
  f.depth = depth;
  if (last - first <= FIN_SMALL)
    fl[atomicAdd(fin, 1u)] = f;
  else
    fl[fin[2] - 1 - atomicAdd(fin + 1, 1u)] = f;  // fin[2] = capacity of the list
}

// the state words of one sort in ONE block of device memory: err[8] (flags, longest heap segment, elements in heap segments, their
// number, the long ones, last slot of the heap list), fin[4] (finisher list: small count, large count, capacity), lvl[4] (live
// segments, live elements, largest live segment, spare) - one small kernel resets them, one copy reads them all back
constexpr uint32_t ST_ERR = 0, ST_FIN = 8, ST_LVL = 12, ST_WORDS = 16;
__global__ void k_se_reset(uint32_t *state, uint32_t fin_cap, uint32_t last_slot)
{
  const uint32_t t = threadIdx.x;
  if (t < ST_WORDS) state[t] = t == ST_FIN + 2 ? fin_cap : (t == ST_ERR + 5 ? last_slot : 0u);
}
// cnt entries: (#segments) | (#elements in them) << 32, scanned together
__global__ void k_se_init(const uint64_t *__restrict__ goff, uint32_t ng, unsigned long long *__restrict__ cnt, FinSeg *__restrict__ fl, uint32_t *__restrict__ fin,
                          uint32_t *__restrict__ max_size)
{
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  uint64_t sz = goff[g + 1] - goff[g];
  cnt[g] = sz > FIN_MAX ? (1ull | (sz << 32)) : 0ull;
  if (sz > FIN_MAX) atomicMax(max_size, (uint32_t) sz);  // largest live segment of level 0
  if (sz > 16 && sz <= FIN_MAX) fin_append(fl, fin, (uint32_t) goff[g], (uint32_t) goff[g + 1], 2 * (63 - __clzll((long long) sz)));
}
__global__ void k_se_init_write(const uint64_t *__restrict__ goff, uint32_t ng, const unsigned long long *__restrict__ off, Seg *__restrict__ segs, uint32_t *__restrict__ tile_seg,
                                uint32_t *__restrict__ lvl)
{
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  if (g == 0)
  {
    lvl[0] = (uint32_t) off[ng];  // live segments / elements of level 0 for the level kernels that read their counts on the device
    lvl[1] = (uint32_t) (off[ng] >> 32);
  }
  uint64_t sz = goff[g + 1] - goff[g];
  if (sz > FIN_MAX)
  {
    Seg s;
    s.first = (uint32_t) goff[g];
    s.last = (uint32_t) goff[g + 1];
    s.pivot = 0;
    s.depth = 2 * (63 - __clzll((long long) sz));
    s.cut = 0;
    s.cbase = (uint32_t) (off[g] >> 32);
    segs[(uint32_t) off[g]] = s;
    lv_mark_tiles(tile_seg, (uint32_t) off[g], s.cbase, (uint32_t) sz);
  }
}

// __move_median_to_first(first, first+1, mid, last-1) of one live segment; a segment whose depth budget is used up goes
// to the heap list instead (depth = -1)
__device__ __forceinline__ void pivot_one(Seg &sg, uint32_t *__restrict__ key, uint32_t *__restrict__ idx, uint32_t *__restrict__ err, uint2 *__restrict__ heap_list)
{
  if (sg.depth == 0)
  {
    // std::sort switches to heapsort here (__partial_sort(first,last,last)); k_se_heapsort finishes the segment
    sg.depth = -1;
    uint32_t slot = atomicAdd(err + 3, 1u);
    heap_list[slot] = make_uint2(sg.first, sg.last);
    // the long ones a second time, from the END of the list downwards (err[5] = its last slot): their kernel takes a whole CU's LDS
    // per workgroup, so it is launched over them only, not over every segment of the list
    if (sg.last - sg.first > HEAP_BIG_MIN) heap_list[err[5] - atomicAdd(err + 4, 1u)] = make_uint2(sg.first, sg.last);
    atomicAdd(err + 2, sg.last - sg.first);
    atomicMax(err + 1, sg.last - sg.first);
    return;
  }
  uint32_t first = sg.first, last = sg.last;
  uint32_t a = first + 1, b = first + (last - first) / 2, c = last - 1;
  uint32_t ka = key[a], kb = key[b], kc = key[c];
  uint32_t pick;
  if (ka < kb)
  {
    if (kb < kc) pick = b;
    else if (ka < kc) pick = c;
    else pick = a;
  }
  else if (ka < kc) pick = a;
  else if (kb < kc) pick = c;
  else pick = b;
  uint32_t kf = key[first], kp = key[pick];
  uint32_t xf = idx[first], xp = idx[pick];
  key[first] = kp;
  key[pick] = kf;
  idx[first] = xp;
  idx[pick] = xf;
  sg.pivot = kp;
  sg.depth = sg.depth - 1;
}
__global__ void k_se_pivot(Seg *__restrict__ segs, uint32_t ns, uint32_t *__restrict__ key, uint32_t *__restrict__ idx, uint32_t *__restrict__ err, uint2 *__restrict__ heap_list)
{
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  Seg sg = segs[s];
  pivot_one(sg, key, idx, err, heap_list);
  segs[s].pivot = sg.pivot;
  segs[s].depth = sg.depth;
}

// ---- heapsort branch of std::sort (__partial_sort(first,last,last) = make_heap + sort_heap) ----------------
// libstdc++'s __adjust_heap moves the hole to the bottom along the larger-child path and pushes the value back
// up with a strict compare; the net effect equals a top-down sift that stops at the first node whose larger
// child is < value (ties between children go to the right child, equal child keeps descending).  In that form
// every write is final when it is made, so
//   * make_heap runs level by level (nodes of one depth own disjoint subtrees), and
//   * the pops of sort_heap are pipelined inside one wavefront: pop t+1 starts two steps behind pop t and
//     stalls while an in-flight pop could still reach the leaf it is about to detach (ancestor test).
// Verified on the host against std::partial_sort on tie-heavy inputs.
struct HeapSeg
{
  uint32_t first, last;
};

// heap entries are packed (key << 32 | idx): one 8-byte access moves an element, the two children of a node are
// adjacent.  Only the key half takes part in comparisons.
typedef unsigned long long hent;
__device__ __forceinline__ uint32_t hkey(hent e) { return (uint32_t) (e >> 32); }

// A second entry format serves heaps of 20 001 .. 65 536 elements: (rank of the key inside the segment) << 16 | local
// index, 4 bytes, so that twice as much of the heap fits LDS (the keys are ranked by one radix sort before the launch).
struct E64
{
  typedef hent T;
  static __device__ __forceinline__ uint32_t key(T e) { return (uint32_t) (e >> 32); }
};
struct E32
{
  typedef uint32_t T;
  static __device__ __forceinline__ uint32_t key(T e) { return e >> 16; }
};

template <class E> struct LdsMemT
{
  typedef typename E::T T;
  T *e;
  static __device__ __forceinline__ uint32_t key(T v) { return E::key(v); }
  __device__ __forceinline__ T ld(uint32_t i) const { return e[i]; }
  __device__ __forceinline__ void st(uint32_t i, T v) const { e[i] = v; }
  __device__ __forceinline__ void step_sync() const { __builtin_amdgcn_wave_barrier(); }
  __device__ __forceinline__ void launch_sync() const { __builtin_amdgcn_wave_barrier(); }
};
// global-memory variant for segments that do not fit LDS.  All lanes belong to one wavefront on one CU, so
// plain accesses are coherent through that CU's write-through L1 (the same guarantee __syncthreads() gives a
// block); every step drains its stores before the next step's loads.
template <class E> struct GlbMemT
{
  typedef typename E::T T;
  T *e;  // plain accesses; the "memory" clobber of step_sync makes the compiler reload after every step
  static __device__ __forceinline__ uint32_t key(T v) { return E::key(v); }
  __device__ __forceinline__ T ld(uint32_t i) const { return e[i]; }
  __device__ __forceinline__ void st(uint32_t i, T v) const { e[i] = v; }
  __device__ __forceinline__ void step_sync() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  // the detached leaf L is out of reach of every in-flight pop (ancestor stall), so its store needs no drain of its own
  __device__ __forceinline__ void launch_sync() const {}
};
typedef LdsMemT<E64> LdsMem;
typedef GlbMemT<E64> GlbMem;

__device__ __forceinline__ bool anc_or_self(uint32_t a, uint32_t b)  // is node a an ancestor of (or equal to) node b
{
  uint32_t A = a + 1, B = b + 1;
  int da = 31 - __clz(A), db = 31 - __clz(B);
  return db >= da && (B >> (db - da)) == A;
}

// one top-down sift step of the value v sitting in `hole`; returns true while the hole keeps descending
template <class M> __device__ __forceinline__ bool sift_step(const M &mem, uint32_t &hole, uint32_t len, typename M::T v)
{
  typedef typename M::T T;
  const uint32_t right = 2 * (hole + 1), left = right - 1;
  T ec = 0;
  uint32_t c = 0;
  bool has = true;
  if (right < len)
  {
    const T el = mem.ld(left), er = mem.ld(right);
    if (M::key(er) < M::key(el))
    {
      c = left;
      ec = el;
    }
    else
    {
      c = right;
      ec = er;
    }
  }
  else if (left < len)
  {
    c = left;
    ec = mem.ld(left);
  }
  else
    has = false;
  if (has && !(M::key(ec) < M::key(v)))
  {
    mem.st(hole, ec);
    hole = c;
    return true;
  }
  mem.st(hole, v);
  return false;
}

constexpr uint32_t HEAP_PAD = 32;
__device__ unsigned long long g_heap_iters[2];
__device__ unsigned long long g_heap_phase[8];  // debug: 10 ns ticks of the phases of the largest ranked heap (BK_DEBUG=sort)  // debug: loop iterations / pops of sort_heap (BK_DEBUG=sort)

// the routines below are executed by one full wavefront (64 lanes, all active)
// make_heap, bottom level first (nodes of one depth own disjoint subtrees)
template <class M> __device__ void make_heap_wave(const M &mem, const uint32_t m)
{
  if (m < 2) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t lastp = (m - 2) / 2;
  for (int d = 31 - __clz(lastp + 1); d >= 0; --d)
  {
    const uint32_t lo = (1u << d) - 1;
    uint32_t hi = (1u << (d + 1)) - 2;
    if (hi > lastp) hi = lastp;
    for (uint32_t base = lo; base <= hi; base += 64)
    {
      const uint32_t p = base + lane;
      if (p <= hi)
      {
        uint32_t hole = p;
        const typename M::T v = mem.ld(p);
        while (sift_step(mem, hole, m, v))
        {
        }
      }
      mem.step_sync();
    }
  }
}

// the same with every wave of the workgroup (NT threads, all of them call this): the nodes of one depth are spread over the
// waves, a barrier separates the depths
template <class M> __device__ void make_heap_block(const M &mem, const uint32_t m, const uint32_t NT)
{
  if (m < 2) return;
  const uint32_t lastp = (m - 2) / 2;
  for (int d = 31 - __clz(lastp + 1); d >= 0; --d)
  {
    const uint32_t lo = (1u << d) - 1;
    uint32_t hi = (1u << (d + 1)) - 2;
    if (hi > lastp) hi = lastp;
    for (uint32_t p = lo + threadIdx.x; p <= hi; p += NT)
    {
      uint32_t hole = p;
      const typename M::T v = mem.ld(p);
      while (sift_step(mem, hole, m, v))
      {
      }
    }
    mem.step_sync();
    __syncthreads();
  }
}

// sort_heap: pops follow each other two steps apart (lag-2 pipeline) until the heap has shrunk to `stop` elements;
// every pop has finished when this returns.  One loop iteration = one sift step of every pop in flight (one lane
// each) + at most one launch.  The wave is alone on its critical path, so the loop is written branch-free: the
// values a launch needs (the root and the leaf about to be detached) are fetched together with the children of
// the holes, idle lanes run the same instructions with len = 0.
template <class M> __device__ void sort_heap_lag2(const M &mem, const uint32_t m, const uint32_t stop)
{
  if (m < 2 || m <= stop) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t t_end = m - (stop < 1 ? 1 : stop) + 1;  // pops t = 1 .. t_end-1 detach leaves m-1 .. stop
  uint32_t hole = 0, len = 0, lvl = 0;  // len == 0 <=> idle lane; lvl = depth of the hole
  uint32_t vk = 0, vx = 0;              // key / index halves of the value being sifted
  uint32_t next_t = 1, since = 2;
  uint32_t budget = m > 0x03000000u ? 0xFFFFFFFFu : 64u * m + 4096u;
  uint32_t L = m - 1;                   // leaf the next launch detaches
  int dL = 31 - __clz((int) (L + 1));   // its depth
  bool more = next_t < t_end;
  while (true)
  {
    const bool may_launch = more & (since >= 1);  // uniform
    hent pre_root = 0, pre_leaf = 0;
    if (may_launch)
    {
      pre_root = mem.ld(0);
      pre_leaf = mem.ld(L);
    }
    const uint32_t left = 2 * hole + 1;
    const bool has_l = left < len;
    const uint32_t a = has_l ? left : 0;
    const hent el = mem.ld(a), er = mem.ld(a + 1);
    const uint32_t kl = hkey(el), kr = hkey(er);
    const bool pr = (left + 1 < len) & (kr >= kl);
    const uint32_t kc = pr ? kr : kl, xc = pr ? (uint32_t) er : (uint32_t) el;
    const bool desc = has_l & (kc >= vk);
    const uint32_t hole0 = hole;
    if (len != 0) mem.st(hole, ((hent) (desc ? kc : vk) << 32) | (desc ? xc : vx));
    const bool was_active = len != 0;
    hole = desc ? left + (pr ? 1u : 0u) : hole;
    lvl += desc ? 1u : 0u;
    len = desc ? len : 0;
    mem.step_sync();
    ++since;
    if (may_launch & (since >= 2))
    {
      // a pop in flight that can still reach L (its hole is L or an ancestor of L), or that wrote L in this step
      // (pre_leaf is stale then), holds the launch back
      const bool anc = ((int) lvl <= dL) & (((L + 1) >> (dL - (int) lvl)) == hole + 1);
      const bool blocks = (was_active & (hole0 == L)) | ((len != 0) & anc);
      if (__ballot(blocks) == 0ull)
      {
        if (lane == (next_t & 63u))
        {
          mem.st(L, pre_root);
          vk = hkey(pre_leaf);
          vx = (uint32_t) pre_leaf;
          hole = 0;
          lvl = 0;
          len = L;
        }
        since = 0;
        ++next_t;
        more = next_t < t_end;
        L = m - next_t;
        dL = 31 - __clz((int) (L + 1));
      }
    }
    else if (!more)
    {
      if (__ballot(len != 0) == 0ull) break;
    }
    if (--budget == 0) break;
  }
  mem.step_sync();
  if (lane == 0)
  {
    atomicAdd(&g_heap_iters[0], (unsigned long long) ((m > 0x03000000u ? 0xFFFFFFFFu : 64u * m + 4096u) - budget));
    atomicAdd(&g_heap_iters[1], (unsigned long long) (t_end - 1));
  }
}

// The same pop pipeline as sort_heap_lag2, written in GCN assembly.  A lone wavefront issues at most one instruction
// every 4 cycles, so the segment's critical path is the instruction count of one loop iteration; the compiler's
// version spends ~80 instructions per iteration on mask bookkeeping, this one ~25 (+~25 in an iteration that
// launches).  Per-lane state: h1 = hole + 1 (v40), len (v41; 0 = idle lane), value (idx v42, key v43).
// Uniform state: next_t (s41), L+1 (s43), clz(L+1) (s44), budget (s45), t_end (s46).
// GLB = false: heap in LDS at byte offset `base`; GLB = true: heap in global memory at `gptr` (base = 0).
// Global variant: the loads of a step follow the stores of the step before in program order through the same L1,
// which keeps them ordered per address for one wavefront; the loop therefore only waits for its loads
// (`s_waitcnt vmcnt(0)` before their use also covers the older stores) and does not drain the store acknowledgements
// at the top of every step (0.34 -> 0.26 us per step).  The caller drains before it reads the result.
// Two ways to keep idle lanes (no pop in flight) harmless.  Global memory: their store is masked by EXEC and len = 0
// marks them.  LDS: an idle lane points at a spare slot behind the heap (h1 = m + 2, v55), so it runs the same
// unmasked instructions as everybody else - its "children" are out of range, its store hits the spare slot, and it
// can never be an ancestor of the leaf to detach; that takes 7 instructions out of an iteration.
#define BK_MASKED_STORE(ST) "v_cmp_ne_u32_e64 s[58:59], 0, v41\n s_and_saveexec_b64 s[56:57], s[58:59]\n" ST "s_mov_b64 exec, s[56:57]\n"
#define BK_PLAIN_STORE(ST) ST
#define BK_IDLE_BY_LEN "v_cndmask_b32_e64 v40, v40, v53, s[54:55]\n v_cndmask_b32_e64 v41, 0, v41, s[54:55]\n"
#define BK_IDLE_BY_SLOT "v_cndmask_b32_e64 v40, v55, v53, s[54:55]\n"
#define BK_CHECK_ACTIVE_LEN "v_cmp_ne_u32_e64 s[60:61], 0, v41\n s_and_b64 vcc, vcc, s[60:61]\n v_cmp_eq_u32_e64 s[60:61], s43, v54\n s_and_b64 s[60:61], s[60:61], s[58:59]\n"
#define BK_CHECK_ACTIVE_SLOT "v_cmp_eq_u32_e64 s[60:61], s43, v54\n"
#define BK_ANY_ACTIVE_LEN "v_cmp_ne_u32_e32 vcc, 0, v41\n"
#define BK_ANY_ACTIVE_SLOT "v_cmp_ne_u32_e32 vcc, v40, v55\n"

// one sift step of every pop in flight
#define BK_HEAP_SIFT(LD2_KIDS, STORE_HOLE, WAIT_LOADS, IDLE_UPD)                                                                         \
  "v_lshlrev_b32 v44, 1, v40\n"                                                                                            \
  "v_cmp_le_u32_e64 s[48:49], v44, v41\n"                                                                                  \
  "v_cmp_lt_u32_e64 s[50:51], v44, v41\n"                                                                                  \
  "v_lshl_add_u32 v45, v44, 3, s40\n"                                                                                      \
  "v_cndmask_b32_e64 v45, v60, v45, s[48:49]\n"                                                                            \
  LD2_KIDS                                                                                                                \
  "v_lshl_add_u32 v52, v40, 3, s40\n"                                                                                      \
  "v_mov_b32 v54, v40\n"                                                                                                   \
  WAIT_LOADS                                                                                                              \
  "v_cmp_ge_u32_e32 vcc, v49, v47\n"                                                                                       \
  "s_and_b64 s[52:53], vcc, s[50:51]\n"                                                                                    \
  "v_cndmask_b32_e64 v51, v47, v49, s[52:53]\n"                                                                            \
  "v_cndmask_b32_e64 v50, v46, v48, s[52:53]\n"                                                                            \
  "v_cmp_ge_u32_e32 vcc, v51, v43\n"                                                                                       \
  "s_and_b64 s[54:55], vcc, s[48:49]\n"                                                                                    \
  "v_cndmask_b32_e64 v51, v43, v51, s[54:55]\n"                                                                            \
  "v_cndmask_b32_e64 v50, v42, v50, s[54:55]\n"                                                                            \
  STORE_HOLE                                                                                                              \
  "v_addc_co_u32_e64 v53, vcc, v44, 0, s[52:53]\n"                                                                         \
  IDLE_UPD

// Loop A = the iteration right after a launch (the next pop may not start yet: lag 2), loop B = iterations that may
// launch: they prefetch the root and the leaf to detach together with the children of the holes.
#define BK_HEAP_ASM(LD1_ROOT, LD1_LEAF, LD2_KIDS, STORE_HOLE, ST_LEAF, WAIT_LOADS, WAIT_ALL, IDLE_UPD, CHECK_ACTIVE, ANY_ACTIVE)                                      \
  "v_mov_b32 v62, %[lane]\n"                                                                                               \
  "s_mov_b32 s62, %[plo]\n s_mov_b32 s63, %[phi]\n"                                                                          \
  "s_sub_u32 s40, %[base], 8\n"                                                                                            \
  "v_mov_b32 v60, %[base]\n"                                                                                               \
  "s_mov_b32 s46, %[tend]\n s_mov_b32 s41, 1\n s_mov_b32 s43, %[m]\n"                                                       \
  "s_flbit_i32_b32 s44, s43\n"                                                                                             \
  "s_lshl_b32 s47, s43, 3\n s_add_u32 s47, s47, s40\n v_mov_b32 v61, s47\n"                                                  \
  "s_mov_b32 s45, %[budget]\n"                                                                                             \
  "s_add_u32 s47, %[m], 2\n v_mov_b32 v55, s47\n v_mov_b32 v40, v55\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"                                               \
  "s_branch BK_B_%=\n"                                                                                                     \
  "BK_A_%=:\n"                                                                                                            \
  WAIT_ALL                                                                                                                \
  BK_HEAP_SIFT(LD2_KIDS, STORE_HOLE, WAIT_LOADS, IDLE_UPD)                                                                             \
  "s_sub_u32 s45, s45, 1\n"                                                                                                \
  "s_cbranch_scc1 BK_DONE_%=\n"                                                                                            \
  "BK_B_%=:\n"                                                                                                            \
  WAIT_ALL LD1_ROOT LD1_LEAF                                                                                              \
  BK_HEAP_SIFT(LD2_KIDS, STORE_HOLE, WAIT_LOADS, IDLE_UPD)                                                                             \
  "s_cmp_ge_u32 s41, s46\n"                                                                                                \
  "s_cbranch_scc1 BK_NOMORE_%=\n"                                                                                          \
  "v_ffbh_u32_e32 v63, v40\n"                                                                                              \
  "v_subrev_u32_e32 v63, s44, v63\n"                                                                                       \
  "v_lshrrev_b32_e64 v64, v63, s43\n"                                                                                      \
  "v_cmp_eq_u32_e32 vcc, v64, v40\n"                                                                                       \
  "v_cmp_gt_u32_e64 s[60:61], 32, v63\n"                                                                                   \
  "s_and_b64 vcc, vcc, s[60:61]\n"                                                                                         \
  CHECK_ACTIVE                                                                                                            \
  "s_or_b64 vcc, vcc, s[60:61]\n"                                                                                          \
  "s_cbranch_vccnz BK_BNEXT_%=\n"                                                                                          \
  "s_and_b32 s47, s41, 63\n"                                                                                               \
  "v_cmp_eq_u32_e32 vcc, s47, v62\n"                                                                                       \
  "s_sub_u32 s47, s43, 1\n"                                                                                                \
  "s_and_saveexec_b64 s[56:57], vcc\n"                                                                                     \
  ST_LEAF                                                                                                                 \
  "v_mov_b32 v42, v58\n v_mov_b32 v43, v59\n v_mov_b32 v40, 1\n v_mov_b32 v41, s47\n"                                         \
  "s_mov_b64 exec, s[56:57]\n"                                                                                             \
  "s_add_u32 s41, s41, 1\n"                                                                                                \
  "s_mov_b32 s43, s47\n"                                                                                                   \
  "s_flbit_i32_b32 s44, s43\n"                                                                                             \
  "v_add_u32_e32 v61, -8, v61\n"                                                                                           \
  "s_sub_u32 s45, s45, 1\n"                                                                                                \
  "s_cbranch_scc0 BK_A_%=\n"                                                                                               \
  "s_branch BK_DONE_%=\n"                                                                                                  \
  "BK_NOMORE_%=:\n"                                                                                                       \
  ANY_ACTIVE                                                                                                              \
  "s_cbranch_vccz BK_DONE_%=\n"                                                                                            \
  "BK_BNEXT_%=:\n"                                                                                                        \
  "s_sub_u32 s45, s45, 1\n"                                                                                                \
  "s_cbranch_scc0 BK_B_%=\n"                                                                                               \
  "BK_DONE_%=:\n"                                                                                                         \
  WAIT_ALL                                                                                                                \
  "s_mov_b32 %[left], s45\n"

#define BK_HEAP_CLOBBERS                                                                                                     \
  "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v56", "v57", "v58", \
      "v55", "v59", "v60", "v61", "v62", "v63", "v64", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",    \
      "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "vcc", "scc", "memory"

template <bool GLB> __device__ __forceinline__ void sort_heap_asm(hent *buf, const uint32_t m, const uint32_t stop)
{
  if (m < 2 || m <= stop) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t t_end = __builtin_amdgcn_readfirstlane(m - (stop < 1 ? 1 : stop) + 1);
  const uint32_t budget = __builtin_amdgcn_readfirstlane(m > 0x03000000u ? 0xFFFFFFFFu : 64u * m + 4096u);
  const uint32_t mm = __builtin_amdgcn_readfirstlane(m);
  const unsigned long long p = (unsigned long long) buf;
  const uint32_t plo = __builtin_amdgcn_readfirstlane((uint32_t) p), phi = __builtin_amdgcn_readfirstlane((uint32_t) (p >> 32));
  const uint32_t base = GLB ? 0u : plo;  // low half of a flat LDS address = byte offset inside LDS
  uint32_t left;
  if (GLB)
    asm volatile(BK_HEAP_ASM("global_load_dwordx2 v[56:57], v60, s[62:63]\n", "global_load_dwordx2 v[58:59], v61, s[62:63]\n",
                             "global_load_dwordx4 v[46:49], v45, s[62:63]\n", BK_MASKED_STORE("global_store_dwordx2 v52, v[50:51], s[62:63]\n"),
                             "global_store_dwordx2 v61, v[56:57], s[62:63]\n", "s_waitcnt vmcnt(0)\n", "", BK_IDLE_BY_LEN, BK_CHECK_ACTIVE_LEN, BK_ANY_ACTIVE_LEN)
                 : [left] "=s"(left)
                 : [lane] "v"(lane), [plo] "s"(plo), [phi] "s"(phi), [base] "s"(base), [tend] "s"(t_end), [m] "s"(mm), [budget] "s"(budget)
                 : BK_HEAP_CLOBBERS);
  else
    asm volatile(BK_HEAP_ASM("ds_read_b64 v[56:57], v60\n", "ds_read_b64 v[58:59], v61\n", "ds_read2_b64 v[46:49], v45 offset1:1\n",
                             BK_PLAIN_STORE("ds_write_b64 v52, v[50:51]\n"), "ds_write_b64 v61, v[56:57]\n", "s_waitcnt lgkmcnt(0)\n", "", BK_IDLE_BY_SLOT,
                             BK_CHECK_ACTIVE_SLOT, BK_ANY_ACTIVE_SLOT)
                 : [left] "=s"(left)
                 : [lane] "v"(lane), [plo] "s"(plo), [phi] "s"(phi), [base] "s"(base), [tend] "s"(t_end), [m] "s"(mm), [budget] "s"(budget)
                 : BK_HEAP_CLOBBERS);
  if (!GLB) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0)
  {
    atomicAdd(&g_heap_iters[0], (unsigned long long) (budget - left));
    atomicAdd(&g_heap_iters[1], (unsigned long long) (t_end - 1));
  }
}

// ---- the same pipeline for 4-byte entries (rank << 16 | local index): value v42, its key v43, children v46/v47 ----
#define BK_HEAP32_SIFT(LD2_KIDS, STORE_HOLE, WAIT_LOADS, IDLE_UPD)                                                                       \
  "v_lshlrev_b32 v44, 1, v40\n"                                                                                            \
  "v_cmp_le_u32_e64 s[48:49], v44, v41\n"                                                                                  \
  "v_cmp_lt_u32_e64 s[50:51], v44, v41\n"                                                                                  \
  "v_lshl_add_u32 v45, v44, 2, s40\n"                                                                                      \
  "v_cndmask_b32_e64 v45, v60, v45, s[48:49]\n"                                                                            \
  LD2_KIDS                                                                                                                \
  "v_lshl_add_u32 v52, v40, 2, s40\n"                                                                                      \
  "v_mov_b32 v54, v40\n"                                                                                                   \
  WAIT_LOADS                                                                                                              \
  "v_lshrrev_b32 v48, 16, v46\n"                                                                                           \
  "v_lshrrev_b32 v49, 16, v47\n"                                                                                           \
  "v_cmp_ge_u32_e32 vcc, v49, v48\n"                                                                                       \
  "s_and_b64 s[52:53], vcc, s[50:51]\n"                                                                                    \
  "v_cndmask_b32_e64 v50, v46, v47, s[52:53]\n"                                                                            \
  "v_cndmask_b32_e64 v51, v48, v49, s[52:53]\n"                                                                            \
  "v_cmp_ge_u32_e32 vcc, v51, v43\n"                                                                                       \
  "s_and_b64 s[54:55], vcc, s[48:49]\n"                                                                                    \
  "v_cndmask_b32_e64 v50, v42, v50, s[54:55]\n"                                                                            \
  STORE_HOLE                                                                                                              \
  "v_addc_co_u32_e64 v53, vcc, v44, 0, s[52:53]\n"                                                                         \
  IDLE_UPD

#define BK_HEAP32_ASM(LD1_ROOT, LD1_LEAF, LD2_KIDS, STORE_HOLE, ST_LEAF, WAIT_LOADS, WAIT_ALL, IDLE_UPD, CHECK_ACTIVE, ANY_ACTIVE)                                    \
  "v_mov_b32 v62, %[lane]\n"                                                                                               \
  "s_mov_b32 s62, %[plo]\n s_mov_b32 s63, %[phi]\n"                                                                          \
  "s_sub_u32 s40, %[base], 4\n"                                                                                            \
  "v_mov_b32 v60, %[base]\n"                                                                                               \
  "s_mov_b32 s46, %[tend]\n s_mov_b32 s41, 1\n s_mov_b32 s43, %[m]\n"                                                       \
  "s_flbit_i32_b32 s44, s43\n"                                                                                             \
  "s_lshl_b32 s47, s43, 2\n s_add_u32 s47, s47, s40\n v_mov_b32 v61, s47\n"                                                  \
  "s_mov_b32 s45, %[budget]\n"                                                                                             \
  "s_add_u32 s47, %[m], 2\n v_mov_b32 v55, s47\n v_mov_b32 v40, v55\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"                                               \
  "s_branch BK_B_%=\n"                                                                                                     \
  "BK_A_%=:\n"                                                                                                            \
  WAIT_ALL                                                                                                                \
  BK_HEAP32_SIFT(LD2_KIDS, STORE_HOLE, WAIT_LOADS, IDLE_UPD)                                                                           \
  "s_sub_u32 s45, s45, 1\n"                                                                                                \
  "s_cbranch_scc1 BK_DONE_%=\n"                                                                                            \
  "BK_B_%=:\n"                                                                                                            \
  WAIT_ALL LD1_ROOT LD1_LEAF                                                                                              \
  BK_HEAP32_SIFT(LD2_KIDS, STORE_HOLE, WAIT_LOADS, IDLE_UPD)                                                                           \
  "s_cmp_ge_u32 s41, s46\n"                                                                                                \
  "s_cbranch_scc1 BK_NOMORE_%=\n"                                                                                          \
  "v_ffbh_u32_e32 v63, v40\n"                                                                                              \
  "v_subrev_u32_e32 v63, s44, v63\n"                                                                                       \
  "v_lshrrev_b32_e64 v64, v63, s43\n"                                                                                      \
  "v_cmp_eq_u32_e32 vcc, v64, v40\n"                                                                                       \
  "v_cmp_gt_u32_e64 s[60:61], 32, v63\n"                                                                                   \
  "s_and_b64 vcc, vcc, s[60:61]\n"                                                                                         \
  CHECK_ACTIVE                                                                                                            \
  "s_or_b64 vcc, vcc, s[60:61]\n"                                                                                          \
  "s_cbranch_vccnz BK_BNEXT_%=\n"                                                                                          \
  "s_and_b32 s47, s41, 63\n"                                                                                               \
  "v_cmp_eq_u32_e32 vcc, s47, v62\n"                                                                                       \
  "s_sub_u32 s47, s43, 1\n"                                                                                                \
  "s_and_saveexec_b64 s[56:57], vcc\n"                                                                                     \
  ST_LEAF                                                                                                                 \
  "v_mov_b32 v42, v58\n v_lshrrev_b32 v43, 16, v58\n v_mov_b32 v40, 1\n v_mov_b32 v41, s47\n"                                 \
  "s_mov_b64 exec, s[56:57]\n"                                                                                             \
  "s_add_u32 s41, s41, 1\n"                                                                                                \
  "s_mov_b32 s43, s47\n"                                                                                                   \
  "s_flbit_i32_b32 s44, s43\n"                                                                                             \
  "v_add_u32_e32 v61, -4, v61\n"                                                                                           \
  "s_sub_u32 s45, s45, 1\n"                                                                                                \
  "s_cbranch_scc0 BK_A_%=\n"                                                                                               \
  "s_branch BK_DONE_%=\n"                                                                                                  \
  "BK_NOMORE_%=:\n"                                                                                                       \
  ANY_ACTIVE                                                                                                              \
  "s_cbranch_vccz BK_DONE_%=\n"                                                                                            \
  "BK_BNEXT_%=:\n"                                                                                                        \
  "s_sub_u32 s45, s45, 1\n"                                                                                                \
  "s_cbranch_scc0 BK_B_%=\n"                                                                                               \
  "BK_DONE_%=:\n"                                                                                                         \
  WAIT_ALL                                                                                                                \
  "s_mov_b32 %[left], s45\n"

template <bool GLB> __device__ __forceinline__ void sort_heap_asm32(uint32_t *buf, const uint32_t m, const uint32_t stop)
{
  if (m < 2 || m <= stop) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t t_end = __builtin_amdgcn_readfirstlane(m - (stop < 1 ? 1 : stop) + 1);
  const uint32_t budget = __builtin_amdgcn_readfirstlane(m > 0x03000000u ? 0xFFFFFFFFu : 64u * m + 4096u);
  const uint32_t mm = __builtin_amdgcn_readfirstlane(m);
  const unsigned long long p = (unsigned long long) buf;
  const uint32_t plo = __builtin_amdgcn_readfirstlane((uint32_t) p), phi = __builtin_amdgcn_readfirstlane((uint32_t) (p >> 32));
  const uint32_t base = GLB ? 0u : plo;
  uint32_t left;
  if (GLB)
    asm volatile(BK_HEAP32_ASM("global_load_dword v56, v60, s[62:63]\n", "global_load_dword v58, v61, s[62:63]\n",
                               "global_load_dwordx2 v[46:47], v45, s[62:63]\n", BK_MASKED_STORE("global_store_dword v52, v50, s[62:63]\n"),
                               "global_store_dword v61, v56, s[62:63]\n", "s_waitcnt vmcnt(0)\n", "", BK_IDLE_BY_LEN, BK_CHECK_ACTIVE_LEN, BK_ANY_ACTIVE_LEN)
                 : [left] "=s"(left)
                 : [lane] "v"(lane), [plo] "s"(plo), [phi] "s"(phi), [base] "s"(base), [tend] "s"(t_end), [m] "s"(mm), [budget] "s"(budget)
                 : BK_HEAP_CLOBBERS);
  else
    asm volatile(BK_HEAP32_ASM("ds_read_b32 v56, v60\n", "ds_read_b32 v58, v61\n", "ds_read2_b32 v[46:47], v45 offset1:1\n",
                               BK_PLAIN_STORE("ds_write_b32 v52, v50\n"), "ds_write_b32 v61, v56\n", "s_waitcnt lgkmcnt(0)\n", "", BK_IDLE_BY_SLOT,
                               BK_CHECK_ACTIVE_SLOT, BK_ANY_ACTIVE_SLOT)
                 : [left] "=s"(left)
                 : [lane] "v"(lane), [plo] "s"(plo), [phi] "s"(phi), [base] "s"(base), [tend] "s"(t_end), [m] "s"(mm), [budget] "s"(budget)
                 : BK_HEAP_CLOBBERS);
  if (!GLB) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0)
  {
    atomicAdd(&g_heap_iters[0], (unsigned long long) (budget - left));
    atomicAdd(&g_heap_iters[1], (unsigned long long) (t_end - 1));
  }
}

// ---- third form of the pop pipeline: ranked entries in LDS, no existence masks ------------------------------------------
// Entries are ((rank + 1) << 16 | local index), never 0.  The heap lives in LDS slots 1..m (slot s at base - 4 + 4 s, so the
// children of hole h are the adjacent slots 2h, 2h + 1); slot 0 is scratch, slots m + 1, m + 2 hold zeros and child
// addresses beyond them are clamped onto them.  A detached leaf slot is ZEROED and the popped root goes straight to the
// output array in global memory (a store nobody waits for) instead of into the freed slot.  A zero reads as "rank below
// every value": the sift stops in front of a child that does not exist (any more) without a per-lane heap length, so the
// two existence masks of the loops above, their SALU round trips and the key extraction are gone:
//   right >= left by rank  <=>  (right | 0xffff) >= left          child >= value by rank  <=>  (child | 0xffff) >= value
// An idle lane sits on hole 0 with value 0xffffffff: it reads slots 0 / 1, never descends, stores into slot 0, and can
// never look like an ancestor of the leaf to detach.  Same lag-2 launch rule and ancestor stall as above (a slot is only
// zeroed when no pop in flight can still reach it).  17 instructions per sift step (was 21 + 2 SALU).
// Per-lane: hole v40, value v42.  Uniform: next_t s41, L s43, clz(L) s44, budget s45, t_end s46, clamp address s42.
#define BK_HEAP32Z_STEP                                                                                                   \
  "v_lshlrev_b32 v44, 1, v40\n"                                                                                            \
  "v_lshl_add_u32 v45, v44, 2, s40\n"                                                                                      \
  "v_min_u32 v45, s42, v45\n"                                                                                              \
  "ds_read2_b32 v[46:47], v45 offset1:1\n"                                                                                 \
  "v_lshl_add_u32 v52, v40, 2, s40\n"                                                                                      \
  "v_mov_b32 v54, v40\n"                                                                                                   \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                 \
  "v_or_b32 v48, 0xffff, v47\n"                                                                                            \
  "v_cmp_ge_u32 vcc, v48, v46\n"                                                                                           \
  "v_cndmask_b32 v50, v46, v47, vcc\n"                                                                                     \
  "v_addc_co_u32 v53, vcc, 0, v44, vcc\n"                                                                                  \
  "v_or_b32 v49, 0xffff, v50\n"                                                                                            \
  "v_cmp_ge_u32 vcc, v49, v42\n"                                                                                           \
  "v_cndmask_b32 v51, v42, v50, vcc\n"                                                                                     \
  "ds_write_b32 v52, v51\n"                                                                                                \
  "v_cndmask_b32 v40, v55, v53, vcc\n"                                                                                     \
  "v_cndmask_b32 v42, v59, v42, vcc\n"
#define BK_HEAP32Z_ASM                                                                                                    \
  "v_mov_b32 v62, %[lane]\n"                                                                                               \
  "s_mov_b32 s62, %[olo]\n s_mov_b32 s63, %[ohi]\n"                                                                        \
  "s_sub_u32 s40, %[base], 4\n"                                                                                            \
  "v_mov_b32 v60, %[base]\n"                                                                                               \
  "s_mov_b32 s46, %[tend]\n s_mov_b32 s41, 1\n s_mov_b32 s43, %[m]\n"                                                     \
  "s_flbit_i32_b32 s44, s43\n"                                                                                             \
  "s_lshl_b32 s47, s43, 2\n s_add_u32 s42, s47, %[base]\n"                                                                 \
  "s_add_u32 s47, s47, s40\n v_mov_b32 v61, s47\n"                                                                         \
  "s_lshl_b32 s47, s43, 2\n s_sub_u32 s47, s47, 4\n v_mov_b32 v57, s47\n"                                                  \
  "s_mov_b32 s45, %[budget]\n"                                                                                             \
  "v_mov_b32 v55, 0\n v_mov_b32 v59, -1\n v_mov_b32 v40, 0\n v_mov_b32 v42, -1\n"                                          \
  "s_branch BK_ZB_%=\n"                                                                                                    \
  "BK_ZA_%=:\n"                                                                                                           \
  BK_HEAP32Z_STEP                                                                                                         \
  "s_sub_u32 s45, s45, 1\n"                                                                                                \
  "s_cbranch_scc1 BK_ZDONE_%=\n"                                                                                           \
  "BK_ZB_%=:\n"                                                                                                           \
  "ds_read_b32 v56, v60\n"                                                                                                 \
  "ds_read_b32 v58, v61\n"                                                                                                 \
  BK_HEAP32Z_STEP                                                                                                         \
  "s_cmp_ge_u32 s41, s46\n"                                                                                                \
  "s_cbranch_scc1 BK_ZNOMORE_%=\n"                                                                                         \
  "v_ffbh_u32 v63, v40\n"                                                                                                  \
  "v_subrev_u32 v63, s44, v63\n"                                                                                           \
  "v_lshrrev_b32_e64 v64, v63, s43\n"                                                                                      \
  "v_cmp_eq_u32 vcc, v64, v40\n"                                                                                           \
  "v_cmp_eq_u32_e64 s[60:61], s43, v54\n"                                                                                  \
  "s_or_b64 vcc, vcc, s[60:61]\n"                                                                                          \
  "s_cbranch_vccnz BK_ZBNEXT_%=\n"                                                                                         \
  "s_and_b32 s47, s41, 63\n"                                                                                               \
  "v_cmp_eq_u32 vcc, s47, v62\n"                                                                                           \
  "s_and_saveexec_b64 s[56:57], vcc\n"                                                                                     \
  "ds_write_b32 v61, v55\n"                                                                                                \
  "global_store_dword v57, v56, s[62:63]\n"                                                                                \
  "v_mov_b32 v42, v58\n v_mov_b32 v40, 1\n"                                                                                \
  "s_mov_b64 exec, s[56:57]\n"                                                                                             \
  "s_add_u32 s41, s41, 1\n"                                                                                                \
  "s_sub_u32 s43, s43, 1\n"                                                                                                \
  "s_flbit_i32_b32 s44, s43\n"                                                                                             \
  "v_add_u32 v61, -4, v61\n"                                                                                               \
  "v_add_u32 v57, -4, v57\n"                                                                                               \
  "s_sub_u32 s45, s45, 1\n"                                                                                                \
  "s_cbranch_scc0 BK_ZA_%=\n"                                                                                              \
  "s_branch BK_ZDONE_%=\n"                                                                                                 \
  "BK_ZNOMORE_%=:\n"                                                                                                      \
  "v_cmp_ne_u32 vcc, 0, v40\n"                                                                                             \
  "s_cbranch_vccz BK_ZDONE_%=\n"                                                                                           \
  "BK_ZBNEXT_%=:\n"                                                                                                       \
  "s_sub_u32 s45, s45, 1\n"                                                                                                \
  "s_cbranch_scc0 BK_ZB_%=\n"                                                                                              \
  "BK_ZDONE_%=:\n"                                                                                                        \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                 \
  "s_mov_b32 %[left], s45\n"

// slot1 = LDS address of slot 1 (slot 0 in front of it and the two zero slots behind slot m belong to the caller); pops
// t = 1 .. m - 1 store the popped roots to out[m - 1] .. out[1]; the last element stays in slot 1
__device__ __forceinline__ void sort_heap_lds_zero(uint32_t *slot1, const uint32_t m, uint32_t *out)
{
  if (m < 2) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t t_end = __builtin_amdgcn_readfirstlane(m);
  const uint32_t budget = __builtin_amdgcn_readfirstlane(64u * m + 4096u);
  const uint32_t mm = __builtin_amdgcn_readfirstlane(m);
  const unsigned long long o = (unsigned long long) out;
  const uint32_t olo = __builtin_amdgcn_readfirstlane((uint32_t) o), ohi = __builtin_amdgcn_readfirstlane((uint32_t) (o >> 32));
  const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t) (unsigned long long) slot1);  // low half of a flat LDS address = byte offset inside LDS
  uint32_t left;
  asm volatile(BK_HEAP32Z_ASM
               : [left] "=s"(left)
               : [lane] "v"(lane), [olo] "s"(olo), [ohi] "s"(ohi), [base] "s"(base), [tend] "s"(t_end), [m] "s"(mm), [budget] "s"(budget)
               : BK_HEAP_CLOBBERS);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane == 0)
  {
    atomicAdd(&g_heap_iters[0], (unsigned long long) (budget - left));
    atomicAdd(&g_heap_iters[1], (unsigned long long) (t_end - 1));
  }
}

// ---- fourth form: the zero-slot loop above, scheduled for a lone wave ------------------------------------------------------
// A wave that is alone on its SIMD issues one instruction every 4 cycles whatever it is, so a pop costs its instruction
// count plus whatever LDS latency is left exposed.  Same data layout, launch rule and ancestor stall as sort_heap_lds_zero:
//   * rank compares on the high halves directly (SDWA WORD_1 selects) instead of `or 0xffff` + compare;
//   * the children of the NEXT step's hole are requested as soon as this step's hole is known (behind this step's store in
//     program order, so they see it); the launch decision, the launch and the bookkeeping of a launch (which moves into the
//     following iteration) run while that read is in flight;
//   * the ancestor test is prepared from the holes BEFORE the step while the children are still on their way: a lane that
//     descends ends one level deeper, so the only ancestor of L it can reach is L >> (clz(hole) - 1 - clz(L)) (a hole that
//     ends below L's level is compared with L itself, which it cannot equal; a lane that sits ON L stops there, so it is
//     compared with 0, the hole of a lane that stopped); after the step ONE vector compare against the new holes and a
//     branch are left - a scalar instruction that combines masks a vector compare has just written waits ~20 clocks for
//     them (tools/ubench/issue.hip);
//   * the launching lane is a rotating one-hot mask in SGPRs; the launches left are counted in the launch path (s46: its
//     borrow ends the launches), so an iteration that may launch does not test for the end; after the last launch a plain
//     loop of steps drains the pops in flight.
// 20 instructions in the iteration after a launch, 36 in a launching one (was 21 and 42, with two exposed LDS round trips).
// Per-lane: hole v40 (0 = idle), value v42 (-1 = idle).  Uniform: L s43, clz(L) + 1 s47, budget s45, clamp address s42,
// launch mask s[58:59].
#define BK_HEAP32Q_STEP                                                                                                   \
  "v_cmp_ge_u32_sdwa vcc, v47, v46 src0_sel:WORD_1 src1_sel:WORD_1\n"                                                      \
  "v_cndmask_b32 v50, v46, v47, vcc\n"                                                                                     \
  "v_addc_co_u32_e32 v53, vcc, v40, v40, vcc\n"                                                                            \
  "v_cmp_ge_u32_sdwa vcc, v50, v42 src0_sel:WORD_1 src1_sel:WORD_1\n"                                                      \
  "v_cndmask_b32 v51, v42, v50, vcc\n"                                                                                     \
  "ds_write_b32 v52, v51\n"                                                                                                \
  "v_cndmask_b32 v40, v55, v53, vcc\n"                                                                                     \
  "v_cndmask_b32 v42, v59, v42, vcc\n"
#define BK_HEAP32Q_NEXT                                                                                                   \
  "v_lshl_add_u32 v45, v40, 3, s40\n"                                                                                      \
  "v_min_u32 v45, s42, v45\n"                                                                                              \
  "ds_read2_b32 v[46:47], v45 offset1:1\n"
#define BK_HEAP32Q_ASM                                                                                                    \
  "s_setprio 3\n"                                                                                                          \
  "s_mov_b64 s[56:57], exec\n"                                                                                             \
  "s_mov_b32 s62, %[olo]\n s_mov_b32 s63, %[ohi]\n"                                                                        \
  "s_sub_u32 s40, %[base], 4\n"                                                                                            \
  "v_mov_b32 v60, %[base]\n"                                                                                               \
  "s_mov_b32 s43, %[m]\n"                                                                                                  \
  "s_sub_u32 s46, s43, 2\n"                                                                                               \
  "s_flbit_i32_b32 s47, s43\n s_add_u32 s47, s47, 1\n"                                                                     \
  "s_lshl_b32 s48, s43, 2\n s_add_u32 s42, s48, %[base]\n"                                                                 \
  "s_add_u32 s48, s48, s40\n v_mov_b32 v61, s48\n"                                                                         \
  "s_lshl_b32 s48, s43, 2\n s_sub_u32 s48, s48, 4\n v_mov_b32 v57, s48\n"                                                  \
  "s_mov_b32 s45, %[budget]\n"                                                                                             \
  "v_mov_b32 v55, 0\n v_mov_b32 v59, -1\n v_mov_b32 v40, 0\n v_mov_b32 v42, -1\n"                                          \
  "s_mov_b64 s[58:59], 1\n"                                                                                                \
  "v_mov_b32 v45, s40\n"                                                                                                   \
  "ds_read2_b32 v[46:47], v45 offset1:1\n"                                                                                 \
  "s_branch BK_QB_%=\n"                                                                                                    \
  "BK_QA_%=:\n"                                                                                                           \
  "v_lshl_add_u32 v52, v40, 2, s40\n"                                                                                      \
  "s_sub_u32 s43, s43, 1\n"                                                                                                \
  "s_flbit_i32_b32 s47, s43\n"                                                                                             \
  "s_add_u32 s47, s47, 1\n"                                                                                                \
  "v_add_u32 v61, -4, v61\n"                                                                                               \
  "v_add_u32 v57, -4, v57\n"                                                                                               \
  "s_lshl_b64 s[58:59], s[58:59], 1\n"                                                                                     \
  "s_cselect_b64 s[58:59], s[58:59], 1\n"                                                                                  \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                 \
  BK_HEAP32Q_STEP                                                                                                         \
  BK_HEAP32Q_NEXT                                                                                                         \
  "BK_QB_%=:\n"                                                                                                           \
  "ds_read_b32 v56, v60\n"                                                                                                 \
  "ds_read_b32 v58, v61\n"                                                                                                 \
  "v_lshl_add_u32 v52, v40, 2, s40\n"                                                                                      \
  "v_ffbh_u32 v63, v40\n"                                                                                                  \
  "v_subrev_u32 v63, s47, v63\n"                                                                                           \
  "v_max_i32 v63, 0, v63\n"                                                                                                \
  "v_lshrrev_b32_e64 v64, v63, s43\n"                                                                                      \
  "v_cmp_eq_u32 vcc, s43, v40\n"                                                                                           \
  "v_cndmask_b32 v64, v64, v55, vcc\n"                                                                                     \
  "s_waitcnt lgkmcnt(2)\n"                                                                                                 \
  BK_HEAP32Q_STEP                                                                                                         \
  "v_cmp_eq_u32 vcc, v64, v40\n"                                                                                           \
  BK_HEAP32Q_NEXT                                                                                                         \
  "s_cbranch_vccnz BK_QBNEXT_%=\n"                                                                                         \
  "s_waitcnt lgkmcnt(2)\n"                                                                                                 \
  "s_mov_b64 exec, s[58:59]\n"                                                                                             \
  "ds_write_b32 v61, v55\n"                                                                                                \
  "global_store_dword v57, v56, s[62:63]\n"                                                                                \
  "v_mov_b32 v42, v58\n"                                                                                                   \
  "v_mov_b32 v40, 1\n"                                                                                                     \
  "ds_read2_b32 v[46:47], v60 offset0:1 offset1:2\n"                                                                       \
  "s_mov_b64 exec, s[56:57]\n"                                                                                                   \
  "s_sub_u32 s46, s46, 1\n"                                                                                               \
  "s_cbranch_scc0 BK_QA_%=\n"                                                                                              \
  "s_branch BK_QDRAIN_%=\n"                                                                                                 \
  "BK_QBNEXT_%=:\n"                                                                                                       \
  "s_sub_u32 s45, s45, 1\n"                                                                                               \
  "s_cbranch_scc0 BK_QB_%=\n"                                                                                             \
  "s_branch BK_QDONE_%=\n"                                                                                                \
  "BK_QDRAIN_%=:\n"                                                                                                       \
  "v_lshl_add_u32 v52, v40, 2, s40\n"                                                                                     \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                \
  BK_HEAP32Q_STEP                                                                                                         \
  BK_HEAP32Q_NEXT                                                                                                         \
  "v_cmp_ne_u32 vcc, 0, v40\n"                                                                                            \
  "s_cbranch_vccz BK_QDONE_%=\n"                                                                                          \
  "s_sub_u32 s45, s45, 1\n"                                                                                               \
  "s_cbranch_scc0 BK_QDRAIN_%=\n"                                                                                         \
  "BK_QDONE_%=:\n"                                                                                                        \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                 \
  "s_setprio 0\n"                                                                                                          \
  "s_mov_b32 %[left], s45\n"

__device__ int g_heap_no_q = 0;  // BK_HEAP_NO_Q=1: sort_heap_lds_zero instead of sort_heap_lds_q (debugging / comparison)

// same contract as sort_heap_lds_zero
__device__ __forceinline__ void sort_heap_lds_q(uint32_t *slot1, const uint32_t m, uint32_t *out)
{
  if (m < 2) return;
  if (g_heap_no_q)
  {
    sort_heap_lds_zero(slot1, m, out);
    return;
  }
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t budget = __builtin_amdgcn_readfirstlane(64u * m + 4096u);
  const uint32_t mm = __builtin_amdgcn_readfirstlane(m);
  const unsigned long long o = (unsigned long long) out;
  const uint32_t olo = __builtin_amdgcn_readfirstlane((uint32_t) o), ohi = __builtin_amdgcn_readfirstlane((uint32_t) (o >> 32));
  const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t) (unsigned long long) slot1);
  uint32_t left;
  asm volatile(BK_HEAP32Q_ASM
               : [left] "=s"(left)
               : [olo] "s"(olo), [ohi] "s"(ohi), [base] "s"(base), [m] "s"(mm), [budget] "s"(budget)
               : BK_HEAP_CLOBBERS);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane == 0)
  {
    // the budget counts the iterations that may launch; every pop adds one iteration that may not
    atomicAdd(&g_heap_iters[0], (unsigned long long) (budget - left) + (m - 1));
    atomicAdd(&g_heap_iters[1], (unsigned long long) (m - 1));
  }
}

// ---- a heap that is larger than the LDS: the same loop with its tail in global memory ------------------------------------
// Slots 1 .. CAP (the CU's LDS) hold the upper levels, slots CAP+1 .. m live in a global array O (O[slot]); m < 2 (CAP + 1), so
// every slot in O is a leaf.  The loop is sort_heap_lds_q with three additions (model: the same schedule was checked against
// libstdc++ on the host before it was written):
//   * after every step one more vector compare asks whether a new hole lies beyond CAP / 2, i.e. has its children in O (or
//     lies in O itself); if none does - most iterations - nothing else changes;
//   * otherwise: a lane whose new hole lies IN O is done - a leaf: it stores its value there one step early and goes idle
//     (if that leaf is L, the value is also the leaf of the next launch); a lane whose children are in O loads them from
//     there (slots beyond m read the zeros behind the heap) over the zeros its clamped LDS read returned;
//   * the leaf to detach is in O: it is requested right after the launch before (two to three iterations ahead of its use)
//     and zeroed there by the launch.  When a lane finishes ON that leaf meanwhile, its value replaces the requested one in
//     v58 - behind a wait for the request, which would otherwise land in v58 afterwards with the slot's old content (the
//     hardware does not order a load's register write against a later VALU write: one payload lost, another doubled, in a few
//     per cent of the runs over a 53 K-element heap before this wait was there).
// Runs the m - CAP pops that bring the heap down to CAP slots, waits for the pops in flight and returns; the caller carries
// on with sort_heap_lds_q on the LDS part.  0.6 us per pop (everything in global memory) -> ~0.25.
#define BK_HEAP32H_SLOW(TAG)                                                                                                  \
  "v_cmp_lt_u32_e64 s[52:53], s67, v40\n"                                                                                  \
  "s_mov_b64 exec, s[52:53]\n"                                                                                             \
  "v_mov_b32 v71, v42\n"                                                                                                   \
  "v_lshlrev_b32 v70, 2, v40\n"                                                                                            \
  "global_store_dword v70, v42, s[64:65]\n"                                                                                \
  "v_cmp_eq_u32_e64 s[54:55], s43, v40\n"                                                                                  \
  "v_mov_b32 v40, 0\n"                                                                                                     \
  "v_mov_b32 v42, -1\n"                                                                                                    \
  "s_mov_b64 exec, s[56:57]\n"                                                                                             \
  "s_cmp_lg_u64 s[54:55], 0\n"                                                                                             \
  "s_cbranch_scc0 BK_HS1_" TAG "_%=\n"                                                                                             \
  "s_ff1_i32_b64 s70, s[54:55]\n"                                                                                          \
  "s_nop 3\n"                                                                                                              \
  "v_readlane_b32 s69, v71, s70\n"                                                                                         \
  "s_waitcnt vmcnt(0)\n"                                                                                                   \
  "v_mov_b32 v58, s69\n"                                                                                                   \
  "BK_HS1_" TAG "_%=:\n"                                                                                                          \
  "v_cmp_lt_u32_e64 s[50:51], s66, v40\n"                                                                                  \
  "s_mov_b64 exec, s[50:51]\n"                                                                                             \
  "v_lshlrev_b32 v70, 1, v40\n"                                                                                            \
  "v_min_u32 v70, s68, v70\n"                                                                                              \
  "v_lshlrev_b32 v70, 2, v70\n"                                                                                            \
  "global_load_dwordx2 v[68:69], v70, s[64:65]\n"                                                                          \
  "s_mov_b64 exec, s[56:57]\n"                                                                                             \
  "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                                                        \
  "v_cndmask_b32_e64 v46, v46, v68, s[50:51]\n"                                                                            \
  "v_cndmask_b32_e64 v47, v47, v69, s[50:51]\n"
#define BK_HEAP32H_ASM                                                                                                    \
  "s_setprio 3\n"                                                                                                          \
  "s_mov_b64 s[56:57], exec\n"                                                                                             \
  "s_mov_b32 s62, %[olo]\n s_mov_b32 s63, %[ohi]\n"                                                                        \
  "s_mov_b32 s64, %[plo]\n s_mov_b32 s65, %[phi]\n"                                                                        \
  "s_sub_u32 s40, %[base], 4\n"                                                                                            \
  "v_mov_b32 v60, %[base]\n"                                                                                               \
  "s_mov_b32 s43, %[m]\n"                                                                                                  \
  "s_mov_b32 s67, %[cap]\n"                                                                                                \
  "s_lshr_b32 s66, s67, 1\n"                                                                                               \
  "s_add_u32 s68, s43, 1\n"                                                                                                \
  "s_sub_u32 s46, s43, s67\n s_sub_u32 s46, s46, 1\n"                                                                      \
  "s_flbit_i32_b32 s47, s43\n s_add_u32 s47, s47, 1\n"                                                                     \
  "s_add_u32 s48, s67, 1\n s_lshl_b32 s48, s48, 2\n s_add_u32 s42, s48, s40\n"                                             \
  "s_lshl_b32 s48, s43, 2\n v_mov_b32 v61, s48\n"                                                                          \
  "s_sub_u32 s48, s48, 4\n v_mov_b32 v57, s48\n"                                                                           \
  "s_mov_b32 s45, %[budget]\n"                                                                                             \
  "v_mov_b32 v55, 0\n v_mov_b32 v59, -1\n v_mov_b32 v40, 0\n v_mov_b32 v42, -1\n"                                          \
  "s_mov_b64 s[58:59], 1\n"                                                                                                \
  "global_load_dword v58, v61, s[64:65]\n"                                                                                 \
  "v_mov_b32 v45, s40\n"                                                                                                   \
  "ds_read2_b32 v[46:47], v45 offset1:1\n"                                                                                 \
  "s_branch BK_HB_%=\n"                                                                                                    \
  "BK_HA_%=:\n"                                                                                                           \
  "v_lshl_add_u32 v52, v40, 2, s40\n"                                                                                      \
  "s_sub_u32 s43, s43, 1\n"                                                                                                \
  "s_flbit_i32_b32 s47, s43\n"                                                                                             \
  "s_add_u32 s47, s47, 1\n"                                                                                                \
  "v_add_u32 v61, -4, v61\n"                                                                                               \
  "v_add_u32 v57, -4, v57\n"                                                                                               \
  "s_lshl_b64 s[58:59], s[58:59], 1\n"                                                                                     \
  "s_cselect_b64 s[58:59], s[58:59], 1\n"                                                                                  \
  "global_load_dword v58, v61, s[64:65]\n"                                                                                 \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                 \
  BK_HEAP32Q_STEP                                                                                                         \
  "v_cmp_lt_u32_e64 s[50:51], s66, v40\n"                                                                                  \
  BK_HEAP32Q_NEXT                                                                                                         \
  "s_cmp_lg_u64 s[50:51], 0\n"                                                                                             \
  "s_cbranch_scc0 BK_HB_%=\n"                                                                                              \
  BK_HEAP32H_SLOW("a")                                                                                                    \
  "BK_HB_%=:\n"                                                                                                           \
  "ds_read_b32 v56, v60\n"                                                                                                 \
  "v_lshl_add_u32 v52, v40, 2, s40\n"                                                                                      \
  "v_ffbh_u32 v63, v40\n"                                                                                                  \
  "v_subrev_u32 v63, s47, v63\n"                                                                                           \
  "v_max_i32 v63, 0, v63\n"                                                                                                \
  "v_lshrrev_b32_e64 v64, v63, s43\n"                                                                                      \
  "s_waitcnt lgkmcnt(1)\n"                                                                                                 \
  BK_HEAP32Q_STEP                                                                                                         \
  "v_cmp_eq_u32 vcc, v64, v40\n"                                                                                           \
  "v_cmp_lt_u32_e64 s[50:51], s66, v40\n"                                                                                  \
  BK_HEAP32Q_NEXT                                                                                                         \
  "s_cmp_lg_u64 s[50:51], 0\n"                                                                                             \
  "s_cbranch_scc0 BK_HB2_%=\n"                                                                                             \
  BK_HEAP32H_SLOW("b")                                                                                                    \
  "v_cmp_eq_u32 vcc, v64, v40\n"                                                                                           \
  "BK_HB2_%=:\n"                                                                                                          \
  "s_cbranch_vccnz BK_HBNEXT_%=\n"                                                                                         \
  "s_waitcnt vmcnt(0) lgkmcnt(2)\n"                                                                                        \
  "s_mov_b64 exec, s[58:59]\n"                                                                                             \
  "global_store_dword v61, v55, s[64:65]\n"                                                                                \
  "global_store_dword v57, v56, s[62:63]\n"                                                                                \
  "v_mov_b32 v42, v58\n"                                                                                                   \
  "v_mov_b32 v40, 1\n"                                                                                                     \
  "ds_read2_b32 v[46:47], v60 offset0:1 offset1:2\n"                                                                       \
  "s_mov_b64 exec, s[56:57]\n"                                                                                             \
  "s_sub_u32 s46, s46, 1\n"                                                                                                \
  "s_cbranch_scc0 BK_HA_%=\n"                                                                                              \
  "s_branch BK_HDRAIN_%=\n"                                                                                                \
  "BK_HBNEXT_%=:\n"                                                                                                       \
  "s_sub_u32 s45, s45, 1\n"                                                                                                \
  "s_cbranch_scc0 BK_HB_%=\n"                                                                                              \
  "s_branch BK_HDONE_%=\n"                                                                                                 \
  "BK_HDRAIN_%=:\n"                                                                                                       \
  "v_lshl_add_u32 v52, v40, 2, s40\n"                                                                                      \
  "s_waitcnt lgkmcnt(0)\n"                                                                                                 \
  BK_HEAP32Q_STEP                                                                                                         \
  "v_cmp_lt_u32_e64 s[50:51], s66, v40\n"                                                                                  \
  BK_HEAP32Q_NEXT                                                                                                         \
  "s_cmp_lg_u64 s[50:51], 0\n"                                                                                             \
  "s_cbranch_scc0 BK_HD2_%=\n"                                                                                             \
  BK_HEAP32H_SLOW("d")                                                                                                    \
  "BK_HD2_%=:\n"                                                                                                          \
  "v_cmp_ne_u32 vcc, 0, v40\n"                                                                                             \
  "s_cbranch_vccz BK_HDONE_%=\n"                                                                                           \
  "s_sub_u32 s45, s45, 1\n"                                                                                                \
  "s_cbranch_scc0 BK_HDRAIN_%=\n"                                                                                          \
  "BK_HDONE_%=:\n"                                                                                                        \
  "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                                                        \
  "s_setprio 0\n"                                                                                                          \
  "s_mov_b32 %[left], s45\n"

__device__ int g_heap_no_hybrid = 0;  // BK_HEAP_NO_HYBRID=1: heaps beyond the LDS pop in global memory until they fit (the earlier loop)

// slot1 = LDS address of slot 1 (slots 0, cap+1, cap+2 zero), ovf = O (ovf[slot] for cap < slot <= m + 2, the last two zero),
// m > cap odd, m < 2 (cap + 1); pops leaves m .. cap+1 to out[m-1 .. cap]; the heap is left in LDS slots 1 .. cap
__device__ __forceinline__ void sort_heap_hybrid(uint32_t *slot1, const uint32_t m, const uint32_t cap, uint32_t *ovf, uint32_t *out)
{
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t budget = __builtin_amdgcn_readfirstlane(64u * (m - cap) + 4096u);
  const uint32_t mm = __builtin_amdgcn_readfirstlane(m), cc = __builtin_amdgcn_readfirstlane(cap);
  const unsigned long long o = (unsigned long long) out, pp = (unsigned long long) ovf;
  const uint32_t olo = __builtin_amdgcn_readfirstlane((uint32_t) o), ohi = __builtin_amdgcn_readfirstlane((uint32_t) (o >> 32));
  const uint32_t plo = __builtin_amdgcn_readfirstlane((uint32_t) pp), phi = __builtin_amdgcn_readfirstlane((uint32_t) (pp >> 32));
  const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t) (unsigned long long) slot1);
  uint32_t left;
  asm volatile(BK_HEAP32H_ASM
               : [left] "=s"(left)
               : [olo] "s"(olo), [ohi] "s"(ohi), [plo] "s"(plo), [phi] "s"(phi), [base] "s"(base), [m] "s"(mm), [cap] "s"(cc), [budget] "s"(budget)
               : BK_HEAP_CLOBBERS, "v68", "v69", "v70", "v71", "s64", "s65", "s66", "s67", "s68", "s69", "s70");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane == 0)
  {
    atomicAdd(&g_heap_iters[0], (unsigned long long) (budget - left) + (m - cap));
    atomicAdd(&g_heap_iters[1], (unsigned long long) (m - cap));
  }
}

constexpr uint32_t HEAP_SMALL = 1024;    // 8 KiB of LDS per wave
constexpr uint32_t HEAP_LARGE = 20000;   // 156 KiB of LDS (one wave per CU)
constexpr size_t HEAP_BIG_LDS = 163800;  // dynamic LDS of the one-per-CU kernel: all of the CU's 160 KiB but the kernel's few static bytes
constexpr uint32_t HEAP_LARGE32 = HEAP_BIG_LDS / 4 - 3; // that LDS in 4-byte ranked entries (+ slot 0 and two zero slots): 40 947
constexpr uint32_t HEAP_RANKED_MIN = 4096;   // level-loop heaps above this size run on ranked entries (sort_heap_lds_zero)
constexpr uint32_t HEAP_RANKED_MAX = 65534;  // rank + 1 must fit 16 bits and stay below the idle marker's 0xffff

// cls 0: len <= HEAP_SMALL (static LDS), 1: <= HEAP_LARGE (dynamic LDS), 2: larger (global scratch of packed entries)
// (lo, hi]: the sizes this launch takes (CLS 1 is launched once per LDS footprint so that small heaps share a CU)
// CLS 2 runs with HEAP_BIG_THREADS threads: the loads, the ranking and the final gather of a 40 000-element segment are
// 700 dependent round trips for a lone wave (0.9 ms) and a fraction of that for four; the heap itself belongs to wave 0,
// the other waves sleep at the barriers meanwhile.
constexpr uint32_t HEAP_BIG_THREADS = 1024;
constexpr uint32_t RK_UNROLL = 4;
// Dense ranks of the m <= 65534 keys of ONE heap segment by the workgroup that is about to heapsort it: entries (key << 16 | position)
// go through four stable 8-bit counting passes between two scratch arrays in global memory (wave w owns a contiguous run of rows
// of 64 entries; a row's lanes find the lanes of the same digit with eight ballots, the first of them moves the wave's running
// base of that digit), then out[position] = (number of smaller DISTINCT keys + 1) << 16 | position: the ranked entries of
// sort_heap_lds_q.  lds: (NW + 1) * 256 words of scratch, wcnt: NW + 1 words behind them (all inside the dynamic LDS the heap is
// loaded into afterwards: `out` may be that LDS - the scratch is dead when the entries are written).
// Replaces the device-wide ranking (count / gather / five radix passes / flags / scan / scatter: ~30 launches and two host looks in
// front of every sort's longest heap) by ~60 us inside the heap's own workgroup.
__device__ void wg_ranked_entries(const uint32_t *gk, const uint32_t m, unsigned long long *A, unsigned long long *B, uint32_t *out, uint32_t *lds, uint32_t *wcnt, const uint32_t NT)
{
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6, NW = NT >> 6;
  for (uint32_t i = tid; i < m; i += NT) A[i] = ((unsigned long long) gk[i] << 16) | i;
  const uint32_t rows = (m + 63) / 64, rpw = (rows + NW - 1) / NW;
  const uint32_t r0 = min(rows, w * rpw), r1 = min(rows, r0 + rpw);
  unsigned long long *src = A, *dst = B;
  __syncthreads();
  for (int pass = 0; pass < 4; ++pass)
  {
    const int sh = 16 + 8 * pass;
    uint32_t *hist = lds + w * 256;
    for (uint32_t d = lane; d < 256; d += 64) hist[d] = 0;
    for (uint32_t r = r0; r < r1; r += RK_UNROLL)
    {
      unsigned long long e[RK_UNROLL];
#pragma unroll
      for (uint32_t u = 0; u < RK_UNROLL; ++u)
      {
        const uint32_t i = (r + u) * 64 + lane;
        e[u] = (r + u < r1 && i < m) ? src[i] : ~0ull;
      }
#pragma unroll
      for (uint32_t u = 0; u < RK_UNROLL; ++u)
        if (e[u] != ~0ull) atomicAdd(&hist[(uint32_t) (e[u] >> sh) & 255u], 1u);
    }
    __syncthreads();
    // bases: digit-major, wave-minor
    if (tid < 256)
    {
      uint32_t tot = 0;
      for (uint32_t k = 0; k < NW; ++k) tot += lds[k * 256 + tid];
      uint32_t inc = tot;
      for (int d = 1; d < 64; d <<= 1)
      {
        const uint32_t o = __shfl_up(inc, d, 64);
        if ((int) lane >= d) inc += o;
      }
      if (lane == 63) wcnt[w] = inc;
      // (tid < 256 = the first four waves: they meet at the barrier below with everybody)
      lds[NW * 256 + tid] = inc - tot;  // exclusive inside the wave
    }
    __syncthreads();
    if (tid < 256)
    {
      uint32_t base = lds[NW * 256 + tid];
      for (uint32_t k = 0; k < w; ++k) base += wcnt[k];
      for (uint32_t k = 0; k < NW; ++k)
      {
        const uint32_t c = lds[k * 256 + tid];
        lds[k * 256 + tid] = base;
        base += c;
      }
    }
    __syncthreads();
    for (uint32_t r = r0; r < r1; r += RK_UNROLL)
    {
      unsigned long long e[RK_UNROLL];
#pragma unroll
      for (uint32_t u = 0; u < RK_UNROLL; ++u)
      {
        const uint32_t i = (r + u) * 64 + lane;
        e[u] = (r + u < r1 && i < m) ? src[i] : ~0ull;
      }
#pragma unroll
      for (uint32_t u = 0; u < RK_UNROLL; ++u)
      {
        const bool valid = e[u] != ~0ull;
        const uint32_t d = (uint32_t) (e[u] >> sh) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit)
        {
          const bool mine = (d >> bit) & 1u;
          const unsigned long long bal = __ballot(mine);
          peers &= mine ? bal : ~bal;
        }
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t) (peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) peers, 0u));
        const uint32_t base = valid ? hist[d] : 0u;
        if (valid) dst[base + before] = e[u];
        if (valid && before == 0) hist[d] = base + (uint32_t) __popcll(peers);  // (one lane per digit of the row; the wave's LDS operations stay in order)
      }
    }
    __syncthreads();
    unsigned long long *t = src;
    src = dst;
    dst = t;
  }
  // src: sorted by key, equal keys in position order.  rank = number of key changes in front of the entry
  uint32_t changes = 0;
  for (uint32_t r = r0; r < r1; ++r)
  {
    const uint32_t i = r * 64 + lane;
    const bool ch = i < m && i > 0 && (src[i] >> 16) != (src[i - 1] >> 16);
    changes += (uint32_t) __popcll(__ballot(ch));
  }
  if (lane == 0) wcnt[w] = changes;
  __syncthreads();
  uint32_t run = 0;
  for (uint32_t k = 0; k < w; ++k) run += wcnt[k];
  __syncthreads();  // (wcnt is read; the caller's LDS may be overwritten from here on)
  for (uint32_t r = r0; r < r1; r += RK_UNROLL)
  {
    unsigned long long e[RK_UNROLL], ep[RK_UNROLL];
#pragma unroll
    for (uint32_t u = 0; u < RK_UNROLL; ++u)
    {
      const uint32_t i = (r + u) * 64 + lane;
      const bool in = r + u < r1 && i < m;
      e[u] = in ? src[i] : ~0ull;
      ep[u] = in && i > 0 ? src[i - 1] : ~0ull;
    }
#pragma unroll
    for (uint32_t u = 0; u < RK_UNROLL; ++u)
    {
      const bool valid = e[u] != ~0ull;
      const bool ch = valid && ep[u] != ~0ull && (e[u] >> 16) != (ep[u] >> 16);
      const unsigned long long mk = __ballot(ch);
      const uint32_t upto = __builtin_amdgcn_mbcnt_hi((uint32_t) (mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mk, 0u)) + (ch ? 1u : 0u);
      if (valid)
      {
        const uint32_t pos = (uint32_t) e[u] & 0xFFFFu;
        out[pos] = ((run + upto + 1u) << 16) | pos;
      }
      run += (uint32_t) __popcll(mk);
    }
  }
}
// One heap segment of at most 4096 elements as packed 8-byte entries in LDS (buf): every thread of the workgroup calls this (NT of
// them), the heap itself belongs to wave 0.
__device__ __forceinline__ void heap_small_body(const uint32_t first, const uint32_t last, uint32_t *key, uint32_t *idx, hent *buf, const uint32_t NT)
{
  const uint32_t m = last - first;
  uint32_t *gk = key + first, *gx = idx + first;
  for (uint32_t i = threadIdx.x; i < m; i += NT) buf[i] = ((hent) gk[i] << 32) | gx[i];
  __syncthreads();
  if (threadIdx.x < 64)
  {
    LdsMem mem{buf};
    make_heap_wave(mem, m);
    sort_heap_asm<false>(buf, m, 1);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < m; i += NT)
  {
    const hent e = buf[i];
    st_through(gk + i, hkey(e));
    st_through(gx + i, (uint32_t) e);
  }
}
// One heap segment of more than HEAP_BIG_MIN elements by a workgroup of NT threads that owns `dyn`: 4 * (cap32 + 3) bytes of LDS
// (cap32 odd; at least 8 * HEAP_LARGE bytes when segments beyond HEAP_RANKED_MAX elements may come).  The loads, the ranking and
// the final gather of a 40 000-element segment are 700 dependent round trips for a lone wave (0.9 ms) and a fraction of that for
// sixteen; the heap itself belongs to wave 0, the other waves sleep at the barriers meanwhile.
__device__ __forceinline__ void heap_big_body(const uint32_t first, const uint32_t last, uint32_t *key, uint32_t *idx, hent *scratch, uint32_t *scratch32, uint32_t *scratch32b,
                                              unsigned long long *rka, unsigned long long *rkb, hent *dyn, const uint32_t cap32, const uint32_t NT)
{
  const uint32_t m = last - first;
  uint32_t *gk = key + first, *gx = idx + first;
  hent *buf = scratch + first;
  const bool w0 = threadIdx.x < 64;  // the wave that owns the heap
  for (uint32_t i = threadIdx.x; i < m; i += NT) buf[i] = ((hent) gk[i] << 32) | gx[i];
  __syncthreads();
  if (m <= HEAP_RANKED_MAX)
  {
    // ranked 4-byte entries ((rank + 1) << 16 | local index, never 0): up to cap32 of them fit LDS (slots 1..m of l32,
    // slot 0 scratch, two zero slots behind).  buf keeps the packed originals; the sorted entries end up in g32.
    const unsigned long long tp0 = wall_clock64();
    uint32_t *l32 = reinterpret_cast<uint32_t *>(dyn);
    uint32_t *g32 = scratch32 + first;
    const bool fits = m <= cap32;
    unsigned long long tp1, tp2, tp3;
    if (fits)
    {
      wg_ranked_entries(gk, m, rka + first, rkb + first, l32 + 1, l32, l32 + (NT / 64 + 1) * 256, NT);
      __syncthreads();
      if (threadIdx.x < 3) l32[threadIdx.x == 0 ? 0 : m + threadIdx.x] = 0;
      __syncthreads();
      tp1 = wall_clock64();
      {
        LdsMemT<E32> mem{l32 + 1};
        make_heap_block(mem, m, NT);
      }
      tp2 = tp3 = wall_clock64();
      if (w0) sort_heap_lds_q(l32 + 1, m, g32);
    }
    else
    {
      wg_ranked_entries(gk, m, rka + first, rkb + first, g32, l32, l32 + (NT / 64 + 1) * 256, NT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      tp1 = wall_clock64();
      {
        GlbMemT<E32> gmem{g32};
        make_heap_block(gmem, m, NT);
      }
      tp2 = wall_clock64();
      if (m < 2 * (cap32 + 1))
      {
        // the upper levels to LDS, the rest (all leaves) to the overflow array; the pops that bring the heap down to the LDS part
        uint32_t *ovf = scratch32b + first;  // ovf[slot], slots cap32 + 1 .. m + 2
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < m; i += NT)
        {
          const uint32_t e = g32[i];
          if (i < cap32) l32[1 + i] = e; else ovf[1 + i] = e;
        }
        if (threadIdx.x < 3) l32[threadIdx.x == 0 ? 0 : cap32 + threadIdx.x] = 0;
        if (threadIdx.x < 2) ovf[m + 1 + threadIdx.x] = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (w0) sort_heap_hybrid(l32 + 1, m, cap32, ovf, g32);
        tp3 = wall_clock64();
      }
      else
      {
        if (w0)
        {
          sort_heap_asm32<true>(g32, m, cap32);  // pops in global memory until the heap fits LDS
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        tp3 = wall_clock64();
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cap32; i += NT) l32[1 + i] = g32[i];
        if (threadIdx.x < 3) l32[threadIdx.x == 0 ? 0 : cap32 + threadIdx.x] = 0;
      }
      __syncthreads();
      if (w0) sort_heap_lds_q(l32 + 1, cap32, g32);
    }
    __syncthreads();
    if (threadIdx.x == 0) g32[0] = l32[1];  // the last element never leaves the root
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long tp4 = wall_clock64();
    if (threadIdx.x == 0 && m > 36000)
    {
      g_heap_phase[0] = m;
      g_heap_phase[1] = tp1 - tp0;
      g_heap_phase[2] = tp2 - tp1;
      g_heap_phase[3] = tp3 - tp2;
      g_heap_phase[4] = tp4 - tp3;
    }
    for (uint32_t i = threadIdx.x; i < m; i += NT)
    {
      const hent e = buf[g32[i] & 0xFFFFu];
      st_through(gk + i, hkey(e));
      st_through(gx + i, (uint32_t) e);
    }
    return;
  }
  // beyond the ranked form: heapify and pop in global memory until the heap fits LDS as packed entries, then finish there
  if (w0)
  {
    GlbMem gmem{buf};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    make_heap_wave(gmem, m);
    sort_heap_asm<true>(buf, m, HEAP_LARGE);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < HEAP_LARGE; i += NT) dyn[i] = buf[i];
  __syncthreads();
  if (w0) sort_heap_asm<false>(dyn, HEAP_LARGE, 1);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < HEAP_LARGE; i += NT) buf[i] = dyn[i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < m; i += NT)
  {
    const hent e = buf[i];
    st_through(gk + i, hkey(e));
    st_through(gx + i, (uint32_t) e);
  }
}
// cls 0: len <= HEAP_SMALL (static LDS), 1: <= HEAP_BIG_MIN (dynamic LDS), 2: larger, one workgroup of HEAP_BIG_THREADS per CU;
// (lo, hi]: the sizes this launch takes
template <int CLS> __global__ __launch_bounds__(CLS == 2 ? HEAP_BIG_THREADS : 64) void k_se_heapsort(const HeapSeg *__restrict__ hs, uint32_t nh, uint32_t *key, uint32_t *idx, hent *scratch, uint32_t lo, uint32_t hi,
                                                                                                    uint32_t *scratch32, uint32_t *scratch32b, unsigned long long *rka, unsigned long long *rkb)
{
  extern __shared__ __attribute__((aligned(16))) hent dyn[];
  __shared__ hent stat[CLS == 0 ? HEAP_SMALL + HEAP_PAD : 1];
  const uint32_t s = blockIdx.x;
  if (s >= nh) return;
  const HeapSeg sg = hs[s];
  const uint32_t m = sg.last - sg.first;
  if (m <= lo || m > hi) return;
  if (CLS == 2)
    heap_big_body(sg.first, sg.last, key, idx, scratch, scratch32, scratch32b, rka, rkb, dyn, HEAP_LARGE32, HEAP_BIG_THREADS);
  else
    heap_small_body(sg.first, sg.last, key, idx, CLS == 0 ? stat : dyn, 64);
}

// ---- a partition level in three passes over the keys --------------------------------------------------------------------
// (The first form materialised the stopper flags, their prefix sums, the owner of every compact index and two position lists in
// seven launches: ~96 B per live element and level.)  A tile of LV_TILE compact indices recomputes its flags from the keys
// wherever it needs them:
//   k_lv_count   tile totals (#L-stoppers | #R-stoppers << 32)
//   k_lv_sums    one workgroup: exclusive scan of the tile totals, grand total -> segbase[ns]
//   k_lv_lists   running counts inside the tile -> the j-th L-stopper of the whole level goes to posL[j], the j-th R-stopper
//                (in forward order) to posR[j]; the thread that owns a segment's first index leaves the running counts
//                there in segbase[s], so a segment's stoppers are posL[segbase[s] ..), posR[segbase[s] >> 32 ..)
//   k_lv_swap    pair j of segment s = (posL[bL + j], posR[bR + nR - 1 - j]): swap while l_j < r_j, then the cut
// ~27 B per element and level, five launches with the children kernel.  A live segment holds more than FIN_MAX = LV_TILE
// elements, so a tile touches at most two segments: one binary search per tile, no owner array.
constexpr uint32_t LV_EPT = LV_TILE / 256;
static_assert(LV_TILE <= FIN_MAX, "a tile must not span more than one segment boundary");
struct LvTile
{
  uint32_t s0;
  Seg a, b;
};
// executed by every thread of the block after a barrier; c0 = first compact index of the tile
__device__ __forceinline__ void lv_tile_setup(const Seg *__restrict__ segs, uint32_t ns, uint32_t c0, LvTile *sh, const uint32_t *__restrict__ tile_seg)
{
  if (threadIdx.x == 0)
  {
    const uint32_t s = tile_seg[c0 / LV_TILE];
    sh->s0 = s;
    sh->a = segs[s];
    Seg z = {};
    z.cbase = 0xFFFFFFFFu;
    z.depth = -1;
    sh->b = s + 1 < ns ? segs[s + 1] : z;
  }
  __syncthreads();
}
__device__ __forceinline__ uint32_t lv_flags(const Seg &sg, uint32_t c, uint32_t k)
{
  uint32_t v = 0;
  if (sg.depth >= 0 && c > sg.cbase)
  {
    if (k >= sg.pivot) v |= 1u;         // !(key < pivot): the left scan stops here
    if (k <= sg.pivot) v |= 1u << 16;   // !(pivot < key): the right scan stops here
  }
  return v;
}
// one tile of a level pass: vb = tile number (blockIdx.x); the shared
// scratch belongs to the caller; every thread of the 256-thread workgroup calls it, ns / na are the level's live counts
__device__ __forceinline__ void lv_count_tile(const Seg *segs, uint32_t ns, const uint32_t *key, uint32_t na, unsigned long long *tile_cnt,
                                              const uint32_t *tile_seg, uint32_t vb, LvTile *sh, uint32_t *s_scan)
{
  const uint32_t c0 = vb * LV_TILE;
  if (c0 >= na) return;
  lv_tile_setup(segs, ns, c0, sh, tile_seg);
  const Seg A = sh->a, B = sh->b;
  uint32_t acc = 0;
#pragma unroll
  for (uint32_t k = 0; k < LV_EPT; ++k)
  {
    const uint32_t c = c0 + k * 256 + threadIdx.x;
    if (c < na)
    {
      const Seg &sg = c >= B.cbase ? B : A;
      acc += lv_flags(sg, c, key[sg.first + (c - sg.cbase)]);
    }
  }
  uint32_t tot;
  (void) prims::block_exclusive_scan(acc, s_scan, tot);
  if (threadIdx.x == 0) tile_cnt[vb] = (unsigned long long) (tot & 0xFFFFu) | ((unsigned long long) (tot >> 16) << 32);
}
__global__ __launch_bounds__(256) void k_lv_count(const Seg *__restrict__ segs, uint32_t ns, const uint32_t *__restrict__ key, uint32_t na, unsigned long long *__restrict__ tile_cnt,
                                                  const uint32_t *__restrict__ lvl, const uint32_t *__restrict__ tile_seg)
{
  __shared__ LvTile sh;
  __shared__ uint32_t s_scan[prims::WAVES];
  if (lvl)
  {
    ns = lvl[0];
    na = lvl[1];
  }
  lv_count_tile(segs, ns, key, na, tile_cnt, tile_seg, blockIdx.x, &sh, s_scan);
}
constexpr uint32_t LV_SUM_THREADS = 1024;
// exclusive scan of the tile totals by ONE workgroup of T threads (wsum: T / 64 entries)
template <uint32_t T> __device__ __forceinline__ void lv_sums_body(unsigned long long *tile_cnt, uint32_t ns, uint32_t na, unsigned long long *segbase,
                                                                   unsigned long long *wsum)
{
  const uint32_t nt = (na + LV_TILE - 1) / LV_TILE, t = threadIdx.x, lane = t & 63, w = t >> 6;
  unsigned long long carry = 0;
  for (uint32_t base = 0; base < nt; base += T)
  {
    const uint32_t i = base + t;
    const unsigned long long v = i < nt ? tile_cnt[i] : 0ull;
    const unsigned long long inc = prims::wave_inclusive_scan(v);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned long long pre = 0, tot = 0;
    for (uint32_t j = 0; j < T / 64; ++j)
    {
      const unsigned long long x = wsum[j];
      if (j < w) pre += x;
      tot += x;
    }
    __syncthreads();
    if (i < nt) tile_cnt[i] = carry + pre + inc - v;
    carry += tot;
  }
  if (t == 0) segbase[ns] = carry;
}
__global__ __launch_bounds__(LV_SUM_THREADS) void k_lv_sums(unsigned long long *__restrict__ tile_cnt, uint32_t ns, uint32_t na, unsigned long long *__restrict__ segbase,
                                                            const uint32_t *__restrict__ lvl)
{
  __shared__ unsigned long long wsum[LV_SUM_THREADS / 64];
  if (lvl)
  {
    ns = lvl[0];
    na = lvl[1];
  }
  lv_sums_body<LV_SUM_THREADS>(tile_cnt, ns, na, segbase, wsum);
}
constexpr uint32_t LV_KEYS_LDS = LV_TILE + LV_TILE / 32;  // one pad word per 32: a thread's LV_EPT consecutive keys spread over the banks
__device__ __forceinline__ void lv_lists_tile(const Seg *segs, uint32_t ns, const uint32_t *key, uint32_t na,
                                              const unsigned long long *tile_base, uint32_t *posL, uint32_t *posR,
                                              unsigned long long *segbase, const uint32_t *tile_seg, uint32_t vb, LvTile *sh, uint32_t *s_scan,
                                              uint32_t *s_key)
{
  const uint32_t c0 = vb * LV_TILE;
  if (c0 >= na) return;
  lv_tile_setup(segs, ns, c0, sh, tile_seg);
  const Seg A = sh->a, B = sh->b;
  const uint32_t s0 = sh->s0;
#pragma unroll
  for (uint32_t k = 0; k < LV_EPT; ++k)
  {
    const uint32_t e = k * 256 + threadIdx.x, c = c0 + e;
    if (c < na)
    {
      const Seg &sg = c >= B.cbase ? B : A;
      s_key[e + (e >> 5)] = key[sg.first + (c - sg.cbase)];
    }
  }
  __syncthreads();
  uint32_t f[LV_EPT], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < LV_EPT; ++k)
  {
    const uint32_t e = threadIdx.x * LV_EPT + k, c = c0 + e;
    uint32_t v = 0;
    if (c < na) v = lv_flags(c >= B.cbase ? B : A, c, s_key[e + (e >> 5)]);
    f[k] = v;
    sum += v;
  }
  uint32_t tot;
  const uint32_t ex = prims::block_exclusive_scan(sum, s_scan, tot);
  const unsigned long long tb = tile_base[vb];
  uint32_t runL = (uint32_t) tb + (ex & 0xFFFFu), runR = (uint32_t) (tb >> 32) + (ex >> 16);
#pragma unroll
  for (uint32_t k = 0; k < LV_EPT; ++k)
  {
    const uint32_t e = threadIdx.x * LV_EPT + k, c = c0 + e;
    if (c >= na) break;
    const bool second = c >= B.cbase;
    const Seg &sg = second ? B : A;
    if (c == sg.cbase) segbase[s0 + (second ? 1u : 0u)] = (unsigned long long) runL | ((unsigned long long) runR << 32);
    const uint32_t p = sg.first + (c - sg.cbase);
    if (f[k] & 1u) posL[runL++] = p;
    if (f[k] >> 16) posR[runR++] = p;
  }
}
__global__ __launch_bounds__(256) void k_lv_lists(const Seg *__restrict__ segs, uint32_t ns, const uint32_t *__restrict__ key, uint32_t na,
                                                  const unsigned long long *__restrict__ tile_base, uint32_t *__restrict__ posL, uint32_t *__restrict__ posR,
                                                  unsigned long long *__restrict__ segbase, const uint32_t *__restrict__ lvl, const uint32_t *__restrict__ tile_seg)
{
  __shared__ LvTile sh;
  __shared__ uint32_t s_scan[prims::WAVES];
  __shared__ uint32_t s_key[LV_KEYS_LDS];
  if (lvl)
  {
    ns = lvl[0];
    na = lvl[1];
  }
  lv_lists_tile(segs, ns, key, na, tile_base, posL, posR, segbase, tile_seg, blockIdx.x, &sh, s_scan, s_key);
}
__device__ __forceinline__ void lv_swap_tile(Seg *segs, uint32_t ns, uint32_t *key, uint32_t *idx, uint32_t na,
                                             const unsigned long long *segbase, const uint32_t *posL, const uint32_t *posR,
                                             const uint32_t *tile_seg, uint32_t vb, LvTile *sh, unsigned long long *s_base)
{
  const uint32_t c0 = vb * LV_TILE;
  if (c0 >= na) return;
  lv_tile_setup(segs, ns, c0, sh, tile_seg);
  const Seg A = sh->a, B = sh->b;
  const uint32_t s0 = sh->s0;
  if (threadIdx.x < 3) s_base[threadIdx.x] = s0 + threadIdx.x <= ns ? segbase[s0 + threadIdx.x] : 0ull;
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < LV_EPT; ++k)
  {
    const uint32_t c = c0 + k * 256 + threadIdx.x;
    if (c >= na) break;
    const uint32_t second = c >= B.cbase ? 1u : 0u;
    const Seg &sg = second ? B : A;
    if (sg.depth < 0) continue;
    const uint32_t first = sg.first, s = s0 + second;
    const unsigned long long base = s_base[second], end = s_base[second + 1];
    const uint32_t bL = (uint32_t) base, bR = (uint32_t) (base >> 32);
    const uint32_t nL = (uint32_t) end - bL, nR = (uint32_t) (end >> 32) - bR;
    const uint32_t m = nL < nR ? nL : nR, j = c - sg.cbase;
    if (j > m) continue;
    const uint32_t INF = 0xFFFFFFFFu;
    const uint32_t lj = j < nL ? posL[bL + j] : INF;
    const uint32_t rj = j < nR ? posR[bR + (nR - 1 - j)] : first;
    if ((j < nL) && (j < nR) && (lj < rj))
    {
      const uint32_t k1 = key[lj], k2 = key[rj], x1 = idx[lj], x2 = idx[rj];
      key[lj] = k2;
      key[rj] = k1;
      idx[lj] = x2;
      idx[rj] = x1;
    }
    else
    {
      bool prev_cont = false;
      uint32_t rprev = 0;
      if (j > 0)
      {
        const uint32_t lp = posL[bL + j - 1];  // j - 1 < m <= nL, nR
        rprev = posR[bR + (nR - j)];
        prev_cont = lp < rprev;
      }
      if (j == 0)
        segs[s].cut = lj;  // J = 0: the left scan's first stop (exists: median-of-3 sentinel)
      else if (prev_cont)
        segs[s].cut = lj < rprev ? lj : rprev;  // cut = min(l_J, r_{J-1})
    }
  }
}
__global__ __launch_bounds__(256) void k_lv_swap(Seg *__restrict__ segs, uint32_t ns, uint32_t *__restrict__ key, uint32_t *__restrict__ idx, uint32_t na,
                                                 const unsigned long long *__restrict__ segbase, const uint32_t *__restrict__ posL, const uint32_t *__restrict__ posR,
                                                 const uint32_t *__restrict__ lvl, const uint32_t *__restrict__ tile_seg)
{
  __shared__ LvTile sh;
  __shared__ unsigned long long s_base[3];
  if (lvl)
  {
    ns = lvl[0];
    na = lvl[1];
  }
  lv_swap_tile(segs, ns, key, idx, na, segbase, posL, posR, tile_seg, blockIdx.x, &sh, s_base);
}

__global__ void k_se_child_count(const Seg *__restrict__ segs, uint32_t ns, unsigned long long *__restrict__ cnt)
{
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  Seg sg = segs[s];
  unsigned long long v = 0;
  if (sg.depth >= 0)
  {
    const uint32_t a = sg.cut - sg.first, b = sg.last - sg.cut;
    if (a > FIN_MAX) v += 1ull | ((unsigned long long) a << 32);
    if (b > FIN_MAX) v += 1ull | ((unsigned long long) b << 32);
  }
  cnt[s] = v;
}
__global__ void k_se_child_write(const Seg *__restrict__ segs, uint32_t ns, const unsigned long long *__restrict__ off, Seg *__restrict__ out, FinSeg *__restrict__ fl,
                                 uint32_t *__restrict__ fin, uint32_t *__restrict__ tile_seg)
{
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  Seg sg = segs[s];
  if (sg.depth < 0) return;
  uint32_t o = (uint32_t) off[s], cb = (uint32_t) (off[s] >> 32);
  const uint32_t a = sg.cut - sg.first, b = sg.last - sg.cut;
  if (a > FIN_MAX)
  {
    Seg c = sg;
    c.last = sg.cut;
    c.cbase = cb;
    lv_mark_tiles(tile_seg, o, cb, a);
    cb += a;
    out[o++] = c;
  }
  else if (a > 16)
    fin_append(fl, fin, sg.first, sg.cut, sg.depth);
  if (b > FIN_MAX)
  {
    Seg c = sg;
    c.first = sg.cut;
    c.cbase = cb;
    lv_mark_tiles(tile_seg, o, cb, b);
    out[o++] = c;
  }
  else if (b > 16)
    fin_append(fl, fin, sg.cut, sg.last, sg.depth);
}

// Late levels hold a few dozen to a few hundred live segments and are bound by launches and by the host round trip
// that sizes the next level.  For up to CHILD_FUSED segments (a few per thread) one workgroup counts the children, scans the counts, writes
// the children, picks their pivots (k_se_child_count + exclusive_scan + k_se_child_write + the next level's k_se_pivot: six
// launches in one) and leaves the next level's {segments, elements} in lvl, so that the host can queue several levels
// before it looks (the level kernels read lvl; their grids are sized for bounds).
constexpr uint32_t CHILD_THREADS = 1024, CHILD_PER = 8, CHILD_FUSED = CHILD_THREADS * CHILD_PER;
template <uint32_t T> __device__ __forceinline__ void children_small_body(const Seg *segs, uint32_t *lvl, Seg *out, FinSeg *fl,
                                                                          uint32_t *fin, uint32_t *key, uint32_t *idx, uint32_t *err,
                                                                          uint2 *heap_list, uint32_t *tile_seg, unsigned long long *wsum, uint32_t *s_max)
{
  if (threadIdx.x == 0) *s_max = 0;
  const uint32_t ns = lvl[0];
  const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
  // thread t takes segments [t * per, (t + 1) * per): one each while they are few
  const uint32_t per = (ns + T - 1) / T;
  unsigned long long v = 0;
  uint32_t lmax = 0;  // largest live child (lvl[2]: the host hands the tail of the loop to k_se_tail once it is small)
  for (uint32_t k = 0; k < per; ++k)
  {
    const uint32_t s = t * per + k;
    if (s >= ns) break;
    const Seg sg = segs[s];
    if (sg.depth < 0) continue;
    const uint32_t a = sg.cut - sg.first, b = sg.last - sg.cut;
    if (a > FIN_MAX) v += 1ull | ((unsigned long long) a << 32);
    if (b > FIN_MAX) v += 1ull | ((unsigned long long) b << 32);
    lmax = max(lmax, max(a > FIN_MAX ? a : 0u, b > FIN_MAX ? b : 0u));
  }
  // exclusive scan of (count | elements << 32) over the workgroup
  unsigned long long inc = v;
  for (int d = 1; d < 64; d <<= 1)
  {
    const unsigned long long o = __shfl_up(inc, d, 64);
    if ((int) lane >= d) inc += o;
  }
  if (lane == 63) wsum[w] = inc;
  if (lmax) atomicMax(s_max, lmax);
  __syncthreads();  // (every thread has read lvl[0] by now)
  unsigned long long base = 0, tot = 0;
  for (uint32_t i = 0; i < T / 64; ++i)
  {
    const unsigned long long x = wsum[i];
    if (i < w) base += x;
    tot += x;
  }
  if (t == 0)
  {
    lvl[0] = (uint32_t) tot;
    lvl[1] = (uint32_t) (tot >> 32);
    lvl[2] = *s_max;
  }
  const unsigned long long off = base + inc - v;
  uint32_t o = (uint32_t) off, cb = (uint32_t) (off >> 32);
  for (uint32_t k = 0; k < per; ++k)
  {
    const uint32_t s = t * per + k;
    if (s >= ns) break;
    const Seg sg = segs[s];
    if (sg.depth < 0) continue;
    const uint32_t a = sg.cut - sg.first, b = sg.last - sg.cut;
    if (a > FIN_MAX)
    {
      Seg c = sg;
      c.last = sg.cut;
      c.cbase = cb;
      lv_mark_tiles(tile_seg, o, cb, a);
      cb += a;
      pivot_one(c, key, idx, err, heap_list);
      out[o++] = c;
    }
    else if (a > 16)
      fin_append(fl, fin, sg.first, sg.cut, sg.depth);
    if (b > FIN_MAX)
    {
      Seg c = sg;
      c.first = sg.cut;
      c.cbase = cb;
      lv_mark_tiles(tile_seg, o, cb, b);
      cb += b;
      pivot_one(c, key, idx, err, heap_list);
      out[o++] = c;
    }
    else if (b > 16)
      fin_append(fl, fin, sg.cut, sg.last, sg.depth);
  }
}
__global__ __launch_bounds__(CHILD_THREADS) void k_se_children_small(const Seg *__restrict__ segs, uint32_t *__restrict__ lvl, Seg *__restrict__ out, FinSeg *__restrict__ fl,
                                                                     uint32_t *__restrict__ fin, uint32_t *__restrict__ key, uint32_t *__restrict__ idx, uint32_t *__restrict__ err,
                                                                     uint2 *__restrict__ heap_list, uint32_t *__restrict__ tile_seg)
{
  __shared__ unsigned long long wsum[CHILD_THREADS / 64];
  __shared__ uint32_t s_max;
  children_small_body<CHILD_THREADS>(segs, lvl, out, fl, fin, key, idx, err, heap_list, tile_seg, wsum, &s_max);
}

#ifndef TL_UNROLL_N
#define TL_UNROLL_N 24
#endif
// ---- the rest of the loop, one workgroup per live segment, in rounds -------------------------------------------------------------
// The device-wide level loop pays five dependent launches per level (~45-70 us whatever the level holds) for up to 2 lg n
// levels; on a WGS sample 25-30 of them exist only for a few dozen degenerate segments that lose a few per cent per level.  Here a
// workgroup of 1024 threads takes ONE live segment and follows its SPINE: it partitions the segment (the same partition as
// k_lv_count / _lists / _swap: stoppers counted per wave over contiguous chunks, their positions written to the segment's own
// slice of posL / posR by wave ballots, pair j swapped while l_j < r_j, cut = min(l_J, r_{J-1})), hands the SMALLER side on - to
// the finisher list, to the heap list when the depth budget is used up (both exactly as the level loop does), or, if it is still
// larger than FIN_MAX, to the segment list of the next round - and carries on with the larger side.  A segment that is handed on is
// at most half of its parent, so floor(lg(largest / FIN_MAX)) + 1 rounds finish every tree whatever its shape, each round one
// launch sized for the bound na / FIN_MAX with the actual count read on the device (no host look in between); a degenerate chain
// of 30 partitions is 30 nodes of ONE workgroup (~8 us each: two passes over the keys and a handful of workgroup barriers)
// instead of 30 levels of five launches.  The sides of a partition never interact, so the order in which the tree is walked does
// not matter for the result.
constexpr uint32_t TL_THREADS = 1024, TL_WAVES = TL_THREADS / 64, TL_UNROLL = TL_UNROLL_N;
// child [first, last) of a node with `depth` left (one lane): 0 = nothing left to do here (went to the finisher list, to the heap list
// or is at most 16 long), 1 = still live: c holds it with its pivot picked
__device__ __forceinline__ int tl_child(uint32_t first, uint32_t last, int32_t depth, uint32_t *key, uint32_t *idx, uint32_t *err, uint2 *heap_list, FinSeg *fl, uint32_t *fin, Seg &c)
{
  const uint32_t sz = last - first;
  if (sz > FIN_MAX)
  {
    c = Seg{};
    c.first = first;
    c.last = last;
    c.depth = depth;
    pivot_one(c, key, idx, err, heap_list);  // depth 0: the segment goes to the heap list (c.depth = -1)
    return c.depth >= 0 ? 1 : 0;
  }
  if (sz > 16) fin_append(fl, fin, first, last, depth);
  return 0;
}
// spine = 0: one partition per workgroup, BOTH sides handed on (the wide top of the trees: a round per level, every node of a level
// at once); spine = 1: the workgroup carries on with the larger side as described above
__global__ __launch_bounds__(TL_THREADS) void k_se_tail_round(const Seg *__restrict__ in, const uint32_t *__restrict__ n_in, Seg *__restrict__ out, uint32_t *__restrict__ n_out,
                                                              uint32_t out_cap, uint32_t *key, uint32_t *idx, uint32_t *posL, uint32_t *posR, FinSeg *fl, uint32_t *fin, uint32_t *err,
                                                              uint2 *heap_list, int spine)
{
  __shared__ Seg s_cur;
  __shared__ Seg s_kid[2];
  __shared__ int s_live[2];
  __shared__ uint32_t s_cnt[2][TL_WAVES];
  __shared__ uint32_t s_cut;
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (blockIdx.x >= *n_in) return;
  if (tid == 0)
  {
    s_cur = in[blockIdx.x];
    s_cut = 0xFFFFFFFFu;
  }
  __syncthreads();
  if (s_cur.depth < 0) return;  // (k_se_pivot sent it to the heap list)
  for (;;)
  {
    const Seg sg = s_cur;
    const uint32_t first = sg.first, last = sg.last, pivot = sg.pivot;
    // stoppers among (first, last): key >= pivot stops the scan from the left, key <= pivot the scan from the right; wave w owns
    // a contiguous chunk (rows of 64 consecutive positions)
    const uint32_t lo = first + 1, total = last - lo;
    const uint32_t chunk = (((total + TL_WAVES - 1) / TL_WAVES) + 63u) & ~63u;
    const uint32_t wbeg = lo + min(total, w * chunk), wend = lo + min(total, (w + 1) * chunk);
    uint32_t cL = 0, cR = 0;
    {
      // (TL_UNROLL rows in flight per wave: a single CU has to stream a segment of 10^5 keys in a few microseconds)
      uint32_t p = wbeg + lane;
      for (; p + (TL_UNROLL - 1) * 64 < wend; p += TL_UNROLL * 64)
      {
        uint32_t k[TL_UNROLL];
#pragma unroll
        for (uint32_t u = 0; u < TL_UNROLL; ++u) k[u] = key[p + u * 64];
#pragma unroll
        for (uint32_t u = 0; u < TL_UNROLL; ++u)
        {
          cL += k[u] >= pivot ? 1u : 0u;
          cR += k[u] <= pivot ? 1u : 0u;
        }
      }
      for (; p < wend; p += 64)
      {
        const uint32_t k = key[p];
        cL += k >= pivot ? 1u : 0u;
        cR += k <= pivot ? 1u : 0u;
      }
    }
    for (int d = 32; d >= 1; d >>= 1)
    {
      cL += __shfl_xor(cL, d, 64);
      cR += __shfl_xor(cR, d, 64);
    }
    if (lane == 0)
    {
      s_cnt[0][w] = cL;
      s_cnt[1][w] = cR;
    }
    __syncthreads();
    uint32_t runL = 0, runR = 0, nL = 0, nR = 0;
    for (uint32_t i = 0; i < TL_WAVES; ++i)
    {
      const uint32_t a = s_cnt[0][i], b = s_cnt[1][i];
      if (i < w)
      {
        runL += a;
        runR += b;
      }
      nL += a;
      nR += b;
    }
    // the segment's own slice of the position lists: at most last - first - 1 entries each
    uint32_t *pL = posL + first, *pR = posR + first;
    {
      auto place = [&](uint32_t p, bool valid, uint32_t k) {
        const bool fL = valid && k >= pivot, fR = valid && k <= pivot;
        const unsigned long long mL = __ballot(fL), mR = __ballot(fR);
        const uint32_t bL = __builtin_amdgcn_mbcnt_hi((uint32_t) (mL >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mL, 0u));
        const uint32_t bR = __builtin_amdgcn_mbcnt_hi((uint32_t) (mR >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mR, 0u));
        if (fL) pL[runL + bL] = p;
        if (fR) pR[runR + bR] = p;
        runL += (uint32_t) __popcll(mL);
        runR += (uint32_t) __popcll(mR);
      };
      uint32_t p0 = wbeg;
      for (; p0 + TL_UNROLL * 64 <= wend; p0 += TL_UNROLL * 64)
      {
        uint32_t k[TL_UNROLL];
#pragma unroll
        for (uint32_t u = 0; u < TL_UNROLL; ++u) k[u] = key[p0 + u * 64 + lane];
#pragma unroll
        for (uint32_t u = 0; u < TL_UNROLL; ++u) place(p0 + u * 64 + lane, true, k[u]);
      }
      for (; p0 < wend; p0 += 64)
      {
        const uint32_t p = p0 + lane;
        const bool valid = p < wend;
        place(p, valid, valid ? key[p] : 0u);
      }
    }
    __syncthreads();
    // pair j = (l_j, r_j): swapped while l_j < r_j; the first pair that is not gives the cut (k_lv_swap, one pair per index)
    const uint32_t mm = nL < nR ? nL : nR;
    // the pairs are swapped until they cross: l_j < r_j holds for j < J only, so the first batch without a swap ends a thread's work
    // early (J is a few per cent of the segment on a degenerate one); four pairs in flight per thread
    constexpr uint32_t SW = 4;
    for (uint32_t j0 = tid; j0 <= mm; j0 += SW * TL_THREADS)
    {
      uint32_t lj[SW], rj[SW];
      bool sw[SW];
#pragma unroll
      for (uint32_t u = 0; u < SW; ++u)
      {
        const uint32_t j = j0 + u * TL_THREADS;
        lj[u] = j < nL ? pL[j] : 0xFFFFFFFFu;
        rj[u] = j < nR ? pR[nR - 1 - j] : first;
      }
      uint32_t k1[SW], k2[SW], x1[SW], x2[SW];
#pragma unroll
      for (uint32_t u = 0; u < SW; ++u)
      {
        const uint32_t j = j0 + u * TL_THREADS;
        sw[u] = j <= mm && (j < nL) && (j < nR) && (lj[u] < rj[u]);
        if (sw[u])
        {
          k1[u] = key[lj[u]];
          k2[u] = key[rj[u]];
          x1[u] = idx[lj[u]];
          x2[u] = idx[rj[u]];
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < SW; ++u)
        if (sw[u])
        {
          key[lj[u]] = k2[u];
          key[rj[u]] = k1[u];
          idx[lj[u]] = x2[u];
          idx[rj[u]] = x1[u];
        }
#pragma unroll
      for (uint32_t u = 0; u < SW; ++u)
      {
        const uint32_t j = j0 + u * TL_THREADS;
        if (j > mm || sw[u]) continue;
        const uint32_t ljj = lj[u];
        bool prev_cont = false;
        uint32_t rprev = 0;
        if (j > 0)
        {
          const uint32_t lp = pL[j - 1];
          rprev = pR[nR - j];
          prev_cont = lp < rprev;
        }
        if (j == 0)
          s_cut = ljj;
        else if (prev_cont)
          s_cut = ljj < rprev ? ljj : rprev;
      }
    }
    __syncthreads();
    const uint32_t cut = s_cut;
    if (cut == 0xFFFFFFFFu || cut <= first || cut > last)
    {
      if (tid == 0) atomicOr(err, 8u);  // (a partition always yields a cut inside the segment: reported, never silent)
      return;
    }
    // the two sides, one lane each (their pivots are two dependent round trips apiece): side 0 = [first, cut), side 1 = [cut, last)
    if (lane == 0 && w < 2)
    {
      Seg c;
      s_live[w] = tl_child(w == 0 ? first : cut, w == 0 ? cut : last, sg.depth, key, idx, err, heap_list, fl, fin, c);
      s_kid[w] = c;
    }
    __syncthreads();
    const int big = (cut - first) >= (last - cut) ? 0 : 1;  // carry on with the larger side, hand the smaller one on
    const bool go_on = spine && s_live[big] != 0;
    if (tid == 0)
    {
      if (!spine && s_live[big])
      {
        const uint32_t slot = atomicAdd(n_out, 1u);
        if (slot < out_cap)
          out[slot] = s_kid[big];
        else
          atomicOr(err, 8u);
      }
      if (s_live[1 - big])
      {
        const uint32_t slot = atomicAdd(n_out, 1u);
        if (slot < out_cap)
          out[slot] = s_kid[1 - big];
        else
          atomicOr(err, 8u);
      }
      if (go_on)
      {
        s_cur = s_kid[big];
        s_cut = 0xFFFFFFFFu;
      }
    }
    __syncthreads();
    if (!go_on) return;
  }
}

// ---- the same introsort loop for one segment of at most FIN_MAX elements, entirely in LDS ---------------------
// Level-synchronous like the device-wide passes (pivot, stopper flags, prefix sums, position lists, swaps + cut,
// children), with the sub-segment of every element tracked incrementally.  Sub-segments that exhaust the depth limit
// are handed to the heapsort kernels through the global heap list, exactly as k_se_pivot does.
struct LSeg
{
  uint16_t first, last, cut, base;  // base = index of the first child in the next table
  uint32_t pivot;
  int32_t depth;
};
constexpr uint16_t FIN_DEAD = 0xFFFFu;

// exclusive scan over the T threads of a finisher block (T = 64: one wave, no barrier; T = 256: prims' block scan)
template <uint32_t T> __device__ __forceinline__ uint32_t fin_scan(uint32_t v, uint32_t *lds, uint32_t &total)
{
  if (T == 64)
  {
    const uint32_t inc = prims::wave_inclusive_scan(v);
    total = __shfl(inc, 63, 64);
    return inc - v;
  }
  return prims::block_exclusive_scan(v, lds, total);
}

// the LDS of one finisher workgroup (FMAX = capacity in elements)
template <uint32_t FMAX> struct FinLds
{
  uint32_t key[FMAX], idx[FMAX];
  uint32_t lr[FMAX + 1];  // exclusive prefix of (L-stopper | R-stopper << 16)
  uint16_t posL[FMAX + 2], posR[FMAX + 2];
  uint16_t segof[FMAX];
  LSeg seg[2][FMAX / 16 + 2];  // > FMAX / 17 live sub-segments
  uint32_t scan[prims::WAVES];
  uint32_t ns;
};
// where the finisher leaves the sub-segments that exhaust the depth limit: the sort's heap list (launch-per-phase form) ...
struct FinHeapToList
{
  uint32_t *err;
  uint2 *heap_list;
  __device__ __forceinline__ void add(uint32_t first, uint32_t last) const
  {
    const uint32_t slot = atomicAdd(err + 3, 1u), sz = last - first;
    heap_list[slot] = make_uint2(first, last);
    atomicAdd(err + 2, sz);
    atomicMax(err + 1, sz);
  }
};
// ... or a list in the workgroup's LDS (resident service: they become tasks once the segment is back in global memory)
struct FinHeapToLds
{
  uint32_t *cnt;  // [0] = entries, [1] = elements in them
  uint2 *list;
  __device__ __forceinline__ void add(uint32_t first, uint32_t last) const
  {
    list[atomicAdd(cnt, 1u)] = make_uint2(first, last);
    atomicAdd(cnt + 1, last - first);
  }
};
// one segment [fs.first, fs.last) by the T threads of a workgroup (all of them call this)
template <uint32_t FMAX, uint32_t T, class HOUT> __device__ __forceinline__ void fin_body(FinLds<FMAX> &L, const FinSeg fs, uint32_t *key, uint32_t *idx, const HOUT &hout)
{
  constexpr uint32_t FIN_EPT = FMAX / T;
  constexpr uint32_t FIN_SEGS = FMAX / 16 + 2;
  static_assert(FIN_SEGS <= T, "one thread per sub-segment");
  uint32_t *s_key = L.key, *s_idx = L.idx, *s_lr = L.lr;
  uint16_t *s_posL = L.posL, *s_posR = L.posR, *s_segof = L.segof;
  LSeg(*s_seg)[FIN_SEGS] = L.seg;
  uint32_t *s_scan = L.scan;
  uint32_t &s_ns = L.ns;
  const uint32_t m = fs.last - fs.first, g0 = fs.first, tid = threadIdx.x;
  for (uint32_t e = tid; e < m; e += T)
  {
    s_key[e] = key[g0 + e];
    s_idx[e] = idx[g0 + e];
    s_segof[e] = 0;
  }
  if (tid == 0)
  {
    LSeg r;
    r.first = 0;
    r.last = (uint16_t) m;
    r.cut = r.base = 0;
    r.pivot = 0;
    r.depth = fs.depth;
    s_seg[0][0] = r;
    s_ns = 1;
  }
  __syncthreads();
  int cur = 0;
  for (int level = 0; level < 200; ++level)
  {
    const uint32_t ns = s_ns;
    if (ns == 0) break;
    LSeg *S = s_seg[cur], *N = s_seg[cur ^ 1];
    // pivot step: __move_median_to_first(first, first+1, mid, last-1), or hand over to heapsort at the depth limit
    if (tid < ns)
    {
      const LSeg sg = S[tid];
      if (sg.depth == 0)
      {
        S[tid].depth = -1;
        hout.add(g0 + sg.first, g0 + sg.last);
      }
      else
      {
        const uint32_t first = sg.first, last = sg.last;
        const uint32_t a = first + 1, b = first + (last - first) / 2, c = last - 1;
        const uint32_t ka = s_key[a], kb = s_key[b], kc = s_key[c];
        uint32_t pick;
        if (ka < kb)
        {
          if (kb < kc) pick = b;
          else if (ka < kc) pick = c;
          else pick = a;
        }
        else if (ka < kc) pick = a;
        else if (kb < kc) pick = c;
        else pick = b;
        const uint32_t kf = s_key[first], kp = s_key[pick], xf = s_idx[first], xp = s_idx[pick];
        s_key[first] = kp;
        s_key[pick] = kf;
        s_idx[first] = xp;
        s_idx[pick] = xf;
        S[tid].pivot = kp;
        S[tid].depth = sg.depth - 1;
      }
    }
    __syncthreads();
    // stopper flags and their prefix sums (FIN_EPT consecutive elements per thread)
    {
      uint32_t loc[FIN_EPT], sum = 0;
      uint16_t ps = FIN_DEAD;  // the FIN_EPT consecutive elements of a thread mostly share their sub-segment
      LSeg sg = {};
#pragma unroll
      for (uint32_t k = 0; k < FIN_EPT; ++k)
      {
        const uint32_t e = tid * FIN_EPT + k;
        uint32_t v = 0;
        if (e < m)
        {
          const uint16_t s = s_segof[e];
          if (s != FIN_DEAD)
          {
            if (s != ps)
            {
              sg = S[s];
              ps = s;
            }
            if (sg.depth >= 0 && e > sg.first)
            {
              const uint32_t kk = s_key[e];
              if (kk >= sg.pivot) v |= 1u;
              if (kk <= sg.pivot) v |= 1u << 16;
            }
          }
        }
        loc[k] = sum;
        sum += v;
      }
      uint32_t total;
      const uint32_t base = fin_scan<T>(sum, s_scan, total);
#pragma unroll
      for (uint32_t k = 0; k < FIN_EPT; ++k)
      {
        const uint32_t e = tid * FIN_EPT + k;
        if (e <= m) s_lr[e] = base + loc[k];
      }
      if (m == FMAX && tid == 0) s_lr[FMAX] = total;
    }
    __syncthreads();
    // position lists: l_j from the left, r_j from the right
    {
      uint16_t ps = FIN_DEAD;
      LSeg sg = {};
      uint32_t bs = 0, en = 0;
#pragma unroll
      for (uint32_t k = 0; k < FIN_EPT; ++k)
      {
        const uint32_t e = tid * FIN_EPT + k;
        if (e >= m) continue;
        const uint16_t s = s_segof[e];
        if (s == FIN_DEAD) continue;
        if (s != ps)
        {
          sg = S[s];
          ps = s;
          bs = s_lr[sg.first];
          en = s_lr[sg.last];
        }
        if (sg.depth < 0 || e <= sg.first) continue;
        const uint32_t kk = s_key[e], here = s_lr[e];
        if (kk >= sg.pivot) s_posL[sg.first + 1 + ((here & 0xFFFFu) - (bs & 0xFFFFu))] = (uint16_t) e;
        if (kk <= sg.pivot)
        {
          const uint32_t nR = (en >> 16) - (bs >> 16), jl = (here >> 16) - (bs >> 16);
          s_posR[sg.first + 1 + (nR - 1 - jl)] = (uint16_t) e;
        }
      }
    }
    __syncthreads();
    // swaps (l_j, r_j) for j < J and the cut
#pragma unroll
    for (uint32_t k = 0; k < FIN_EPT; ++k)
    {
      const uint32_t e = tid * FIN_EPT + k;
      if (e >= m) continue;
      const uint16_t s = s_segof[e];
      if (s == FIN_DEAD) continue;
      const LSeg sg = S[s];
      if (sg.depth < 0) continue;
      const uint32_t first = sg.first, bs = s_lr[first], en = s_lr[sg.last];
      const uint32_t nL = (en & 0xFFFFu) - (bs & 0xFFFFu), nR = (en >> 16) - (bs >> 16);
      const uint32_t mm = nL < nR ? nL : nR, j = e - first;
      if (j > mm) continue;
      const uint32_t lj = j < nL ? s_posL[first + 1 + j] : 0xFFFFFFFFu;
      const uint32_t rj = j < nR ? s_posR[first + 1 + j] : first;
      if ((j < nL) && (j < nR) && (lj < rj))
      {
        const uint32_t k1 = s_key[lj], k2 = s_key[rj], x1 = s_idx[lj], x2 = s_idx[rj];
        s_key[lj] = k2;
        s_key[rj] = k1;
        s_idx[lj] = x2;
        s_idx[rj] = x1;
      }
      else
      {
        bool prev_cont = false;
        uint32_t rprev = 0;
        if (j > 0)
        {
          const uint32_t lp = s_posL[first + j];
          rprev = s_posR[first + j];
          prev_cont = lp < rprev;
        }
        if (j == 0)
          S[s].cut = (uint16_t) lj;
        else if (prev_cont)
          S[s].cut = (uint16_t) (lj < rprev ? lj : rprev);
      }
    }
    __syncthreads();
    // children (> 16 elements) form the next table
    {
      uint32_t cnt = 0, a = 0, b = 0;
      LSeg sg;
      sg.depth = -1;
      if (tid < ns)
      {
        sg = S[tid];
        if (sg.depth >= 0)
        {
          a = (uint32_t) sg.cut - sg.first;
          b = (uint32_t) sg.last - sg.cut;
          cnt = (a > 16 ? 1u : 0u) + (b > 16 ? 1u : 0u);
        }
      }
      uint32_t total;
      uint32_t o = fin_scan<T>(cnt, s_scan, total);
      if (tid < ns && sg.depth >= 0)
      {
        S[tid].base = (uint16_t) o;
        LSeg c = sg;
        c.cut = c.base = 0;
        if (a > 16)
        {
          c.first = sg.first;
          c.last = sg.cut;
          N[o++] = c;
        }
        if (b > 16)
        {
          c.first = sg.cut;
          c.last = sg.last;
          N[o] = c;
        }
      }
      if (tid == 0) s_ns = total;
    }
    __syncthreads();
    // every element moves to its child (or retires)
#pragma unroll
    for (uint32_t k = 0; k < FIN_EPT; ++k)
    {
      const uint32_t e = tid * FIN_EPT + k;
      if (e >= m) continue;
      const uint16_t s = s_segof[e];
      if (s == FIN_DEAD) continue;
      const LSeg sg = S[s];
      uint16_t nx = FIN_DEAD;
      if (sg.depth >= 0)
      {
        const uint32_t a = (uint32_t) sg.cut - sg.first, b = (uint32_t) sg.last - sg.cut;
        if (e < sg.cut)
        {
          if (a > 16) nx = sg.base;
        }
        else if (b > 16)
          nx = (uint16_t) (sg.base + (a > 16 ? 1 : 0));
      }
      s_segof[e] = nx;
    }
    __syncthreads();
    cur ^= 1;
  }
  for (uint32_t e = tid; e < m; e += T)
  {
    st_through(key + g0 + e, s_key[e]);
    st_through(idx + g0 + e, s_idx[e]);
  }
}
// `step` = +1 / -1 walks the list from the front / the back
template <uint32_t FMAX, uint32_t T> __global__ __launch_bounds__(T) void k_se_finish(const FinSeg *__restrict__ fl, uint32_t nf, int step, uint32_t *__restrict__ key,
                                                                                        uint32_t *__restrict__ idx, uint32_t *__restrict__ err, uint2 *__restrict__ heap_list)
{
  __shared__ FinLds<FMAX> L;
  if (blockIdx.x >= nf) return;
  fin_body<FMAX, T>(L, fl[(long long) step * blockIdx.x], key, idx, FinHeapToList{err, heap_list});
}

#include "sortsvc.inc"
}  // namespace

// __final_insertion_sort.  What the introsort loop (and the heapsorts) leave is ordered between segments and arbitrary
// only inside the left-over segments of at most 16 elements, so the stable sort by key is local: every such segment
// lies entirely inside a 32-element window of one of two tilings (offset 0 and offset 16), and a stable sort of every
// window of both tilings sorts the array (a window sort never disturbs what is already in order).  One half-wave per
// window: rank by counting over the 32 elements, (group, key) compared so that windows may straddle groups.
namespace
{
__global__ __launch_bounds__(256) void k_se_window_sort(uint32_t *__restrict__ key, uint32_t *__restrict__ idx, const uint32_t *__restrict__ gof, uint32_t n, uint32_t offset)
{
  const uint32_t lane = threadIdx.x & 63, half = lane & 32u, wl = lane & 31u;
  const uint64_t p = (uint64_t) offset + ((uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane;
  unsigned long long k64 = ~0ull;
  uint32_t kk = 0, xx = 0;
  if (p < n)
  {
    kk = key[p];
    xx = idx[p];
    k64 = ((unsigned long long) gof[p] << 32) | kk;
  }
  // already in order (heapsorted stretches, sorted input): nothing to do for this wave
  const unsigned long long prev = __shfl_up(k64, 1, 64);
  if (__ballot(wl != 0 && prev > k64) == 0ull) return;
  uint32_t rank = 0;
#pragma unroll 8
  for (uint32_t j = 0; j < 32; ++j)
  {
    const unsigned long long o = __shfl(k64, (int) (half + j), 64);
    rank += (o < k64 || (o == k64 && j < wl)) ? 1u : 0u;
  }
  if (p < n)
  {
    const uint64_t d = p - wl + rank;  // elements beyond n carry the largest key and rank last
    key[d] = kk;
    idx[d] = xx;
  }
}
}  // namespace

// BK_DEBUG=sortcheck (debugging): is idx still a permutation of 0 .. n-1 and does every element still carry its own key?
// (callers whose idx is not such a permutation must not set it)
__global__ __launch_bounds__(256) void k_chk_count(const uint32_t *__restrict__ idx, uint32_t n, uint32_t *__restrict__ cnt, uint32_t *__restrict__ bad)
{
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t x = idx[p];
  if (x >= n)
  {
    atomicAdd(&bad[0], 1u);
    atomicMin(&bad[2], p);
    return;
  }
  if (atomicAdd(&cnt[x], 1u) != 0u)
  {
    atomicAdd(&bad[1], 1u);
    atomicMin(&bad[3], p);
  }
}
__global__ __launch_bounds__(256) void k_chk_keys(const uint32_t *__restrict__ key, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ key0, uint32_t n, uint32_t *__restrict__ bad)
{
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t x = idx[p];
  if (x < n && key0[x] != key[p])
  {
    atomicAdd(&bad[4], 1u);
    atomicMin(&bad[5], p);
  }
}
static void sort_check(const char *phase, const uint32_t *key, const uint32_t *idx, const uint32_t *key0, uint32_t n, const uint64_t *goff, uint32_t ng, hipStream_t st, SortEmuBufs &b)
{
  uint32_t *c = b.chk_cnt.as<uint32_t>(n), *bd = b.chk_bad.as<uint32_t>(8);
  const uint32_t init[8] = {0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0xFFFFFFFFu, 0, 0};
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemset(c, 0, (size_t) n * 4));
  HIP_CHECK(hipMemcpy(bd, init, 32, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_chk_count, dim3(cdiv(n, 256)), dim3(256), 0, st, idx, n, c, bd);
  hipLaunchKernelGGL(k_chk_keys, dim3(cdiv(n, 256)), dim3(256), 0, st, key, idx, key0, n, bd);
  uint32_t h[8];
  HIP_CHECK(hipMemcpyAsync(h, bd, 32, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  if (h[0] || h[1] || h[4])
  {
    std::vector<uint64_t> go((size_t) ng + 1);
    HIP_CHECK(hipMemcpy(go.data(), goff, ((size_t) ng + 1) * 8, hipMemcpyDeviceToHost));
    auto grp = [&](uint32_t p) { return p == 0xFFFFFFFFu ? -1 : (int) (std::upper_bound(go.begin(), go.end(), (uint64_t) p) - go.begin()) - 1; };
    fprintf(stderr, "[sortemu] CHECK FAILED %s: %u payloads out of range (first at %u), %u duplicated payloads (first at position %u, group %d), %u elements whose key is not their own (first at %u, group %d); n=%u\n",
            phase, h[0], h[2], h[1], h[3], grp(h[3]), h[4], h[5], grp(h[5]), n);
  }
}

// ---- the resident sort service (sortsvc.inc), host side -----------------------------------------------------------------------
static SvcParams svc_params(SortService &S)
{
  SvcParams P;
  uint32_t *ctl = S.ctl.get<uint32_t>();
  for (int k = 0; k < 2; ++k)
  {
    P.q[k].head = ctl + 64 * k;
    P.q[k].tail = ctl + 64 * k + 32;
    P.q[k].slots = S.slots[k].get<SvcTask>();
    P.q[k].seq = S.seq[k].get<uint32_t>();
    P.q[k].mask = S.cap[k] - 1;
    P.q[k].release = 1u;  // an agent-scope release in front of every push (sortsvc.inc, visibility)
  }
  P.error = ctl + 128;
  P.stats = ctl + 160;
  P.jobs = S.jobs.get<SvcJob>();
  P.quit_d = ctl + 192;
  P.host = S.quit_dev;
  P.cap32 = S.cap32;
  P.quit_word = S.quit_word;
  P.dbg = S.dbg.get<uint32_t>();
  for (int k = 0; k < 2; ++k)
  {
    P.pos[k] = S.pos[k].get<uint32_t>();
    P.pos_cap[k] = S.pos_cap[k];
  }
  P.timeout_ticks = SVC_TIMEOUT_TICKS;
  return P;
}
void SortService::start(uint64_t n_bound, uint64_t max_group, hipStream_t after)
{
  if (running) return;
  quit_stream = after;  // (a stream of the stage that the stage probes: the word that ends the service is written by a kernel on it)
  int dev = 0, cus = 0;
  HIP_CHECK(hipGetDevice(&dev));
  HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // a quarter of the CUs to the wide workgroups (each takes a whole CU's LDS), two narrow ones on each of the others
  const int n_wide = std::min(1024, std::max(4, cus / 4));
  const int n_narrow = std::max(8, 2 * (cus - n_wide));
  if (!quit_host)
  {
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&quit_host), SVC_H_WORDS * 4, hipHostMallocMapped));  // SvcParams::host
    HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&quit_dev), quit_host, 0));
    for (int k = 0; k < 2; ++k) HIP_CHECK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&st_copy, hipStreamNonBlocking));
    hipFuncAttributes fa;
    HIP_CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_sort_service<true>)));
    // all of a CU's 160 KB but the kernel's static LDS; the ranked entries that fit (+ slot 0 and two zero slots), an odd count
    wide_lds = (size_t) ((160u * 1024u - (uint32_t) fa.sharedSizeBytes) & ~15u);
    cap32 = (uint32_t) (wide_lds / 4 - 3);
    if ((cap32 & 1u) == 0) --cap32;
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sort_service<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) wide_lds));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sort_service<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) SVC_NARROW_LDS));
  }
  // rings: every live segment holds more than 16 elements; the wide ring only sees segments above HEAP_BIG_MIN elements
  auto pow2 = [](uint64_t v) {
    uint32_t c = 1024;
    while (c < v && c < (1u << 30)) c <<= 1;
    return c;
  };
  cap[0] = pow2(n_bound / HEAP_BIG_MIN + 4096);
  cap[1] = pow2(n_bound / 16 + 4096);
  (void) ctl.as<uint32_t>(256);
  for (int k = 0; k < 2; ++k)
  {
    (void) slots[k].as<SvcTask>(cap[k]);
    (void) seq[k].as<uint32_t>(cap[k]);
  }
  (void) jobs.as<SvcJob>(SVC_MAX_JOBS);
  // every workgroup's own position lists (two per partition node): a wide one may be handed a whole group, a narrow one a node
  // of at most SVC_WIDE_MIN elements
  pos_cap[0] = (uint32_t) std::max<uint64_t>(max_group, SVC_WIDE_MIN) + 64;
  pos_cap[1] = SVC_WIDE_MIN + 64;
  (void) pos[0].as<uint32_t>(2ull * pos_cap[0] * (uint64_t) n_wide);
  (void) pos[1].as<uint32_t>(2ull * pos_cap[1] * (uint64_t) n_narrow);
  if (bk_debug("svc"))
  {
    (void) dbg.as<uint32_t>(12 * 8192 + 48);
    HIP_CHECK(hipMemsetAsync(dbg.p, 0, (12 * 8192 + 48) * 4, after));
  }
  __atomic_store_n(quit_host, 0u, __ATOMIC_SEQ_CST);
  quit_word = 0xC0DE0000u | (++starts & 0xFFFFu);
  for (uint32_t k = 1; k < SVC_H_WORDS; ++k) quit_host[k] = 0u;
  next_slot = 0;
  const SvcParams P = svc_params(*this);
  hipLaunchKernelGGL(k_svc_reset, dim3(cdiv(std::max(cap[0], cap[1]), 256)), dim3(256), 0, after, P);
  HIP_CHECK(hipStreamSynchronize(after));
  DeferredFrees::begin();
  running = true;
  hipLaunchKernelGGL(k_sort_service<true>, dim3(n_wide), dim3(1024), wide_lds, st[0], P);
  HIP_CHECK(hipGetLastError());
  // a wide workgroup needs a CU to itself: the narrow ones, which fit anywhere, are only launched when every wide one has its CU
  // (the other way round they could sit on every CU and keep the wide ones out for good)
  {
    const auto t0 = std::chrono::steady_clock::now();
    volatile uint32_t *started = quit_host + SVC_H_STARTED;
    for (;;)
    {
      int have = 0;
      for (int k = 0; k < n_wide; ++k) have += started[k] != 0u;
      if (have == n_wide) break;
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.25) break;  // (a busy device: the rest start when CUs free up)
    }
  }
  hipLaunchKernelGGL(k_sort_service<false>, dim3(n_narrow), dim3(256), SVC_NARROW_LDS, st[1], P);
  HIP_CHECK(hipGetLastError());
}
bool SortService::narrow_running(double seconds) const
{
  const auto t0 = std::chrono::steady_clock::now();
  volatile const uint32_t *flag = quit_host + SVC_H_NARROW;
  while (*flag == 0u)
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
  return true;
}
void SortService::stop()
{
  if (!running) return;
  running = false;
  // the word that ends the service, two ways: a kernel on a stream of the stage (the stage has probed it) and a copy from page-locked
  // memory on a stream of its own (a DMA engine's business, not a compute queue's) - whichever arrives first; a stage that gives
  // the service up BECAUSE one of its streams sits behind a persistent kernel's queue must not wait for that very stream
  __atomic_store_n(quit_host, quit_word, __ATOMIC_SEQ_CST);
  (void) hipMemcpyAsync(ctl.get<uint32_t>() + 192, quit_host, 4, hipMemcpyHostToDevice, st_copy);
  hipLaunchKernelGGL(k_svc_quit, dim3(1), dim3(1), 0, quit_stream, ctl.get<uint32_t>() + 192, quit_word);
  HIP_CHECK(hipStreamSynchronize(st[0]));
  HIP_CHECK(hipStreamSynchronize(st[1]));
  uint32_t h[40] = {};
  HIP_CHECK(hipMemcpy(h, ctl.get<uint32_t>() + 128, sizeof h, hipMemcpyDeviceToHost));
  for (int k = 0; k < 8; ++k) stats[k] = h[32 + k];
  DeferredFrees::end();
  if (bk_debug("svc"))
  {
    uint32_t c[128] = {};
    SvcJob j0 = {};
    HIP_CHECK(hipMemcpy(c, ctl.get<uint32_t>(), sizeof c, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(&j0, jobs.get<SvcJob>(), sizeof j0, hipMemcpyDeviceToHost));
    if (dbg.p)
    {
      std::vector<uint32_t> d(12 * 8192 + 48);
      HIP_CHECK(hipMemcpy(d.data(), dbg.p, d.size() * 4, hipMemcpyDeviceToHost));
      {
        const uint32_t *ph = d.data() + 12 * 8192 + 32;
        const double lv = ph[7] ? (double) ph[7] : 1.0;
        fprintf(stderr, "[svc]   wide partition nodes: %u levels; per level: count pass %.2f us, place pass %.2f us, swaps + drain %.2f us, cut, release, pushes and the next pivot %.2f us\n", ph[7], ph[0] * 1e-2 / lv,
                ph[1] * 1e-2 / lv, ph[2] * 1e-2 / lv, ph[3] * 1e-2 / lv);
      }
      for (int kind = 0; kind < 2; ++kind)
      {
        fprintf(stderr, "[svc]   %s tasks by size (2^k ..):", kind ? "narrow heap" : "finisher");
        for (int k = 4; k < 13; ++k) fprintf(stderr, " %d:%u", k, d[12 * 8192 + 16 * kind + k]);
        fprintf(stderr, "\n");
      }
      for (int kind = 0; kind < 2; ++kind)
      {
        // the workgroups' own accounts: time waiting and inside tasks by type (sums over the workgroups, ms), the longest task
        const uint32_t nw = kind == 0 ? stats[2] : stats[3];
        double sum[8] = {};
        uint32_t longest = 0;
        for (uint32_t w = 0; w < nw && w < 4096; ++w)
        {
          const uint32_t *a = d.data() + 4 * 8192 + 8 * (w + 4096 * kind);
          for (int k = 0; k < 7; ++k) sum[k] += a[k];
          longest = std::max(longest, a[7]);
        }
        fprintf(stderr, "[svc]   %s workgroups (%u): waiting %.2f ms; partition nodes %.0f in %.2f ms, finisher %.0f in %.2f ms, heaps %.0f in %.2f ms (sums over the workgroups); longest task %.3f ms\n", kind ? "narrow" : "wide", nw,
                sum[0] * 1e-5, sum[4], sum[1] * 1e-5, sum[5], sum[2] * 1e-5, sum[6], sum[3] * 1e-5, longest * 1e-5);
        std::map<uint32_t, int> hist;
        const uint32_t n = kind == 0 ? stats[2] : stats[3];
        for (uint32_t w = 0; w < n && w < 4096; ++w) hist[d[4 * (w + 4096 * kind)]]++;
        fprintf(stderr, "[svc]   %s workgroups by last state (1 polling, 5 took a task, 9 left):", kind ? "narrow" : "wide");
        for (auto &kv : hist) fprintf(stderr, " %u:%d", kv.first, kv.second);
        fprintf(stderr, "; first ones (state, polls, time of the last poll in 10 ns, how it left 6 error 7 quit 8 timeout):");
        for (uint32_t w = 0; w < 6 && w < n; ++w) fprintf(stderr, " [%u %u %u %u]", d[4 * (w + 4096 * kind)], d[4 * (w + 4096 * kind) + 1], d[4 * (w + 4096 * kind) + 2], d[4 * (w + 4096 * kind) + 3]);
        fprintf(stderr, "\n");
      }
    }
    uint32_t sq[4] = {};
    HIP_CHECK(hipMemcpy(sq, seq[1].get<uint32_t>(), sizeof sq, hipMemcpyDeviceToHost));
    fprintf(stderr, "[svc] tasks %u wide / %u narrow; workgroups started %u / %u, left on their own %u / %u, odd quit words read %u; error %u; wide queue head %u tail %u, narrow queue head %u tail %u (seq %u %u %u %u); job 0: remaining %u done %u heaps %u (longest %u)\n",
            stats[0], stats[1], stats[2], stats[3], stats[4], stats[5], stats[6], h[0], c[0], c[32], c[64], c[96], sq[0], sq[1], sq[2], sq[3], j0.remaining, j0.done, j0.n_heap, j0.max_heap);
  }
  if (h[0]) throw bk_error(h[0] & 2u ? BK_ERR_LIMIT : BK_ERR_HIP, "sort service: task error " + std::to_string(h[0]) + " (2 = ring overflow, 4 = a job timed out, 8 = a cut outside its segment, 16 = a task in the wrong queue, 32 = a node beyond the position lists, 64 = the finisher); first failing task: code " + std::to_string(h[8]) + ", " + std::to_string(h[9]) + " " + std::to_string(h[10]) + " " + std::to_string(h[11]));
}
SortService::~SortService()
{
  if (running)
  {
    running = false;
    if (quit_host && st_copy)
    {
      __atomic_store_n(quit_host, quit_word, __ATOMIC_SEQ_CST);
      (void) hipMemcpyAsync(ctl.get<uint32_t>() + 192, quit_host, 4, hipMemcpyHostToDevice, st_copy);
    }
    if (quit_stream) hipLaunchKernelGGL(k_svc_quit, dim3(1), dim3(1), 0, quit_stream, ctl.get<uint32_t>() + 192, quit_word);
    for (int k = 0; k < 2; ++k)
      if (st[k]) (void) hipStreamSynchronize(st[k]);
    DeferredFrees::end();
  }
  for (int k = 0; k < 2; ++k)
    if (st[k]) (void) hipStreamDestroy(st[k]);
  if (st_copy) (void) hipStreamDestroy(st_copy);
  if (quit_host) (void) hipHostFree(quit_host);
}

static void window_sorts(uint32_t *key, uint32_t *idx, const uint32_t *gof, uint32_t n, hipStream_t st);

// the sort as ONE job of the resident service: submit, wait (both on the caller's stream), then the insertion sort's windows
static void std_sort_groups_svc(uint32_t *key, uint32_t *idx, const uint32_t *gof, const uint64_t *goff, uint32_t ng, uint32_t n, SortEmuBufs &b, hipStream_t st)
{
  SortService &S = *b.svc;
  if (b.svc_slot == 0xFFFFFFFFu)
  {
    b.svc_slot = S.new_slot();
    if (b.svc_slot >= SVC_MAX_JOBS) throw bk_error(BK_ERR_LIMIT, "sort service: more than 64 callers");
  }
  SvcJob d = {};
  d.key = key;
  d.idx = idx;
  d.posL = b.posL.as<uint32_t>((uint64_t) n + 2);
  d.posR = b.posR.as<uint32_t>((uint64_t) n + 2);
  d.hscratch = b.heap_scratch.as<hent>((uint64_t) n + HEAP_PAD);
  d.scratch32 = b.scratch32.as<uint32_t>((uint64_t) n + HEAP_PAD);
  d.scratch32b = b.scratch32b.as<uint32_t>((uint64_t) n + HEAP_PAD);
  d.rka = b.rk_a.as<unsigned long long>((uint64_t) n + HEAP_PAD);
  d.rkb = b.rk_b.as<unsigned long long>((uint64_t) n + HEAP_PAD);
  d.epoch = (b.svc_epoch++ % 0xFFFFFu) + 1u;  // (never 0)
  const SvcParams P = svc_params(S);
  hipLaunchKernelGGL(k_svc_submit, dim3(1), dim3(256), 0, st, P, d, b.svc_slot, goff, ng);
  HIP_CHECK(hipGetLastError());
  // The caller's thread waits, not its stream: a kernel that spins on the stream until the job is done keeps the stream's hardware
  // queue busy, and the command processor serves the other queues' dependent launches the slower the more queues hold a running
  // kernel (twelve lanes' wait kernels: 21 us per dependent launch against 3, tools/ubench/beside.hip - the lanes' ~160 other
  // kernels then cost more than their sorts).
  {
    volatile const uint32_t *done = S.quit_host + SVC_H_DONE + b.svc_slot, *err = S.quit_host + SVC_H_ERROR;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; *done != d.epoch; ++spins)
    {
      if (*err != 0u)
        throw bk_error(BK_ERR_HIP, "sort service: a task failed (error " + std::to_string(*err) + "; first failing task: code " + std::to_string(err[1]) + ", " + std::to_string(err[2]) + " " + std::to_string(err[3]) + " " + std::to_string(err[4]) + ")");
      if ((spins & 1023u) == 1023u)
      {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0) throw bk_error(BK_ERR_HIP, "sort service: a job did not finish");
        std::this_thread::yield();
      }
      else
        __builtin_ia32_pause();
    }
  }
  if (bk_debug("svc"))
  {
    SvcJob j;
    HIP_CHECK(hipMemcpy(&j, S.jobs.get<SvcJob>() + b.svc_slot, sizeof j, hipMemcpyDeviceToHost));
    fprintf(stderr, "[svc] slot %u sort %u: %u elements in %u groups; partitions and finisher done after %.3f ms, job after %.3f ms; %u elements in heaps, the longest %u, the heap that ended last %u elements in %.3f ms\n", b.svc_slot,
            b.svc_epoch, n, ng, (double) (j.t_parts - j.t_submit) * 1e-5, (double) (j.t_done - j.t_submit) * 1e-5, j.n_heap, j.max_heap, (uint32_t) j.last_heap, (double) (j.last_heap >> 32) * 1e-5);
  }
  window_sorts(key, idx, gof, n, st);
}

void std_sort_groups(uint32_t *key, uint32_t *idx, const uint32_t *gof, const uint64_t *goff, uint32_t ng, uint64_t n64, SortEmuBufs &b, hipStream_t st)
{
  if (n64 == 0 || ng == 0) return;
  if (n64 > 0x7FFFFFF0ull) throw bk_error(BK_ERR_LIMIT, "std_sort_groups: more than 2^31 pairs");
  {
    // the two debugging switches live in __device__ variables, i.e. once per DEVICE: every device this process sorts on gets them
    // (one sample over several GPUs runs one host thread per device; the lanes of one device share the entry under the lock)
    static std::mutex flag_m;
    static bool flag_set[64] = {};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> l(flag_m);
    if (dev >= 0 && dev < 64 && !flag_set[dev])
    {
      const int nq = getenv("BK_HEAP_NO_Q") != nullptr;
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_heap_no_q), &nq, sizeof nq));
      const int nhy = nq;
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_heap_no_hybrid), &nhy, sizeof nhy));
      flag_set[dev] = true;
    }
  }
  const uint32_t n = (uint32_t) n64;
  if (b.svc && b.svc->running)
  {
    std_sort_groups_svc(key, idx, gof, goff, ng, n, b, st);
    return;
  }
  static const bool chk = bk_debug("sortcheck");
  DevBuf &chk_key0 = b.chk_key0;  // (BK_DEBUG=sortcheck: per buffer set, i.e. per lane and device)
  uint32_t *key0 = nullptr;
  if (chk)
  {
    // key0[idx] = the key that belongs to payload idx (as the sort receives them)
    key0 = chk_key0.as<uint32_t>(n);
    std::vector<uint32_t> hk(n), hx(n), k0(n);
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipMemcpy(hk.data(), key, (size_t) n * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(hx.data(), idx, (size_t) n * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i)
      if (hx[i] < n) k0[hx[i]] = hk[i];
    HIP_CHECK(hipMemcpy(key0, k0.data(), (size_t) n * 4, hipMemcpyHostToDevice));
    sort_check("at entry", key, idx, key0, n, goff, ng, st, b);
  }
  unsigned long long *cnt = b.cnt.as<unsigned long long>((uint64_t) (n / 8 + ng) + 32);
  if (const char *dump = getenv("BK_DEBUG_SORT_DUMP"))
  {
    // keys and group offsets as they arrive at this sort (debugging aid): <dump>.<call>.keys.u32 / .goff.u64, first 5 calls
    static int call = 0;
    if (call < 5)
    {
      std::vector<uint64_t> go((size_t) ng + 1);
      std::vector<uint32_t> kk(n);
      HIP_CHECK(hipStreamSynchronize(st));
      HIP_CHECK(hipMemcpy(go.data(), goff, ((size_t) ng + 1) * 8, hipMemcpyDeviceToHost));
      HIP_CHECK(hipMemcpy(kk.data(), key, (size_t) n * 4, hipMemcpyDeviceToHost));
      char name[512];
      snprintf(name, sizeof name, "%s.%d.keys.u32", dump, call);
      if (FILE *f = fopen(name, "wb"))
      {
        fwrite(kk.data(), 4, kk.size(), f);
        fclose(f);
      }
      snprintf(name, sizeof name, "%s.%d.goff.u64", dump, call);
      if (FILE *f = fopen(name, "wb"))
      {
        fwrite(go.data(), 8, go.size(), f);
        fclose(f);
      }
    }
    ++call;
  }
  // level 0 segments = groups larger than 16
  const uint32_t fin_cap = (uint32_t) ((uint64_t) n / 16 + ng + 16);  // every entry holds more than 16 elements
  FinSeg *fin_list = b.fin_list.as<FinSeg>(fin_cap);
  size_t max_segs = (size_t) n / 8 + ng + 16;
  Seg *segs = b.segs_a.as<Seg>(max_segs), *segs2 = b.segs_b.as<Seg>(max_segs);
  uint2 *heap_list = b.heap_list.as<uint2>(max_segs);
  uint32_t *state = b.err.as<uint32_t>(ST_WORDS);
  uint32_t *err = state + ST_ERR, *fin = state + ST_FIN, *lvl = state + ST_LVL;
  hipLaunchKernelGGL(k_se_reset, dim3(1), dim3(64), 0, st, state, fin_cap, (uint32_t) (max_segs - 1));
  hipLaunchKernelGGL(k_se_init, dim3(cdiv(ng, 256)), dim3(256), 0, st, goff, ng, cnt, fin_list, fin, lvl + 2);
  prims::exclusive_scan<unsigned long long>(cnt, cnt, ng, b.scan_tmp, st);
  unsigned long long tot = 0;
  uint32_t max_live = 0xFFFFFFFFu;  // largest live segment (known at level 0 and after a batch of levels)
  HIP_CHECK(hipMemcpyAsync(&tot, cnt + ng, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&max_live, lvl + 2, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  uint32_t ns = (uint32_t) tot, na = (uint32_t) (tot >> 32);  // live segments, elements in them
  if (ns)
  {
    hipLaunchKernelGGL(k_se_init_write, dim3(cdiv(ng, 256)), dim3(256), 0, st, goff, ng, cnt, segs, b.lv_tileseg.as<uint32_t>((uint64_t) n / LV_TILE + 2), lvl);
    unsigned long long *tile_cnt = b.lv_tile.as<unsigned long long>((uint64_t) n / LV_TILE + 2);
    unsigned long long *segbase = b.lv_segbase.as<unsigned long long>(max_segs + 1);
    uint32_t *tile_seg = b.lv_tileseg.as<uint32_t>((uint64_t) n / LV_TILE + 2);
    uint32_t *posL = b.posL.as<uint32_t>((uint64_t) n + 2), *posR = b.posR.as<uint32_t>((uint64_t) n + 2);
    int level = 0;
    static const bool dbg_levels = bk_debug("sort");
    double t_loop0 = 0;
    if (dbg_levels)
    {
      HIP_CHECK(hipStreamSynchronize(st));
      t_loop0 = now_ms();
    }
    bool pivoted = false;  // the live segments already carry their pivots (the fused child kernel picked them)
    while (ns)
    {
      if (ns > max_segs) throw bk_error(BK_ERR_LIMIT, "std_sort_groups: segment list overflow");
      // how many levels may be queued before the host has to look: the segment count at most doubles per level and
      // the fused child kernel takes CHILD_FUSED segments
      // every live segment to a workgroup of its own for the rest of its introsort tree (k_se_tail) as soon as the largest one is
      // small enough for a single CU to stream (2^19 elements; BK_SORT_NO_TAIL=1: the level loop to the
      // end) and the wide top of the trees is done (from level 4 on: the first levels hold thousands of segments and
      // are one pass over everything for the device-wide kernels, while a single workgroup per ROOT would walk ~10^2 nodes; measured
      // with four lanes: level 2 41.3 ms, level 4 39.6, level 6 42.2 for the stage, 42.0-42.7 without the tail kernel)
      static const bool no_tail = getenv("BK_SORT_NO_TAIL") != nullptr;
      constexpr uint32_t tail_max = 1u << 19;
      constexpr int tail_level = 4;
      if (!no_tail && level >= tail_level && max_live <= tail_max)
      {
        if (!pivoted) hipLaunchKernelGGL(k_se_pivot, dim3(cdiv(ns, 256)), dim3(256), 0, st, segs, ns, key, idx, err, heap_list);
        // spine rounds: a segment that is handed on is at most half of its parent and larger than FIN_MAX
        int rounds = 1;
        for (uint64_t sz = max_live; sz / 2 > FIN_MAX; sz /= 2) ++rounds;
        uint32_t *rc = b.lv_bar.as<uint32_t>((uint64_t) rounds + 2);
        HIP_CHECK(hipMemsetAsync(rc, 0, ((size_t) rounds + 2) * 4, st));
        HIP_CHECK(hipMemcpyAsync(rc, &ns, 4, hipMemcpyHostToDevice, st));
        const uint32_t cap = (uint32_t) std::min<uint64_t>(max_segs, (uint64_t) na / FIN_MAX + 1);
        for (int r = 0; r < rounds; ++r)
        {
          const uint64_t bound = r < 1 ? std::min<uint64_t>(cap, ns) : cap;
          hipLaunchKernelGGL(k_se_tail_round, dim3((unsigned) std::max<uint64_t>(1, bound)), dim3(TL_THREADS), 0, st, (const Seg *) segs, (const uint32_t *) (rc + r), segs2, rc + r + 1, cap, key,
                             idx, posL, posR, fin_list, fin, err, heap_list, 1);
          std::swap(segs, segs2);
        }
        if (dbg_levels)
        {
          HIP_CHECK(hipStreamSynchronize(st));
          fprintf(stderr, "[sortemu]   levels %d.. by one workgroup per live segment in %d rounds: %u segments, %u live elements, largest %u, %.3f ms so far\n", level, rounds, ns, na, max_live, now_ms() - t_loop0);
        }
        break;
      }
      int batch = 0;
      constexpr int max_batch = 6;
      {
        // a live segment holds more than FIN_MAX elements and the live elements never grow: na / FIN_MAX bounds the
        // segment count of every later level
        if (ns <= CHILD_FUSED && na / FIN_MAX <= CHILD_FUSED)
          batch = max_batch;
        else
          for (uint32_t cap = ns; cap <= CHILD_FUSED && batch < max_batch; cap *= 2) ++batch;
      }
      if (batch == 0)
      {
        if (!pivoted) hipLaunchKernelGGL(k_se_pivot, dim3(cdiv(ns, 256)), dim3(256), 0, st, segs, ns, key, idx, err, heap_list);
        pivoted = false;
        {
          const unsigned nbt = cdiv(na, LV_TILE);
          hipLaunchKernelGGL(k_lv_count, dim3(nbt), dim3(256), 0, st, segs, ns, key, na, tile_cnt, (const uint32_t *) nullptr, (const uint32_t *) tile_seg);
          hipLaunchKernelGGL(k_lv_sums, dim3(1), dim3(LV_SUM_THREADS), 0, st, tile_cnt, ns, na, segbase, (const uint32_t *) nullptr);
          hipLaunchKernelGGL(k_lv_lists, dim3(nbt), dim3(256), 0, st, segs, ns, key, na, tile_cnt, posL, posR, segbase, (const uint32_t *) nullptr, (const uint32_t *) tile_seg);
          hipLaunchKernelGGL(k_lv_swap, dim3(nbt), dim3(256), 0, st, segs, ns, key, idx, na, segbase, posL, posR, (const uint32_t *) nullptr, (const uint32_t *) tile_seg);
        }
        hipLaunchKernelGGL(k_se_child_count, dim3(cdiv(ns, 256)), dim3(256), 0, st, segs, ns, cnt);
        prims::exclusive_scan<unsigned long long>(cnt, cnt, ns, b.scan_tmp, st);
        // the children are written while the host waits for their count (at most 2 per segment: 2 * ns <= capacity)
        if (2ull * ns > max_segs) throw bk_error(BK_ERR_LIMIT, "std_sort_groups: segment list overflow");
        hipLaunchKernelGGL(k_se_child_write, dim3(cdiv(ns, 256)), dim3(256), 0, st, segs, ns, cnt, segs2, fin_list, fin, tile_seg);
        HIP_CHECK(hipMemcpyAsync(&tot, cnt + ns, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        std::swap(segs, segs2);
        ++level;
      }
      else
      {
        // `batch` levels without a host round trip: counts live in lvl, grids are sized for the counts at the start
        if (std::min<uint64_t>((uint64_t) ns << batch, (uint64_t) na / FIN_MAX + 1) > max_segs) throw bk_error(BK_ERR_LIMIT, "std_sort_groups: segment list overflow");
        const uint32_t cur[2] = {ns, na};
        if (level > 0) HIP_CHECK(hipMemcpyAsync(lvl, cur, 8, hipMemcpyHostToDevice, st));  // (level 0: k_se_init_write left them; later batches: the children kernel did, but an unbatched level may lie between)
        if (!pivoted) hipLaunchKernelGGL(k_se_pivot, dim3(cdiv(ns, 256)), dim3(256), 0, st, segs, ns, key, idx, err, heap_list);
        const unsigned nbt = cdiv(na, LV_TILE);
        uint32_t ns_bound = ns;
        for (int l = 0; l < batch; ++l)
        {
          {
            hipLaunchKernelGGL(k_lv_count, dim3(nbt), dim3(256), 0, st, segs, ns_bound, key, na, tile_cnt, (const uint32_t *) lvl, (const uint32_t *) tile_seg);
            hipLaunchKernelGGL(k_lv_sums, dim3(1), dim3(LV_SUM_THREADS), 0, st, tile_cnt, ns_bound, na, segbase, (const uint32_t *) lvl);
            hipLaunchKernelGGL(k_lv_lists, dim3(nbt), dim3(256), 0, st, segs, ns_bound, key, na, tile_cnt, posL, posR, segbase, (const uint32_t *) lvl, (const uint32_t *) tile_seg);
            hipLaunchKernelGGL(k_lv_swap, dim3(nbt), dim3(256), 0, st, segs, ns_bound, key, idx, na, segbase, posL, posR, (const uint32_t *) lvl, (const uint32_t *) tile_seg);
          }
          hipLaunchKernelGGL(k_se_children_small, dim3(1), dim3(CHILD_THREADS), 0, st, segs, lvl, segs2, fin_list, fin, key, idx, err, heap_list, tile_seg);
          std::swap(segs, segs2);
          ns_bound = ns_bound * 2 < CHILD_FUSED ? ns_bound * 2 : CHILD_FUSED;
          ++level;
        }
        pivoted = true;
        uint32_t now[3] = {0, 0, 0};
        HIP_CHECK(hipMemcpyAsync(now, lvl, 12, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        tot = (unsigned long long) now[0] | ((unsigned long long) now[1] << 32);
        max_live = now[2];
      }
      const uint32_t ns2 = (uint32_t) tot;
      if (dbg_levels && (level % 4 == 0 || ns2 == 0))
      {
        HIP_CHECK(hipStreamSynchronize(st));
        fprintf(stderr, "[sortemu]   level %d: %u segments -> %u (%u live elements, largest %u), %.3f ms so far\n", level, ns, ns2, (uint32_t) (tot >> 32), max_live, now_ms() - t_loop0);
      }
      ns = ns2;
      na = (uint32_t) (tot >> 32);
      if (level > 200) throw bk_error(BK_ERR_LIMIT, "std_sort_groups: runaway recursion");
    }
  }
  if (chk) sort_check("after the partition levels", key, idx, key0, n, goff, ng, st, b);
  // What is left: (1) segments of at most FIN_MAX elements - the rest of their introsort loop runs in LDS, one workgroup
  // each (the finisher; it may add small segments to the heap list) - and (2) the segments that exhausted introsort's
  // depth limit in the level loop, which are heapsorted (they are final: no children).  The longest heap segment is the
  // critical path of the whole sort, so the big heaps are started first, on side streams, and the finisher runs beside them.
  uint32_t hstate[ST_WORDS] = {};
  HIP_CHECK(hipMemcpyAsync(hstate, state, sizeof hstate, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  uint32_t nfin2[2] = {hstate[ST_FIN], hstate[ST_FIN + 1]};
  uint32_t e[8];
  for (int k = 0; k < 8; ++k) e[k] = hstate[ST_ERR + k];
  if (e[0] & 8u) throw bk_error(BK_ERR_HIP, "std_sort_groups: k_se_tail_round lost a segment (list overflow or a cut outside its segment)");
  const uint32_t nfin = nfin2[0] + nfin2[1];
  const uint32_t nh1 = e[3], max1 = e[1], n_big = e[4];  // heap segments of the level loop; those above HEAP_BIG_MIN
  const HeapSeg *hl = reinterpret_cast<const HeapSeg *>(heap_list);
  static const bool dbg = bk_debug("sort");
  hent *hscratch = nullptr;
  uint32_t *scratch32 = nullptr, *scratch32b = nullptr;
  unsigned long long *rka = nullptr, *rkb = nullptr;  // scratch of the ranking inside the big heaps' own workgroups (wg_ranked_entries)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (dbg)
  {
    HIP_CHECK(hipEventCreate(&ev0));
    HIP_CHECK(hipEventCreate(&ev1));
    HIP_CHECK(hipEventRecord(ev0, st));
  }
  if (nh1 || nfin) hscratch = b.heap_scratch.as<hent>((uint64_t) n + HEAP_PAD);
  if (b.heavy && nh1 && (max1 > HEAP_RANKED_MIN || b.heavy_all))
  {
    // the caller balances its lanes of groups on the longest heap segment per group
    std::vector<HeapSeg> hh(nh1);
    std::vector<uint64_t> go((size_t) ng + 1);
    HIP_CHECK(hipMemcpyAsync(hh.data(), hl, (size_t) nh1 * sizeof(HeapSeg), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(go.data(), goff, ((size_t) ng + 1) * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (b.heavy->size() < ng) b.heavy->resize(ng, 0u);
    for (const HeapSeg &h : hh)
    {
      const uint32_t m = h.last - h.first;
      if (m <= HEAP_RANKED_MIN && !b.heavy_all) continue;
      const uint32_t g = (uint32_t) (std::upper_bound(go.begin(), go.end(), (uint64_t) h.first) - go.begin()) - 1;
      if (g < ng && (*b.heavy)[g] < m) (*b.heavy)[g] = m;
    }
  }
  bool forked = false;
  if (nh1 && max1 > HEAP_SMALL)
  {
    // heaps above HEAP_BIG_MIN elements: ranked 4-byte entries, one workgroup per CU, on a side stream (the longest of them is the
    // critical path of the sort); the mid-size heaps of the level loop (~1.5 ms) in front of the finisher on the caller's own stream
    if (max1 > HEAP_BIG_MIN)
    {
      rka = b.rk_a.as<unsigned long long>((uint64_t) n + HEAP_PAD);
      rkb = b.rk_b.as<unsigned long long>((uint64_t) n + HEAP_PAD);
      scratch32 = b.scratch32.as<uint32_t>((uint64_t) n + HEAP_PAD);
      if (max1 > HEAP_LARGE32) scratch32b = b.scratch32b.as<uint32_t>((uint64_t) n + HEAP_PAD);  // overflow slots of the heaps beyond the LDS
      if (!b.fork) HIP_CHECK(hipEventCreateWithFlags(&b.fork, hipEventDisableTiming));
      if (!b.aux[0])
      {
        HIP_CHECK(hipStreamCreateWithFlags(&b.aux[0], hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&b.join[0], hipEventDisableTiming));
      }
      HIP_CHECK(hipEventRecord(b.fork, st));
      HIP_CHECK(hipStreamWaitEvent(b.aux[0], b.fork, 0));
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_se_heapsort<2>), hipFuncAttributeMaxDynamicSharedMemorySize, HEAP_BIG_LDS));
      hipLaunchKernelGGL(k_se_heapsort<2>, dim3(n_big), dim3(HEAP_BIG_THREADS), HEAP_BIG_LDS, b.aux[0], hl + (max_segs - n_big), n_big, key, idx, hscratch, HEAP_BIG_MIN, 0xFFFFFFFFu, scratch32, scratch32b, rka,
                         rkb);  // (the list of the long ones only)
      HIP_CHECK(hipEventRecord(b.join[0], b.aux[0]));
      forked = true;
    }
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_se_heapsort<1>), hipFuncAttributeMaxDynamicSharedMemorySize, ((size_t) HEAP_BIG_MIN + HEAP_PAD) * 8));
    hipLaunchKernelGGL(k_se_heapsort<1>, dim3(nh1), dim3(64), ((size_t) HEAP_BIG_MIN + HEAP_PAD) * 8, st, hl, nh1, key, idx, hscratch, HEAP_SMALL, HEAP_BIG_MIN, scratch32, scratch32b, rka, rkb);
  }
  if (nfin2[1]) hipLaunchKernelGGL((k_se_finish<FIN_MAX, 256>), dim3(nfin2[1]), dim3(256), 0, st, fin_list + (fin_cap - 1), nfin2[1], -1, key, idx, err, heap_list);
  if (nfin2[0]) hipLaunchKernelGGL((k_se_finish<FIN_SMALL, 64>), dim3(nfin2[0]), dim3(64), 0, st, fin_list, nfin2[0], 1, key, idx, err, heap_list);
  if (nh1 || nfin)
  {
    // small heaps (level loop and finisher) and the finisher's own segments above HEAP_SMALL (at most FIN_MAX elements)
    if (nfin)
    {
      HIP_CHECK(hipMemcpyAsync(e, err, 16, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
    }
    const uint32_t nh2 = e[3];
    if (nh2)
    {
      // the finisher's segments (at most FIN_MAX elements) in ONE launch whatever their size, then the few short segments the level
      // loop itself left
      if (nh2 > nh1)
      {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_se_heapsort<1>), hipFuncAttributeMaxDynamicSharedMemorySize, ((size_t) HEAP_BIG_MIN + HEAP_PAD) * 8));
        hipLaunchKernelGGL(k_se_heapsort<1>, dim3(nh2 - nh1), dim3(64), ((size_t) FIN_MAX + HEAP_PAD) * 8, st, hl + nh1, nh2 - nh1, key, idx, hscratch, 0u, HEAP_BIG_MIN, scratch32, scratch32b, rka, rkb);
      }
      if (nh1) hipLaunchKernelGGL(k_se_heapsort<0>, dim3(nh1), dim3(64), 0, st, hl, nh1, key, idx, hscratch, 0u, HEAP_SMALL, scratch32, scratch32b, rka, rkb);
    }
    if (forked) HIP_CHECK(hipStreamWaitEvent(st, b.join[0], 0));
    if (dbg)
    {
      float ms = 0;
      HIP_CHECK(hipEventRecord(ev1, st));
      HIP_CHECK(hipEventSynchronize(ev1));
      HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
      unsigned long long it[2] = {0, 0}, zero[2] = {0, 0};
      HIP_CHECK(hipMemcpyFromSymbol(it, HIP_SYMBOL(g_heap_iters), 16));
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_heap_iters), zero, 16));
      fprintf(stderr, "[sortemu] finisher (%u + %u segments) beside the heapsort kernels: %.3f ms (%.3f us per element of the largest heap segment); %llu iterations for %llu pops\n", nfin2[1], nfin2[0], ms,
              e[1] ? ms * 1e3 / e[1] : 0.0, it[0], it[1]);
      fprintf(stderr, "[sortemu] n=%u groups=%u heap segments=%u (%u from the level loop) elements=%u max=%u\n", n, ng, e[3], nh1, e[2], e[1]);
      unsigned long long ph[8] = {0}, zz[8] = {0};
      HIP_CHECK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_heap_phase), 64));
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_heap_phase), zz, 64));
      if (ph[0]) fprintf(stderr, "[sortemu]   a ranked heap of %llu: load + rank %.2f ms, make_heap %.2f ms, pops in global memory %.2f ms, pops in LDS %.2f ms\n", ph[0], ph[1] * 1e-5, ph[2] * 1e-5, ph[3] * 1e-5, ph[4] * 1e-5);
    }
  }
  if (ev0) (void) hipEventDestroy(ev0);
  if (ev1) (void) hipEventDestroy(ev1);
  if (chk) sort_check("after the heaps and the finisher", key, idx, key0, n, goff, ng, st, b);
  // __final_insertion_sort == stable sort by key of what the introsort loop left: two tilings of 32-element windows
  window_sorts(key, idx, gof, n, st);
}
static void window_sorts(uint32_t *key, uint32_t *idx, const uint32_t *gof, uint32_t n, hipStream_t st)
{
  hipLaunchKernelGGL(k_se_window_sort, dim3(cdiv(n, 256)), dim3(256), 0, st, key, idx, gof, n, 0u);
  if (n > 16) hipLaunchKernelGGL(k_se_window_sort, dim3(cdiv(n - 16, 256)), dim3(256), 0, st, key, idx, gof, n, 16u);
}
