// Device-wide primitives for gfx950 (wave64): exclusive scan and a stable LSD radix sort.
// Hand-written (no rocPRIM): the sort ranks keys with wavefront ballots and LDS-staged digit bins.
#pragma once
#include "bk_common.h"

namespace prims
{
constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / BK_WAVE;

// ---- wave / block scan ----------------------------------------------------------------------------
template <class T> __device__ __forceinline__ T wave_inclusive_scan(T v)
{
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    T o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// exclusive scan over the 256 threads of a block; returns exclusive prefix, `total` = block sum.
// lds must hold WAVES elements of T.  Contains two __syncthreads().
template <class T> __device__ __forceinline__ T block_exclusive_scan(T v, T *lds, T &total)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  T inc = wave_inclusive_scan(v);
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < WAVES; ++i)
  {
    T s = lds[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return base + inc - v;
}

// ---- device-wide exclusive scan: out has n+1 entries, out[n] = total ------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;

// (n_dev != nullptr: the element count is read from device memory, n is only the bound the grid was sized for)
template <class T> __global__ __launch_bounds__(BLOCK) void k_scan_reduce(const T *__restrict__ in, T *__restrict__ sums, uint64_t n, const uint32_t *__restrict__ n_dev = nullptr)
{
  __shared__ T lds[WAVES];
  if (n_dev) n = *n_dev;
  uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE;
  T acc = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
  {
    uint64_t i = base + (uint64_t) k * BLOCK + threadIdx.x;
    if (i < n) acc += in[i];
  }
  T tot;
  (void) block_exclusive_scan(acc, lds, tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single block: exclusive scan of sums[0..nb) in place, total -> *total_out
template <class T> __global__ __launch_bounds__(BLOCK) void k_scan_sums(T *sums, uint32_t nb, T *total_out, const uint32_t *__restrict__ n_dev = nullptr)
{
  __shared__ T lds[WAVES];
  if (n_dev) total_out += *n_dev;  // total_out = the output array then
  T carry = 0;
  for (uint32_t base = 0; base < nb; base += BLOCK)
  {
    uint32_t i = base + threadIdx.x;
    T v = i < nb ? sums[i] : (T) 0;
    T tot;
    T ex = block_exclusive_scan(v, lds, tot);
    if (i < nb) sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

template <class T> __global__ __launch_bounds__(BLOCK) void k_scan_apply(const T *__restrict__ in, T *__restrict__ out, const T *__restrict__ sums, uint64_t n, const uint32_t *__restrict__ n_dev = nullptr)
{
  __shared__ T lds[WAVES];
  if (n_dev) n = *n_dev;
  uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE;
  T carry = sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
  {
    uint64_t i = base + (uint64_t) k * BLOCK + threadIdx.x;
    T v = i < n ? in[i] : (T) 0;
    T tot;
    T ex = block_exclusive_scan(v, lds, tot);
    if (i < n) out[i] = carry + ex;
    carry += tot;
  }
}

// the whole scan by ONE workgroup (tiles one after the other, the carry in a register): a short array is bound by the three
// dependent launches of the general form, not by its elements - and every dispatch of a process shares one command processor
constexpr uint64_t SCAN_ONE_MAX = 16 * SCAN_TILE;
template <class T> __global__ __launch_bounds__(BLOCK) void k_scan_one(const T *in, T *out, uint64_t n, const uint32_t *__restrict__ n_dev = nullptr)
{
  __shared__ T lds[WAVES];
  if (n_dev) n = *n_dev;
  T carry = 0;
  for (uint64_t base = 0; base < n; base += BLOCK)
  {
    const uint64_t i = base + threadIdx.x;
    const T v = i < n ? in[i] : (T) 0;
    T tot;
    const T ex = block_exclusive_scan(v, lds, tot);
    if (i < n) out[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) out[n] = carry;
}

// in/out may alias.  tmp grows as needed.
template <class T> inline void exclusive_scan(const T *in, T *out, uint64_t n, DevBuf &tmp, hipStream_t st)
{
  if (n == 0)
  {
    HIP_CHECK(hipMemsetAsync(out, 0, sizeof(T), st));
    return;
  }
  if (n <= SCAN_ONE_MAX)
  {
    hipLaunchKernelGGL(k_scan_one<T>, dim3(1), dim3(BLOCK), 0, st, in, out, n, (const uint32_t *) nullptr);
    return;
  }
  uint32_t nb = cdiv(n, SCAN_TILE);
  T *sums = tmp.as<T>(nb + 1);
  hipLaunchKernelGGL(k_scan_reduce<T>, dim3(nb), dim3(BLOCK), 0, st, in, sums, n, (const uint32_t *) nullptr);
  hipLaunchKernelGGL(k_scan_sums<T>, dim3(1), dim3(BLOCK), 0, st, sums, nb, out + n, (const uint32_t *) nullptr);
  hipLaunchKernelGGL(k_scan_apply<T>, dim3(nb), dim3(BLOCK), 0, st, in, out, sums, n, (const uint32_t *) nullptr);
}

// the same with the element count in device memory (*n_dev <= n_bound); out[*n_dev] = total
template <class T> inline void exclusive_scan_devn(const T *in, T *out, uint64_t n_bound, const uint32_t *n_dev, DevBuf &tmp, hipStream_t st)
{
  if (n_bound <= SCAN_ONE_MAX)
  {
    hipLaunchKernelGGL(k_scan_one<T>, dim3(1), dim3(BLOCK), 0, st, in, out, n_bound, n_dev);
    return;
  }
  uint32_t nb = cdiv(n_bound ? n_bound : 1, SCAN_TILE);
  T *sums = tmp.as<T>(nb + 1);
  hipLaunchKernelGGL(k_scan_reduce<T>, dim3(nb), dim3(BLOCK), 0, st, in, sums, n_bound, n_dev);
  hipLaunchKernelGGL(k_scan_sums<T>, dim3(1), dim3(BLOCK), 0, st, sums, nb, out, n_dev);
  hipLaunchKernelGGL(k_scan_apply<T>, dim3(nb), dim3(BLOCK), 0, st, in, out, sums, n_bound, n_dev);
}

// ---- stable LSD radix sort, u64 keys + u32 values, 8-bit digits -----------------------------------
constexpr int RS_ROWS = 8;                           // rows of 64 keys per wave
constexpr int RS_TILE = BLOCK * RS_ROWS;             // 2048 keys per block

static __global__ __launch_bounds__(BLOCK) void k_radix_hist(const uint64_t *__restrict__ keys, uint32_t *__restrict__ ghist, uint64_t n, int shift, uint32_t nb)
{
  __shared__ uint32_t hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  uint64_t base = (uint64_t) blockIdx.x * RS_TILE;
#pragma unroll
  for (int k = 0; k < RS_ROWS; ++k)
  {
    uint64_t i = base + (uint64_t) k * BLOCK + threadIdx.x;
    if (i < n) atomicAdd(&hist[(keys[i] >> shift) & 255], 1u);
  }
  __syncthreads();
  ghist[(uint64_t) threadIdx.x * nb + blockIdx.x] = hist[threadIdx.x];
}

static __global__ __launch_bounds__(BLOCK) void k_radix_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint64_t *__restrict__ kout,
                                                         uint32_t *__restrict__ vout, const uint32_t *__restrict__ goff, uint64_t n, int shift, uint32_t nb)
{
  __shared__ uint32_t whist[WAVES][256];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < WAVES; ++i) whist[i][threadIdx.x] = 0;
  __syncthreads();
  const uint64_t tile = (uint64_t) blockIdx.x * RS_TILE + (uint64_t) w * RS_ROWS * 64;
  uint64_t key[RS_ROWS];
  uint32_t val[RS_ROWS], rank[RS_ROWS];
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < RS_ROWS; ++r)
  {
    uint64_t i = tile + (uint64_t) r * 64 + lane;
    bool valid = i < n;
    key[r] = valid ? kin[i] : 0ull;
    val[r] = valid ? vin[i] : 0u;
    uint32_t d = (uint32_t) (key[r] >> shift) & 255u;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b)
    {
      bool bit = (d >> b) & 1u;
      uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    uint32_t rk = __popcll(peers & lt);
    uint32_t cnt = __popcll(peers);
    int leader = valid ? (__ffsll((long long) peers) - 1) : lane;
    uint32_t old = 0;
    if (valid && lane == leader) old = atomicAdd(&whist[w][d], cnt);  // rows are issued in order by this wave
    old = __shfl(old, leader, 64);
    rank[r] = old + rk;
  }
  __syncthreads();
  {
    // per digit: exclusive prefix over waves + global offset of this (digit, block)
    uint32_t run = goff[(uint64_t) threadIdx.x * nb + blockIdx.x];
#pragma unroll
    for (int i = 0; i < WAVES; ++i)
    {
      uint32_t c = whist[i][threadIdx.x];
      whist[i][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ROWS; ++r)
  {
    uint64_t i = tile + (uint64_t) r * 64 + lane;
    if (i < n)
    {
      uint32_t d = (uint32_t) (key[r] >> shift) & 255u;
      uint32_t p = whist[w][d] + rank[r];
      kout[p] = key[r];
      vout[p] = val[r];
    }
  }
}

struct RadixBufs
{
  DevBuf keys_alt, vals_alt, hist, scan_tmp;
};

// Sorts (keys, vals) by bits [begin_bit, end_bit) of the key, stable.  On return the sorted data is in
// (*keys_out, *vals_out), which point either at the inputs or at the alternate buffers.
inline void radix_sort_pairs(uint64_t *keys, uint32_t *vals, uint64_t n, int begin_bit, int end_bit, RadixBufs &rb, hipStream_t st,
                             uint64_t **keys_out, uint32_t **vals_out)
{
  *keys_out = keys;
  *vals_out = vals;
  if (n == 0 || end_bit <= begin_bit) return;
  if (n > 0xFFFFFFF0ull) throw bk_error(BK_ERR_LIMIT, "radix_sort_pairs: more than 2^32 items");
  uint64_t *ka = keys, *kb = rb.keys_alt.as<uint64_t>(n);
  uint32_t *va = vals, *vb = rb.vals_alt.as<uint32_t>(n);
  uint32_t nb = cdiv(n, RS_TILE);
  uint32_t *hist = rb.hist.as<uint32_t>((uint64_t) 256 * nb + 1);
  for (int shift = begin_bit; shift < end_bit; shift += 8)
  {
    hipLaunchKernelGGL(k_radix_hist, dim3(nb), dim3(BLOCK), 0, st, ka, hist, n, shift, nb);
    exclusive_scan<uint32_t>(hist, hist, (uint64_t) 256 * nb, rb.scan_tmp, st);
    hipLaunchKernelGGL(k_radix_scatter, dim3(nb), dim3(BLOCK), 0, st, ka, va, kb, vb, hist, n, shift, nb);
    std::swap(ka, kb);
    std::swap(va, vb);
  }
  *keys_out = ka;
  *vals_out = va;
}

}  // namespace prims
