// BGZF inflate on the GPU (RFC 1951 DEFLATE, one wavefront per BGZF block).
//
// the host decoder is the feed limit (all cores of the box: ~3.6 GB/s of inflated BAM).  BGZF blocks are
// independent <= 64 KiB deflate streams, so a file is thousands to millions of independent jobs.  One wave takes one
// block; all 64 lanes run the Huffman decoder redundantly on the same state (every load is a broadcast, the control
// flow is uniform), lane 0 stores literals, and a match of `len` bytes at distance `dist` is copied by the whole wave
// at once: out[o + k] = out[o - dist + k mod dist] reads only bytes that are already final.  The decoder is a serial
// chain of LDS round trips (~0.3 us per symbol), so throughput comes from many blocks in flight and everything hot sits
// in a small LDS footprint: the last 16 KiB of output as a ring (flushed to global memory in aligned 256-byte chunks;
// the rare match that reaches farther back reads the flushed bytes from global memory), a 512-byte window of the
// compressed input refilled by the whole wave, and the decode tables (a 10-bit direct table for literal/length codes,
// 8-bit for distances, canonical count/symbol arrays for the rare longer codes, as in zlib's puff.c).  Output offsets of
// the blocks are multiples of 256 bytes, so every flush is 64 aligned dword stores.  (Earlier versions: input and
// output in global memory - every refill and every match copy paid a full memory round trip, 29 ms per block; the
// whole 64 KiB block staged in LDS - only two waves per CU.)
#include "bk_common.h"
#include "bgzf_gpu.h"

namespace
{
constexpr int LIT_FAST = 10, DIST_FAST = 8;
constexpr uint32_t IN_WIN = 512;  // bytes of compressed input held in LDS
constexpr uint32_t RING = 16384;  // bytes of recent output held in LDS (power of two; 8 KiB measured ~10 % faster on synthetic data, but real BAMs match farther back)
constexpr uint32_t FLUSH = 256;   // the ring is written out in aligned chunks of this size
constexpr uint32_t NEAR = RING - 2 * FLUSH - 258;  // matches up to this distance read the ring, farther ones read global memory

struct HuffLds
{
  uint16_t lfast[1 << LIT_FAST];   // (symbol << 4) | length, 0 = longer than LIT_FAST bits
  uint16_t dfast[1 << DIST_FAST];
  uint16_t lsym[288], dsym[32];    // symbols ordered by (length, symbol)
  uint16_t lcount[16], dcount[16];
  uint8_t lens[320];               // code lengths while a dynamic header is read
  uint32_t win[IN_WIN / 4 + 4];    // input window [win_base, win_base + IN_WIN) + slack for the 8-byte reads
};

struct BitReader
{
  const uint8_t *in;   // compressed stream (global)
  uint32_t *win;       // LDS window
  uint32_t pos, end;   // byte offsets of the next unread byte / the end of the stream
  uint32_t win_base;   // stream offset of win[0] (multiple of 4 relative to `in`'s own alignment)
  uint64_t buf;
  int cnt;
  // the whole wave loads IN_WIN bytes starting at stream offset `at` (rounded down to the window grid)
  __device__ __forceinline__ void load_window(uint32_t at)
  {
    const uint32_t lane = threadIdx.x & 63;
    win_base = at & ~(IN_WIN - 1);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k = lane; k < IN_WIN / 4 + 4; k += 64)
    {
      const uint32_t off = win_base + 4 * k;
      uint32_t v = 0;
      if (off + 4 <= end)
        v = (uint32_t) in[off] | ((uint32_t) in[off + 1] << 8) | ((uint32_t) in[off + 2] << 16) | ((uint32_t) in[off + 3] << 24);
      else
        for (uint32_t b = 0; b < 4; ++b)
          if (off + b < end) v |= (uint32_t) in[off + b] << (8 * b);
      win[k] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // tops the bit buffer up to >= 56 bits with one unaligned 8-byte read of the window (three aligned words)
  __device__ __forceinline__ void refill()
  {
    if (pos - win_base >= IN_WIN) load_window(pos);
    const uint32_t r = pos - win_base, i = r >> 2, sh = (r & 3) * 8;
    // (readfirstlane: every lane holds the same value; saying so moves the decoder's arithmetic to the scalar unit)
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(win[i]), w1 = __builtin_amdgcn_readfirstlane(win[i + 1]), w2 = __builtin_amdgcn_readfirstlane(win[i + 2]);
    uint64_t v = ((uint64_t) w1 << 32) | w0;
    if (sh) v = (v >> sh) | ((uint64_t) w2 << (64 - sh));
    buf |= v << cnt;                     // bits beyond 64 fall off; they are read again next time
    const uint32_t took = (uint32_t) (63 - cnt) >> 3;  // whole bytes that fitted
    pos += took;
    cnt += (int) took * 8;
  }
  __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t) (buf & ((1ull << n) - 1)); }
  __device__ __forceinline__ void drop(int n)
  {
    buf >>= n;
    cnt -= n;
  }
  __device__ __forceinline__ uint32_t bits(int n)
  {
    if (cnt < n) refill();
    const uint32_t v = peek(n);
    drop(n);
    return v;
  }
};

__device__ __forceinline__ uint32_t bitrev(uint32_t c, int len) { return __brev(c) >> (32 - len); }

// canonical Huffman tables from code lengths (all lanes run it; the LDS writes are identical)
__device__ __forceinline__ bool build_tables(const uint8_t *lens, int n, uint16_t *fast, int fast_bits, uint16_t *sym, uint16_t *count)
{
  for (int i = 0; i < 16; ++i) count[i] = 0;
  for (int i = 0; i < n; ++i) count[lens[i]]++;
  for (int i = 0; i < (1 << fast_bits); ++i) fast[i] = 0;
  if (count[0] == n) return true;  // no codes at all (a distance tree of a literal-only block)
  int left = 1;
  for (int len = 1; len < 16; ++len)
  {
    left <<= 1;
    left -= count[len];
    if (left < 0) return false;  // over-subscribed
  }
  uint16_t offs[16], nextc[16];
  offs[1] = 0;
  for (int len = 1; len < 15; ++len) offs[len + 1] = offs[len] + count[len];
  uint32_t code = 0;
  for (int len = 1; len < 16; ++len)
  {
    nextc[len] = (uint16_t) code;
    code = (code + count[len]) << 1;
  }
  for (int s = 0; s < n; ++s)
  {
    const int len = lens[s];
    if (!len) continue;
    sym[offs[len]++] = (uint16_t) s;
    const uint32_t c = nextc[len]++;
    if (len <= fast_bits)
    {
      const uint32_t r = bitrev(c, len);
      for (uint32_t j = r; j < (1u << fast_bits); j += 1u << len) fast[j] = (uint16_t) ((s << 4) | len);
    }
  }
  return true;
}

// one symbol: direct table, else the canonical walk of puff.c (bits arrive LSB first, codes are MSB first)
__device__ __forceinline__ int decode_sym(BitReader &br, const uint16_t *fast, int fast_bits, const uint16_t *sym, const uint16_t *count)
{
  const uint32_t e = __builtin_amdgcn_readfirstlane((uint32_t) fast[br.peek(fast_bits)]);
  if (e)
  {
    br.drop(e & 15);
    return e >> 4;
  }
  int code = 0, first = 0, index = 0;
  uint64_t b = br.buf;
  for (int len = 1; len < 16; ++len)
  {
    code |= (int) (b & 1);
    b >>= 1;
    const int c = __builtin_amdgcn_readfirstlane((int) count[len]);
    if (code - c < first)
    {
      br.drop(len);
      return __builtin_amdgcn_readfirstlane((int) sym[index + (code - first)]);
    }
    index += c;
    first += c;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

// base value and extra bits of length code 257 + ls / distance code ds (RFC 1951 section 3.2.5), computed: a table in
// constant memory costs a memory round trip per match on this serial path
__device__ __forceinline__ void len_code(int ls, uint32_t &base, int &extra)
{
  if (ls < 8)
  {
    base = 3u + (uint32_t) ls;
    extra = 0;
  }
  else if (ls == 28)
  {
    base = 258;
    extra = 0;
  }
  else
  {
    extra = (ls - 4) >> 2;
    base = 3u + ((4u + (uint32_t) (ls & 3)) << extra);
  }
}
__device__ __forceinline__ void dist_code(int ds, uint32_t &base, int &extra)
{
  if (ds < 4)
  {
    base = 1u + (uint32_t) ds;
    extra = 0;
  }
  else
  {
    extra = (ds - 2) >> 1;
    base = 1u + ((2u + (uint32_t) (ds & 1)) << extra);
  }
}
__constant__ uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// returns the number of bytes produced, or ~0u on a malformed stream
__device__ __forceinline__ uint32_t inflate_wave(const uint8_t *in, uint32_t in_len, uint8_t *ring, uint8_t *gout, uint32_t out_cap, HuffLds &h)
{
  const uint32_t lane = threadIdx.x & 63;
  BitReader br;
  br.in = in;
  br.win = h.win;
  br.pos = 0;
  br.end = in_len;
  br.buf = 0;
  br.cnt = 0;
  br.load_window(0);
  uint32_t o = 0, flushed = 0;
  constexpr uint32_t M = RING - 1;
  // completed FLUSH-byte chunks leave the ring: 64 lanes x 4 bytes, aligned on both sides
  auto flush = [&]() {
    while (flushed + FLUSH <= o)
    {
      const uint32_t v = *reinterpret_cast<const uint32_t *>(ring + ((flushed + 4 * lane) & M));
      *reinterpret_cast<uint32_t *>(gout + flushed + 4 * lane) = v;
      flushed += FLUSH;
    }
  };
  for (int guard = 0; guard < 4096; ++guard)
  {
    const uint32_t last = br.bits(1), type = br.bits(2);
    if (type == 0)
    {
      br.drop(br.cnt & 7);  // to the byte boundary
      if (br.cnt < 32) br.refill();
      const uint32_t len = br.bits(16), nlen = br.bits(16);
      if ((len ^ 0xFFFFu) != nlen || o + len > out_cap) return ~0u;
      // the bytes still in the bit buffer come first, the rest straight from the input
      uint32_t src = br.pos - (uint32_t) (br.cnt >> 3);
      if (src + len > br.end) return ~0u;
      // stored bytes pass through the ring in chunks so that the flush logic stays the only writer of the output
      for (uint32_t done = 0; done < len;)
      {
        const uint32_t n = len - done < FLUSH ? len - done : FLUSH;
        for (uint32_t k = lane; k < n; k += 64) ring[(o + k) & M] = in[src + done + k];
        __builtin_amdgcn_wave_barrier();
        o += n;
        done += n;
        flush();
      }
      br.pos = src + len;
      br.buf = 0;
      br.cnt = 0;
    }
    else if (type == 1 || type == 2)
    {
      int nlen, ndist;
      if (type == 1)
      {
        for (int i = 0; i < 144; ++i) h.lens[i] = 8;
        for (int i = 144; i < 256; ++i) h.lens[i] = 9;
        for (int i = 256; i < 280; ++i) h.lens[i] = 7;
        for (int i = 280; i < 288; ++i) h.lens[i] = 8;
        for (int i = 0; i < 30; ++i) h.lens[288 + i] = 5;
        nlen = 288;
        ndist = 30;
      }
      else
      {
        nlen = (int) br.bits(5) + 257;
        ndist = (int) br.bits(5) + 1;
        const int ncode = (int) br.bits(4) + 4;
        if (nlen > 286 || ndist > 30) return ~0u;
        uint8_t cl[19];
        for (int i = 0; i < 19; ++i) cl[i] = 0;
        for (int i = 0; i < ncode; ++i) cl[CL_ORDER[i]] = (uint8_t) br.bits(3);
        // the code-length code uses the distance tables' storage for a moment
        __builtin_amdgcn_wave_barrier();
        if (!build_tables(cl, 19, h.dfast, 7, h.dsym, h.dcount)) return ~0u;
        __builtin_amdgcn_wave_barrier();
        int idx = 0;
        while (idx < nlen + ndist)
        {
          if (br.cnt < 32) br.refill();  // a code (<= 7 bits) + its repeat count (<= 7 bits)
          const int s = decode_sym(br, h.dfast, 7, h.dsym, h.dcount);
          if (s < 0) return ~0u;
          if (s < 16)
            h.lens[idx++] = (uint8_t) s;
          else
          {
            int prev = 0, rep;
            if (s == 16)
            {
              if (idx == 0) return ~0u;
              prev = h.lens[idx - 1];
              rep = 3 + (int) br.bits(2);
            }
            else if (s == 17)
              rep = 3 + (int) br.bits(3);
            else
              rep = 11 + (int) br.bits(7);
            if (idx + rep > nlen + ndist) return ~0u;
            while (rep--) h.lens[idx++] = (uint8_t) prev;
          }
        }
        if (h.lens[256] == 0) return ~0u;  // no end-of-block code
        // the distance lengths move behind a fixed offset so that both builds read their own range
        __builtin_amdgcn_wave_barrier();
        uint8_t dl[32];
        for (int i = 0; i < ndist; ++i) dl[i] = h.lens[nlen + i];
        for (int i = 0; i < ndist; ++i) h.lens[288 + i] = dl[i];
      }
      __builtin_amdgcn_wave_barrier();
      if (!build_tables(h.lens, nlen, h.lfast, LIT_FAST, h.lsym, h.lcount)) return ~0u;
      if (!build_tables(h.lens + 288, ndist, h.dfast, DIST_FAST, h.dsym, h.dcount)) return ~0u;
      __builtin_amdgcn_wave_barrier();
      while (true)
      {
        if (br.cnt < 48) br.refill();  // literal/length code + extra + distance code + extra <= 15 + 5 + 15 + 13 bits
        const int s = decode_sym(br, h.lfast, LIT_FAST, h.lsym, h.lcount);
        if (s < 0) return ~0u;
        if (s < 256)
        {
          if (o >= out_cap) return ~0u;
          if (lane == 0) ring[o & M] = (uint8_t) s;
          ++o;
          if ((o & (FLUSH - 1)) == 0) flush();
        }
        else if (s == 256)
          break;
        else
        {
          const int ls = s - 257;
          if (ls >= 29) return ~0u;
          uint32_t lbase, dbase;
          int lextra, dextra;
          len_code(ls, lbase, lextra);
          const uint32_t len = lbase + br.bits(lextra);
          const int ds = decode_sym(br, h.dfast, DIST_FAST, h.dsym, h.dcount);
          if (ds < 0 || ds >= 30) return ~0u;
          dist_code(ds, dbase, dextra);
          const uint32_t dist = dbase + br.bits(dextra);
          if (dist > o || o + len > out_cap) return ~0u;
          // the whole wave copies the match; sources lie in the finished part of the output
          if (dist <= NEAR)
          {
            if (dist >= len)
            {
              for (uint32_t k = lane; k < len; k += 64) ring[(o + k) & M] = ring[(o - dist + k) & M];
            }
            else
            {
              for (uint32_t k = lane; k < len; k += 64) ring[(o + k) & M] = ring[(o - dist + k % dist) & M];
            }
          }
          else
          {
            // farther back than the ring keeps: those bytes were flushed long ago (dist > NEAR > len)
            for (uint32_t k = lane; k < len; k += 64) ring[(o + k) & M] = gout[o - dist + k];
          }
          __builtin_amdgcn_wave_barrier();
          o += len;
          if ((o ^ (o - len)) >= FLUSH) flush();
        }
      }
    }
    else
      return ~0u;
    if (last)
    {
      flush();
      for (uint32_t k = flushed + lane; k < o; k += 64) gout[k] = ring[k & M];  // ragged tail
      return o;
    }
  }
  return ~0u;
}

__global__ __launch_bounds__(64) void k_bgzf_inflate(const uint8_t *__restrict__ file, const BgzfBlock *__restrict__ blk, uint32_t nblk, uint8_t *__restrict__ out,
                                                     uint32_t *__restrict__ err)
{
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[RING];
  __shared__ HuffLds s_h;
  const uint32_t w = blockIdx.x;
  if (w >= nblk) return;
  const BgzfBlock b = blk[w];
  if (b.isize == 0) return;
  const uint32_t got = inflate_wave(file + b.in_off, b.clen, s_ring, out + b.out_off, b.isize, s_h);
  if (got != b.isize && threadIdx.x == 0) atomicOr(err, 1u);
}
}  // namespace

void launch_bgzf_inflate(const uint8_t *file_dev, const BgzfBlock *blk_dev, uint32_t nblk, uint8_t *out_dev, uint32_t *err_dev, hipStream_t st)
{
  if (nblk == 0) return;
  hipLaunchKernelGGL(k_bgzf_inflate, dim3(nblk), dim3(64), 0, st, file_dev, blk_dev, nblk, out_dev, err_dev);
}
