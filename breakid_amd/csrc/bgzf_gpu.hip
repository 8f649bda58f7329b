// BGZF inflate on the GPU (RFC 1951 DEFLATE) in two kernels (decode, resolve).
//
// the host decoder is the feed limit (all cores of the box: ~3.6 GB/s of inflated BAM).  BGZF blocks are
// independent <= 64 KiB deflate streams, so a file is thousands to millions of independent jobs, but inside a block
// the Huffman decode is one serial chain (a code's position depends on every code before it) and the LZ77 copies
// depend on earlier output.  The two dependences are separated:
//
//  k_bgzf_decode<1>  one wavefront per block; the 64 lanes WALK 64 parts of a Huffman block symbol by symbol, each from a
//                  guessed bit, and the entries are made true by handing every lane its neighbour's exit (Huffman streams
//                  synchronise) - "one Huffman block, every lane walking its own part" below.  Literals and match tokens
//                  as in <0>; a block whose parts do not synchronise is finished by the rounds of <0>.
//  k_bgzf_decode<0>  (rounds 1-2; BK_BGZF_ROUNDS=1, and the fallback of <1>)
//                  one wavefront per block walks the bit stream, 64 bit positions per round.  Lane k decodes the
//                  symbol that WOULD start at bit k of the unread input - literal/length code from a 10-bit direct
//                  table whose entries carry the pre-computed base value and extra-bit count, the length's extra
//                  bits, the distance code (8-bit direct table) and its extra bits - and so knows where the next
//                  symbol would start.  The scalar unit then only follows that chain from bit 0 (one readlane per
//                  symbol) and collects the mask of lanes that hold real symbols.  Those lanes get their output
//                  positions from a wave prefix sum of the output lengths; literal lanes store their byte, match lanes
//                  store one token (position, length, distance).  No byte is copied here, so there is no output
//                  window in LDS and no load->store chain; ~7 KiB of LDS per wave (tables, 768-byte input window fed by
//                  a register prefetch).  Codes longer than the direct tables, and the block headers, take a
//                  scalar path.  The Huffman tables are built by the 64 lanes in parallel (counts by LDS atomics,
//                  canonical ranks by ballots).  (Measured before this layout: the whole decoder on the scalar unit,
//                  ~110 scalar instructions per symbol; the CU's scalar issue rate was the limit whatever the
//                  occupancy: 33 ms for 8736 blocks.)
//  k_bgzf_resolve  one 1024-thread workgroup per block builds parent[p] for all positions in LDS (p itself for
//                  literals, p - dist inside a match) and runs pointer jumping, parent[p] = parent[parent[p]], in place
//                  and without barriers until every position points at a literal (a jump only ever replaces an
//                  ancestor by an older ancestor, so racing reads are harmless); ~log2(longest copy chain) sweeps.
//                  Then out[p] = out[parent[p]].
//
// (Earlier versions, 8736 blocks / 570 MB inflated: matches copied through an LDS ring of the last 16-32 KiB of output -
// 62-90 ms, bound by waves per CU and the copy's LDS round trips; input and output in global memory - every refill and
// every match copy paid a full memory round trip, 29 ms per block.)
#include "bk_common.h"
#include "bgzf_gpu.h"
#include <cstdio>
#include <cstdlib>

namespace
{
constexpr int LIT_FAST = 11, DIST_FAST = 9;
// symbol statistics (build with -DBGZF_STATS, run with BK_DEBUG=bgzf); off by default
#ifdef BGZF_STATS
__device__ unsigned long long g_bgzf_stats[8];
__device__ unsigned long long g_lane_stats2[4];  // steps with a long code, lanes with a long code, walking lanes over all steps
__device__ unsigned long long g_lane_stats[8];  // walk clocks, steps, write clocks, windows, long passes after a window's first, passes, windows left to the rounds, slowest block
#define ST(x) x
#else
#define ST(x)
#endif
constexpr uint32_t CHUNK_DW = 64;            // one dword per lane
constexpr uint32_t WIN_DW = 3 * CHUNK_DW;    // input window: three chunks in LDS, the fourth on its way in a register
#ifndef LANE_PART_BITS
constexpr uint32_t LANE_STAGE_DW = 2120;
#else
constexpr uint32_t LANE_STAGE_DW = 64 * (LANE_PART_BITS / 32) + 4 + (64 * (LANE_PART_BITS / 32) + 4) / 32 + 4;
#endif
#ifndef BGZF_SHARED_WAVES
#define BGZF_SHARED_WAVES 4  // waves per SIMD of the lanes decoder inside the streaming feed (its 15.5 KiB of LDS allow 10 waves per CU; 2 or 3 per SIMD measured the same: tools/gpu_lane_variants2.sh)
#endif
constexpr uint32_t RESOLVE_THREADS = 256;
constexpr uint32_t RES_WIN = 16384;            // output positions whose parents are in LDS at a time (k_bgzf_resolve)

// direct-table entries (u16)
//   literal/length: bits 0-3 code length (0 = longer than the table); bit 4 clear: literal in bits 8-15; bit 4 set: bits 5-7 extra
//                   bits x, bits 8-15 m with length = 3 + (m << x) + extra value; x = 7 marks end of block (m = 0) / an invalid symbol (m = 1)
//   distance:       bits 0-3 code length, 4-7 extra bits x, 8-9 m with distance = 1 + (m << x) + extra value, bit 10 invalid symbol
//   code-length code: (symbol << 4) | code length
enum : int { T_LITLEN = 0, T_DIST = 1, T_CLEN = 2 };
constexpr uint32_t E_EOB = 0x00F0u, E_BAD = 0x01F0u;  // | code length

struct HuffLds
{
  uint16_t lfast[1 << LIT_FAST];
  uint16_t dfast[1 << DIST_FAST];
  uint16_t lent[288], dent[32];    // table entries of all symbols ordered by (length, symbol): codes longer than the direct tables
  uint32_t llim[16], lbas[16];     // literal/length code: codes of length L have 15-bit left-justified values < llim[L], slot = lbas[L] + (value >> (15 - L))
  uint32_t dlim[16], dbas[16];     // the same for the distance code
  uint32_t lcount[16], dcount[16];
  uint32_t offs[16], nextc[16];    // table build scratch
  uint8_t lens[344];               // code lengths: literal/length [0,288), distance [288,320), code-length code [320,339)
  uint32_t win[WIN_DW];
  uint32_t stage[LANE_STAGE_DW];   // the window of the stream the 64 lanes walk (lanes_block), rows of 32 dwords + 1 pad
};

// every lane holds the same decoder state; saying so keeps it in scalar registers (the compiler cannot prove it
// across LDS round trips)
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }
__device__ __forceinline__ uint32_t bitrev(uint32_t c, uint32_t len) { return __brev(c) >> (32 - len); }
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u)); }

// inclusive prefix sum over the wave (DPP row shifts + row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xF, 0xF, false);  // row_shr:1
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return v;
}

// RFC 1951 section 3.2.5 in the table's terms: length code 257 + ls -> 3 + (m << x), distance code ds -> 1 + (m << x)
template <int T>
__device__ __forceinline__ uint32_t make_entry(uint32_t s, uint32_t len)
{
  if (T == T_CLEN) return (s << 4) | len;
  if (T == T_DIST)
  {
    if (s >= 30) return len | (1u << 10);
    const uint32_t x = s < 4 ? 0u : (s - 2) >> 1, m = s < 4 ? s : 2u + (s & 1u);
    return len | (x << 4) | (m << 8);
  }
  if (s < 256) return len | (s << 8);
  if (s == 256) return len | E_EOB;
  if (s > 285) return len | E_BAD;
  const uint32_t ls = s - 257;
  const uint32_t x = (ls < 8 || ls == 28) ? 0u : (ls - 4) >> 2, m = ls < 8 ? ls : ls == 28 ? 255u : 4u + (ls & 3u);
  return len | 16u | (x << 5) | (m << 8);
}

// Canonical Huffman tables from n code lengths, built by the wave: counts per length (LDS atomics), first code and
// first slot of every length (15 serial steps), then 64 symbols at a time: a symbol's rank among the symbols of its
// length is the number of such symbols in earlier chunks plus the lanes below it in a ballot.  Symbols whose code fits
// the direct table fill their 2^(fast_bits - len) slots.  Returns false for an over-subscribed set of lengths.
template <int T>
__device__ __noinline__ bool build_tables(HuffLds &h, const uint8_t *lens, uint32_t n, uint16_t *fast, uint32_t fast_bits, uint16_t *sym, uint32_t *count)
{
  const uint32_t lane = threadIdx.x & 63;
  if (lane < 16) count[lane] = 0;
  for (uint32_t i = lane; i < (1u << fast_bits); i += 64) fast[i] = 0;
  __builtin_amdgcn_wave_barrier();
  for (uint32_t s = lane; s < n; s += 64) atomicAdd(&count[lens[s]], 1u);
  __builtin_amdgcn_wave_barrier();
  if (uni(count[0]) == n) return true;  // no codes at all (the distance tree of a literal-only block)
  int left = 1;
  uint32_t code = 0, off = 0;
  for (uint32_t len = 1; len < 16; ++len)
  {
    const uint32_t c = uni(count[len]);
    left = (left << 1) - (int) c;
    if (left < 0) return false;
    if (lane == 0)
    {
      h.offs[len] = off;
      h.nextc[len] = code;
      if (T == T_LITLEN)
      {
        h.llim[len] = (code + c) << (15 - len);
        h.lbas[len] = off - code;
      }
      if (T == T_DIST)
      {
        h.dlim[len] = (code + c) << (15 - len);
        h.dbas[len] = off - code;
      }
    }
    off += c;
    code = (code + c) << 1;
  }
  __builtin_amdgcn_wave_barrier();
  uint32_t seen[16];  // symbols of each length in earlier chunks (uniform; the loops below are fully unrolled)
#pragma unroll
  for (int len = 1; len < 16; ++len) seen[len] = 0;
  for (uint32_t base = 0; base < n; base += 64)
  {
    const uint32_t s = base + lane;
    const uint32_t mylen = s < n ? lens[s] : 0u;
    uint32_t rank = 0;
#pragma unroll
    for (int len = 1; len < 16; ++len)
    {
      const unsigned long long m = __builtin_amdgcn_ballot_w64(mylen == (uint32_t) len);
      if (mylen == (uint32_t) len) rank = seen[len] + lanes_below(m);
      seen[len] += (uint32_t) __popcll(m);
    }
    if (mylen)
    {
      const uint16_t e = (uint16_t) make_entry<T>(s, mylen);
      sym[h.offs[mylen] + rank] = e;
      if (mylen <= fast_bits)
      {
        for (uint32_t j = bitrev(h.nextc[mylen] + rank, mylen); j < (1u << fast_bits); j += 1u << mylen) fast[j] = e;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  return true;
}

struct Decoder
{
  HuffLds &h;
  const uint32_t *in32;  // the stream's dwords, from the 4-byte boundary at or below its first byte (global)
  uint32_t bitpos;       // next unread bit, relative to in32
  uint32_t end_bit;      // end of the stream
  uint32_t end_dw;       // dwords that may be loaded
  uint32_t win_dw;       // dword index of win[0]
  uint32_t pre;          // prefetched dword win_dw + WIN_DW + lane
  uint32_t lane;

  __device__ __forceinline__ uint32_t load_dw(uint32_t d) const { return d < end_dw ? in32[d] : 0u; }
  // window = [at, at + WIN_DW) dwords, prefetch of the following chunk under way
  __device__ __forceinline__ void seek(uint32_t at)
  {
    win_dw = at;
    const uint32_t a = load_dw(at + lane), b = load_dw(at + CHUNK_DW + lane), c = load_dw(at + 2 * CHUNK_DW + lane);
    pre = load_dw(at + WIN_DW + lane);
    __builtin_amdgcn_wave_barrier();
    h.win[lane] = a;
    h.win[CHUNK_DW + lane] = b;
    h.win[2 * CHUNK_DW + lane] = c;
    __builtin_amdgcn_wave_barrier();
  }
  // keeps the read position inside the first chunk + a few dwords: the window slides by one chunk, the prefetched
  // chunk lands in LDS and the next one is requested
  __device__ __forceinline__ void ensure()
  {
    uint32_t ahead = (bitpos >> 5) - win_dw;
    if (ahead < CHUNK_DW) return;
    if (ahead >= 2 * CHUNK_DW)
    {
      seek(bitpos >> 5);
      return;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t b = h.win[CHUNK_DW + lane], c = h.win[2 * CHUNK_DW + lane];
    h.win[lane] = b;
    h.win[CHUNK_DW + lane] = c;
    h.win[2 * CHUNK_DW + lane] = pre;
    win_dw += CHUNK_DW;
    pre = load_dw(win_dw + WIN_DW + lane);
    __builtin_amdgcn_wave_barrier();
  }
  // the next 32 unread bits (uniform)
  __device__ __forceinline__ uint32_t peek()
  {
    ensure();
    const uint32_t b = bitpos - (win_dw << 5), di = b >> 5, sh = b & 31;
    const uint32_t w0 = uni(h.win[di]), w1 = uni(h.win[di + 1]);
    return (uint32_t) ((((uint64_t) w1 << 32) | w0) >> sh);
  }
  __device__ __forceinline__ uint32_t bits(uint32_t n)
  {
    const uint32_t v = peek() & ((1u << n) - 1u);
    bitpos += n;
    return v;
  }
};

// canonical walk of puff.c over the bits in w (LSB first; codes are MSB first): the symbol's table entry, or 0
__device__ __forceinline__ uint32_t walk_code(uint32_t w, const uint16_t *ent, const uint32_t *count)
{
  int code = 0, first = 0, index = 0;
  for (uint32_t len = 1; len < 16; ++len)
  {
    code |= (int) (w & 1u);
    w >>= 1;
    const int c = (int) uni(count[len]);
    if (code - c < first) return uni(ent[index + (code - first)]);
    index += c;
    first += c;
    first <<= 1;
    code <<= 1;
  }
  return 0;
}

// ---- one Huffman block, every lane walking its own part of the bit stream ---------------------------------------------
// The rounds below spend 64 lanes on 64 bit positions of which ~7 start a symbol.  Here the block is taken in windows of
// 64 x LANE_PART bits (8 KiB of the stream, staged in LDS) and lane i walks part i of the window symbol by symbol, one Huffman
// code per step (a literal/length code with the length's extra bits, or - after a length - the distance code with its extra
// bits: one table lookup per step, never more than 28 bits, two dwords of the staged stream).  Lane i does not know where a
// code starts inside its part: it starts at the part's first bit, and Huffman streams synchronise - after some symbols a walk
// from a wrong bit is on the true chain.  The walk of lane i ends at the first literal/length code at or beyond the end of its
// part; that exit is where lane i+1 must really enter.
// Every walk remembers where it stood, and what it had counted, at its first code at or beyond bit LANE_CHECK of its part
// (checkpoint).  A lane whose entry changes walks from the new entry to the checkpoint bit only: at the same code there, the
// rest of the old walk - its exit, its counts from the checkpoint on - holds for the new entry too; elsewhere the lane walks on
// to the end of its part and that walk is its reference from then on.  Entries are handed on until none changes: lane 0's entry
// is the true one (the block's first code, or the exit of the window before), so after pass k lanes 0..k-1 are final whatever the
// data (a fixed point in at most 64 passes); when every speculative walk synchronised before its checkpoint - the usual case -
// the second pass is 64 short walks and the last.  The walks count output bytes and matches, a wave prefix sum turns the counts
// into positions, one more walk writes, and the exit of lane 63 is the first code of the next window.
//
// A stretch whose codes all have the same length (random bytes) never synchronises: a walk from a wrong bit stays wrong and the
// true entries move on by one lane per pass.  After LANE_FULL_WALKS passes that needed walks beyond the checkpoint the rest of
// the block is left to the rounds, which do not depend on the data (X_RETRY: nothing of the window has been written by then).
constexpr uint32_t X_EOB = 1u << 30, X_BAD = 1u << 31, X_NONE = 0xFFFFFFFFu, X_POS = (1u << 30) - 1u, X_RETRY = 0xFFFFFFFEu;
#ifndef LANE_PART_BITS
#define LANE_PART_BITS 1024
#define LANE_CHECK_BITS 384
#endif
constexpr uint32_t LANE_PART = LANE_PART_BITS;    // bits of a lane's part: a padded row of the staged window
constexpr uint32_t LANE_CHECK = LANE_CHECK_BITS;  // bits from the start of a part to its checkpoint
constexpr int LANE_FULL_WALKS = 8;     // per window; a long pass costs ~1/10 of the rounds over a window
static_assert(LANE_STAGE_DW >= 64 * (LANE_PART / 32) + 4 + (64 * (LANE_PART / 32) + 4) / 32 + 1, "the staged window, the dwords a last code may reach into, one pad dword per row");
// dword d of the window lives at d + d / 32: the lanes are 32 dwords apart, the pad spreads them over the banks
__device__ __forceinline__ uint32_t stage_at(uint32_t d) { return d + (d >> 5); }

struct LaneWalk
{
  uint32_t pos;   // bit position of the next code
  uint32_t opos;  // output position (counting walks: bytes so far)
  uint32_t nm;    // matches so far (writing walk: index of the next token)
  uint32_t stop;  // X_EOB / X_BAD once the walk has ended for good
};

// lanes with `run` walk on until their next literal/length code is at or beyond `until`, or the walk ends (k.stop).
// One step = one code of every walking lane; straight-line apart from the long codes and the stores: a lane that has finished
// keeps executing with an advance of 0 bits.  wbit0 = bit position of the window's first staged dword.
template <int WRITE>
__device__ __forceinline__ void lane_segment(LaneWalk &k, const HuffLds &h, uint32_t wbit0, uint32_t end_bit, uint32_t until, bool run, uint8_t *__restrict__ gout,
                                             unsigned long long *__restrict__ tok, uint32_t &bad)
{
  const uint16_t *tab = h.lfast;  // dfast follows it
  uint32_t pos = k.pos, opos = k.opos, nm = k.nm, kstop = k.stop;
  uint32_t want = 0, len = 0;
  uint32_t act = (uint32_t) (run && !kstop && pos < until);
  ST(uint32_t st_steps = 0; uint32_t st_long = 0; uint32_t st_long_lanes = 0; uint32_t st_lanes = 0;)
  if (__builtin_amdgcn_ballot_w64(act != 0u))
  {
    do
    {
      ST(++st_steps; st_lanes += (uint32_t) __popcll(__builtin_amdgcn_ballot_w64(act != 0u));)
      const uint32_t rel = pos - wbit0, d = rel >> 5;
      const uint32_t lo = h.stage[stage_at(d)], hi = h.stage[stage_at(d + 1u)];
      const uint32_t w = __builtin_amdgcn_alignbit(hi, lo, rel & 31u);
      uint32_t e = tab[want ? (1u << LIT_FAST) | (w & ((1u << DIST_FAST) - 1u)) : (w & ((1u << LIT_FAST) - 1u))];
      if (__builtin_amdgcn_ballot_w64((e & 15u) == 0u && act))
      {
        ST(++st_long; st_long_lanes += (uint32_t) __popcll(__builtin_amdgcn_ballot_w64((e & 15u) == 0u && act));)
        // codes longer than the direct table: the length from the canonical limits, then the symbol's entry
        if ((e & 15u) == 0u)
        {
          const uint32_t v = __brev(w) >> 17;
          uint32_t Ll = LIT_FAST + 1, Ld = DIST_FAST + 1;
#pragma unroll
          for (uint32_t q = LIT_FAST + 1; q < 15; ++q) Ll += (uint32_t) (v >= h.llim[q]);
#pragma unroll
          for (uint32_t q = DIST_FAST + 1; q < 15; ++q) Ld += (uint32_t) (v >= h.dlim[q]);
          const uint32_t L = want ? Ld : Ll;
          const uint32_t sl = (want ? h.dbas[L] : h.lbas[L]) + (v >> (15u - L));
          const bool ok = want ? (v < h.dlim[15] && sl < 32u) : (v < h.llim[15] && sl < 288u);
          e = ok ? (want ? h.dent[sl] : h.lent[sl]) : 0u;
        }
      }
      // the fields of either entry kind: x extra bits, value = (m << x) + extra (the literal itself, length - 3, distance - 1)
      const uint32_t cl = e & 15u;
      const uint32_t is_lit = ((want | (e >> 4)) & 1u) ^ 1u;  // literal/length state and bit 4 clear
      const uint32_t xr = want ? (e >> 4) & 15u : (e >> 5) & 7u;
      const uint32_t special = (want ^ 1u) & (is_lit ^ 1u) & (uint32_t) (xr == 7u);  // end of block / invalid symbol
      const uint32_t x = (is_lit | special) ? 0u : xr;
      const uint32_t m = want ? (e >> 8) & 3u : e >> 8;
      const uint32_t val = (m << x) + __builtin_amdgcn_ubfe(w, cl, x);
      pos += act ? cl + x : 0u;
      const uint32_t isbad = (uint32_t) (cl == 0u) | (want & (e >> 10)) | (special & (uint32_t) (m != 0u)) | (uint32_t) (pos > end_bit);
      const uint32_t stop = (isbad & 1u) ? X_BAD : special ? X_EOB : 0u;
      const uint32_t ok = act & ((isbad & 1u) ^ 1u);
      const uint32_t done_match = ok & want, put_lit = ok & (want ^ 1u) & is_lit, got_len = ok & (want ^ 1u) & (is_lit ^ 1u) & (special ^ 1u);
      if (WRITE)
      {
        if (done_match)
        {
          if (val >= opos) bad = 1;  // distance = val + 1 beyond the start of the output
          tok[nm] = (unsigned long long) opos | ((unsigned long long) len << 16) | ((unsigned long long) (val + 1u) << 32);
        }
        if (put_lit) gout[opos] = (uint8_t) val;
      }
      opos += done_match ? len : put_lit;
      nm += done_match;
      len = got_len ? val + 3u : len;
      want = got_len;
      kstop = act ? stop : kstop;
      act = act & (uint32_t) (stop == 0u) & (want | (uint32_t) (pos < until));
    } while (__builtin_amdgcn_ballot_w64(act != 0u));
  }
  k.pos = pos;
  k.opos = opos;
  k.nm = nm;
  k.stop = kstop;
  ST(if (threadIdx.x == 0) { atomicAdd(&g_lane_stats[1], (unsigned long long) st_steps); atomicAdd(&g_lane_stats2[0], (unsigned long long) st_long); atomicAdd(&g_lane_stats2[1], (unsigned long long) st_long_lanes); atomicAdd(&g_lane_stats2[2], (unsigned long long) st_lanes); })
}

// the Huffman block whose first code is at `bitpos` (tables built): literals and match tokens written, o / ntok advanced;
// returns the bit behind the end-of-block code, X_RETRY with the first code that has not been decoded in `bitpos` (see above),
// or ~0u (malformed, or more output than out_cap)
__device__ __forceinline__ uint32_t lanes_block(HuffLds &h, const uint32_t *__restrict__ in32, uint32_t end_dw, uint32_t end_bit, uint32_t &bitpos, uint8_t *__restrict__ gout,
                                                unsigned long long *__restrict__ tok, uint32_t out_cap, uint32_t &o, uint32_t &ntok)
{
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t last_dw = end_dw - 1u;  // dwords behind the stream are not read: a code that would need them ends beyond end_bit (X_BAD)
  uint32_t cur = bitpos;
  for (;;)
  {
    int full_walks = 0;
    if (cur >= end_bit) return ~0u;
    const uint32_t d0 = cur >> 5, wbit0 = d0 << 5;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j = lane; j < 64u * (LANE_PART / 32u) + 4u; j += 64) h.stage[stage_at(j)] = in32[d0 + j < last_dw ? d0 + j : last_dw];
    __builtin_amdgcn_wave_barrier();
    const uint32_t s = cur + lane * LANE_PART;
    const uint32_t limit = s + LANE_PART < end_bit ? s + LANE_PART : end_bit;
    const uint32_t check = s + LANE_CHECK < limit ? s + LANE_CHECK : limit;
    // the reference walk of this lane: its entry, its checkpoint (position, counts there), its exit and counts at the exit
    uint32_t ref_entry = X_NONE, cp_pos = X_NONE, cp_out = 0, cp_m = 0, ref_exit = X_NONE, ref_out = 0, ref_m = 0;
    // what holds for the current entry
    uint32_t entry = s < end_bit ? s : X_NONE, exitv = X_NONE, nout = 0, nm = 0, bad = 0;
    bool need = true;
    for (int pass = 0; pass < 66; ++pass)
    {
      const bool walk = need && entry != X_NONE && entry < limit && entry != ref_entry;
      ST(if (threadIdx.x == 0) { atomicAdd(&g_lane_stats[5], 1ull); if (pass == 0) atomicAdd(&g_lane_stats[3], 1ull); })
      ST(const uint64_t st_w0 = wall_clock64();)
      LaneWalk k = {walk ? entry : wbit0, 0u, 0u, 0u};
      lane_segment<0>(k, h, wbit0, end_bit, check, walk, nullptr, nullptr, bad);
      // at the reference walk's checkpoint: the rest of that walk holds
      const bool joined = walk && !k.stop && cp_pos != X_NONE && k.pos == cp_pos;
      const bool on = walk && !joined;
      const uint32_t my_cp_pos = k.stop ? X_NONE : k.pos, my_cp_out = k.opos, my_cp_m = k.nm;
      if (__builtin_amdgcn_ballot_w64(on && !k.stop && k.pos < limit))
      {
        ST(if (threadIdx.x == 0 && pass) atomicAdd(&g_lane_stats[4], 1ull);)
        if (++full_walks > LANE_FULL_WALKS)
        {
          ST(if (threadIdx.x == 0) atomicAdd(&g_lane_stats[6], 1ull);)
          bitpos = cur;
          return X_RETRY;
        }
        lane_segment<0>(k, h, wbit0, end_bit, limit, on, nullptr, nullptr, bad);
      }
      ST(if (threadIdx.x == 0) atomicAdd(&g_lane_stats[0], (unsigned long long) (wall_clock64() - st_w0));)
      if (on)
      {
        ref_entry = entry;
        cp_pos = my_cp_pos;
        cp_out = my_cp_out;
        cp_m = my_cp_m;
        ref_exit = k.pos | k.stop;
        ref_out = k.opos;
        ref_m = k.nm;
      }
      if (need)
      {
        if (entry == X_NONE || entry >= limit)
        {
          exitv = entry;  // no entry: no exit; an entry beyond the part: nothing to walk, the neighbour enters there
          nout = nm = 0;
        }
        else if (joined)
        {
          exitv = ref_exit;
          nout = k.opos + (ref_out - cp_out);
          nm = k.nm + (ref_m - cp_m);
        }
        else  // the reference walk itself (just made, or met again)
        {
          exitv = ref_exit;
          nout = ref_out;
          nm = ref_m;
        }
      }
      const uint32_t prev = (uint32_t) __shfl_up((int) exitv, 1);
      uint32_t ne = lane == 0 ? cur : ((prev & (X_EOB | X_BAD)) ? X_NONE : prev);
      if (ne != X_NONE && ne >= end_bit) ne = X_NONE;
      need = ne != entry;
      entry = ne;
      if (!__builtin_amdgcn_ballot_w64(need)) break;
    }
    const bool real = entry != X_NONE;
    const unsigned long long reals = __builtin_amdgcn_ballot_w64(real);
    const unsigned long long stops = __builtin_amdgcn_ballot_w64(real && (exitv & (X_EOB | X_BAD)) != 0u);
    uint32_t fx = 0;
    if (stops)
    {
      const uint32_t fs = (uint32_t) __ffsll((long long) stops) - 1u;
      fx = (uint32_t) __builtin_amdgcn_readlane((int) exitv, (int) fs);
      if (fx & X_BAD) return ~0u;
    }
    else if (~reals)
      return ~0u;  // the stream ends inside the window without an end-of-block code
    const uint32_t my_out = real ? nout : 0u, my_m = real ? nm : 0u;
    const uint32_t incl = wave_incl_scan(my_out), mincl = wave_incl_scan(my_m);
    const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63), mtotal = (uint32_t) __builtin_amdgcn_readlane((int) mincl, 63);
    if (o + total > out_cap || ntok + mtotal > BGZF_TOKENS_PER_BLOCK) return ~0u;
    ST(const uint64_t st_w1 = wall_clock64();)
    {
      const bool wr = real && entry < limit;
      LaneWalk k = {wr ? entry : wbit0, o + incl - my_out, ntok + mincl - my_m, 0u};
      lane_segment<1>(k, h, wbit0, end_bit, limit, wr, gout, tok, bad);
    }
    ST(if (threadIdx.x == 0) atomicAdd(&g_lane_stats[2], (unsigned long long) (wall_clock64() - st_w1));)
    if (__builtin_amdgcn_ballot_w64(bad != 0u)) return ~0u;
    o += total;
    ntok += mtotal;
    if (stops) return fx & X_POS;
    const uint32_t nxt = (uint32_t) __builtin_amdgcn_readlane((int) exitv, 63);  // lane 63 is real and went on to the end of its part
    if (nxt <= cur) return ~0u;
    cur = nxt;
  }
}

// Walks one deflate stream: literals -> gout, matches -> tok[] (position | length << 16 | distance << 32).  Returns the
// number of output positions (ntok = number of tokens), or ~0u on a malformed stream.
template <int LANES>
__device__ __forceinline__ uint32_t decode_wave(const uint8_t *file, uint64_t in_off, uint32_t in_len, uint8_t *gout, unsigned long long *tok, uint32_t out_cap, HuffLds &h, uint32_t &ntok)
{
  const uint32_t lane = threadIdx.x & 63;
  Decoder dc = {h};
  dc.lane = lane;
  dc.in32 = reinterpret_cast<const uint32_t *>(file + (in_off & ~3ull));
  const uint8_t *in8 = reinterpret_cast<const uint8_t *>(dc.in32);
  dc.bitpos = (uint32_t) (in_off & 3ull) * 8u;
  dc.end_bit = dc.bitpos + in_len * 8u;
  dc.end_dw = (dc.end_bit + 31u) >> 5;
  dc.seek(0);
  uint32_t o = 0;
  ntok = 0;
  ST(uint32_t st_fix = 0; uint32_t st_ll = 0; uint32_t st_dlong = 0; uint32_t st_rounds = 0; uint32_t st_slow = 0; uint32_t st_dyn = 0; uint64_t st_tb = 0; const uint64_t st_t0 = wall_clock64();)
  for (int guard = 0; guard < 4096; ++guard)
  {
    const uint32_t hdr = dc.bits(3), last = hdr & 1u, type = hdr >> 1;
    if (type == 0)
    {
      dc.bitpos = (dc.bitpos + 7u) & ~7u;  // to the byte boundary
      const uint32_t len = dc.bits(16), nlen = dc.bits(16);
      if ((len ^ 0xFFFFu) != nlen || o + len > out_cap || dc.bitpos + 8u * len > dc.end_bit) return ~0u;
      const uint32_t src = dc.bitpos >> 3;
      for (uint32_t k = lane; k < len; k += 64) gout[o + k] = in8[src + k];
      o += len;
      dc.bitpos += 8u * len;
    }
    else if (type == 1 || type == 2)
    {
      uint32_t nlen, ndist;
      ST(const uint64_t tb0 = wall_clock64();)
      if (type == 1)
      {
        for (uint32_t i = lane; i < 288; i += 64) h.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        if (lane < 32) h.lens[288 + lane] = lane < 30 ? 5 : 0;
        nlen = 288;
        ndist = 30;
      }
      else
      {
        const uint32_t hd = dc.bits(14);
        nlen = (hd & 31u) + 257u;
        ndist = ((hd >> 5) & 31u) + 1u;
        const uint32_t ncode = (hd >> 10) + 4u;
        if (nlen > 286 || ndist > 30) return ~0u;
        static constexpr uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        if (lane < 19) h.lens[320 + lane] = 0;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = 0; i < ncode; ++i) h.lens[320 + ORDER[i]] = (uint8_t) dc.bits(3);
        // the code-length code uses the distance tables' storage for a moment (its codes are at most 7 bits: all direct)
        __builtin_amdgcn_wave_barrier();
        if (!uni(build_tables<T_CLEN>(h, h.lens + 320, 19, h.dfast, 7, h.dent, h.dcount))) return ~0u;
        uint32_t idx = 0;
        while (idx < nlen + ndist)
        {
          const uint32_t w = dc.peek();
          const uint32_t e = uni(h.dfast[w & 127u]);
          if (e == 0) return ~0u;
          const uint32_t s = e >> 4, cl = e & 15u;
          if (s < 16)
          {
            h.lens[idx++] = (uint8_t) s;
            dc.bitpos += cl;
          }
          else
          {
            uint32_t prev = 0, rep;
            if (s == 16)
            {
              if (idx == 0) return ~0u;
              prev = uni(h.lens[idx - 1]);
              rep = 3 + ((w >> cl) & 3u);
              dc.bitpos += cl + 2;
            }
            else if (s == 17)
            {
              rep = 3 + ((w >> cl) & 7u);
              dc.bitpos += cl + 3;
            }
            else
            {
              rep = 11 + ((w >> cl) & 127u);
              dc.bitpos += cl + 7;
            }
            if (idx + rep > nlen + ndist) return ~0u;
            for (uint32_t k = lane; k < rep; k += 64) h.lens[idx + k] = (uint8_t) prev;
            idx += rep;
          }
        }
        __builtin_amdgcn_wave_barrier();
        if (uni(h.lens[256]) == 0) return ~0u;  // no end-of-block code
        // the distance lengths move to their fixed place
        const uint32_t dl = lane < ndist ? h.lens[nlen + lane] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (lane < 32) h.lens[288 + lane] = (uint8_t) dl;
      }
      __builtin_amdgcn_wave_barrier();
      ST(++st_dyn;)
      if (!uni(build_tables<T_LITLEN>(h, h.lens, nlen, h.lfast, LIT_FAST, h.lent, h.lcount))) return ~0u;
      if (!uni(build_tables<T_DIST>(h, h.lens + 288, ndist, h.dfast, DIST_FAST, h.dent, h.dcount))) return ~0u;
      ST(st_tb += wall_clock64() - tb0;)
      bool walked = false;
      if (LANES)
      {
        uint32_t at = uni(dc.bitpos);
        const uint32_t r = uni(lanes_block(h, dc.in32, dc.end_dw, dc.end_bit, at, gout, tok, out_cap, o, ntok));
        if (r == ~0u) return ~0u;
        o = uni(o);
        ntok = uni(ntok);
        dc.bitpos = r != X_RETRY ? r : uni(at);
        walked = r != X_RETRY;
      }
      // (the walks of the lanes did not agree within LANE_FULL_WALKS long passes: nothing was written, the rounds decode the block)
      while (!walked)
      {
        dc.ensure();
        dc.bitpos = uni(dc.bitpos);
        dc.win_dw = uni(dc.win_dw);
        o = uni(o);
        ntok = uni(ntok);
        // ---- all 64 bit positions at once (straight-line code: flags are combined with & and |)
        const uint32_t lb = dc.bitpos - (dc.win_dw << 5) + lane, di = lb >> 5, sh = lb & 31u;
        const uint32_t w0 = h.win[di], w1 = h.win[di + 1], w2 = h.win[di + 2];
        const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
        uint32_t e = h.lfast[lo & ((1u << LIT_FAST) - 1u)];
        uint32_t other, is_len, len, dist, nxt1;
        auto from_entry = [&]() {
          const uint32_t cl = e & 15u, lextra = (e >> 5) & 7u;
          const uint32_t t = cl + lextra;
          const uint32_t dbits = __builtin_amdgcn_alignbit(hi, lo, t);
          const uint32_t d = h.dfast[dbits & ((1u << DIST_FAST) - 1u)];
          const uint32_t dl = d & 15u, dextra = (d >> 4) & 15u;
          len = 3u + ((e >> 8) << lextra) + __builtin_amdgcn_ubfe(lo, cl, lextra);
          dist = 1u + (((d >> 8) & 3u) << dextra) + __builtin_amdgcn_ubfe(dbits, dl, dextra);
          other = (e >> 4) & 1u;
          is_len = other & (uint32_t) (lextra != 7u);
          const uint32_t is_eob = (uint32_t) ((e & 0xFFF0u) == E_EOB), is_bad = (uint32_t) ((e & 0xFFF0u) == E_BAD);
          const uint32_t n = lane + (is_len ? t + dl + dextra : cl);  // < 128
          const uint32_t ok = (uint32_t) (cl != 0u) & (is_bad ^ 1u) & ((is_len ^ 1u) | ((uint32_t) (dl != 0u) & (((d >> 10) & 1u) ^ 1u))) & (uint32_t) (dc.bitpos + n <= dc.end_bit);
          nxt1 = ok ? (n | (is_eob << 8)) - 1u : 0xFFFFFFFFu;
        };
        from_entry();
        // ---- the chain of real symbols from bit 0: sel = lanes that start one; the walk ends at a symbol that cannot be
        // decoded here (nn = ~0), at the end of block, or at one that ends at or beyond bit 64 (both: nn >= 63)
        uint32_t k = 0, nn;
        unsigned long long sel = 0;
        auto hop = [&]() {
          asm volatile("1:\n\t"
                       "v_readlane_b32 %2, %3, %1\n\t"
                       "s_cmp_ge_u32 %2, 63\n\t"
                       "s_cbranch_scc1 2f\n\t"
                       "s_bitset1_b64 %0, %1\n\t"
                       "s_add_u32 %1, %2, 1\n\t"
                       "v_readlane_b32 %2, %3, %1\n\t"
                       "s_cmp_ge_u32 %2, 63\n\t"
                       "s_cbranch_scc1 2f\n\t"
                       "s_bitset1_b64 %0, %1\n\t"
                       "s_add_u32 %1, %2, 1\n\t"
                       "v_readlane_b32 %2, %3, %1\n\t"
                       "s_cmp_ge_u32 %2, 63\n\t"
                       "s_cbranch_scc1 2f\n\t"
                       "s_bitset1_b64 %0, %1\n\t"
                       "s_add_u32 %1, %2, 1\n\t"
                       "v_readlane_b32 %2, %3, %1\n\t"
                       "s_cmp_ge_u32 %2, 63\n\t"
                       "s_cbranch_scc1 2f\n\t"
                       "s_bitset1_b64 %0, %1\n\t"
                       "s_add_u32 %1, %2, 1\n\t"
                       "s_branch 1b\n"
                       "2:"
                       : "+s"(sel), "+s"(k), "=&s"(nn)
                       : "v"(nxt1)
                       : "scc");
        };
        hop();
        if (nn == 0xFFFFFFFFu && ((uint32_t) __builtin_amdgcn_readlane((int) e, (int) k) & 15u) == 0u)
        {
          // the chain ran into a code longer than the direct table: every lane that looks at such a code decodes it
          // from the canonical limits (codes of length L lie below llim[L] when left-justified in 15 bits), the lanes
          // are re-evaluated and the walk goes on where it stopped
          ST(++st_fix;)
          const uint32_t v = __brev(lo) >> 17;
          uint32_t L = LIT_FAST + 1;
#pragma unroll
          for (uint32_t q = LIT_FAST + 1; q < 15; ++q) L += (uint32_t) (v >= h.llim[q]);
          const uint32_t slot = h.lbas[L] + (v >> (15u - L));
          if ((e & 15u) == 0u && v < h.llim[15] && slot < 288u) e = h.lent[slot];
          from_entry();
          hop();
        }
        if (nn != 0xFFFFFFFFu)
        {
          sel |= 1ull << k;
          k = (nn + 1u) & 127u;
        }
        ST(++st_rounds;)
        if (sel)
        {
          const uint32_t start = (uint32_t) (sel >> lane) & 1u;
          const uint32_t is_lit = start & (other ^ 1u), is_match = start & is_len;
          const uint32_t olen = is_lit | (is_match ? len : 0u);
          const uint32_t incl = wave_incl_scan(olen), opos = o + incl - olen;
          const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
          if (o + total > out_cap || __builtin_amdgcn_ballot_w64(is_match & (uint32_t) (dist > opos))) return ~0u;
          if (is_lit) gout[opos] = (uint8_t) (e >> 8);
          const unsigned long long mm = __builtin_amdgcn_ballot_w64(is_match != 0u);
          if (is_match) tok[ntok + lanes_below(mm)] = (unsigned long long) opos | ((unsigned long long) len << 16) | ((unsigned long long) dist << 32);
          ntok += (uint32_t) __popcll(mm);
          o += total;
          dc.bitpos += k;
          if (nn != 0xFFFFFFFFu && ((nn + 1u) >> 8)) break;  // end of block
          continue;
        }
        // ---- the symbol at the head did not decode from the direct tables (a long code): scalar path
        ST(++st_slow;)
        uint32_t w = dc.peek();
        uint32_t se = uni(h.lfast[w & ((1u << LIT_FAST) - 1u)]);
        if ((se & 15u) == 0)
        {
          ST(++st_ll;)
          se = walk_code(w, h.lent, h.lcount);
          if (se == 0) return ~0u;
        }
        const uint32_t scl = se & 15u;
        if ((se & 0xFFF0u) == E_BAD) return ~0u;
        if ((se & 16u) == 0)
        {
          if (o >= out_cap) return ~0u;
          if (lane == 0) gout[o] = (uint8_t) (se >> 8);
          ++o;
          dc.bitpos += scl;
        }
        else if ((se & 0xFFF0u) == E_EOB)
        {
          dc.bitpos += scl;
          if (dc.bitpos > dc.end_bit) return ~0u;
          break;
        }
        else
        {
          const uint32_t sx = (se >> 5) & 7u;
          const uint32_t slen = 3u + ((se >> 8) << sx) + ((w >> scl) & ((1u << sx) - 1u));
          dc.bitpos += scl + sx;
          w = dc.peek();
          uint32_t sd = uni(h.dfast[w & ((1u << DIST_FAST) - 1u)]);
          if ((sd & 15u) == 0)
          {
            ST(++st_dlong;)
            sd = walk_code(w, h.dent, h.dcount);
            if (sd == 0) return ~0u;
          }
          if (sd & (1u << 10)) return ~0u;
          const uint32_t sdl = sd & 15u, sdx = (sd >> 4) & 15u;
          const uint32_t sdist = 1u + (((sd >> 8) & 3u) << sdx) + ((w >> sdl) & ((1u << sdx) - 1u));
          dc.bitpos += sdl + sdx;
          if (sdist > o || o + slen > out_cap) return ~0u;
          if (lane == 0) tok[ntok] = (unsigned long long) o | ((unsigned long long) slen << 16) | ((unsigned long long) sdist << 32);
          ++ntok;
          o += slen;
        }
        if (dc.bitpos > dc.end_bit) return ~0u;
      }
    }
    else
      return ~0u;
    if (last)
    {
#ifdef BGZF_STATS
      if (lane == 0)
      {
        atomicAdd(&g_bgzf_stats[1], (unsigned long long) st_rounds);
        atomicAdd(&g_bgzf_stats[2], (unsigned long long) ntok);
        atomicAdd(&g_bgzf_stats[5], (unsigned long long) st_slow);
        atomicAdd(&g_bgzf_stats[6], (unsigned long long) st_dyn);
        atomicMax(&g_lane_stats[7], (unsigned long long) (wall_clock64() - st_t0));
        atomicAdd(&g_bgzf_stats[0], (unsigned long long) st_ll | ((unsigned long long) st_fix << 32));
        atomicAdd(&g_bgzf_stats[4], (unsigned long long) st_dlong);
        atomicAdd(&g_bgzf_stats[3], (unsigned long long) st_tb);
        atomicAdd(&g_bgzf_stats[7], (unsigned long long) (wall_clock64() - st_t0));
      }
#endif
      return o;
    }
  }
  return ~0u;
}

// blk[first + blockIdx.x]: literals into out, match tokens into slab block blockIdx.x, their number into ntok[blockIdx.x]
template <int LANES>
__device__ __forceinline__ void decode_block(const uint8_t *__restrict__ file, const BgzfBlock *__restrict__ blk, uint32_t first, uint32_t nblk, uint8_t *__restrict__ out,
                                             unsigned long long *__restrict__ slab, uint32_t *__restrict__ ntok, uint32_t *__restrict__ err, HuffLds &s_h)
{
  if (blockIdx.x >= nblk) return;
  const BgzfBlock b = blk[first + blockIdx.x];
  uint32_t got = ~0u, nt = 0;
  if (b.isize == 0)
    got = 0;
  else if (b.isize <= 65536u)
    got = decode_wave<LANES>(file, b.in_off, b.clen, out + b.out_off, slab + (size_t) blockIdx.x * BGZF_TOKENS_PER_BLOCK, b.isize, s_h, nt);
  const bool bad = got != b.isize;
  if (threadIdx.x == 0)
  {
    ntok[blockIdx.x] = bad ? 0u : nt;
    if (bad) atomicOr(err, 1u);
  }
}
// one launch has the GPU to itself (a whole file at once): as many waves per CU as the LDS holds (22)
template <int LANES>
__global__ __launch_bounds__(64) void k_bgzf_decode(const uint8_t *__restrict__ file, const BgzfBlock *__restrict__ blk, uint32_t first, uint32_t nblk, uint8_t *__restrict__ out,
                                                    unsigned long long *__restrict__ slab, uint32_t *__restrict__ ntok, uint32_t *__restrict__ err)
{
  __shared__ HuffLds s_h;
  decode_block<LANES>(file, blk, first, nblk, out, slab, ntok, err, s_h);
}
// the chunks of the streaming feed: decoders of several chunks and the resolve blocks of others share the CUs.  Four waves per
// SIMD (the register allocation is rounded up to enforce it) leave 45 KiB of LDS and a wave slot per SIMD with 96 registers
// on every CU - room for a resolve block; without the cap the decoders fill the LDS (22 x 7 KiB) and the resolve blocks of
// a chunk wait until the other chunks' decoders have drained.
template <int LANES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, LANES ? BGZF_SHARED_WAVES : 4))) void k_bgzf_decode_shared(const uint8_t *__restrict__ file, const BgzfBlock *__restrict__ blk, uint32_t first,
                                                                                                      uint32_t nblk, uint8_t *__restrict__ out, unsigned long long *__restrict__ slab,
                                                                                                      uint32_t *__restrict__ ntok, uint32_t *__restrict__ err)
{
  __shared__ HuffLds s_h;
  decode_block<LANES>(file, blk, first, nblk, out, slab, ntok, err, s_h);
}

// LZ77 resolution of one block by pointer jumping in LDS (file comment), a window of RES_WIN output positions at a time.
// A match reaches back at most 32 KiB, but nothing here depends on that: a parent in front of the window's first position is
// final already (the earlier windows wrote it), so it ends a chain like a literal does.  32 KiB of LDS and 4 waves per block
// instead of 128 KiB and 16: a resolve block finds room on a CU that also holds decode waves of the neighbouring chunks
// (with the whole block's parents in LDS it had to wait until a CU was nearly empty, i.e. for the other chunks' decoders).
// Thread t owns window positions t + RESOLVE_THREADS i.  Tokens are in output order (the decoder writes them so).
__global__ __launch_bounds__(RESOLVE_THREADS) void k_bgzf_resolve(const BgzfBlock *__restrict__ blk, uint32_t first, uint32_t nblk, uint8_t *__restrict__ out,
                                                                  const unsigned long long *__restrict__ slab, const uint32_t *__restrict__ ntok)
{
  extern __shared__ __attribute__((aligned(16))) uint16_t s_par[];  // RES_WIN parents (absolute positions)
  __shared__ uint32_t s_hi;
  if (blockIdx.x >= nblk) return;
  const BgzfBlock b = blk[first + blockIdx.x];
  const uint32_t n = b.isize <= 65536u ? b.isize : 0u, t = threadIdx.x, nt = ntok[blockIdx.x];
  if (n == 0 || nt == 0) return;  // literals only: the decoder wrote every byte
  const unsigned long long *tok = slab + (size_t) blockIdx.x * BGZF_TOKENS_PER_BLOCK;
  uint8_t *o = out + b.out_off;
  const bool packed = (b.out_off & 3u) != 0;
  constexpr uint32_t PER = RES_WIN / RESOLVE_THREADS;
  static_assert(PER <= 64 && RES_WIN % (8 * RESOLVE_THREADS) == 0, "one todo bit per owned position");
  uint32_t tok_lo = 0;  // first token that may reach into the window
  for (uint32_t base = 0; base < n; base += RES_WIN)
  {
    const uint32_t end = base + RES_WIN < n ? base + RES_WIN : n;
    for (uint32_t q = 8 * t; q < RES_WIN; q += 8 * RESOLVE_THREADS)
    {
      const uint32_t a = ((base + q) & 0xFFFFu) | (((base + q + 1) & 0xFFFFu) << 16);
      *reinterpret_cast<uint4 *>(s_par + q) = make_uint4(a, a + 0x00020002u, a + 0x00040004u, a + 0x00060006u);
    }
    if (t == 0) s_hi = nt;
    __syncthreads();
    {
      uint32_t i = tok_lo + t;
      for (; i < nt; i += RESOLVE_THREADS)
      {
        const unsigned long long k = tok[i];
        const uint32_t pos = (uint32_t) k & 0xFFFFu, len = (uint32_t) (k >> 16) & 0xFFFFu, dist = (uint32_t) (k >> 32);
        if (pos >= end) break;
        const uint32_t j0 = pos < base ? base - pos : 0u, j1 = pos + len > end ? end - pos : len;
        for (uint32_t j = j0; j < j1; ++j) s_par[pos + j - base] = (uint16_t) (pos + j - dist);
      }
      if (i < nt) atomicMin(&s_hi, i);  // the first token of a later window, as far as this thread saw
    }
    __syncthreads();
    unsigned long long todo = 0;
    for (uint32_t i = 0; i < PER; ++i)
    {
      const uint32_t q = t + RESOLVE_THREADS * i;
      if (base + q < end && s_par[q] != ((base + q) & 0xFFFFu)) todo |= 1ull << i;
    }
    while (todo)
    {
      unsigned long long m = todo;
      while (m)
      {
        const uint32_t i = (uint32_t) __ffsll((long long) m) - 1u;
        m &= m - 1ull;
        const uint32_t q = t + RESOLVE_THREADS * i;
        const uint32_t sp = s_par[q];
        if (sp < base)
          todo &= ~(1ull << i);  // final since an earlier window
        else
        {
          const uint32_t r = s_par[sp - base];
          if (r == sp)
            todo &= ~(1ull << i);  // a literal
          else
            s_par[q] = (uint16_t) r;
        }
      }
    }
    __syncthreads();
    if (packed)
    {
      // packed output (blocks at arbitrary offsets): only the copied bytes are written, one at a time
      for (uint32_t q = t; base + q < end; q += RESOLVE_THREADS)
      {
        const uint32_t sp = s_par[q];
        if (sp != base + q) o[base + q] = o[sp];
      }
    }
    else
    {
      // four output bytes per thread and store
      const uint32_t w4 = (end - base) & ~3u;
      for (uint32_t q = 4 * t; q < w4; q += 4 * RESOLVE_THREADS)
      {
        const uint2 sp = *reinterpret_cast<const uint2 *>(s_par + q);
        const uint32_t v = (uint32_t) o[sp.x & 0xFFFFu] | ((uint32_t) o[sp.x >> 16] << 8) | ((uint32_t) o[sp.y & 0xFFFFu] << 16) | ((uint32_t) o[sp.y >> 16] << 24);
        *reinterpret_cast<uint32_t *>(o + base + q) = v;
      }
      if (t < end - base - w4)
      {
        const uint32_t q = w4 + t, sp = s_par[q];
        if (sp != base + q) o[base + q] = o[sp];
      }
    }
    const uint32_t hi = s_hi;
    tok_lo = hi ? hi - 1 : 0;  // the token in front of the next window's first may reach into it
    // the next window reads what this one wrote (same CU, same L1) and reuses the LDS
    __threadfence_block();
    __syncthreads();
  }
}
}  // namespace

uint32_t bgzf_scratch_blocks(uint32_t nblk) { return nblk < BGZF_BATCH_BLOCKS ? nblk : BGZF_BATCH_BLOCKS; }

void launch_bgzf_inflate(const uint8_t *file_dev, const BgzfBlock *blk_dev, uint32_t nblk, uint8_t *out_dev, void *scratch_dev, uint32_t *err_dev, hipStream_t st, bool shared_gpu)
{
  if (nblk == 0) return;
  static bool attr_set = false;
  if (!attr_set)
  {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bgzf_resolve), hipFuncAttributeMaxDynamicSharedMemorySize, RES_WIN * 2));
    attr_set = true;
  }
  static const bool rounds = getenv("BK_BGZF_ROUNDS") != nullptr;  // the decoder that walks 64 bit positions per round (comparison)
  const uint32_t cap = bgzf_scratch_blocks(nblk);
  unsigned long long *slab = static_cast<unsigned long long *>(scratch_dev);
  uint32_t *ntok = reinterpret_cast<uint32_t *>(slab + (size_t) cap * BGZF_TOKENS_PER_BLOCK);
  for (uint32_t first = 0; first < nblk; first += BGZF_BATCH_BLOCKS)
  {
    const uint32_t nb = nblk - first < BGZF_BATCH_BLOCKS ? nblk - first : BGZF_BATCH_BLOCKS;
    if (shared_gpu)
      hipLaunchKernelGGL(rounds ? k_bgzf_decode_shared<0> : k_bgzf_decode_shared<1>, dim3(nb), dim3(64), 0, st, file_dev, blk_dev, first, nb, out_dev, slab, ntok, err_dev);
    else
      hipLaunchKernelGGL(rounds ? k_bgzf_decode<0> : k_bgzf_decode<1>, dim3(nb), dim3(64), 0, st, file_dev, blk_dev, first, nb, out_dev, slab, ntok, err_dev);
    hipLaunchKernelGGL(k_bgzf_resolve, dim3(nb), dim3(RESOLVE_THREADS), RES_WIN * 2, st, blk_dev, first, nb, out_dev, slab, ntok);
  }
#ifdef BGZF_STATS
  if (bk_debug("bgzf"))
  {
    unsigned long long h[8], z[8] = {0};
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bgzf_stats), 64));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_bgzf_stats), z, 64));
    unsigned long long l[8];
    HIP_CHECK(hipMemcpyFromSymbol(l, HIP_SYMBOL(g_lane_stats), 64));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stats), z, 64));
    unsigned long long l2[4];
    HIP_CHECK(hipMemcpyFromSymbol(l2, HIP_SYMBOL(g_lane_stats2), 32));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stats2), z, 32));
    fprintf(stderr, "[bgzf lanes] steps: %.1f %% with a code longer than the direct table (%.2f lanes of those steps), %.1f of 64 lanes walking on average\n", 100.0 * l2[0] / (l[1] ? l[1] : 1), (double) l2[1] / (l2[0] ? l2[0] : 1),
            (double) l2[2] / (l[1] ? l[1] : 1));
    fprintf(stderr, "[bgzf lanes] per block: %.1f us in counting walks, %.1f us in writing walks, %.1f us in table builds, %.1f us in all (slowest block %.1f us); %.1f steps, %.2f windows, %.2f passes; %llu long passes after a window's first, %llu windows left to the rounds\n",
            l[0] * 0.01 / nblk, l[2] * 0.01 / nblk, h[3] * 0.01 / nblk, h[7] * 0.01 / nblk, l[7] * 0.01, (double) l[1] / nblk, (double) l[3] / nblk, (double) l[5] / nblk, l[4], l[6]);
    fprintf(stderr, "[bgzf] %u blocks: %llu long-code fix-ups, %llu rounds, %llu symbols on the scalar path (%llu long literal/length codes, %llu long distance codes), %llu matches, %llu Huffman blocks; per block %.1f us in table builds of %.1f us\n", nblk, h[0] >> 32, h[1],
            h[5], h[0] & 0xFFFFFFFFull, h[4], h[2], h[6], h[3] * 0.01 / nblk, h[7] * 0.01 / nblk);
  }
#endif
}
