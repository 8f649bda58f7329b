// Streaming pass over the columnar record table (one lane per record, coalesced column reads):
//   A1  insert-size sums            BreakID.cc:1929-1941
//   A2  discordant-candidate filter BreakID.cc:1419-1420  (wave-ballot compaction)
//   A12 per-read SA/CIGAR evidence  BreakID.cc:895-1016 + CigarRoller.cc / Cigar.cc semantics
// plus the bit-exact replay of the reference's order-dependent sd accumulation (BreakID.cc:1942-1946).
// HBM-bound integer/byte work: no MFMA.  gfx950 only.
#include "bk_common.h"
#include "prims.h"
#include "stream.h"
#include <cstdlib>

namespace
{
// qhash, mtid, mpos, qcheck of record i: one 32-byte row when the table has the side layout (a 16-byte and a 4-byte load out of
// one sector), else four columns
__global__ __launch_bounds__(256) void k_make_side(const uint64_t *__restrict__ qhash, const int32_t *__restrict__ mtid, const int32_t *__restrict__ mpos,
                                                   const uint32_t *__restrict__ qcheck, uint64_t n, bk_side *__restrict__ side)
{
  const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bk_side r;
  r.qhash = qhash[i];
  r.mtid = mtid[i];
  r.mpos = mpos[i];
  r.qcheck = qcheck ? qcheck[i] : 0u;
  r.reserved[0] = r.reserved[1] = r.reserved[2] = 0;
  side[i] = r;
}
__device__ __forceinline__ void cand_fields(const StreamArgs &a, uint64_t i, uint64_t &qhash, int32_t &mtid, int32_t &mpos, uint32_t &qcheck)
{
  if (a.side)
  {
    const uint4 v = *reinterpret_cast<const uint4 *>(a.side + i);
    qhash = (uint64_t) v.x | ((uint64_t) v.y << 32);
    mtid = (int32_t) v.z;
    mpos = (int32_t) v.w;
    qcheck = a.side[i].qcheck;
  }
  else
  {
    qhash = a.qhash[i];
    mtid = a.mtid[i];
    mpos = a.mpos[i];
    qcheck = a.qcheck ? a.qcheck[i] : 0u;
  }
}

// ---- CIGAR roll-up (CigarRoller.cc:26-46 operator+=, :67-116 Add, Cigar.cc:80-144) ------------------
// classes: 1 match, 3 insert, 4 del, 5 skip, 6 softClip, 7 hardClip, 8 pad (Cigar.h:65-76)
struct Roll
{
  int n_ops;          // number of rolled operations
  int last;           // class of the last rolled op
  int op0, op1;       // classes of the first two rolled ops
  uint32_t c0, c1;    // their counts
  int reflen, nmatch; // getExpectedReferenceBaseCount / getNumMatches (int arithmetic like the reference)
  int begin, tail;    // getNumBeginClips / getNumEndClips
  bool lead;
  __device__ void init()
  {
    n_ops = 0; last = 0; op0 = op1 = 0; c0 = c1 = 0; reflen = nmatch = 0; begin = tail = 0; lead = true;
  }
  __device__ void add(int cls, uint32_t count)
  {
    if (count == 0) return;
    if (n_ops == 0 || last != cls)
    {
      if (n_ops == 0) { op0 = cls; c0 = count; }
      else if (n_ops == 1) { op1 = cls; c1 = count; }
      ++n_ops;
      last = cls;
    }
    else
    {
      if (n_ops == 1) c0 += count;
      else if (n_ops == 2) c1 += count;
    }
    bool clip = (cls == 6 || cls == 7);
    if (cls == 1 || cls == 4 || cls == 5) reflen += (int) count;
    if (cls == 1) nmatch += (int) count;
    if (clip)
    {
      if (lead) begin += (int) count;
      tail += (int) count;
    }
    else
    {
      lead = false;
      tail = 0;
    }
  }
  __device__ void add_code(int c, int count)  // Add(char operation, int count): BAM codes and letters
  {
    int cls;
    switch (c)
    {
    case 0: case 'M': cls = 1; break;
    case 1: case 'I': cls = 3; break;
    case 2: case 'D': cls = 4; break;
    case 3: case 'N': cls = 5; break;
    case 4: case 'S': cls = 6; break;
    case 5: case 'H': cls = 7; break;
    case 6: case 'P': cls = 8; break;
    case 7: case '=': cls = 1; break;
    case 8: case 'X': cls = 1; break;
    default: return;  // reference logs an error and ignores the op
    }
    add(cls, (uint32_t) count);
  }
  // rolled string matches ([0-9]+[MS]){2}
  __device__ bool two_op_ms() const { return n_ops == 2 && (op0 == 1 || op0 == 6) && (op1 == 1 || op1 == 6); }
};

__device__ __forceinline__ void roll_bam(Roll &r, const uint32_t *__restrict__ w, uint32_t n)
{
  r.init();
  for (uint32_t k = 0; k < n; ++k)
  {
    uint32_t v = w[k];
    r.add_code((int) (v & 15u), (int) (v >> 4));
  }
}

// CigarRoller::Add(const char*) (:119-136): digits -> strtol, any other byte -> Add(byte, count)
__device__ void roll_text(Roll &r, const uint8_t *__restrict__ s, uint32_t len)
{
  r.init();
  int count = 0;
  uint32_t i = 0;
  while (i < len && s[i])
  {
    uint8_t ch = s[i];
    if (ch >= '0' && ch <= '9')
    {
      long long v = 0;
      while (i < len && s[i] >= '0' && s[i] <= '9')
      {
        v = v * 10 + (s[i] - '0');
        if (v > 0x7fffffffffffffLL) v = 0x7fffffffffffffLL;
        ++i;
      }
      count = (int) v;
    }
    else
    {
      r.add_code((int) ch, count);
      ++i;
    }
  }
}

// raw text fully matches ([0-9]+[MS]){2}  (CigarRoller.cc:326)
__device__ bool text_two_op_ms(const uint8_t *__restrict__ s, uint32_t len)
{
  uint32_t i = 0;
#pragma unroll 1
  for (int k = 0; k < 2; ++k)
  {
    uint32_t d = i;
    while (i < len && s[i] >= '0' && s[i] <= '9') ++i;
    if (i == d) return false;
    if (i >= len || (s[i] != 'M' && s[i] != 'S')) return false;
    ++i;
  }
  return i == len;
}

__device__ __forceinline__ uint64_t fnv_bytes(const uint8_t *__restrict__ s, uint32_t len)
{
  uint64_t h = 0xCBF29CE484222325ull;
  for (uint32_t i = 0; i < len; ++i)
  {
    h ^= s[i];
    h *= 0x100000001B3ull;
  }
  return h;
}
__device__ __forceinline__ void fnv_push(uint64_t &h, uint8_t c)
{
  h ^= c;
  h *= 0x100000001B3ull;
}
__device__ void fnv_push_dec(uint64_t &h, uint32_t v)  // decimal text of a count (sprintf "%d")
{
  char buf[10];
  int n = 0;
  do
  {
    buf[n++] = (char) ('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) fnv_push(h, (uint8_t) buf[--n]);
}
__device__ __forceinline__ char cls_char(int cls) { return cls == 1 ? 'M' : 'S'; }

__device__ int32_t intern_name(const NameTableDev &nt, const uint8_t *__restrict__ s, uint32_t len)
{
  uint64_t h = fnv_bytes(s, len);
  if (h == 0) h = 1;
  uint32_t slot = (uint32_t) h & nt.mask;
  for (uint32_t probe = 0; probe <= nt.mask; ++probe)
  {
    uint64_t e = nt.hash[slot];
    if (e == 0) break;
    if (e == h) return nt.id[slot];
    slot = (slot + 1) & nt.mask;
  }
  return (int32_t) (0x40000000u | (uint32_t) (fnv_bytes(s, len) & 0x3FFFFFFFu));
}

// bam_endpos (sam.c:344-350) with BAM_CIGAR_TYPE 0x3C1A7
__device__ __forceinline__ int32_t cigar_reflen_hts(const uint32_t *__restrict__ w, uint32_t n)
{
  int l = 0;
  for (uint32_t k = 0; k < n; ++k)
  {
    uint32_t v = w[k], op = v & 15u;
    if ((0x3C1A7u >> (op << 1)) & 2u) l += (int) (v >> 4);
  }
  return l;
}

// CigarRoller::is_complementary_cigar (CigarRoller.cc:323-346): c1 = this read's rolled cigar, c2 = the SA entry's raw text
// (rolled into `sac`); both must match ([0-9]+[MS]){2} - c1 as its rolled string, c2 as written
__device__ __forceinline__ bool is_complementary(const Roll &c1, const Roll &sac, const uint8_t *__restrict__ c2, uint32_t c2len, int e)
{
  if (!c1.two_op_ms() || !text_two_op_ms(c2, c2len)) return false;
  int c1_m = c1.nmatch, c2_m = sac.nmatch;
  int c1_s = c1.begin + c1.tail, c2_s = sac.tail + sac.begin;
  return (c1_m <= c2_s + e && c1_m >= c2_s - e) && (c1_m + c1_s == c2_m + c2_s);
}

// Evidence tuple of one record (BreakID.cc:895-1016).  false = record yields no tuple.
__device__ bool record_split(const StreamArgs &a, uint64_t i, uint16_t flag, int32_t tid, int32_t pos, uint32_t c0, uint32_t c1,
                             int32_t endpos, bk_split &t)
{
  uint32_t a0 = a.aux_off[i], a1 = a.aux_off[i + 1];
  if (a1 <= a0) return false;                       // sa_tag == ""
  if ((flag & 0x400) || !(flag & 1)) return false;  // :898
  const uint8_t *blob = a.aux + a0;
  uint32_t blen = a1 - a0;
  // aux blob: [OC '\t'] SA
  uint32_t oc_len = 0, sa_beg = 0;
  for (uint32_t k = 0; k < blen; ++k)
  {
    uint8_t ch = blob[k];
    if (ch == ',') break;
    if (ch == '\t')
    {
      oc_len = k;
      sa_beg = k + 1;
      break;
    }
  }
  const uint8_t *sa = blob + sa_beg;
  uint32_t sa_len = blen - sa_beg;
  if (sa_len == 0) return false;
  // split_string(sa, ","), empty tokens dropped (util_bed.cc:194-222): tokens 0,1,3
  uint32_t tb[4], te[4];
  int nt = 0;
  {
    uint32_t st = 0;
    while (nt < 4)
    {
      uint32_t e = st;
      while (e < sa_len && sa[e] != ',') ++e;
      if (e > st)
      {
        tb[nt] = st;
        te[nt] = e;
        ++nt;
      }
      if (e >= sa_len) break;
      st = e + 1;
    }
  }
  if (nt < 4) return false;
  const uint8_t *c2 = sa + tb[3];
  uint32_t c2len = te[3] - tb[3];
  Roll own, sac, tmp;
  roll_bam(own, a.cigar + c0, c1 - c0);
  roll_text(sac, c2, c2len);
  if (oc_len)
    roll_text(tmp, blob, oc_len);
  else
    tmp = own;
  if (!is_complementary(tmp, sac, c2, c2len, 10)) return false;  // is_complementary_cigar(sa[3], 10), BreakID.cc:915
  t.rec = a.rec_base + i;
  t.reserved = 0;
  t.tid = tid;
  t.pos = pos;
  t.endpos = endpos;
  t.qhash = a.side ? a.side[i].qhash : a.qhash[i];
  bool secondary = (flag & 0x100) != 0;
  uint32_t flags = secondary ? 1u : 0u;
  // stoi(sa[1])
  uint32_t sa_start;
  {
    const uint8_t *p = sa + tb[1];
    uint32_t l = te[1] - tb[1], k = 0;
    while (k < l && (p[k] == ' ' || (p[k] >= 9 && p[k] <= 13))) ++k;
    bool neg = false;
    if (k < l && (p[k] == '+' || p[k] == '-'))
    {
      neg = p[k] == '-';
      ++k;
    }
    long long v = 0;
    while (k < l && p[k] >= '0' && p[k] <= '9')
    {
      v = v * 10 + (p[k] - '0');
      if (v > 0x7fffffffLL) v = 0x7fffffffLL;
      ++k;
    }
    sa_start = (uint32_t) (int) (neg ? -v : v);
  }
  uint32_t sa_end = sa_start + (uint32_t) sac.reflen - 1u;  // CigarRoller.cc:316-321
  long long a_start = (long long) pos + 1;                    // BamAlignment.cc:172-175
  long long a_end = (own.reflen == 0 ? (long long) pos : (long long) pos + own.reflen - 1) + 1;
  int32_t own_chr = (tid >= 0 && tid < a.names.n_targets) ? a.names.own_id[tid] : a.names.empty_id;
  int32_t sa_chr = intern_name(a.names, sa + tb[0], te[0] - tb[0]);
  // a contig name that is neither in the header nor chr1..22,X,Y travels as a 30-bit hash id; its second hash rides in `reserved`
  // so that two such names are only equal when 62 bits agree (the reference compares the strings, BreakID.cc:627-637)
  if (sa_chr & 0x40000000) t.reserved = qname_check32(sa + tb[0], te[0] - tb[0]);
  uint32_t own_end_val, own_bp = 0, sa_bp = 0;
  uint64_t own_cig;
  bool poison = false;
  if (oc_len)
  {
    own_cig = cigar_text_code(blob, oc_len);
    own_end_val = (uint32_t) a_start + (uint32_t) tmp.reflen - 1u;
  }
  else
  {
    // text of the rolled BAM cigar (getCigarString): exactly two ops here
    uint8_t txt[24];
    uint32_t tl = 0;
    for (int k = 0; k < 2; ++k)
    {
      uint32_t v = k ? own.c1 : own.c0;
      uint8_t tmp[10];
      int nd = 0;
      do
      {
        tmp[nd++] = (uint8_t) ('0' + v % 10);
        v /= 10;
      } while (v);
      while (nd) txt[tl++] = tmp[--nd];
      txt[tl++] = (uint8_t) cls_char(k ? own.op1 : own.op0);
    }
    own_cig = cigar_text_code(txt, tl);
    own_end_val = (uint32_t) a_end;
  }
  if (tmp.begin != 0)
    own_bp = (uint32_t) a_start;
  else if (tmp.tail != 0)
    own_bp = (uint32_t) a_end;
  else
    poison = true;
  if (sac.begin != 0)
    sa_bp = sa_start;
  else if (sac.tail != 0)
    sa_bp = sa_end;
  else
    poison = true;
  uint64_t sa_cig = cigar_text_code(c2, c2len);
  if (!secondary)
  {
    t.prim_chr = own_chr; t.prim_start = (uint32_t) a_start; t.prim_end = own_end_val; t.prim_cigar = own_cig; t.prim_bp = own_bp;
    t.sec_chr = sa_chr; t.sec_start = sa_start; t.sec_end = sa_end; t.sec_cigar = sa_cig; t.sec_bp = sa_bp;
  }
  else
  {
    t.prim_chr = sa_chr; t.prim_start = sa_start; t.prim_end = sa_end; t.prim_cigar = sa_cig; t.prim_bp = sa_bp;
    t.sec_chr = own_chr; t.sec_start = (uint32_t) a_start; t.sec_end = own_end_val; t.sec_cigar = own_cig; t.sec_bp = own_bp;
  }
  if (poison) flags |= 2u;
  t.flags = flags;
  t.qcheck = a.side ? a.side[i].qcheck : (a.qcheck ? a.qcheck[i] : 0u);
  return true;
}

// ---------------------------------------------------------------------------------------------------
// Streaming kernel.  Each lane owns 4 consecutive records per iteration (16-byte column loads), so one
// wave-instruction moves 1 KiB per 32-bit column.  Outputs that are rare per record (5 % candidates,
// ~1 % SA-bearing records) are staged in LDS bins and flushed with one global atomic per ~half-full bin:
// a single device counter saturates near 90 returning atomics/us (MI355X_MICROARCH.md, dequeue row),
// which a per-wave append would hit at this record rate.
constexpr int ST_V = STREAM_V;      // records per lane per iteration (stream.h)
constexpr int ST_CAND_CAP = 640;    // LDS candidate bin (25 KiB of 40-byte candidates: 5 workgroups per CU still fit)
constexpr int ST_SA_CAP = 1024;     // LDS bin of SA-bearing record indices (4 KiB)

struct StreamAcc
{
  unsigned long long isum, icnt;
  double isq;
  unsigned int span, vmax, unsorted;
};

__device__ __forceinline__ void stream_record(const StreamArgs &a, uint64_t i, uint16_t flag, uint8_t mapq, int32_t tid, int32_t pos, int32_t isize, uint32_t c0, uint32_t c1,
                                              uint32_t a0, uint32_t a1, uint32_t ptid, int32_t ppos, bool has_prev, StreamAcc &acc, bool &cand, bool &sa)
{
  // A1 (:1932): PAIRED && PROPER && !(UNMAP|SECONDARY|QCFAIL|DUP)
  if ((flag & 1) && (flag & 2) && !(flag & (0x4 | 0x100 | 0x200 | 0x400)))
  {
    int v = isize < 0 ? -isize : isize;
    acc.isum += (unsigned long long) (long long) v;
    acc.icnt += 1;
    acc.isq += (double) v * (double) v;
    acc.vmax = max(acc.vmax, (unsigned int) v);
  }
  // bam_endpos span bound for the later region selects (reads the CIGAR words: 4 B per op, algorithmic)
  int sp = 1;
  if (!(flag & 4) && c1 > c0) sp = cigar_reflen_hts(a.cigar + c0, c1 - c0);
  acc.span = max(acc.span, (unsigned int) (sp < 0 ? 0 : sp));
  if (has_prev && (ptid > (uint32_t) tid || (ptid == (uint32_t) tid && ppos > pos))) acc.unsorted = 1;
  // A2 (:1419-1420): mapq >= q && !DUP && !SECONDARY && PAIRED && !PROPER_PAIR
  cand = ((int) mapq >= a.mapq_min) && !(flag & 0x400) && !(flag & 0x100) && (flag & 1) && !(flag & 2);
  // A12 gate (:898): SA tag present, !DUP, PAIRED -> evaluated by k_split_records
  sa = (a1 > a0) && !(flag & 0x400) && (flag & 1);
}

__global__ __launch_bounds__(256) void k_stream(StreamArgs a)
{
  __shared__ __attribute__((aligned(16))) Cand s_cand[ST_CAND_CAP];
  __shared__ uint32_t s_sa[ST_SA_CAP];
  __shared__ unsigned int s_ncand, s_nsa;
  __shared__ unsigned long long s_gbase[2];
  __shared__ unsigned long long s_sum[4], s_n[4];
  __shared__ double s_sq[4];
  __shared__ unsigned int s_span[4], s_vmax[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0)
  {
    s_ncand = 0;
    s_nsa = 0;
  }
  __syncthreads();
  StreamAcc acc;
  acc.isum = acc.icnt = 0;
  acc.isq = 0.0;
  acc.span = 1;
  acc.vmax = 0;
  acc.unsorted = 0;
  const uint64_t nq = a.n / ST_V;  // full quads
  const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
  // this launch covers the quads [a.q_begin, nq): the feed hands the table over in pieces (bk_bam_decode_device_ctx)
  const uint64_t q_round = a.q_begin + ((nq - a.q_begin + stride - 1) / stride) * stride;  // whole blocks stay in the loop (barriers inside)
  unsigned int iter_no = 0;
  for (uint64_t q = a.q_begin + (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; q < q_round; q += stride)
  {
    const bool live = q < nq;
    const uint64_t i0 = q * ST_V;
    bool cand[ST_V], sa[ST_V];
    int32_t tidv[ST_V], posv[ST_V];
    uint16_t flv[ST_V];
    uint8_t mqv[ST_V];
#pragma unroll
    for (int k = 0; k < ST_V; ++k) cand[k] = sa[k] = false;
    if (live)
    {
      // ST_V consecutive records of every column: 32-bit columns as ST_V / 4 16-byte loads, flag as one 16-byte load (8 records) or one
      // 8-byte load (4), mapq as one 8- / 4-byte load
      int32_t izv[ST_V];
      uint32_t cov[ST_V + 1], aov[ST_V + 1];
#pragma unroll
      for (int h = 0; h < ST_V / 4; ++h)
      {
        const int4 t4 = reinterpret_cast<const int4 *>(a.tid)[q * (ST_V / 4) + h];
        const int4 p4 = reinterpret_cast<const int4 *>(a.pos)[q * (ST_V / 4) + h];
        const int4 z4 = reinterpret_cast<const int4 *>(a.isize)[q * (ST_V / 4) + h];
        const uint4 co = reinterpret_cast<const uint4 *>(a.cigar_off)[q * (ST_V / 4) + h];
        const uint4 ao = reinterpret_cast<const uint4 *>(a.aux_off)[q * (ST_V / 4) + h];
        tidv[4 * h + 0] = t4.x; tidv[4 * h + 1] = t4.y; tidv[4 * h + 2] = t4.z; tidv[4 * h + 3] = t4.w;
        posv[4 * h + 0] = p4.x; posv[4 * h + 1] = p4.y; posv[4 * h + 2] = p4.z; posv[4 * h + 3] = p4.w;
        izv[4 * h + 0] = z4.x; izv[4 * h + 1] = z4.y; izv[4 * h + 2] = z4.z; izv[4 * h + 3] = z4.w;
        cov[4 * h + 0] = co.x; cov[4 * h + 1] = co.y; cov[4 * h + 2] = co.z; cov[4 * h + 3] = co.w;
        aov[4 * h + 0] = ao.x; aov[4 * h + 1] = ao.y; aov[4 * h + 2] = ao.z; aov[4 * h + 3] = ao.w;
      }
      if (ST_V == 8)
      {
        const uint4 f8 = reinterpret_cast<const uint4 *>(a.flag)[q];
        const uint2 m8 = reinterpret_cast<const uint2 *>(a.mapq)[q];
        const uint32_t fw[4] = {f8.x, f8.y, f8.z, f8.w}, mw[2] = {m8.x, m8.y};
#pragma unroll
        for (int k = 0; k < ST_V; ++k)
        {
          flv[k] = (uint16_t) (fw[k >> 1] >> (16 * (k & 1)));
          mqv[k] = (uint8_t) (mw[k >> 2] >> (8 * (k & 3)));
        }
      }
      else
      {
        const ushort4 f4 = reinterpret_cast<const ushort4 *>(a.flag)[q];
        const uchar4 m4 = reinterpret_cast<const uchar4 *>(a.mapq)[q];
        flv[0] = f4.x; flv[1] = f4.y; flv[2] = f4.z; flv[3] = f4.w;
        mqv[0] = m4.x; mqv[1] = m4.y; mqv[2] = m4.z; mqv[3] = m4.w;
      }
      // the neighbours' values (offset that ends the group, record in front of it) sit in the neighbouring lanes' registers:
      // only the lanes at the edge of the wave (or of the table) load them
      uint32_t co_n = (uint32_t) __shfl_down((int) cov[0], 1, 64), ao_n = (uint32_t) __shfl_down((int) aov[0], 1, 64);
      uint32_t ptid = (uint32_t) __shfl_up(tidv[ST_V - 1], 1, 64);
      int32_t ppos = __shfl_up(posv[ST_V - 1], 1, 64);
      if (lane == 63 || q + 1 >= nq)
      {
        co_n = a.cigar_off[i0 + ST_V];
        ao_n = a.aux_off[i0 + ST_V];
      }
      if (lane == 0)
      {
        ptid = 0;
        ppos = 0;
        if (i0 > 0)
        {
          ptid = (uint32_t) a.tid[i0 - 1];
          ppos = a.pos[i0 - 1];
        }
      }
      cov[ST_V] = co_n;
      aov[ST_V] = ao_n;
#pragma unroll
      for (int k = 0; k < ST_V; ++k)
      {
        stream_record(a, i0 + k, flv[k], mqv[k], tidv[k], posv[k], izv[k], cov[k], cov[k + 1], aov[k], aov[k + 1], k ? (uint32_t) tidv[k - 1] : ptid,
                      k ? posv[k - 1] : ppos, k ? true : (i0 > 0), acc, cand[k], sa[k]);
      }
    }
    // ---- stage candidates: wave prefix over the per-lane counts, one LDS atomic per wave ----
    {
      unsigned int c = 0;
#pragma unroll
      for (int k = 0; k < ST_V; ++k) c += cand[k] ? 1u : 0u;
      unsigned int inc = c;
      for (int d = 1; d < 64; d <<= 1)
      {
        unsigned int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
      }
      unsigned int total = __shfl(inc, 63, 64);
      if (total)
      {
        unsigned int base = 0;
        if (lane == 63) base = atomicAdd(&s_ncand, total);
        base = __shfl(base, 63, 64);
        unsigned int slot = base + inc - c;
#pragma unroll
        for (int k = 0; k < ST_V; ++k)
        {
          if (!cand[k]) continue;
          const uint64_t i = i0 + k;
          Cand cd;
          cand_fields(a, i, cd.qhash, cd.mtid, cd.mpos, cd.qcheck);
          cd.rec = a.rec_base + i;
          cd.tid = tidv[k];
          cd.pos = posv[k];
          cd.flag = flv[k];
          cd.mapq = mqv[k];
          cd.pad = 0;
          if (slot < ST_CAND_CAP)
            s_cand[slot] = cd;
          else
          {
            // bin overflow (denser than ~75 % candidates in this block iteration): direct append
            unsigned long long g = atomicAdd(&a.counters->n_cand, 1ull);
            if (g < a.cand_cap) a.cand[g] = cd;
          }
          ++slot;
        }
      }
    }
    // ---- stage SA-bearing record indices ----
    {
      unsigned int c = 0;
#pragma unroll
      for (int k = 0; k < ST_V; ++k) c += sa[k] ? 1u : 0u;
      uint64_t any = __ballot(c != 0);
      if (any)
      {
        unsigned int inc = c;
        for (int d = 1; d < 64; d <<= 1)
        {
          unsigned int o = __shfl_up(inc, d, 64);
          if (lane >= d) inc += o;
        }
        unsigned int total = __shfl(inc, 63, 64);
        unsigned int base = 0;
        if (lane == 63) base = atomicAdd(&s_nsa, total);
        base = __shfl(base, 63, 64);
        unsigned int slot = base + inc - c;
#pragma unroll
        for (int k = 0; k < ST_V; ++k)
        {
          if (!sa[k]) continue;
          if (slot < ST_SA_CAP)
            s_sa[slot] = (uint32_t) (i0 + k);
          else
          {
            unsigned long long g = atomicAdd(&a.counters->n_sa, 1ull);
            if (g < a.sa_cap) a.sa_list[g] = (uint32_t) (i0 + k);
          }
          ++slot;
        }
      }
    }
    // ---- flush bins that are at least half full (or on the last iteration); looked at every other iteration ----
    const bool last_iter = q + stride >= q_round;  // uniform per block: all lanes share the iteration index
    ++iter_no;
    if (!(last_iter || (iter_no & 3u) == 0)) continue;
    __syncthreads();
    const unsigned int nc = min(s_ncand, (unsigned int) ST_CAND_CAP), ns = min(s_nsa, (unsigned int) ST_SA_CAP);
    const bool flush_c = nc && (nc >= ST_CAND_CAP / 2 || last_iter), flush_s = ns && (ns >= ST_SA_CAP / 2 || last_iter);
    // every wave must take the flush decision from the same counts: nobody adds to the bins before all have read them
    __syncthreads();
    if (flush_c || flush_s)
    {
      if (threadIdx.x == 0)
      {
        if (flush_c) s_gbase[0] = atomicAdd(&a.counters->n_cand, (unsigned long long) nc);
        if (flush_s) s_gbase[1] = atomicAdd(&a.counters->n_sa, (unsigned long long) ns);
      }
      __syncthreads();
      if (flush_c)
      {
        const unsigned long long g = s_gbase[0];
        // 40-byte records as five 8-byte words: contiguous 8 B stores across the block
        const uint2 *src = reinterpret_cast<const uint2 *>(s_cand);
        uint2 *dst = reinterpret_cast<uint2 *>(a.cand);
        for (unsigned int h = threadIdx.x; h < nc * 5; h += 256)
          if (g + h / 5 < a.cand_cap) dst[g * 5 + h] = src[h];
      }
      if (flush_s)
      {
        const unsigned long long g = s_gbase[1];
        for (unsigned int h = threadIdx.x; h < ns; h += 256)
          if (g + h < a.sa_cap) a.sa_list[g + h] = s_sa[h];
      }
      __syncthreads();
      if (threadIdx.x == 0)
      {
        if (flush_c) s_ncand = 0;
        if (flush_s) s_nsa = 0;
      }
      __syncthreads();
    }
  }
  // ---- tail records (n % ST_V) : block 0, first lanes, straight to global ----
  if (blockIdx.x == 0 && threadIdx.x < (a.n - nq * ST_V))  // (a piece that is not the last one ends on a quad: no tail)
  {
    const uint64_t i = nq * ST_V + threadIdx.x;
    bool cand, sa;
    const uint16_t flag = a.flag[i];
    const uint8_t mapq = a.mapq[i];
    const int32_t tid = a.tid[i], pos = a.pos[i];
    uint32_t ptid = 0;
    int32_t ppos = 0;
    if (i > 0)
    {
      ptid = (uint32_t) a.tid[i - 1];
      ppos = a.pos[i - 1];
    }
    stream_record(a, i, flag, mapq, tid, pos, a.isize[i], a.cigar_off[i], a.cigar_off[i + 1], a.aux_off[i], a.aux_off[i + 1], ptid, ppos, i > 0, acc, cand, sa);
    if (cand)
    {
      Cand cd;
      cand_fields(a, i, cd.qhash, cd.mtid, cd.mpos, cd.qcheck);
      cd.rec = a.rec_base + i; cd.tid = tid; cd.pos = pos;
      cd.flag = flag; cd.mapq = mapq; cd.pad = 0;
      unsigned long long g = atomicAdd(&a.counters->n_cand, 1ull);
      if (g < a.cand_cap) a.cand[g] = cd;
    }
    if (sa)
    {
      unsigned long long g = atomicAdd(&a.counters->n_sa, 1ull);
      if (g < a.sa_cap) a.sa_list[g] = (uint32_t) i;
    }
  }
  // ---- block reduction of the scalar accumulators ----
  unsigned long long isum = acc.isum, icnt = acc.icnt;
  double isq = acc.isq;
  unsigned int span = acc.span, vmax = acc.vmax, unsorted = acc.unsorted;
  for (int d = 32; d; d >>= 1)
  {
    isum += __shfl_down(isum, d, 64);
    icnt += __shfl_down(icnt, d, 64);
    isq += __shfl_down(isq, d, 64);
    span = max(span, (unsigned int) __shfl_down((int) span, d, 64));
    vmax = max(vmax, (unsigned int) __shfl_down((int) vmax, d, 64));
    unsorted |= (unsigned int) __shfl_down((int) unsorted, d, 64);
  }
  if (lane == 0)
  {
    s_sum[w] = isum;
    s_n[w] = icnt;
    s_sq[w] = isq;
    s_span[w] = span;
    s_vmax[w] = vmax;
    if (unsorted) atomicOr(&a.counters->unsorted, 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    unsigned long long ts = 0, tn = 0;
    double tq = 0;
    unsigned int sp = 1, vm = 0;
    for (int k = 0; k < 4; ++k)
    {
      ts += s_sum[k];
      tn += s_n[k];
      tq += s_sq[k];
      sp = max(sp, s_span[k]);
      vm = max(vm, s_vmax[k]);
    }
    if (tn)
    {
      atomicAdd(&a.counters->isize_sum, ts);
      atomicAdd(&a.counters->isize_n, tn);
      atomicAdd(&a.sd->sumsq, tq);
      atomicMax(&a.sd->vmax, vm);
    }
    atomicMax(&a.counters->max_span, sp);
  }
}

// rare path: one lane per SA-bearing record (the list k_stream compacted), full evidence evaluation
__global__ __launch_bounds__(256) void k_split_records(StreamArgs a, unsigned long long n_sa)
{
  unsigned long long j = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_sa) return;
  const uint64_t i = a.sa_list[j];
  const uint16_t flag = a.flag[i];
  const int32_t tid = a.tid[i], pos = a.pos[i];
  const uint32_t c0 = a.cigar_off[i], c1 = a.cigar_off[i + 1];
  int32_t endpos;
  if (!(flag & 4) && c1 > c0)
    endpos = pos + cigar_reflen_hts(a.cigar + c0, c1 - c0);
  else
    endpos = pos + 1;
  bk_split t;
  if (record_split(a, i, flag, tid, pos, c0, c1, endpos, t))
  {
    unsigned long long slot = atomicAdd(&a.counters->n_split, 1ull);
    if (slot < a.split_cap) a.split[slot] = t;
  }
}

// test hook (bk_debug_cigar): the CIGAR model on rows of (c1 as text or BAM words, c2 text, e); out = n_ops, begin clips, end clips,
// reference length, matches, complementary - the quantities ref_units prints for the reference's CigarRoller
__global__ void k_debug_cigar(const uint8_t *__restrict__ kind, const uint32_t *__restrict__ c1_off, const uint8_t *__restrict__ c1, const uint32_t *__restrict__ c2_off,
                              const uint8_t *__restrict__ c2, const int32_t *__restrict__ e, uint32_t n, int32_t *__restrict__ out)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Roll r, sac;
  const uint32_t a0 = c1_off[i], a1 = c1_off[i + 1], b0 = c2_off[i], b1 = c2_off[i + 1];
  if (kind[i])
    roll_bam(r, reinterpret_cast<const uint32_t *>(c1 + a0), (a1 - a0) / 4);
  else
    roll_text(r, c1 + a0, a1 - a0);
  roll_text(sac, c2 + b0, b1 - b0);
  int32_t *o = out + 6 * (size_t) i;
  o[0] = r.n_ops; o[1] = r.begin; o[2] = r.tail; o[3] = r.reflen; o[4] = r.nmatch;
  o[5] = is_complementary(r, sac, c2 + b0, b1 - b0, e[i]) ? 1 : 0;
}

// ---- bit-exact sd replay ----------------------------------------------------------------------------
// The reference accumulates  long T += (v-mean)^2  with a truncation after every add (:1942-1946).
// For T < 2^51 one step adds floor(d) unless d lies within half an ulp(T+d) below an integer, where the
// double add rounds up to the next integer ("exception").  floor(d) sums are order independent; only the
// rare exceptions need the running T, so they are compacted in record order with their exact prefix
// sum and replayed by one wave.
constexpr int SD_ITEMS = 8;
constexpr int SD_TILE = 256 * SD_ITEMS;

__device__ __forceinline__ bool sd_elem(const uint16_t *__restrict__ flag, const int32_t *__restrict__ isize, uint64_t i, uint64_t n, double mean,
                                        double thr, double &d, unsigned long long &fd, bool &exc)
{
  d = 0;
  fd = 0;
  exc = false;
  if (i >= n) return false;
  uint16_t f = flag[i];
  if (!((f & 1) && (f & 2) && !(f & (0x4 | 0x100 | 0x200 | 0x400)))) return false;
  int v = isize[i];
  v = v < 0 ? -v : v;
  double a = __dsub_rn((double) v, mean);
  d = __dmul_rn(a, a);
  double fl = floor(d);
  fd = (unsigned long long) fl;
  double gap = __dsub_rn(__dadd_rn(fl, 1.0), d);  // distance to the next integer (exact for d < 2^52)
  exc = gap <= thr;
  return true;
}

// one record's term from values already loaded (sd_elem without the loads)
__device__ __forceinline__ void sd_term(uint32_t f, int32_t v, double mean, double thr, unsigned long long &L, unsigned long long &E)
{
  if (!((f & 1) && (f & 2) && !(f & (0x4 | 0x100 | 0x200 | 0x400)))) return;
  v = v < 0 ? -v : v;
  const double a = __dsub_rn((double) v, mean);
  const double d = __dmul_rn(a, a);
  const double fl = floor(d);
  L += (unsigned long long) fl;
  const double gap = __dsub_rn(__dadd_rn(fl, 1.0), d);
  E += gap <= thr ? 1ull : 0ull;
}

// totals of a tile of SD_TILE records (the tiling of k_sd_emit): the sums do not depend on the order inside the tile, so a
// thread takes SD_ITEMS consecutive records with three 16-byte loads
__global__ __launch_bounds__(256) void k_sd_count(const uint16_t *__restrict__ flag, const int32_t *__restrict__ isize, uint64_t n, double mean, double thr,
                                                  unsigned long long *__restrict__ blockL, unsigned long long *__restrict__ blockE)
{
  static_assert(SD_ITEMS == 8, "eight records per thread: one uint4 of flags, two int4 of insert sizes");
  __shared__ unsigned long long lds[8];
  unsigned long long L = 0, E = 0;
  const uint64_t i0 = (uint64_t) blockIdx.x * SD_TILE + (uint64_t) threadIdx.x * SD_ITEMS;
  if (i0 + SD_ITEMS <= n)
  {
    const uint4 f8 = *reinterpret_cast<const uint4 *>(flag + i0);
    const int4 za = *reinterpret_cast<const int4 *>(isize + i0), zb = *reinterpret_cast<const int4 *>(isize + i0 + 4);
    sd_term(f8.x & 0xFFFFu, za.x, mean, thr, L, E);
    sd_term(f8.x >> 16, za.y, mean, thr, L, E);
    sd_term(f8.y & 0xFFFFu, za.z, mean, thr, L, E);
    sd_term(f8.y >> 16, za.w, mean, thr, L, E);
    sd_term(f8.z & 0xFFFFu, zb.x, mean, thr, L, E);
    sd_term(f8.z >> 16, zb.y, mean, thr, L, E);
    sd_term(f8.w & 0xFFFFu, zb.z, mean, thr, L, E);
    sd_term(f8.w >> 16, zb.w, mean, thr, L, E);
  }
  else
    for (uint64_t i = i0; i < n && i < i0 + SD_ITEMS; ++i) sd_term(flag[i], isize[i], mean, thr, L, E);
  for (int d = 32; d; d >>= 1)
  {
    L += __shfl_down(L, d, 64);
    E += __shfl_down(E, d, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
  {
    lds[w] = L;
    lds[4 + w] = E;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    blockL[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
    blockE[blockIdx.x] = lds[4] + lds[5] + lds[6] + lds[7];
  }
}

__global__ __launch_bounds__(256) void k_sd_emit(const uint16_t *__restrict__ flag, const int32_t *__restrict__ isize, uint64_t n, double mean, double thr,
                                                 const unsigned long long *__restrict__ baseL, const unsigned long long *__restrict__ baseE,
                                                 SdException *__restrict__ out)
{
  __shared__ unsigned long long lds[4];
  if (baseE[blockIdx.x + 1] == baseE[blockIdx.x]) return;  // block has no exception
  unsigned long long carryL = baseL[blockIdx.x], carryE = baseE[blockIdx.x];
  uint64_t base = (uint64_t) blockIdx.x * SD_TILE;
#pragma unroll 1
  for (int k = 0; k < SD_ITEMS; ++k)
  {
    double d;
    unsigned long long fd;
    bool exc;
    sd_elem(flag, isize, base + (uint64_t) k * 256 + threadIdx.x, n, mean, thr, d, fd, exc);
    unsigned long long totL, totE;
    unsigned long long exL = prims::block_exclusive_scan(fd, lds, totL);
    unsigned long long exE = prims::block_exclusive_scan(exc ? 1ull : 0ull, lds, totE);
    if (exc)
    {
      SdException e;
      e.l_before = carryL + exL;
      e.d = d;
      out[carryE + exE] = e;
    }
    carryL += totL;
    carryE += totE;
  }
}

// one wave: replay the exceptions in record order
__global__ __launch_bounds__(64) void k_sd_walk(const SdException *__restrict__ ex, unsigned long long n_ex, unsigned long long l_total, SdState *s)
{
  const int lane = threadIdx.x;
  long long corr = 0;
  for (unsigned long long base = 0; base < n_ex; base += 64)
  {
    unsigned long long i = base + lane;
    unsigned long long lb = 0;
    double d = 0;
    if (i < n_ex)
    {
      lb = ex[i].l_before;
      d = ex[i].d;
    }
    int cnt = (int) min((unsigned long long) 64, n_ex - base);
    for (int j = 0; j < cnt; ++j)
    {
      unsigned long long lbj = __shfl(lb, j, 64);
      double dj = __shfl(d, j, 64);
      long long tprev = (long long) lbj + corr;
      long long tnew = (long long) __dadd_rn((double) tprev, dj);  // long += double
      corr += (tnew - tprev) - (long long) floor(dj);
    }
  }
  if (lane == 0) s->t_final = (long long) l_total + corr;
}
}  // namespace

// ---- host side ----------------------------------------------------------------------------------------
void launch_stream(const StreamArgs &a, hipStream_t st)
{
  if (a.n <= (uint64_t) STREAM_V * a.q_begin) return;
  unsigned blocks = cdiv(a.n / STREAM_V - a.q_begin + 1, 256);
  // exactly one resident set of blocks (grid-stride loop inside): a block more than fits leaves a tail that runs at
  // a fraction of the occupancy (measured 4.7 vs 6.1 TB/s at 6 vs 5 blocks per CU, LDS bins ~29 KiB per block)
  static unsigned resident = 0;
  if (!resident)
  {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_stream, 256, 0));
    resident = (unsigned) (per_cu < 1 ? 1 : per_cu) * (unsigned) prop.multiProcessorCount;
  }
  if (blocks > resident) blocks = resident;
  hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, st, a);
}

void launch_split_records(const StreamArgs &a, unsigned long long n_sa, hipStream_t st)
{
  if (n_sa == 0) return;
  hipLaunchKernelGGL(k_split_records, dim3(cdiv(n_sa, 256)), dim3(256), 0, st, a, n_sa);
}

// per-table part: floor sums and the ordered exception list (l_before relative to this table)
void launch_sd_local(const uint16_t *flag, const int32_t *isize, uint64_t n, double mean, double thr, SdBufs &b, hipStream_t st, unsigned long long *l_total,
                     unsigned long long *n_ex_out)
{
  uint32_t nb = cdiv(n, SD_TILE);
  if (nb == 0) nb = 1;
  unsigned long long *blockL = b.blockL.as<unsigned long long>(nb + 1);
  unsigned long long *blockE = b.blockE.as<unsigned long long>(nb + 1);
  hipLaunchKernelGGL(k_sd_count, dim3(nb), dim3(256), 0, st, flag, isize, n, mean, thr, blockL, blockE);
  prims::exclusive_scan<unsigned long long>(blockL, blockL, nb, b.scan_tmp, st);
  prims::exclusive_scan<unsigned long long>(blockE, blockE, nb, b.scan_tmp2, st);
  // exception capacity: read the totals back (tiny sync) so the list can be sized exactly
  unsigned long long host[2] = {0, 0};
  HIP_CHECK(hipMemcpyAsync(&host[0], blockE + nb, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&host[1], blockL + nb, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  SdException *ex = b.exceptions.as<SdException>(host[0] + 1);
  if (host[0]) hipLaunchKernelGGL(k_sd_emit, dim3(nb), dim3(256), 0, st, flag, isize, n, mean, thr, blockL, blockE, ex);
  b.last_exceptions = host[0];
  *n_ex_out = host[0];
  *l_total = host[1];
}

// replay of the (possibly gathered) exception list: l_before must already be global
void launch_sd_walk(const SdException *ex, unsigned long long n_ex, unsigned long long l_total, SdState *sd, hipStream_t st)
{
  hipLaunchKernelGGL(k_sd_walk, dim3(1), dim3(64), 0, st, ex, n_ex, l_total, sd);
}

void launch_sd(const uint16_t *flag, const int32_t *isize, uint64_t n, double mean, double thr, SdState *sd, SdBufs &b, hipStream_t st)
{
  unsigned long long l_total = 0, n_ex = 0;
  launch_sd_local(flag, isize, n, mean, thr, b, st, &l_total, &n_ex);
  launch_sd_walk(b.exceptions.get<SdException>(), n_ex, l_total, sd, st);
}

void debug_cigar(const uint8_t *kind, const uint32_t *c1_off, const uint8_t *c1, const uint32_t *c2_off, const uint8_t *c2, const int32_t *e, uint32_t n, int32_t *out,
                 hipStream_t st)
{
  if (n) hipLaunchKernelGGL(k_debug_cigar, dim3(cdiv(n, 64)), dim3(64), 0, st, kind, c1_off, c1, c2_off, c2, e, n, out);
}

void launch_make_side(const uint64_t *qhash, const int32_t *mtid, const int32_t *mpos, const uint32_t *qcheck, uint64_t n, bk_side *side, hipStream_t st)
{
  if (n) hipLaunchKernelGGL(k_make_side, dim3(cdiv(n, 256)), dim3(256), 0, st, qhash, mtid, mpos, qcheck, n, side);
}
