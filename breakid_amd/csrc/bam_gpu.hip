// GPU feed: BGZF file image -> inflated stream -> columnar record table, all on the device (see bgzf_gpu.hip).
#include "bk_common.h"
#include <functional>
#include "bgzf_gpu.h"
#include "../../include/breakid_hip.h"
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

namespace
{
inline uint32_t rd32h(const uint8_t *p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24); }
inline uint16_t rd16h(const uint8_t *p) { return (uint16_t) (p[0] | (p[1] << 8)); }
}  // namespace

// hops over BGZF block headers (18 bytes each) from file offset `off` until at least max_bytes of the file are covered or
// the file ends; appends to blocks (in_off = absolute file offset of the deflate stream, out_off = total before the block,
// total grows by the block sizes rounded up to `align`); false + why on a malformed file
bool bgzf_scan_range(const uint8_t *file, uint64_t n, uint64_t &off, uint64_t max_bytes, std::vector<BgzfBlock> &blocks, uint64_t &total, std::string &why, uint64_t align = BGZF_OUT_ALIGN)
{
  const uint64_t start = off;
  while (off < n && off - start < max_bytes)
  {
    if (off + 18 > n)
    {
      why = "truncated BGZF header";
      return false;
    }
    const uint8_t *h = file + off;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4))
    {
      why = "not a BGZF block";
      return false;
    }
    const uint16_t xlen = rd16h(h + 10);
    if (off + 12 + (uint64_t) xlen > n)
    {
      why = "truncated BGZF header";
      return false;
    }
    int bsize = -1;
    for (uint64_t k = 0; k + 4 <= xlen;)
    {
      const uint16_t slen = rd16h(h + 12 + k + 2);
      if (h[12 + k] == 66 && h[12 + k + 1] == 67 && slen == 2 && k + 6 <= xlen) bsize = rd16h(h + 12 + k + 4);
      k += 4 + (uint64_t) slen;
    }
    if (bsize < 0 || off + (uint64_t) bsize + 1 > n || (uint64_t) bsize + 1 < 12 + (uint64_t) xlen + 8)
    {
      why = "bad BGZF block size";
      return false;
    }
    BgzfBlock b;
    b.in_off = off + 12 + xlen;
    b.clen = (uint32_t) ((uint64_t) bsize + 1 - (12 + (uint64_t) xlen) - 8);
    b.isize = rd32h(h + bsize + 1 - 4);
    if (b.isize > 65536)
    {
      why = "BGZF block larger than 64 KiB";
      return false;
    }
    b.out_off = total;
    total += (b.isize + align - 1) / align * align;
    blocks.push_back(b);
    off += (uint64_t) bsize + 1;
  }
  return true;
}

// the whole file image
bool bgzf_scan_blocks(const uint8_t *file, uint64_t n, std::vector<BgzfBlock> &blocks, uint64_t &total, std::string &why)
{
  blocks.clear();
  total = 0;
  uint64_t off = 0;
  return bgzf_scan_range(file, n, off, ~0ull, blocks, total, why);
}

// Test / measurement hook: inflates a whole BGZF file image on the GPU and hands the bytes back.
extern "C" int bk_debug_bgzf_inflate(const void *file, uint64_t n, void *out, uint64_t out_cap, uint64_t *out_len, float *kernel_ms, char *err, size_t errlen)
{
  try
  {
    std::vector<BgzfBlock> blocks;
    uint64_t total = 0;
    std::string why;
    if (!file || !out_len) throw bk_error(BK_ERR_ARG, "bk_debug_bgzf_inflate: null argument");
    if (!bgzf_scan_blocks((const uint8_t *) file, n, blocks, total, why)) throw bk_error(BK_ERR_IO, why);
    uint64_t packed = 0;
    for (auto &bb : blocks) packed += bb.isize;
    *out_len = packed;
    if (packed > out_cap) throw bk_error(BK_ERR_ARG, "bk_debug_bgzf_inflate: output buffer too small");
    DevBuf dfile, dblk, dout, derr, dslab;
    uint8_t *f = dfile.as<uint8_t>(n + 8);
    BgzfBlock *b = dblk.as<BgzfBlock>(blocks.size() + 1);
    uint8_t *o = dout.as<uint8_t>(total + 8);
    uint32_t *e = derr.as<uint32_t>(1);
    uint8_t *slab = dslab.as<uint8_t>(bgzf_scratch_bytes((uint32_t) blocks.size()));
    HIP_CHECK(hipMemcpy(f, file, n, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(b, blocks.data(), blocks.size() * sizeof(BgzfBlock), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(e, 0, 4));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, nullptr));
    launch_bgzf_inflate(f, b, (uint32_t) blocks.size(), o, slab, e, nullptr);
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    if (kernel_ms) *kernel_ms = ms;
    uint32_t he = 0;
    HIP_CHECK(hipMemcpy(&he, e, 4, hipMemcpyDeviceToHost));
    if (he) throw bk_error(BK_ERR_IO, "inflate failed");
    if (out && total)
    {
      std::vector<uint8_t> tmp(total);
      HIP_CHECK(hipMemcpy(tmp.data(), o, total, hipMemcpyDeviceToHost));
      uint64_t w = 0;
      for (auto &bb : blocks)
      {
        memcpy((uint8_t *) out + w, tmp.data() + bb.out_off, bb.isize);
        w += bb.isize;
      }
    }
    return BK_OK;
  }
  catch (const bk_error &ex)
  {
    if (err && errlen) snprintf(err, errlen, "%s", ex.what());
    return ex.code;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// BAM records -> columns, on the device.  htslib never lets a record straddle two BGZF blocks (bgzf_flush_try before
// every record), so the blocks of such a file are independent jobs here as well: one lane walks the records of one
// block (the 4-byte length chain is serial only inside a block).  A first pass validates the chain (it must end
// exactly at the end of the block, every record must be well formed) and counts records / CIGAR words / aux bytes per
// block; exclusive scans place the blocks; a second pass writes the columns.  Files whose records do straddle blocks
// (other writers) are reported with BK_ERR_IO "not block aligned" and take the host decoder.
namespace
{
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24); }
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8); }

struct BamCols
{
  int32_t *tid, *pos, *mtid, *mpos, *isize;
  uint16_t *flag;
  uint8_t *mapq;
  uint64_t *qhash;
  uint32_t *qcheck;
  uint32_t *cigar_off, *cigar, *aux_off;
  uint8_t *aux;
};
struct BlockCount
{
  uint32_t n_rec, n_cig, n_aux, bad;
};

// aux walk of one record (sam.c:1267-1279 semantics, as bam_reader.cc): first SA:Z and OC:Z
__device__ __forceinline__ void aux_scan(const uint8_t *r, uint32_t q, uint32_t bs, uint32_t &sa_at, uint32_t &sa_len, uint32_t &oc_at, uint32_t &oc_len)
{
  sa_at = oc_at = 0;
  sa_len = oc_len = 0;
  bool has_sa = false, has_oc = false;
  while (q + 3 <= bs)
  {
    const uint8_t t0 = r[q], t1 = r[q + 1], type = r[q + 2];
    q += 3;
    uint32_t len;
    switch (type)
    {
    case 'A': case 'c': case 'C': len = 1; break;
    case 's': case 'S': len = 2; break;
    case 'i': case 'I': case 'f': len = 4; break;
    case 'd': len = 8; break;
    case 'Z': case 'H':
    {
      uint32_t e = q;
      while (e < bs && r[e]) ++e;
      if (type == 'Z')
      {
        if (!has_sa && t0 == 'S' && t1 == 'A')
        {
          has_sa = true;
          sa_at = q;
          sa_len = e - q;
        }
        if (!has_oc && t0 == 'O' && t1 == 'C')
        {
          has_oc = true;
          oc_at = q;
          oc_len = e - q;
        }
      }
      len = e - q + 1;
      break;
    }
    case 'B':
    {
      if (q + 5 > bs) return;
      const uint8_t sub = r[q];
      const uint32_t cnt = ld32(r + q + 1);
      const uint32_t es = (sub == 'c' || sub == 'C') ? 1u : (sub == 's' || sub == 'S') ? 2u : 4u;
      const unsigned long long l64 = 5ull + (unsigned long long) cnt * es;
      if (l64 > bs) return;
      len = (uint32_t) l64;
      break;
    }
    default:
      return;
    }
    q += len;
  }
}

__device__ __forceinline__ uint64_t qname_hash_dev(const uint8_t *name, uint32_t l_name)
{
  uint64_t h = 0xCBF29CE484222325ull;
  for (uint32_t i = 0; i < l_name; ++i)
  {
    const uint8_t c = name[i];
    if (!c) break;  // bam_get_qname is a C string
    h ^= c;
    h *= 0x100000001B3ull;
  }
  h ^= h >> 30;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 27;
  h *= 0x94D049BB133111EBull;
  h ^= h >> 31;
  return h;
}

// where a record lies in a chunk's inflated data and what the chunk holds in front of it (k_bam_blocks<2> writes one per record,
// k_bam_emit reads them: the chain of record lengths is serial inside a block, writing the columns is not)
struct RecIndex
{
  uint64_t at;        // offset of the record's length word in the chunk's inflated data
  uint32_t cig, aux;  // CIGAR words / aux bytes of the chunk's records before this one
};

// columns of one record: r = the record behind its length word bs, ri / ci / ai = its row and the running CIGAR / aux offsets
__device__ __forceinline__ void emit_record(const BamCols &c, uint64_t ri, uint64_t ci, uint64_t ai, const uint8_t *r, int32_t tid, uint32_t l_name, uint32_t n_cig, uint32_t sa_at,
                                            uint32_t sa_len, uint32_t oc_at, uint32_t oc_len)
{
  c.tid[ri] = tid;
  c.pos[ri] = (int32_t) ld32(r + 4);
  c.mapq[ri] = r[9];
  c.flag[ri] = (uint16_t) ld16(r + 14);
  c.mtid[ri] = (int32_t) ld32(r + 20);
  c.mpos[ri] = (int32_t) ld32(r + 24);
  c.isize[ri] = (int32_t) ld32(r + 28);
  c.qhash[ri] = qname_hash_dev(r + 32, l_name);
  {
    uint32_t ql = 0;  // bam_get_qname is a C string
    while (ql < l_name && r[32 + ql]) ++ql;
    c.qcheck[ri] = qname_check32(r + 32, ql);
  }
  c.cigar_off[ri] = (uint32_t) ci;
  c.aux_off[ri] = (uint32_t) ai;
  const uint8_t *cg = r + 32 + l_name;
  for (uint32_t k = 0; k < n_cig; ++k) c.cigar[ci + k] = ld32(cg + 4 * k);
  if (sa_len)
  {
    uint64_t w = ai;
    if (oc_len)
    {
      for (uint32_t k = 0; k < oc_len; ++k) c.aux[w++] = r[oc_at + k];
      c.aux[w++] = '\t';
    }
    for (uint32_t k = 0; k < sa_len; ++k) c.aux[w++] = r[sa_at + k];
  }
}

// MODE 0: validate + count; MODE 1: write the columns (bases from the scans); MODE 2: write one RecIndex per record (chunk-relative
// bases from the scans) for k_bam_emit.
// Aligned files (entry == nullptr): every block starts with a record and no record leaves its block.  Packed mode
// (entry != nullptr, the inflated stream is contiguous, `total` bytes): the lane of block b takes the records that START
// in b, from entry[b] (the block's size = none), wherever they end; next_abs[b] = stream offset where its walk stopped.
// allow_tail: a record that does not fit the stream is not an error but the end of the walk (the next chunk starts with it).
template <int MODE> __global__ __launch_bounds__(64) void k_bam_blocks(const uint8_t *__restrict__ data, const BgzfBlock *__restrict__ blk, uint32_t nblk, uint32_t first_blk,
                                                                         uint32_t first_off, int32_t n_ref, BlockCount *__restrict__ cnt, const uint64_t *__restrict__ rec_base,
                                                                         const uint64_t *__restrict__ cig_base, const uint64_t *__restrict__ aux_base, uint64_t rec0, uint64_t cig0, uint64_t aux0, BamCols c,
                                                                         const uint32_t *__restrict__ entry = nullptr, uint64_t total = 0, uint64_t *__restrict__ next_abs = nullptr, int allow_tail = 0,
                                                                         RecIndex *__restrict__ rindex = nullptr)
{
  constexpr bool EMIT = MODE == 1, INDEX = MODE == 2;
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  BlockCount bc = {0, 0, 0, 0};
  uint64_t stop = 0;
  if (b >= first_blk)
  {
    const BgzfBlock bb = blk[b];
    const uint8_t *d = data + bb.out_off;
    // bytes a record that starts in this block may use
    const uint64_t room = entry ? total - bb.out_off : (uint64_t) bb.isize;
    uint64_t p = entry ? entry[b] : (b == first_blk ? first_off : 0u);
    uint64_t ri = MODE ? rec0 + rec_base[b] : 0, ci = MODE ? cig0 + cig_base[b] : 0, ai = MODE ? aux0 + aux_base[b] : 0;  // rec0.. = totals of the chunks before this one
    while (p < bb.isize)
    {
      if (p + 36 > room)
      {
        bc.bad = allow_tail ? 0 : 1;  // allow_tail: the stream goes on in the next chunk, this record is its first
        break;
      }
      const uint32_t bs = ld32(d + p);
      const uint8_t *r = d + p + 4;
      if (bs < 32 || p + 4 + bs > room)
      {
        bc.bad = (allow_tail && bs >= 32) ? 0 : 1;  // a record that continues in the next block (aligned mode), or garbage
        break;
      }
      const uint32_t l_name = r[8], n_cig = ld16(r + 12), l_seq = ld32(r + 16);
      const unsigned long long need = 32ull + l_name + 4ull * n_cig + ((unsigned long long) l_seq + 1) / 2 + l_seq;
      const int32_t tid = (int32_t) ld32(r);
      if (need > bs || tid < -1 || tid >= n_ref)
      {
        bc.bad = 1;
        break;
      }
      const uint32_t q = (uint32_t) need;
      uint32_t sa_at, sa_len, oc_at, oc_len;
      aux_scan(r, q, bs, sa_at, sa_len, oc_at, oc_len);
      const uint32_t blob = sa_len ? sa_len + (oc_len ? oc_len + 1 : 0) : 0;
      if (EMIT) emit_record(c, ri, ci, ai, r, tid, l_name, n_cig, sa_at, sa_len, oc_at, oc_len);
      if (INDEX)
      {
        RecIndex x;
        x.at = bb.out_off + p;
        x.cig = (uint32_t) ci;
        x.aux = (uint32_t) ai;
        rindex[ri] = x;
      }
      ++ri;
      ci += n_cig;
      ai += blob;
      ++bc.n_rec;
      bc.n_cig += n_cig;
      bc.n_aux += blob;
      p += 4 + bs;
    }
    stop = bb.out_off + p;
  }
  if (MODE == 0)
  {
    cnt[b] = bc;
    if (next_abs) next_abs[b] = stop;
  }
}

// one lane per record of a chunk (the records were validated by the count pass and located by the index pass)
__global__ __launch_bounds__(256) void k_bam_emit(const uint8_t *__restrict__ data, const RecIndex *__restrict__ rindex, uint64_t n, uint64_t rec0, uint64_t cig0, uint64_t aux0, BamCols c)
{
  const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const RecIndex x = rindex[i];
  const uint32_t bs = ld32(data + x.at);
  const uint8_t *r = data + x.at + 4;
  const uint32_t l_name = r[8], n_cig = ld16(r + 12), l_seq = ld32(r + 16);
  const uint32_t q = 32u + l_name + 4u * n_cig + (l_seq + 1) / 2 + l_seq;
  uint32_t sa_at, sa_len, oc_at, oc_len;
  aux_scan(r, q, bs, sa_at, sa_len, oc_at, oc_len);
  emit_record(c, rec0 + i, cig0 + x.cig, aux0 + x.aux, r, (int32_t) ld32(r), l_name, n_cig, sa_at, sa_len, oc_at, oc_len);
}

// the columns of the n records a walk counted: once more over the length chains for where the records lie, then one lane per
// record (the first form - the walk itself writes the columns, one lane per block - took 2.4-2.9 ms per chunk against 0.4)
static void launch_emit(const uint8_t *dd, const BgzfBlock *db, uint32_t nb, uint32_t first_blk, uint32_t first_off, int32_t n_ref, BlockCount *dc, const uint64_t *nr, const uint64_t *nc,
                        const uint64_t *na, uint64_t rec0, uint64_t cig0, uint64_t aux0, const BamCols &c, const uint32_t *entry, uint64_t total, int allow_tail, uint64_t n, DevBuf &index,
                        hipStream_t st)
{
  if (n == 0)
  {
    hipLaunchKernelGGL(k_bam_blocks<1>, dim3(cdiv(nb, 64)), dim3(64), 0, st, dd, db, nb, first_blk, first_off, n_ref, dc, nr, nc, na, rec0, cig0, aux0, c, entry, total, nullptr, allow_tail, nullptr);
    return;
  }
  RecIndex *rx = index.as<RecIndex>(n);
  BamCols none = {};
  hipLaunchKernelGGL(k_bam_blocks<2>, dim3(cdiv(nb, 64)), dim3(64), 0, st, dd, db, nb, first_blk, first_off, n_ref, dc, nr, nc, na, 0ull, 0ull, 0ull, none, entry, total, nullptr, allow_tail, rx);
  hipLaunchKernelGGL(k_bam_emit, dim3((unsigned) cdiv(n, 256)), dim3(256), 0, st, dd, (const RecIndex *) rx, n, rec0, cig0, aux0, c);
}

__global__ void k_bam_count_split(const BlockCount *__restrict__ cnt, uint32_t nblk, uint64_t *__restrict__ nr, uint64_t *__restrict__ nc, uint64_t *__restrict__ na, uint32_t *__restrict__ err)
{
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  nr[b] = cnt[b].n_rec;
  nc[b] = cnt[b].n_cig;
  na[b] = cnt[b].n_aux;
  if (cnt[b].bad) atomicOr(err, 2u);
}

// ---- record boundaries of a packed stream (files whose records straddle BGZF blocks: htsjdk / Picard / GATK writers)
// Is there a well-formed record at stream offset `at`?  (the checks of the count kernel + the ones a random offset fails:
// mate reference, positions, NUL-terminated name)
__device__ __forceinline__ bool plausible_record(const uint8_t *d, uint64_t at, uint64_t total, int32_t n_ref, uint64_t &next)
{
  if (at + 36 > total) return false;
  const uint32_t bs = ld32(d + at);
  if (bs < 32 || at + 4 + bs > total) return false;
  const uint8_t *r = d + at + 4;
  const int32_t tid = (int32_t) ld32(r), pos = (int32_t) ld32(r + 4), mtid = (int32_t) ld32(r + 20), mpos = (int32_t) ld32(r + 24);
  const uint32_t l_name = r[8], n_cig = ld16(r + 12), l_seq = ld32(r + 16);
  if (tid < -1 || tid >= n_ref || mtid < -1 || mtid >= n_ref || pos < -1 || mpos < -1 || l_name < 1) return false;
  const unsigned long long need = 32ull + l_name + 4ull * n_cig + ((unsigned long long) l_seq + 1) / 2 + l_seq;
  if (need > bs || r[32 + l_name - 1] != 0) return false;
  next = at + 4 + bs;
  return true;
}

// One wave per block: entry[b] = the first offset in the block from which GUESS_CHAIN records in a row are well formed
// (or the chain reaches the end of the stream); isize = no record starts here.  A guess, made exact by k_bam_verify.
constexpr int GUESS_CHAIN = 4;
__global__ __launch_bounds__(64) void k_bam_guess(const uint8_t *__restrict__ data, const BgzfBlock *__restrict__ blk, uint32_t nblk, uint32_t first_blk, uint32_t first_off,
                                                  int32_t n_ref, uint64_t total, uint32_t *__restrict__ entry)
{
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  if (b >= nblk) return;
  const BgzfBlock bb = blk[b];
  if (b < first_blk || (b == first_blk && first_off != 0xFFFFFFFFu))
  {
    if (lane == 0) entry[b] = b == first_blk ? first_off : bb.isize;  // the header's blocks; the first record is known
    return;
  }
  // (first_off = 0xFFFFFFFF: a part of a file - its first block has to guess like the others, decode_packed_part)
  uint32_t found = bb.isize;
  for (uint32_t base = 0; base < bb.isize; base += 64)
  {
    const uint32_t s = base + lane;
    bool ok = s < bb.isize;
    uint64_t at = bb.out_off + s;
    for (int k = 0; ok && k < GUESS_CHAIN && at < total; ++k) ok = plausible_record(data, at, total, n_ref, at);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
    if (m)
    {
      found = base + (uint32_t) __ffsll((long long) m) - 1u;
      break;
    }
  }
  if (lane == 0) entry[b] = found;
}

// The guesses are exact iff the walks chain: the walk of every block that has an entry stops exactly at the entry of the
// next block that has one (the first entry is the known first record, the last walk stops at the end of the stream).
// By induction from the first record every entry then IS a record boundary.
__global__ void k_bam_verify(const BgzfBlock *__restrict__ blk, uint32_t nblk, uint32_t first_blk, const uint32_t *__restrict__ entry, const uint64_t *__restrict__ next_abs, uint64_t total,
                             uint32_t *__restrict__ err, int allow_tail = 0, unsigned long long *__restrict__ tail_out = nullptr)
{
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk || b < first_blk || entry[b] >= blk[b].isize) return;
  uint32_t nb = b + 1;
  while (nb < nblk && entry[nb] >= blk[nb].isize) ++nb;
  if (nb < nblk)
  {
    if (next_abs[b] != blk[nb].out_off + entry[nb]) atomicOr(err, 4u);
  }
  else if (allow_tail)
  {
    // the last walk of a chunk stops at the record that continues in the next chunk (or at the end of the stream)
    if (next_abs[b] > total) atomicOr(err, 4u);
    if (tail_out) *tail_out = next_abs[b];
  }
  else
  {
    if (next_abs[b] != total) atomicOr(err, 4u);
    if (tail_out) *tail_out = next_abs[b];
  }
}
}  // namespace

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <cerrno>
#include <zlib.h>
#include <cstdio>
#include <ctime>
#include "prims.h"

struct bk_bam_dev
{
  std::vector<std::string> names;
  std::vector<const char *> name_ptrs;
  std::vector<uint32_t> lens;
  DevBuf tid, pos, mtid, mpos, isize, flag, mapq, qhash, qcheck, cigar_off, cigar, aux_off, aux;
};

namespace
{
struct MappedFile
{
  const uint8_t *p = nullptr;
  size_t n = 0;
  int fd = -1;
  explicit MappedFile(const char *path)
  {
    fd = open(path, O_RDONLY);
    if (fd < 0) throw bk_error(BK_ERR_IO, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0)
    {
      close(fd);
      throw bk_error(BK_ERR_IO, "cannot stat the BAM file");
    }
    n = (size_t) st.st_size;
    if (n)
    {
      void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED)
      {
        close(fd);
        throw bk_error(BK_ERR_IO, "cannot map the BAM file");
      }
      p = (const uint8_t *) m;
      (void) madvise(m, n, MADV_SEQUENTIAL);
    }
  }
  ~MappedFile()
  {
    if (p) munmap((void *) p, n);
    if (fd >= 0) close(fd);
  }
  MappedFile(const MappedFile &) = delete;
  MappedFile &operator=(const MappedFile &) = delete;
  const uint8_t *data() const { return p; }
  size_t size() const { return n; }
  int descriptor() const { return fd; }
};
double now_s2()
{
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}
bool host_inflate_block(const uint8_t *file, const BgzfBlock &b, std::vector<uint8_t> &out)
{
  const size_t base = out.size();
  out.resize(base + b.isize);
  if (!b.isize) return true;
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (inflateInit2(&zs, -15) != Z_OK) return false;
  zs.next_in = const_cast<Bytef *>(file + b.in_off);
  zs.avail_in = b.clen;
  zs.next_out = out.data() + base;
  zs.avail_out = b.isize;
  const int rc = inflate(&zs, Z_FINISH);
  inflateEnd(&zs);
  return rc == Z_STREAM_END;
}
}  // namespace

namespace
{
// Host staging of the mapped file: from the fourth chunk on, a producer thread copies fixed ranges of the file
// ([k C, (k + 1) C + 64 KiB + 64): every block that starts inside chunk k ends inside the range) into page-locked buffers
// with a few helper threads, ahead of the thread that drives the GPU.  The page faults of the mapping and the copy run on
// those threads; the H2D copy out of a staging buffer is a plain asynchronous DMA.  The first three chunks are copied
// straight from the mapping (the runtime pins and unpins the pages of every piece on the calling thread: 16-19 GB/s)
// because a fresh staging buffer costs more than that (first touch + registration, ~10 ms per 64 MiB); the buffers stay
// with the process for the next file.
struct StageCache
{
  std::mutex mu;
  std::vector<std::pair<uint8_t *, uint64_t>> idle;  // registered buffers and their sizes
  uint8_t *take(uint64_t bytes)
  {
    {
      std::lock_guard<std::mutex> g(mu);
      for (size_t i = 0; i < idle.size(); ++i)
        if (idle[i].second == bytes)
        {
          uint8_t *p = idle[i].first;
          idle.erase(idle.begin() + (long) i);
          return p;
        }
    }
    uint8_t *p = (uint8_t *) aligned_alloc(2u << 20, (bytes + (2u << 20) - 1) / (2u << 20) * (2u << 20));
    if (!p) return nullptr;
    (void) madvise(p, bytes, MADV_HUGEPAGE);  // fewer first-touch faults and a shorter registration where the kernel allows it
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess)
    {
      free(p);
      return nullptr;
    }
    return p;
  }
  void give(uint8_t *p, uint64_t bytes)
  {
    std::lock_guard<std::mutex> g(mu);
    idle.emplace_back(p, bytes);
  }
  void release_all()
  {
    std::lock_guard<std::mutex> g(mu);
    for (auto &e : idle)
    {
      (void) hipHostUnregister(e.first);
      free(e.first);
    }
    idle.clear();
  }
  int idle_of(uint64_t bytes)
  {
    std::lock_guard<std::mutex> g(mu);
    int n = 0;
    for (auto &e : idle) n += e.second == bytes;
    return n;
  }
};
StageCache &stage_cache()
{
  static StageCache *c = new StageCache();  // never destroyed: the buffers go with the process
  return *c;
}

// The threads that copy a chunk of the file into its staging buffer: made once per process (a process with many mappings pays
// ~0.1 ms per thread it starts; a file of 31 chunks used to start 250 of them), handed pieces of 1 MiB.
struct CopyPool
{
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> th;
  std::function<void(uint64_t)> fn;
  uint64_t n_items = 0, next = 0, done = 0, gen = 0;
  void loop()
  {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> g(mu);
    for (;;)
    {
      cv_work.wait(g, [&] { return gen != seen; });
      seen = gen;
      while (next < n_items)
      {
        const uint64_t i = next++;
        auto f = fn;
        g.unlock();
        f(i);
        g.lock();
        if (++done == n_items) cv_done.notify_all();
      }
    }
  }
  // f(0) .. f(n - 1) on `threads` threads (the caller is one of them); returns when all have run
  std::mutex one_batch;  // feeds of several devices in one process take turns (the copies are bound by memory bandwidth anyway)
  void run(uint64_t n, int threads, const std::function<void(uint64_t)> &f)
  {
    std::lock_guard<std::mutex> turn(one_batch);
    std::unique_lock<std::mutex> g(mu);
    while ((int) th.size() + 1 < threads)
    {
      th.emplace_back([this] { loop(); });
      th.back().detach();  // the pool lives as long as the process
    }
    fn = f;
    n_items = n;
    next = 0;
    done = 0;
    ++gen;
    cv_work.notify_all();
    while (next < n_items)
    {
      const uint64_t i = next++;
      g.unlock();
      f(i);
      g.lock();
      ++done;
    }
    cv_done.wait(g, [&] { return done == n_items; });
    n_items = 0;
  }
};
CopyPool &copy_pool()
{
  static CopyPool *c = new CopyPool();
  return *c;
}

struct StagePool
{
  static constexpr int NB = 3;
  static constexpr uint64_t FIRST = 3;  // a process's first file: chunks before this one are copied from the mapping
  static constexpr uint64_t SLACK = 65536 + 64;
  struct Buf
  {
    uint8_t *p = nullptr;
    int state = 0;  // 0 free, 1 filled (chunk), 2 in flight (ev recorded by the consumer)
    uint64_t chunk = 0, lo = 0, n = 0;
    hipEvent_t ev = nullptr;
    // scan_lo < scan_hi: the producer has hopped over the headers of the blocks that start in the chunk (offsets relative to
    // the chunk), total = their inflated bytes (aligned), rel_end = where the first block of the next chunk starts
    bool scanned = false;
    std::vector<BgzfBlock> blocks;
    uint64_t total = 0, rel_end = 0;
  };
  uint64_t first = FIRST;        // chunks before first_k + first come from the mapping (0 once the process has its staging buffers)
  uint64_t scan_off = 0, scan_hi = 0;  // first == 0: the producer follows the chain of block headers from scan_off to scan_hi
  uint64_t scan_align = BGZF_OUT_ALIGN;  // alignment of the blocks' places in the inflated stream (1: one contiguous stream)
  Buf buf[NB];
  const uint8_t *file;
  int fd;  // >= 0: the staging threads read() the file (page cache -> buffer, no page tables to fill and to tear down again)
  uint64_t size, chunk_bytes, first_k, nchunks, buf_bytes;  // chunks first_k .. nchunks - 1 are wanted; the first FIRST of them come straight from the mapping
  int threads, device;
  std::mutex mu;
  std::condition_variable cv;
  double t_wait_buf = 0, t_copy = 0, t_scan_join = 0;  // the producer's time: waiting for a free buffer, copying, waiting for the scan before
  std::thread producer, scanner;
  bool stop = false;
  std::string error;

  // b_lo .. b_hi: the block starts the caller decodes (b_hi = 0: the caller hops over the headers itself)
  StagePool(const uint8_t *f, int fd_, uint64_t n, uint64_t cb, int th, int dev, uint64_t k_first, uint64_t k_end, uint64_t b_lo = 0, uint64_t b_hi = 0, uint64_t align = BGZF_OUT_ALIGN)
      : file(f), fd(fd_), size(n), chunk_bytes(cb), first_k(k_first), nchunks(k_end), buf_bytes((cb + SLACK + 4095) / 4096 * 4096), threads(th), device(dev)
  {
    scan_align = align;
    for (auto &b : buf) HIP_CHECK(hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
    // the staging buffers of an earlier file are at hand: every chunk goes through them (a plain DMA each, no pageable copy on
    // the driver thread), and the producer hops over the block headers of a chunk as soon as it has read it
    if (stage_cache().idle_of(buf_bytes) >= NB)
    {
      first = 0;
      scan_off = b_lo;
      scan_hi = b_hi;
    }
    if (nchunks > first_k + first) producer = std::thread([this] { run(); });
  }
  ~StagePool() { shutdown(); }
  void shutdown()
  {
    {
      std::lock_guard<std::mutex> g(mu);
      stop = true;
    }
    cv.notify_all();
    if (producer.joinable()) producer.join();
    for (auto &b : buf)
    {
      if (b.state == 2 && b.ev) (void) hipEventSynchronize(b.ev);
      if (b.p) stage_cache().give(b.p, buf_bytes);
      if (b.ev) (void) hipEventDestroy(b.ev);
      b.p = nullptr;
      b.ev = nullptr;
      b.state = 0;
    }
  }
  void run()
  {
    run_chunks();
    if (scanner.joinable()) scanner.join();
  }
  void run_chunks()
  {
    (void) hipSetDevice(device);
    for (uint64_t k = first_k + first; k < nchunks; ++k)
    {
      Buf &b = buf[k % NB];
      const double tp0 = now_s2();
      {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return stop || b.state == 0 || b.state == 2; });
        if (stop) return;
        if (b.state == 2)
        {
          g.unlock();
          (void) hipEventSynchronize(b.ev);  // the DMA out of this buffer is done
          g.lock();
          b.state = 0;
        }
      }
      const double tp1 = now_s2();
      t_wait_buf += tp1 - tp0;
      const uint64_t lo = k * chunk_bytes, n = std::min(size - lo, chunk_bytes + SLACK);
      if (!b.p) b.p = stage_cache().take(buf_bytes);
      if (!b.p)
      {
        std::lock_guard<std::mutex> g(mu);
        error = "no page-locked host memory for the feed's staging buffers";
        cv.notify_all();
        return;
      }
      const uint64_t per = 1u << 20;
      std::atomic<bool> short_read{false};
      auto fetch = [&](uint64_t at, uint64_t len) {
        if (fd < 0)
        {
          memcpy(b.p + at, file + lo + at, len);
          return;
        }
        uint64_t got = 0;
        while (got < len)
        {
          const ssize_t r = pread(fd, b.p + at + got, len - got, (off_t) (lo + at + got));
          if (r <= 0)
          {
            if (r < 0 && errno == EINTR) continue;
            short_read = true;
            return;
          }
          got += (uint64_t) r;
        }
      };
      copy_pool().run((n + per - 1) / per, threads, [&](uint64_t i) { fetch(i * per, std::min(per, n - i * per)); });
      const double tp2 = now_s2();
      t_copy += tp2 - tp1;
      if (short_read)
      {
        std::lock_guard<std::mutex> g(mu);
        error = "cannot read the BAM file";
        cv.notify_all();
        return;
      }
      // the header hops of this chunk (a serial chain over its ~1100 blocks, 0.25 ms) run beside the reads of the next one
      if (scanner.joinable()) scanner.join();
      t_scan_join += now_s2() - tp2;
      {
        std::lock_guard<std::mutex> g(mu);
        if (!error.empty()) return;
      }
      scanner = std::thread([this, &b, k, lo, n] {
        b.scanned = false;
        if (scan_hi)
        {
          b.blocks.clear();
          b.total = 0;
          uint64_t rel = scan_off - lo;
          std::string why;
          if (rel < chunk_bytes && scan_off < scan_hi && !bgzf_scan_range(b.p, n, rel, std::min(chunk_bytes - rel, scan_hi - scan_off), b.blocks, b.total, why, scan_align))
          {
            std::lock_guard<std::mutex> g(mu);
            error = "c" + why;  // ('c': an input error, see get())
            cv.notify_all();
            return;
          }
          scan_off = lo + rel;
          b.rel_end = rel;
          b.scanned = true;
        }
        {
          std::lock_guard<std::mutex> g(mu);
          b.chunk = k;
          b.lo = lo;
          b.n = n;
          b.state = 1;
        }
        cv.notify_all();
      });
    }
    if (scanner.joinable()) scanner.join();
  }
  // chunk k >= FIRST: blocks until it is staged
  Buf &get(uint64_t k)
  {
    Buf &b = buf[k % NB];
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return !error.empty() || (b.state == 1 && b.chunk == k); });
    if (!error.empty()) throw bk_error(error[0] == 'c' ? BK_ERR_IO : BK_ERR_LIMIT, error[0] == 'c' && error.compare(0, 6, "cannot") != 0 ? error.substr(1) : error);
    return b;
  }
  // the consumer has queued its copy out of b on st
  void release(Buf &b, hipStream_t st)
  {
    HIP_CHECK(hipEventRecord(b.ev, st));
    {
      std::lock_guard<std::mutex> g(mu);
      b.state = 2;
    }
    cv.notify_all();
  }
};

// BAM header (magic, text, reference list) from the first BGZF blocks, inflated on the host one by one until the
// list is complete; first_blk / first_off = where the first record starts
void parse_bam_header(const uint8_t *fdata, const std::vector<BgzfBlock> &blocks, bool more_file, bk_bam_dev *h, uint32_t &n_ref, uint32_t &first_blk, uint32_t &first_off)
{
  std::vector<uint8_t> head;
  size_t hb = 0;
  auto need = [&](size_t bytes) {
    while (head.size() < bytes)
    {
      if (hb >= blocks.size()) throw bk_error(hb && more_file ? BK_ERR_LIMIT : BK_ERR_IO, "truncated BAM header (or a header larger than one feed chunk)");
      if (!host_inflate_block(fdata, blocks[hb], head)) throw bk_error(BK_ERR_IO, "inflate failed");
      ++hb;
    }
  };
  need(12);
  if (memcmp(head.data(), "BAM\1", 4) != 0) throw bk_error(BK_ERR_IO, "not a BAM file");
  size_t p = 4;
  const uint32_t l_text = rd32h(head.data() + p);
  p += 4 + (size_t) l_text;
  need(p + 4);
  n_ref = rd32h(head.data() + p);
  p += 4;
  h->names.clear();
  h->name_ptrs.clear();
  h->lens.clear();
  for (uint32_t i = 0; i < n_ref; ++i)
  {
    need(p + 4);
    const uint32_t l_name = rd32h(head.data() + p);
    p += 4;
    need(p + l_name + 4);
    h->names.emplace_back((const char *) head.data() + p, l_name ? l_name - 1 : 0);
    p += l_name;
    h->lens.push_back(rd32h(head.data() + p));
    p += 4;
  }
  for (auto &nm : h->names) h->name_ptrs.push_back(nm.c_str());
  uint64_t acc = 0;
  first_blk = 0;
  while (first_blk < blocks.size() && acc + blocks[first_blk].isize <= p) acc += blocks[first_blk++].isize;
  first_off = (uint32_t) (p - acc);
}

// The header of a mapped file, however many blocks it takes: reference list into h, and where the first record starts
// (in_off of its block = absolute file offset of the block's deflate stream, offset inside the inflated block).
void parse_bam_header_of_file(const uint8_t *file, uint64_t size, bk_bam_dev *h, uint32_t &n_ref, uint64_t &first_in_off, uint32_t &first_off)
{
  std::vector<BgzfBlock> blocks;
  uint64_t total = 0, off = 0;
  std::string why;
  while (true)
  {
    if (!bgzf_scan_range(file, size, off, 1u << 20, blocks, total, why)) throw bk_error(BK_ERR_IO, why);
    uint32_t first_blk = 0;
    try
    {
      parse_bam_header(file, blocks, off < size, h, n_ref, first_blk, first_off);
    }
    catch (const bk_error &e)
    {
      if (e.code == BK_ERR_LIMIT && off < size) continue;  // the header goes on in blocks not hopped over yet
      throw;
    }
    first_in_off = first_blk < blocks.size() ? blocks[first_blk].in_off : ~0ull;  // ~0: no record at all
    return;
  }
}

struct not_block_aligned
{
};

// one chunk of the file in flight: its compressed bytes, inflated bytes, match tokens and per-block counts
struct FeedSlot
{
  DevBuf dfile, dblk, ddata, dslab, dcnt, dnr, dnc, dna, dscan, derr, dindex;
  hipStream_t st = nullptr;
  hipEvent_t ev_count = nullptr, ev_emit = nullptr;
  uint64_t *tot = nullptr;  // pinned: records, CIGAR words, aux bytes, error flags of the chunk
  std::vector<BgzfBlock> blocks;
  uint32_t first_blk = 0, first_off = 0;
  bool used = false;
  ~FeedSlot()
  {
    if (st) (void) hipStreamDestroy(st);
    if (ev_count) (void) hipEventDestroy(ev_count);
    if (ev_emit) (void) hipEventDestroy(ev_emit);
    if (tot) (void) hipHostFree(tot);
  }
};
// The slots of a finished decode (streams, events, device buffers of the chunk size) stay with the process for the next file
// of the device: freeing and allocating them costs 7-9 ms per file.  A decode that failed drops its slots.
template <class Slot> struct SlotCacheOf
{
  std::mutex mu;
  std::vector<std::pair<int, std::unique_ptr<Slot>>> idle;
  std::unique_ptr<Slot> take(int device)
  {
    std::lock_guard<std::mutex> g(mu);
    for (size_t i = 0; i < idle.size(); ++i)
      if (idle[i].first == device)
      {
        std::unique_ptr<Slot> r = std::move(idle[i].second);
        idle.erase(idle.begin() + (long) i);
        return r;
      }
    return std::unique_ptr<Slot>(new Slot());
  }
  void release_all()
  {
    std::vector<std::pair<int, std::unique_ptr<Slot>>> gone;
    {
      std::lock_guard<std::mutex> g(mu);
      gone.swap(idle);
    }
    for (auto &e : gone)
    {
      (void) hipSetDevice(e.first);
      e.second.reset();  // the slot's buffers, stream and events
    }
  }
  void give(int device, std::unique_ptr<Slot> s)
  {
    s->used = false;
    s->blocks.clear();
    std::lock_guard<std::mutex> g(mu);
    if (idle.size() < 16) idle.emplace_back(device, std::move(s));
  }
};
typedef SlotCacheOf<FeedSlot> SlotCache;
SlotCache &slot_cache()
{
  static SlotCache *c = new SlotCache();  // never destroyed: the buffers go with the process
  return *c;
}
// buffer of at least `need` bytes that keeps its first `used` bytes (device-to-device copy when it moves)
void grow_keep(DevBuf &b, size_t used, size_t need)
{
  if (need <= b.cap) return;
  DevBuf nb;
  (void) nb.ensure(need);
  if (used) HIP_CHECK(hipMemcpy(nb.p, b.p, used, hipMemcpyDeviceToDevice));
  b = std::move(nb);
}
}  // namespace

// The file is taken in chunks of 32 or 64 MiB of BGZF blocks (below).  Per chunk: H2D copy of the staged bytes, inflate,
// per-block record counts + scans, D2H of the three totals - all on the chunk's own stream - and, once the totals are on the
// host, a second walk of the record-length chains that leaves where every record lies and the emit kernel, one lane per
// record, at the running offsets of the columns.  Four to eight chunks are in flight, so the copy of one overlaps the
// inflate of the ones before and the emit of the ones before those, and device memory holds those chunks plus the columns
// whatever the size of the file.  The columns are sized from the first chunk (records per compressed byte x file size) and
// grow by copying when that was too small.
// block start >= target reached by hopping over the block headers from the block start `off` (the end of the file counts as one)
static void bgzf_hop_to(const uint8_t *file, uint64_t n, uint64_t &off, uint64_t target)
{
  std::vector<BgzfBlock> tmp;
  std::string why;
  uint64_t total = 0;
  while (off < n && off < target)
  {
    tmp.clear();
    if (!bgzf_scan_range(file, n, off, std::min<uint64_t>(target - off, 64u << 20), tmp, total, why)) throw bk_error(BK_ERR_IO, why);
  }
}

// part / parts: only the BGZF blocks that start in [b_lo, b_hi) are decoded, b_x = the first block start at or behind
// x / parts of the file - every one of `parts` callers finds the same boundaries, so the parts tile the file (one rank of a
// sharded run each, bk_bam_decode_device_part).  Records of a block-aligned file never leave their block.
static void decode_chunked(const MappedFile &file, int device, bk_bam_dev *h, bk_soa *cols, const FeedConsumer *fc = nullptr, int part = 0, int parts = 1)
{
  {
    const double t0 = now_s2();
    if (file.size() == 0) throw bk_error(BK_ERR_IO, "empty file");
    // A chunk's inflate kernel lasts as long as its slowest block (~3 ms) however few blocks it has, so the rate comes from
    // the chunks in flight: eight slots of 32 MiB when the process has a hardware queue for each of them (the runtime reads
    // GPU_MAX_HW_QUEUES when it starts, default 4: streams that share a queue make each other's launches wait), else four
    // slots of 64 MiB.  BREAKID_FEED_CHUNK_MB overrides the chunk size (tests: many chunks from a small file).
    const int hw_queues = getenv("GPU_MAX_HW_QUEUES") ? atoi(getenv("GPU_MAX_HW_QUEUES")) : 4;
    const bool wide = hw_queues >= 8;
    uint64_t chunk_bytes = wide ? 32ull << 20 : 64ull << 20;
    if (const char *e = getenv("BREAKID_FEED_CHUNK_MB"))
      if (atof(e) > 0) chunk_bytes = (uint64_t) (atof(e) * 1048576.0);
    chunk_bytes = std::max<uint64_t>(chunk_bytes, 70000) / 4096 * 4096 + 4096;  // a chunk is longer than the longest block
    int copy_threads = 8;
    if (const char *e = getenv("BREAKID_THREADS"))
      if (atoi(e) > 0) copy_threads = std::min(atoi(e), 16);
    const bool stage_from_mapping = false;  // (the staging threads read() the file; copying out of the mapping was the first form)
    // chunks in flight; how far the driver thread runs ahead of the totals it waits for (a slot is reused LAG + 1 .. NS chunks later)
    constexpr int NS_MAX = 12;
    const int NS = wide ? 8 : 4, LAG = wide ? 6 : 2;
    std::unique_ptr<FeedSlot> slot_store[NS_MAX];
    struct SlotRef  // slot[k] as before
    {
      std::unique_ptr<FeedSlot> *a;
      FeedSlot &operator[](int k) const { return *a[k]; }
    } slot = {slot_store};
    const bool keep_slots = true;
    for (int k = 0; k < NS; ++k)
    {
      slot_store[k] = keep_slots ? slot_cache().take(device) : std::unique_ptr<FeedSlot>(new FeedSlot());
      FeedSlot &s = slot[k];
      if (!s.st) HIP_CHECK(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
      if (!s.ev_count) HIP_CHECK(hipEventCreateWithFlags(&s.ev_count, hipEventDisableTiming));
      if (!s.ev_emit) HIP_CHECK(hipEventCreateWithFlags(&s.ev_emit, hipEventDisableTiming));
      if (!s.tot) HIP_CHECK(hipHostMalloc((void **) &s.tot, 4 * sizeof(uint64_t), hipHostMallocDefault));
    }
    uint64_t first_in_off = 0;
    uint32_t hdr_first_off = 0, n_ref = 0;
    parse_bam_header_of_file(file.data(), file.size(), h, n_ref, first_in_off, hdr_first_off);
    if (fc && fc->on_header) fc->on_header(fc->user, (int) h->names.size(), h->name_ptrs.data(), h->lens.data());
    uint64_t est_total = 0;
    uint64_t b_lo = 0, b_hi = file.size();
    if (parts > 1)
    {
      uint64_t o = 0;
      bgzf_hop_to(file.data(), file.size(), o, (uint64_t) ((unsigned __int128) file.size() * (unsigned) part / (unsigned) parts));
      b_lo = o;
      if (part + 1 < parts) bgzf_hop_to(file.data(), file.size(), o, (uint64_t) ((unsigned __int128) file.size() * (unsigned) (part + 1) / (unsigned) parts));
      b_hi = part + 1 < parts ? o : file.size();
    }
    const uint64_t k0 = b_lo / chunk_bytes, k1 = b_hi > b_lo ? (b_hi + chunk_bytes - 1) / chunk_bytes : k0;  // chunks k0 .. k1 - 1
    StagePool pool(file.data(), stage_from_mapping ? -1 : file.descriptor(), file.size(), chunk_bytes, copy_threads, device, k0, k1, b_lo, b_hi);
    uint64_t off = b_lo, n_rec = 0, n_cig = 0, n_aux = 0, cap_rec = 0, cap_cig = 0, cap_aux = 0, nblk_all = 0, first_bytes = 0;
    double t_h2d = 0, t_alloc = 0, t_scan = 0, t_reserve = 0, t_stage_wait = 0, t_launch = 0, t_emit = 0, t_slot_wait = 0;
    const double t_setup_done = now_s2();
    std::string why;
    BamCols c = {};
    auto sync_all = [&]() {
      for (int k = 0; k < NS; ++k) HIP_CHECK(hipStreamSynchronize(slot[k].st));
    };
    // columns for at least (r, g, a) records / CIGAR words / aux bytes; the emits in flight finish before a buffer moves
    double t_grow = 0, t_grow_sync = 0;
    auto reserve = [&](uint64_t r, uint64_t g, uint64_t a) {
      if (r <= cap_rec && g <= cap_cig && a <= cap_aux) return;
      const double tg0 = now_s2();
      // (the first allocation moves nothing: no emit has been launched yet, the chunks in flight go on)
      const bool moves = cap_rec || cap_cig || cap_aux;
      if (moves) sync_all();
      t_grow_sync += now_s2() - tg0;
      struct GrowTimer
      {
        double &acc, t0;
        ~GrowTimer() { acc += now_s2() - t0; }
      } grow_timer{t_grow, tg0};
      if (moves && fc && fc->before_move) fc->before_move(fc->user);
      if (r > cap_rec)
      {
        const uint64_t nc = std::max(r, cap_rec + cap_rec / 2) + 1024;
        grow_keep(h->tid, n_rec * 4, (nc + 4) * 4);
        grow_keep(h->pos, n_rec * 4, (nc + 4) * 4);
        grow_keep(h->mtid, n_rec * 4, (nc + 4) * 4);
        grow_keep(h->mpos, n_rec * 4, (nc + 4) * 4);
        grow_keep(h->isize, n_rec * 4, (nc + 4) * 4);
        grow_keep(h->flag, n_rec * 2, (nc + 4) * 2);
        grow_keep(h->mapq, n_rec, nc + 4);
        grow_keep(h->qhash, n_rec * 8, (nc + 4) * 8);
        grow_keep(h->qcheck, n_rec * 4, (nc + 4) * 4);
        grow_keep(h->cigar_off, n_rec * 4, (nc + 4) * 4);
        grow_keep(h->aux_off, n_rec * 4, (nc + 4) * 4);
        cap_rec = nc;
      }
      if (g > cap_cig)
      {
        const uint64_t nc = std::max(g, cap_cig + cap_cig / 2) + 1024;
        grow_keep(h->cigar, n_cig * 4, (nc + 4) * 4);
        cap_cig = nc;
      }
      if (a > cap_aux)
      {
        const uint64_t nc = std::max(a, cap_aux + cap_aux / 2) + 1024;
        grow_keep(h->aux, n_aux, nc + 4);
        cap_aux = nc;
      }
      c.tid = h->tid.get<int32_t>();
      c.pos = h->pos.get<int32_t>();
      c.mtid = h->mtid.get<int32_t>();
      c.mpos = h->mpos.get<int32_t>();
      c.isize = h->isize.get<int32_t>();
      c.flag = h->flag.get<uint16_t>();
      c.mapq = h->mapq.get<uint8_t>();
      c.qhash = h->qhash.get<uint64_t>();
      c.qcheck = h->qcheck.get<uint32_t>();
      c.cigar_off = h->cigar_off.get<uint32_t>();
      c.aux_off = h->aux_off.get<uint32_t>();
      c.cigar = h->cigar.get<uint32_t>();
      c.aux = h->aux.get<uint8_t>();
    };
    // chunk k -> slot: hop over the block headers in the staged bytes, copy, inflate, count
    auto stage = [&](FeedSlot &s, uint64_t k) {
      const double tsw = now_s2();
      if (s.used) HIP_CHECK(hipEventSynchronize(s.ev_emit));
      s.used = true;
      s.blocks.clear();
      s.first_blk = 0;
      s.first_off = 0;
      const double ts00 = now_s2();
      t_slot_wait += ts00 - tsw;
      StagePool::Buf *sb = k >= k0 + pool.first ? &pool.get(k) : nullptr;
      const uint64_t src_lo = k * chunk_bytes, src_n = std::min<uint64_t>(file.size() - src_lo, chunk_bytes + StagePool::SLACK);
      const uint8_t *fdata = sb ? sb->p : file.data() + src_lo;
      const double ts0 = now_s2();
      t_stage_wait += ts0 - ts00;
      // the blocks that start inside [k C, (k + 1) C); offsets are relative to the start of the range
      uint64_t total = 0, rel = off - src_lo;
      if (sb && sb->scanned)
      {
        s.blocks.swap(sb->blocks);
        total = sb->total;
        rel = sb->rel_end;
      }
      else if (rel < chunk_bytes && off < b_hi && !bgzf_scan_range(fdata, src_n, rel, std::min(chunk_bytes - rel, b_hi - off), s.blocks, total, why))
        throw bk_error(BK_ERR_IO, why);
      off = src_lo + rel;
      // blocks of the header hold no records; the first record's block starts at first_off
      while (s.first_blk < s.blocks.size() && src_lo + s.blocks[s.first_blk].in_off < first_in_off) ++s.first_blk;
      if (s.first_blk < s.blocks.size() && src_lo + s.blocks[s.first_blk].in_off == first_in_off) s.first_off = hdr_first_off;
      const uint32_t nb = (uint32_t) s.blocks.size();
      nblk_all += nb;
      if (nb == 0)
      {
        if (sb) pool.release(*sb, s.st);
        return;
      }
      const double ts1 = now_s2();
      t_scan += ts1 - ts0;
      uint8_t *df = s.dfile.as<uint8_t>(rel + 8);
      BgzfBlock *db = s.dblk.as<BgzfBlock>((uint64_t) nb + 1);
      uint8_t *dd = s.ddata.as<uint8_t>(total + 64);
      uint32_t *de = s.derr.as<uint32_t>(1);
      uint8_t *slab = s.dslab.as<uint8_t>(bgzf_scratch_bytes(nb));
      BlockCount *dc = s.dcnt.as<BlockCount>((uint64_t) nb + 1);
      uint64_t *nr = s.dnr.as<uint64_t>((uint64_t) nb + 1), *nc = s.dnc.as<uint64_t>((uint64_t) nb + 1), *na = s.dna.as<uint64_t>((uint64_t) nb + 1);
      const double ta = now_s2();
      t_alloc += ta - ts1;
      HIP_CHECK(hipMemcpyAsync(df, fdata, rel, hipMemcpyHostToDevice, s.st));
      if (sb) pool.release(*sb, s.st);
      HIP_CHECK(hipMemcpyAsync(db, s.blocks.data(), (size_t) nb * sizeof(BgzfBlock), hipMemcpyHostToDevice, s.st));
      const double tl0 = now_s2();
      t_h2d += tl0 - ta;
      HIP_CHECK(hipMemsetAsync(de, 0, 4, s.st));
      launch_bgzf_inflate(df, db, nb, dd, slab, de, s.st, true);
      BamCols none = {};
      hipLaunchKernelGGL(k_bam_blocks<false>, dim3(cdiv(nb, 64)), dim3(64), 0, s.st, dd, db, nb, s.first_blk, s.first_off, (int32_t) n_ref, dc, nullptr, nullptr, nullptr, 0ull, 0ull, 0ull, none);
      hipLaunchKernelGGL(k_bam_count_split, dim3(cdiv(nb, 256)), dim3(256), 0, s.st, dc, nb, nr, nc, na, de);
      prims::exclusive_scan<unsigned long long>((unsigned long long *) nr, (unsigned long long *) nr, nb, s.dscan, s.st);
      prims::exclusive_scan<unsigned long long>((unsigned long long *) nc, (unsigned long long *) nc, nb, s.dscan, s.st);
      prims::exclusive_scan<unsigned long long>((unsigned long long *) na, (unsigned long long *) na, nb, s.dscan, s.st);
      s.tot[3] = 0;
      HIP_CHECK(hipMemcpyAsync(&s.tot[0], nr + nb, 8, hipMemcpyDeviceToHost, s.st));
      HIP_CHECK(hipMemcpyAsync(&s.tot[1], nc + nb, 8, hipMemcpyDeviceToHost, s.st));
      HIP_CHECK(hipMemcpyAsync(&s.tot[2], na + nb, 8, hipMemcpyDeviceToHost, s.st));
      HIP_CHECK(hipMemcpyAsync(&s.tot[3], de, 4, hipMemcpyDeviceToHost, s.st));
      HIP_CHECK(hipEventRecord(s.ev_count, s.st));
      t_launch += now_s2() - tl0;
    };
    // totals of the chunk are known: room in the columns, emit at the running offsets
    auto finish = [&](FeedSlot &s, bool first, bool more) {
      const uint32_t nb = (uint32_t) s.blocks.size();
      if (nb == 0) return;
      const double tw0 = now_s2();
      HIP_CHECK(hipEventSynchronize(s.ev_count));
      if (s.tot[3] & 1u) throw bk_error(BK_ERR_IO, "inflate failed");
      if (s.tot[3] & 2u) throw not_block_aligned();
      const uint64_t r = n_rec + s.tot[0], g = n_cig + s.tot[1], a = n_aux + s.tot[2];
      if (r >= 0xFFFFFFF0ull || g >= 0xFFFFFFF0ull || a >= 0xFFFFFFF0ull) throw bk_error(BK_ERR_LIMIT, "more than 2^32 records / CIGAR words / SA bytes in one BAM");
      if (first && more)
      {
        // the rest of the file at the first chunk's densities, 5 % on top
        const double scale = 1.05 * (double) (b_hi - b_lo) / (double) std::max<uint64_t>(first_bytes, 1);
        reserve((uint64_t) (r * scale), (uint64_t) (g * scale), (uint64_t) (a * scale));
        est_total = (uint64_t) (r * scale);
      }
      reserve(r, g, a);
      const double te0 = now_s2();
      t_reserve += te0 - tw0;
      launch_emit(s.ddata.get<uint8_t>(), s.dblk.get<BgzfBlock>(), nb, s.first_blk, s.first_off, (int32_t) n_ref, s.dcnt.get<BlockCount>(), s.dnr.get<uint64_t>(), s.dnc.get<uint64_t>(),
                  s.dna.get<uint64_t>(), n_rec, n_cig, n_aux, c, nullptr, 0ull, 0, s.tot[0], s.dindex, s.st);
      HIP_CHECK(hipEventRecord(s.ev_emit, s.st));
      n_rec = r;
      n_cig = g;
      n_aux = a;
      if (fc && fc->on_chunk)
      {
        bk_soa v;
        memset(&v, 0, sizeof v);
        v.n = n_rec;
        v.tid = c.tid; v.pos = c.pos; v.mtid = c.mtid; v.mpos = c.mpos; v.isize = c.isize; v.flag = c.flag; v.mapq = c.mapq; v.qhash = c.qhash; v.qcheck = c.qcheck;
        v.cigar_off = c.cigar_off; v.cigar = c.cigar; v.aux_off = c.aux_off; v.aux = c.aux;
        v.n_cigar_words = n_cig;
        v.n_aux_bytes = n_aux;
        fc->on_chunk(fc->user, &v, n_rec, std::max(est_total, n_rec), s.ev_emit);
      }
      t_emit += now_s2() - te0;
    };
    // the driver thread stages chunk ci, then takes the totals of chunk ci - LAG
    const uint64_t nchunk = k1 - k0;
    for (uint64_t ci = 0; ci < nchunk + LAG; ++ci)
    {
      if (ci < nchunk) stage(slot[ci % NS], k0 + ci);
      if (ci == 0) first_bytes = off - b_lo;
      if (ci >= LAG) finish(slot[(ci - LAG) % NS], ci == LAG, ci - LAG + 1 < nchunk);
    }
    if (off != b_hi) throw bk_error(BK_ERR_IO, "BGZF blocks do not end at the end of the file");
    const double t_end_loop = now_s2();
    if (cap_rec == 0) reserve(n_rec + 1, n_cig + 1, n_aux + 1);  // no record at all (an empty part): the columns exist all the same, with their end entries
    reserve(n_rec, n_cig, n_aux);
    sync_all();
    const uint32_t ends[2] = {(uint32_t) n_cig, (uint32_t) n_aux};
    HIP_CHECK(hipMemcpy(c.cigar_off + n_rec, &ends[0], 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(c.aux_off + n_rec, &ends[1], 4, hipMemcpyHostToDevice));
    const double t3 = now_s2();
    memset(cols, 0, sizeof *cols);
    cols->n = n_rec;
    cols->tid = c.tid; cols->pos = c.pos; cols->mtid = c.mtid; cols->mpos = c.mpos; cols->isize = c.isize;
    cols->flag = c.flag; cols->mapq = c.mapq; cols->qhash = c.qhash; cols->qcheck = c.qcheck;
    cols->cigar_off = c.cigar_off; cols->cigar = c.cigar; cols->aux_off = c.aux_off; cols->aux = c.aux;
    cols->n_cigar_words = (uint32_t) n_cig;
    cols->n_aux_bytes = (uint32_t) n_aux;
    const double td0 = now_s2();
    pool.shutdown();
    for (int k = 0; k < NS; ++k)
      if (keep_slots)
        slot_cache().give(device, std::move(slot_store[k]));  // (every stream is idle: sync_all above)
      else
        for (DevBuf *b : {&slot[k].dfile, &slot[k].dblk, &slot[k].ddata, &slot[k].dslab, &slot[k].dcnt, &slot[k].dnr, &slot[k].dnc, &slot[k].dna, &slot[k].dscan, &slot[k].derr, &slot[k].dindex}) b->release();
    const double td2 = now_s2();
    if (bk_debug("feed"))
      fprintf(stderr, "[feed/gpu] columns: %.3f s allocating / growing them, of which %.3f s waiting for the chunks in flight first\n", t_grow, t_grow_sync);
    if (bk_debug("feed"))
      fprintf(stderr, "[feed/gpu] producer: %.3f s waiting for a free staging buffer, %.3f s copying (%d threads), %.3f s waiting for the header hops of the chunk before\n", pool.t_wait_buf, pool.t_copy, copy_threads,
              pool.t_scan_join);
    if (bk_debug("feed"))
      fprintf(stderr, "[feed/gpu] %llu records, %.1f MB file, %llu BGZF blocks in %llu chunks: file -> device table %.3f s (driver thread: setup %.3f s, waiting for a free slot %.3f s, waiting for staged bytes %.3f s, header hops %.3f s, buffers %.3f s, H2D calls %.3f s, kernel launches %.3f s, waiting for chunk totals + column growth %.3f s, emit launches %.3f s, final sync %.3f s, teardown %.3f s)\n",
              (unsigned long long) n_rec, file.size() / 1e6, (unsigned long long) nblk_all, (unsigned long long) nchunk, t3 - t0, t_setup_done - t0, t_slot_wait, t_stage_wait, t_scan, t_alloc, t_h2d, t_launch, t_reserve, t_emit, t3 - t_end_loop, td2 - td0);
  }
}

// Files whose records straddle BGZF blocks (htsjdk / Picard / GATK writers; htslib keeps records inside blocks).  One
// batch: the whole file image goes to HBM, the blocks are inflated into ONE contiguous stream, every block guesses its
// first record boundary (k_bam_guess), the walks are verified to chain (k_bam_verify: then the guesses are exact) and the
// columns are emitted as in the aligned case.
static void decode_packed(const MappedFile &file, int device, bk_bam_dev *h, bk_soa *cols)
{
  (void) device;
  const double t0 = now_s2();
  std::vector<BgzfBlock> blocks;
  uint64_t total = 0, off = 0;
  std::string why;
  if (!bgzf_scan_range(file.data(), file.size(), off, ~0ull, blocks, total, why, 1)) throw bk_error(BK_ERR_IO, why);
  uint32_t n_ref = 0, first_blk = 0, first_off = 0;
  uint64_t first_in_off = 0;
  parse_bam_header_of_file(file.data(), file.size(), h, n_ref, first_in_off, first_off);
  const uint32_t nblk = (uint32_t) blocks.size();
  while (first_blk < nblk && blocks[first_blk].in_off < first_in_off) ++first_blk;
  size_t free_b = 0, total_b = 0;
  HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
  if ((double) file.size() + (double) total * 1.3 + (double) bgzf_scratch_bytes(nblk) > 0.8 * (double) free_b)
    throw bk_error(BK_ERR_LIMIT, "BAM with records across BGZF blocks is too large for the one-batch GPU decoder (use bk_bam_open / bk_bam_decode)");
  DevBuf dfile, dblk, ddata, derr, dcnt, dnr, dnc, dna, dscan, dslab, dentry, dnext;
  uint8_t *df = dfile.as<uint8_t>(file.size() + 8);
  BgzfBlock *db = dblk.as<BgzfBlock>((uint64_t) nblk + 1);
  uint8_t *dd = ddata.as<uint8_t>(total + 64);
  uint32_t *de = derr.as<uint32_t>(1);
  uint8_t *slab = dslab.as<uint8_t>(bgzf_scratch_bytes(nblk));
  uint32_t *entry = dentry.as<uint32_t>((uint64_t) nblk + 1);
  uint64_t *next_abs = dnext.as<uint64_t>((uint64_t) nblk + 1);
  BlockCount *dc = dcnt.as<BlockCount>((uint64_t) nblk + 1);
  uint64_t *nr = dnr.as<uint64_t>((uint64_t) nblk + 1), *nc = dnc.as<uint64_t>((uint64_t) nblk + 1), *na = dna.as<uint64_t>((uint64_t) nblk + 1);
  hipStream_t st = nullptr;
  HIP_CHECK(hipMemcpy(df, file.data(), file.size(), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(db, blocks.data(), (size_t) nblk * sizeof(BgzfBlock), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemset(de, 0, 4));
  HIP_CHECK(hipMemset(dd + total, 0, 64));
  launch_bgzf_inflate(df, db, nblk, dd, slab, de, st);
  BamCols none = {};
  hipLaunchKernelGGL(k_bam_guess, dim3(nblk), dim3(64), 0, st, dd, db, nblk, first_blk, first_off, (int32_t) n_ref, total, entry);
  hipLaunchKernelGGL(k_bam_blocks<false>, dim3(cdiv(nblk, 64)), dim3(64), 0, st, dd, db, nblk, first_blk, first_off, (int32_t) n_ref, dc, nullptr, nullptr, nullptr, 0ull, 0ull, 0ull, none, entry,
                     total, next_abs);
  hipLaunchKernelGGL(k_bam_count_split, dim3(cdiv(nblk, 256)), dim3(256), 0, st, dc, nblk, nr, nc, na, de);
  hipLaunchKernelGGL(k_bam_verify, dim3(cdiv(nblk, 256)), dim3(256), 0, st, db, nblk, first_blk, entry, next_abs, total, de);
  prims::exclusive_scan<unsigned long long>((unsigned long long *) nr, (unsigned long long *) nr, nblk, dscan, st);
  prims::exclusive_scan<unsigned long long>((unsigned long long *) nc, (unsigned long long *) nc, nblk, dscan, st);
  prims::exclusive_scan<unsigned long long>((unsigned long long *) na, (unsigned long long *) na, nblk, dscan, st);
  uint64_t tot[3] = {0, 0, 0};
  uint32_t he = 0;
  HIP_CHECK(hipMemcpyAsync(&tot[0], nr + nblk, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&tot[1], nc + nblk, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&tot[2], na + nblk, 8, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&he, de, 4, hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  if (he & 1u) throw bk_error(BK_ERR_IO, "inflate failed");
  if (he & 2u) throw bk_error(BK_ERR_IO, "corrupt BAM record");
  if (he & 4u) throw bk_error(BK_ERR_IO, "the record boundaries of this BAM could not be established on the GPU: use the host decoder");
  const uint64_t n = tot[0];
  if (n >= 0xFFFFFFF0ull || tot[1] >= 0xFFFFFFF0ull || tot[2] >= 0xFFFFFFF0ull) throw bk_error(BK_ERR_LIMIT, "more than 2^32 records / CIGAR words / SA bytes in one BAM");
  BamCols c;
  c.tid = h->tid.as<int32_t>(n + 4);
  c.pos = h->pos.as<int32_t>(n + 4);
  c.mtid = h->mtid.as<int32_t>(n + 4);
  c.mpos = h->mpos.as<int32_t>(n + 4);
  c.isize = h->isize.as<int32_t>(n + 4);
  c.flag = h->flag.as<uint16_t>(n + 4);
  c.mapq = h->mapq.as<uint8_t>(n + 4);
  c.qhash = h->qhash.as<uint64_t>(n + 4);
  c.qcheck = h->qcheck.as<uint32_t>(n + 4);
  c.cigar_off = h->cigar_off.as<uint32_t>(n + 4);
  c.aux_off = h->aux_off.as<uint32_t>(n + 4);
  c.cigar = h->cigar.as<uint32_t>(tot[1] + 4);
  c.aux = h->aux.as<uint8_t>(tot[2] + 4);
  DevBuf dindex;
  launch_emit(dd, db, nblk, first_blk, first_off, (int32_t) n_ref, dc, nr, nc, na, 0ull, 0ull, 0ull, c, entry, total, 0, n, dindex, st);
  const uint32_t ends[2] = {(uint32_t) tot[1], (uint32_t) tot[2]};
  HIP_CHECK(hipMemcpyAsync(c.cigar_off + n, &ends[0], 4, hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemcpyAsync(c.aux_off + n, &ends[1], 4, hipMemcpyHostToDevice, st));
  HIP_CHECK(hipStreamSynchronize(st));
  memset(cols, 0, sizeof *cols);
  cols->n = n;
  cols->tid = c.tid; cols->pos = c.pos; cols->mtid = c.mtid; cols->mpos = c.mpos; cols->isize = c.isize;
  cols->flag = c.flag; cols->mapq = c.mapq; cols->qhash = c.qhash; cols->qcheck = c.qcheck;
  cols->cigar_off = c.cigar_off; cols->cigar = c.cigar; cols->aux_off = c.aux_off; cols->aux = c.aux;
  cols->n_cigar_words = (uint32_t) tot[1];
  cols->n_aux_bytes = (uint32_t) tot[2];
  if (bk_debug("feed"))
    fprintf(stderr, "[feed/gpu] %llu records, %.1f MB file, %u BGZF blocks, records across blocks (one batch, boundaries guessed and verified): file -> device table %.3f s\n",
            (unsigned long long) n, file.size() / 1e6, nblk, now_s2() - t0);
}

// One rank's part of such a file (bk_bam_decode_device_part: the blocks that start in the part-th of `parts` equal byte ranges).
// The part's blocks are inflated into one contiguous stream together with the blocks BEHIND the part that hold up to
// PACKED_PART_EXT more bytes: the last record that starts in the part ends there.  Every block of a part behind the first guesses its
// first record boundary, the part's very first block included; the walks are verified to chain up to the first record of the
// blocks behind the part.  That block is the first block of the NEXT part, where the same bytes give the same guess: the first
// boundary of part 0 is known (the header), its chain ends at part 1's guess, which is therefore a real boundary, and so on - no
// rank needs another rank's answer, and a part whose chain does not close is an error (the caller takes the host decoder), never a
// wrong table.  A rank takes the records that START in its blocks.
constexpr uint64_t PACKED_PART_EXT = 8u << 20;
static void decode_packed_part(const MappedFile &file, int device, bk_bam_dev *h, bk_soa *cols, int part, int parts)
{
  (void) device;
  const double t0 = now_s2();
  uint32_t n_ref = 0, hdr_first_off = 0;
  uint64_t hdr_first_in_off = 0;
  parse_bam_header_of_file(file.data(), file.size(), h, n_ref, hdr_first_in_off, hdr_first_off);
  uint64_t b_lo = 0, b_hi = file.size();
  {
    uint64_t o = 0;
    bgzf_hop_to(file.data(), file.size(), o, (uint64_t) ((unsigned __int128) file.size() * (unsigned) part / (unsigned) parts));
    b_lo = o;
    if (part + 1 < parts) bgzf_hop_to(file.data(), file.size(), o, (uint64_t) ((unsigned __int128) file.size() * (unsigned) (part + 1) / (unsigned) parts));
    b_hi = part + 1 < parts ? o : file.size();
  }
  std::vector<BgzfBlock> blocks;
  uint64_t total = 0, off = b_lo;
  std::string why;
  if (b_hi > b_lo && !bgzf_scan_range(file.data(), file.size(), off, b_hi - b_lo, blocks, total, why, 1)) throw bk_error(BK_ERR_IO, why);
  if (off != b_hi) throw bk_error(BK_ERR_IO, "BGZF blocks do not end where the part ends");
  const uint32_t nown = (uint32_t) blocks.size();
  const uint64_t total_own = total;
  // the blocks behind the part, until PACKED_PART_EXT bytes of their stream are at hand (or the file ends)
  while (nown && off < file.size() && total - total_own < PACKED_PART_EXT)
    if (!bgzf_scan_range(file.data(), file.size(), off, 1u << 20, blocks, total, why, 1)) throw bk_error(BK_ERR_IO, why);
  const bool to_eof = off >= file.size();
  const uint64_t b_end = off;
  const uint32_t nblk = (uint32_t) blocks.size();
  // the part's first record: known for the part that holds the header's end, guessed otherwise
  uint32_t first_blk = 0, first_off = 0xFFFFFFFFu;
  if (nown && hdr_first_in_off >= blocks[0].in_off)
  {
    while (first_blk < nblk && blocks[first_blk].in_off < hdr_first_in_off) ++first_blk;
    first_off = hdr_first_off;
    if (first_blk >= nown) first_blk = nown;  // (the header runs beyond this part: no record starts here)
  }
  uint64_t tot[3] = {0, 0, 0};
  BamCols c = {};
  DevBuf dfile, dblk, ddata, derr, dcnt, dnr, dnc, dna, dscan, dslab, dentry, dnext, dindex;
  hipStream_t st = nullptr;
  if (nown && first_blk < nown)
  {
    size_t free_b = 0, total_b = 0;
    HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    if ((double) (b_end - b_lo) + (double) total * 1.3 + (double) bgzf_scratch_bytes(nblk) > 0.8 * (double) free_b)
      throw bk_error(BK_ERR_LIMIT, "this part of a BAM with records across BGZF blocks is too large for the GPU decoder (use bk_bam_open / bk_bam_decode)");
    for (BgzfBlock &b : blocks) b.in_off -= b_lo;  // offsets into the part's image
    uint8_t *df = dfile.as<uint8_t>(b_end - b_lo + 8);
    BgzfBlock *db = dblk.as<BgzfBlock>((uint64_t) nblk + 1);
    uint8_t *dd = ddata.as<uint8_t>(total + 64);
    uint32_t *de = derr.as<uint32_t>(1);
    uint8_t *slab = dslab.as<uint8_t>(bgzf_scratch_bytes(nblk));
    uint32_t *entry = dentry.as<uint32_t>((uint64_t) nblk + 1);
    uint64_t *next_abs = dnext.as<uint64_t>((uint64_t) nblk + 1);
    BlockCount *dc = dcnt.as<BlockCount>((uint64_t) nblk + 1);
    uint64_t *nr = dnr.as<uint64_t>((uint64_t) nblk + 1), *nc = dnc.as<uint64_t>((uint64_t) nblk + 1), *na = dna.as<uint64_t>((uint64_t) nblk + 1);
    HIP_CHECK(hipMemcpy(df, file.data() + b_lo, b_end - b_lo, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(db, blocks.data(), (size_t) nblk * sizeof(BgzfBlock), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(de, 0, 4));
    HIP_CHECK(hipMemset(dd + total, 0, 64));
    launch_bgzf_inflate(df, db, nblk, dd, slab, de, st);
    const int allow_tail = to_eof ? 0 : 1;  // (the stream is cut behind the extension: its last walk stops at a record that does not fit)
    BamCols none = {};
    hipLaunchKernelGGL(k_bam_guess, dim3(nblk), dim3(64), 0, st, dd, db, nblk, first_blk, first_off, (int32_t) n_ref, total, entry);
    hipLaunchKernelGGL(k_bam_blocks<false>, dim3(cdiv(nblk, 64)), dim3(64), 0, st, dd, db, nblk, first_blk, first_off, (int32_t) n_ref, dc, nullptr, nullptr, nullptr, 0ull, 0ull, 0ull, none, entry,
                       total, next_abs, allow_tail);
    // counts and columns of the part's OWN blocks; the chain is verified over the blocks behind it as well
    hipLaunchKernelGGL(k_bam_count_split, dim3(cdiv(nown, 256)), dim3(256), 0, st, dc, nown, nr, nc, na, de);
    hipLaunchKernelGGL(k_bam_verify, dim3(cdiv(nblk, 256)), dim3(256), 0, st, db, nblk, first_blk, entry, next_abs, total, de, allow_tail, (unsigned long long *) nullptr);
    prims::exclusive_scan<unsigned long long>((unsigned long long *) nr, (unsigned long long *) nr, nown, dscan, st);
    prims::exclusive_scan<unsigned long long>((unsigned long long *) nc, (unsigned long long *) nc, nown, dscan, st);
    prims::exclusive_scan<unsigned long long>((unsigned long long *) na, (unsigned long long *) na, nown, dscan, st);
    uint32_t he = 0, e0 = 0;
    HIP_CHECK(hipMemcpyAsync(&tot[0], nr + nown, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(&tot[1], nc + nown, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(&tot[2], na + nown, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(&he, de, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(&e0, entry + first_blk, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (he & 1u) throw bk_error(BK_ERR_IO, "inflate failed");
    if (he & 2u) throw bk_error(BK_ERR_IO, "corrupt BAM record");
    if (he & 4u) throw bk_error(BK_ERR_IO, "the record boundaries of this part of the BAM could not be established on the GPU: use the host decoder");
    // the blocks behind the part must hold the end of its last record: a walk of an own block that ran out of stream would have
    // been counted as "does not fit" (allow_tail) - its record is longer than the extension
    if (!to_eof)
    {
      std::vector<uint64_t> na_h(nown);
      HIP_CHECK(hipMemcpy(na_h.data(), next_abs, (size_t) nown * 8, hipMemcpyDeviceToHost));
      std::vector<uint32_t> en(nown);
      HIP_CHECK(hipMemcpy(en.data(), entry, (size_t) nown * 4, hipMemcpyDeviceToHost));
      for (uint32_t b = first_blk; b < nown; ++b)
        if (en[b] < blocks[b].isize && na_h[b] < blocks[b].out_off + blocks[b].isize)
          throw bk_error(BK_ERR_LIMIT, "a record at the end of this part of the BAM is longer than the 8 MiB behind the part: use the host decoder");
    }
    (void) e0;
    const uint64_t n = tot[0];
    if (n >= 0xFFFFFFF0ull || tot[1] >= 0xFFFFFFF0ull || tot[2] >= 0xFFFFFFF0ull) throw bk_error(BK_ERR_LIMIT, "more than 2^32 records / CIGAR words / SA bytes in one part of a BAM");
    c.tid = h->tid.as<int32_t>(n + 4);
    c.pos = h->pos.as<int32_t>(n + 4);
    c.mtid = h->mtid.as<int32_t>(n + 4);
    c.mpos = h->mpos.as<int32_t>(n + 4);
    c.isize = h->isize.as<int32_t>(n + 4);
    c.flag = h->flag.as<uint16_t>(n + 4);
    c.mapq = h->mapq.as<uint8_t>(n + 4);
    c.qhash = h->qhash.as<uint64_t>(n + 4);
    c.qcheck = h->qcheck.as<uint32_t>(n + 4);
    c.cigar_off = h->cigar_off.as<uint32_t>(n + 4);
    c.aux_off = h->aux_off.as<uint32_t>(n + 4);
    c.cigar = h->cigar.as<uint32_t>(tot[1] + 4);
    c.aux = h->aux.as<uint8_t>(tot[2] + 4);
    if (n) launch_emit(dd, db, nown, first_blk, first_off, (int32_t) n_ref, dc, nr, nc, na, 0ull, 0ull, 0ull, c, entry, total, allow_tail, n, dindex, st);
    HIP_CHECK(hipStreamSynchronize(st));
  }
  else
  {
    // no record starts in this part: the columns exist all the same, with their end entries
    c.tid = h->tid.as<int32_t>(4);
    c.pos = h->pos.as<int32_t>(4);
    c.mtid = h->mtid.as<int32_t>(4);
    c.mpos = h->mpos.as<int32_t>(4);
    c.isize = h->isize.as<int32_t>(4);
    c.flag = h->flag.as<uint16_t>(4);
    c.mapq = h->mapq.as<uint8_t>(4);
    c.qhash = h->qhash.as<uint64_t>(4);
    c.qcheck = h->qcheck.as<uint32_t>(4);
    c.cigar_off = h->cigar_off.as<uint32_t>(4);
    c.aux_off = h->aux_off.as<uint32_t>(4);
    c.cigar = h->cigar.as<uint32_t>(4);
    c.aux = h->aux.as<uint8_t>(4);
  }
  const uint64_t n = tot[0];
  const uint32_t ends[2] = {(uint32_t) tot[1], (uint32_t) tot[2]};
  HIP_CHECK(hipMemcpy(c.cigar_off + n, &ends[0], 4, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(c.aux_off + n, &ends[1], 4, hipMemcpyHostToDevice));
  memset(cols, 0, sizeof *cols);
  cols->n = n;
  cols->tid = c.tid; cols->pos = c.pos; cols->mtid = c.mtid; cols->mpos = c.mpos; cols->isize = c.isize;
  cols->flag = c.flag; cols->mapq = c.mapq; cols->qhash = c.qhash; cols->qcheck = c.qcheck;
  cols->cigar_off = c.cigar_off; cols->cigar = c.cigar; cols->aux_off = c.aux_off; cols->aux = c.aux;
  cols->n_cigar_words = (uint32_t) tot[1];
  cols->n_aux_bytes = (uint32_t) tot[2];
  if (bk_debug("feed"))
    fprintf(stderr, "[feed/gpu] part %d of %d: %llu records from %u BGZF blocks (+ %u behind the part), records across blocks (boundaries guessed and verified): file -> device table %.3f s\n", part, parts,
            (unsigned long long) n, nown, nblk - nown, now_s2() - t0);
}

// Growing device columns of the record table.
struct ColumnSink
{
  bk_bam_dev *h;
  BamCols c = {};
  uint64_t n_rec = 0, n_cig = 0, n_aux = 0, cap_rec = 0, cap_cig = 0, cap_aux = 0;
  // room for (r, g, a) records / CIGAR words / aux bytes; `quiesce` must wait for every kernel that writes the columns
  template <class F> void reserve(uint64_t r, uint64_t g, uint64_t a, F quiesce)
  {
    if (r <= cap_rec && g <= cap_cig && a <= cap_aux) return;
    if (cap_rec || cap_cig || cap_aux) quiesce();  // (the first allocation moves nothing: no kernel writes the columns yet)
    if (r > cap_rec)
    {
      const uint64_t nc = std::max(r, cap_rec + cap_rec / 2) + 1024;
      for (DevBuf *b : {&h->tid, &h->pos, &h->mtid, &h->mpos, &h->isize, &h->cigar_off, &h->aux_off}) grow_keep(*b, n_rec * 4, (nc + 4) * 4);
      grow_keep(h->flag, n_rec * 2, (nc + 4) * 2);
      grow_keep(h->mapq, n_rec, nc + 4);
      grow_keep(h->qhash, n_rec * 8, (nc + 4) * 8);
        grow_keep(h->qcheck, n_rec * 4, (nc + 4) * 4);
      cap_rec = nc;
    }
    if (g > cap_cig)
    {
      const uint64_t nc = std::max(g, cap_cig + cap_cig / 2) + 1024;
      grow_keep(h->cigar, n_cig * 4, (nc + 4) * 4);
      cap_cig = nc;
    }
    if (a > cap_aux)
    {
      const uint64_t nc = std::max(a, cap_aux + cap_aux / 2) + 1024;
      grow_keep(h->aux, n_aux, nc + 4);
      cap_aux = nc;
    }
    c.tid = h->tid.get<int32_t>();
    c.pos = h->pos.get<int32_t>();
    c.mtid = h->mtid.get<int32_t>();
    c.mpos = h->mpos.get<int32_t>();
    c.isize = h->isize.get<int32_t>();
    c.flag = h->flag.get<uint16_t>();
    c.mapq = h->mapq.get<uint8_t>();
    c.qhash = h->qhash.get<uint64_t>();
      c.qcheck = h->qcheck.get<uint32_t>();
    c.cigar_off = h->cigar_off.get<uint32_t>();
    c.aux_off = h->aux_off.get<uint32_t>();
    c.cigar = h->cigar.get<uint32_t>();
    c.aux = h->aux.get<uint8_t>();
  }
  void finish(bk_soa *cols)
  {
    const uint32_t ends[2] = {(uint32_t) n_cig, (uint32_t) n_aux};
    HIP_CHECK(hipMemcpy(c.cigar_off + n_rec, &ends[0], 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(c.aux_off + n_rec, &ends[1], 4, hipMemcpyHostToDevice));
    memset(cols, 0, sizeof *cols);
    cols->n = n_rec;
    cols->tid = c.tid; cols->pos = c.pos; cols->mtid = c.mtid; cols->mpos = c.mpos; cols->isize = c.isize;
    cols->flag = c.flag; cols->mapq = c.mapq; cols->qhash = c.qhash; cols->qcheck = c.qcheck;
    cols->cigar_off = c.cigar_off; cols->cigar = c.cigar; cols->aux_off = c.aux_off; cols->aux = c.aux;
    cols->n_cigar_words = (uint32_t) n_cig;
    cols->n_aux_bytes = (uint32_t) n_aux;
  }
};

// Records across BGZF blocks, file too large for one batch: the file is taken in chunks like the aligned case, every
// chunk is inflated into a contiguous stream behind PACKED_RESERVE free bytes.  The walk of a chunk ends at the record
// that does not fit any more; its bytes (the "carry") are copied in front of the next chunk's stream, whose first block is
// extended backwards over them - so every chunk starts at a KNOWN record boundary, the other blocks guess theirs and
// the chain is verified per chunk exactly as in the one-batch variant.  The record phase of chunk c + 1 therefore waits
// for the totals of chunk c; its H2D copy and inflate do not.
constexpr uint64_t PACKED_RESERVE = 8u << 20;  // longest record that may cross a chunk boundary

struct PackedSlot
{
  DevBuf dfile, dblk, dblk2, ddata, dslab, dcnt, dnr, dnc, dna, dscan, derr, dentry, dnext, dtail, dindex;
  hipStream_t st = nullptr;
  hipEvent_t ev_emit = nullptr;
  uint64_t *tot = nullptr;  // pinned: records, CIGAR words, aux bytes, error flags, tail
  std::vector<BgzfBlock> blocks, blocks2;  // as inflated; as walked (first block extended over the carry)
  uint64_t total = 0;       // bytes of the chunk's inflated stream
  uint64_t file_hi = 0;     // file offset behind the chunk
  uint32_t first_blk = 0, first_off = 0;
  bool used = false;
  ~PackedSlot()
  {
    if (st) (void) hipStreamDestroy(st);
    if (ev_emit) (void) hipEventDestroy(ev_emit);
    if (tot) (void) hipHostFree(tot);
  }
};

static SlotCacheOf<PackedSlot> &packed_slot_cache()
{
  static SlotCacheOf<PackedSlot> *c = new SlotCacheOf<PackedSlot>();  // never destroyed: the buffers go with the process
  return *c;
}

extern "C" void bk_feed_release_caches(void)
{
  int dev = 0;
  const bool have = hipGetDevice(&dev) == hipSuccess;
  slot_cache().release_all();
  packed_slot_cache().release_all();
  stage_cache().release_all();
  if (have) (void) hipSetDevice(dev);
}

static void decode_packed_chunked(const MappedFile &file, int device, bk_bam_dev *h, bk_soa *cols)
{
  (void) device;
  const double t0 = now_s2();
  uint64_t chunk_bytes = 64ull << 20;
  if (const char *e = getenv("BREAKID_FEED_CHUNK_MB"))
    if (atof(e) > 0) chunk_bytes = (uint64_t) (atof(e) * 1048576.0);
  chunk_bytes = std::max<uint64_t>(chunk_bytes, 70000) / 4096 * 4096 + 4096;  // a chunk is longer than the longest block
  // the inflates run LAG chunks ahead of the record phases (which follow each other: a chunk starts with the carry of the one before)
  constexpr int NS_MAX = 8;
  const int NS = 4, LAG = 2;
  SlotCacheOf<PackedSlot> *packed_cache = &packed_slot_cache();  // as slot_cache(): a finished decode leaves its slots to the next file
  const bool keep_slots = true;
  struct SlotSet
  {
    std::unique_ptr<PackedSlot> p[NS_MAX];
    PackedSlot &operator[](size_t k) { return *p[k]; }
  } slot;
  for (int k = 0; k < NS; ++k)
  {
    slot.p[k] = keep_slots ? packed_cache->take(device) : std::unique_ptr<PackedSlot>(new PackedSlot());
    PackedSlot &s = slot[k];
    if (!s.st) HIP_CHECK(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
    if (!s.ev_emit) HIP_CHECK(hipEventCreateWithFlags(&s.ev_emit, hipEventDisableTiming));
    if (!s.tot) HIP_CHECK(hipHostMalloc((void **) &s.tot, 8 * sizeof(uint64_t), hipHostMallocDefault));
  }
  hipEvent_t ev_carry;
  HIP_CHECK(hipEventCreateWithFlags(&ev_carry, hipEventDisableTiming));
  struct EvGuard
  {
    hipEvent_t e;
    ~EvGuard() { (void) hipEventDestroy(e); }
  } evg{ev_carry};
  DevBuf dcarry;
  uint8_t *carry = dcarry.as<uint8_t>(PACKED_RESERVE);
  uint64_t carry_len = 0, first_in_off = 0, off = 0, nblk_all = 0, nchunk = 0;
  uint32_t hdr_first_off = 0, n_ref = 0;
  parse_bam_header_of_file(file.data(), file.size(), h, n_ref, first_in_off, hdr_first_off);
  ColumnSink sink{h};
  std::string why;
  double t_stage = 0, t_records = 0, t_sync = 0;
  auto quiesce = [&]() {
    for (int k = 0; k < NS; ++k) HIP_CHECK(hipStreamSynchronize(slot[k].st));
  };
  int copy_threads = 8;
  if (const char *e = getenv("BREAKID_THREADS"))
    if (atoi(e) > 0) copy_threads = std::min(atoi(e), 16);
  // the bytes of chunk k = file range [k C, (k + 1) C + slack) arrive in page-locked staging buffers, read() by the pool's threads
  // (the first three straight from the mapping while the buffers are being registered) - as in the aligned case
  const uint64_t nchunks = (file.size() + chunk_bytes - 1) / chunk_bytes;
  StagePool pool(file.data(), file.descriptor(), file.size(), chunk_bytes, copy_threads, device, 0, nchunks, 0, file.size(), 1);
  // chunk k -> slot: the blocks that start inside it: hop over their headers in the staged bytes, copy, inflate behind the reserve
  auto stage = [&](PackedSlot &s, uint64_t k) {
    if (s.used) HIP_CHECK(hipEventSynchronize(s.ev_emit));
    s.used = true;
    s.blocks.clear();
    s.total = 0;
    s.first_blk = 0;
    s.first_off = 0;
    StagePool::Buf *sb = k >= pool.first ? &pool.get(k) : nullptr;
    const uint64_t src_lo = k * chunk_bytes, src_n = std::min<uint64_t>(file.size() - src_lo, chunk_bytes + StagePool::SLACK);
    const uint8_t *fdata = sb ? sb->p : file.data() + src_lo;
    uint64_t rel = off - src_lo;  // offsets are relative to the start of the range from here on
    if (sb && sb->scanned)
    {
      s.blocks.swap(sb->blocks);
      s.total = sb->total;
      rel = sb->rel_end;
    }
    else if (rel < chunk_bytes && !bgzf_scan_range(fdata, src_n, rel, chunk_bytes - rel, s.blocks, s.total, why, 1))
      throw bk_error(BK_ERR_IO, why);
    off = src_lo + rel;
    const uint32_t nb = (uint32_t) s.blocks.size();
    nblk_all += nb;
    while (s.first_blk < nb && src_lo + s.blocks[s.first_blk].in_off < first_in_off) ++s.first_blk;
    if (s.first_blk < nb && src_lo + s.blocks[s.first_blk].in_off == first_in_off) s.first_off = hdr_first_off;
    for (auto &b : s.blocks) b.out_off += PACKED_RESERVE;
    s.file_hi = off;
    if (nb == 0)
    {
      if (sb) pool.release(*sb, s.st);
      return;
    }
    uint8_t *df = s.dfile.as<uint8_t>(rel + 8);
    BgzfBlock *db = s.dblk.as<BgzfBlock>((uint64_t) nb + 1);
    uint8_t *dd = s.ddata.as<uint8_t>(PACKED_RESERVE + s.total + 64);
    uint32_t *de = s.derr.as<uint32_t>(1);
    uint8_t *slab = s.dslab.as<uint8_t>(bgzf_scratch_bytes(nb));
    HIP_CHECK(hipMemcpyAsync(df, fdata, rel, hipMemcpyHostToDevice, s.st));
    if (sb) pool.release(*sb, s.st);
    HIP_CHECK(hipMemcpyAsync(db, s.blocks.data(), (size_t) nb * sizeof(BgzfBlock), hipMemcpyHostToDevice, s.st));
    HIP_CHECK(hipMemsetAsync(de, 0, 4, s.st));
    HIP_CHECK(hipMemsetAsync(dd + PACKED_RESERVE + s.total, 0, 64, s.st));
    launch_bgzf_inflate(df, db, nb, dd, slab, de, s.st, true);
  };
  // the carry of the chunk before is known: boundaries, counts and totals of this chunk
  auto records = [&](PackedSlot &s, bool last) {
    const uint32_t nb = (uint32_t) s.blocks.size();
    if (nb == 0) return;
    uint8_t *dd = s.ddata.get<uint8_t>();
    std::vector<BgzfBlock> &b2 = s.blocks2;
    b2 = s.blocks;
    // the block that holds the first record is extended backwards over the carried bytes
    if (carry_len)
    {
      if (s.first_blk >= nb) throw bk_error(BK_ERR_IO, "corrupt BAM record");
      HIP_CHECK(hipStreamWaitEvent(s.st, ev_carry, 0));
      HIP_CHECK(hipMemcpyAsync(dd + PACKED_RESERVE - carry_len, carry, carry_len, hipMemcpyDeviceToDevice, s.st));
      // blocks before it hold no record starts (they are empty: isize 0)
      b2[s.first_blk].out_off -= carry_len;
      b2[s.first_blk].isize += (uint32_t) carry_len;
    }
    BgzfBlock *db2 = s.dblk2.as<BgzfBlock>((uint64_t) nb + 1);
    HIP_CHECK(hipMemcpyAsync(db2, b2.data(), (size_t) nb * sizeof(BgzfBlock), hipMemcpyHostToDevice, s.st));
    const uint64_t total = PACKED_RESERVE + s.total;
    uint32_t *de = s.derr.get<uint32_t>();
    uint32_t *entry = s.dentry.as<uint32_t>((uint64_t) nb + 1);
    uint64_t *next_abs = s.dnext.as<uint64_t>((uint64_t) nb + 1);
    unsigned long long *tail = s.dtail.as<unsigned long long>(1);
    BlockCount *dc = s.dcnt.as<BlockCount>((uint64_t) nb + 1);
    uint64_t *nr = s.dnr.as<uint64_t>((uint64_t) nb + 1), *nc = s.dnc.as<uint64_t>((uint64_t) nb + 1), *na = s.dna.as<uint64_t>((uint64_t) nb + 1);
    s.tot[5] = total;  // (no record starts in the chunk: nothing to carry)
    HIP_CHECK(hipMemcpyAsync(tail, &s.tot[5], 8, hipMemcpyHostToDevice, s.st));
    BamCols none = {};
    hipLaunchKernelGGL(k_bam_guess, dim3(nb), dim3(64), 0, s.st, dd, db2, nb, s.first_blk, s.first_off, (int32_t) n_ref, total, entry);
    hipLaunchKernelGGL(k_bam_blocks<false>, dim3(cdiv(nb, 64)), dim3(64), 0, s.st, dd, db2, nb, s.first_blk, s.first_off, (int32_t) n_ref, dc, nullptr, nullptr, nullptr, 0ull, 0ull, 0ull, none,
                       entry, total, next_abs, last ? 0 : 1);
    hipLaunchKernelGGL(k_bam_count_split, dim3(cdiv(nb, 256)), dim3(256), 0, s.st, dc, nb, nr, nc, na, de);
    hipLaunchKernelGGL(k_bam_verify, dim3(cdiv(nb, 256)), dim3(256), 0, s.st, db2, nb, s.first_blk, entry, next_abs, total, de, last ? 0 : 1, tail);
    prims::exclusive_scan<unsigned long long>((unsigned long long *) nr, (unsigned long long *) nr, nb, s.dscan, s.st);
    prims::exclusive_scan<unsigned long long>((unsigned long long *) nc, (unsigned long long *) nc, nb, s.dscan, s.st);
    prims::exclusive_scan<unsigned long long>((unsigned long long *) na, (unsigned long long *) na, nb, s.dscan, s.st);
    s.tot[3] = 0;
    HIP_CHECK(hipMemcpyAsync(&s.tot[0], nr + nb, 8, hipMemcpyDeviceToHost, s.st));
    HIP_CHECK(hipMemcpyAsync(&s.tot[1], nc + nb, 8, hipMemcpyDeviceToHost, s.st));
    HIP_CHECK(hipMemcpyAsync(&s.tot[2], na + nb, 8, hipMemcpyDeviceToHost, s.st));
    HIP_CHECK(hipMemcpyAsync(&s.tot[3], de, 4, hipMemcpyDeviceToHost, s.st));
    HIP_CHECK(hipMemcpyAsync(&s.tot[4], tail, 8, hipMemcpyDeviceToHost, s.st));
    const double tsy = now_s2();
    HIP_CHECK(hipStreamSynchronize(s.st));
    t_sync += now_s2() - tsy;
    if (s.tot[3] & 1u) throw bk_error(BK_ERR_IO, "inflate failed");
    if (s.tot[3] & 2u) throw bk_error(BK_ERR_IO, "corrupt BAM record");
    if (s.tot[3] & 4u) throw bk_error(BK_ERR_IO, "the record boundaries of this BAM could not be established on the GPU: use the host decoder");
    // what is left of the stream behind the last complete record travels to the next chunk
    const uint64_t new_carry = total - s.tot[4];
    if (new_carry > PACKED_RESERVE) throw bk_error(BK_ERR_LIMIT, "a BAM record longer than 8 MiB crosses a feed chunk: use the host decoder");
    if (last && new_carry) throw bk_error(BK_ERR_IO, "truncated BAM record at the end of the file");
    if (new_carry)
    {
      HIP_CHECK(hipMemcpyAsync(carry, dd + s.tot[4], new_carry, hipMemcpyDeviceToDevice, s.st));
      HIP_CHECK(hipEventRecord(ev_carry, s.st));
    }
    carry_len = new_carry;
    const uint64_t r = sink.n_rec + s.tot[0], g = sink.n_cig + s.tot[1], a = sink.n_aux + s.tot[2];
    if (r >= 0xFFFFFFF0ull || g >= 0xFFFFFFF0ull || a >= 0xFFFFFFF0ull) throw bk_error(BK_ERR_LIMIT, "more than 2^32 records / CIGAR words / SA bytes in one BAM");
    if (sink.cap_rec == 0 && !last)
    {
      const double scale = 1.05 * (double) file.size() / (double) std::max<uint64_t>(s.file_hi, 1);
      sink.reserve((uint64_t) (r * scale), (uint64_t) (g * scale), (uint64_t) (a * scale), quiesce);
    }
    sink.reserve(r, g, a, quiesce);
    launch_emit(dd, db2, nb, s.first_blk, s.first_off, (int32_t) n_ref, dc, nr, nc, na, sink.n_rec, sink.n_cig, sink.n_aux, sink.c, entry, total, last ? 0 : 1, s.tot[0], s.dindex, s.st);
    HIP_CHECK(hipEventRecord(s.ev_emit, s.st));
    sink.n_rec = r;
    sink.n_cig = g;
    sink.n_aux = a;
  };
  // the inflates of chunks c + 1 .. c + LAG are queued before the host waits for the totals of chunk c
  nchunk = nchunks;
  const double t_loop0 = now_s2();
  for (uint64_t k = 0; k < nchunks + LAG; ++k)
  {
    const double ta = now_s2();
    if (k < nchunks) stage(slot[k % NS], k);
    const double tb = now_s2();
    t_stage += tb - ta;
    // (the last chunk with blocks is the one whose hop reached the end of the file: chunks behind it hold no block start)
    if (k >= (uint64_t) LAG) records(slot[(k - LAG) % NS], slot[(k - LAG) % NS].file_hi == file.size());
    t_records += now_s2() - tb;
  }
  const double t_loop1 = now_s2();
  if (off != file.size()) throw bk_error(BK_ERR_IO, "BGZF blocks do not end at the end of the file");
  if (carry_len) throw bk_error(BK_ERR_IO, "truncated BAM record at the end of the file");
  sink.reserve(sink.n_rec, sink.n_cig, sink.n_aux, quiesce);
  quiesce();
  sink.finish(cols);
  pool.shutdown();
  if (keep_slots)
    for (int k = 0; k < NS; ++k) packed_cache->give(device, std::move(slot.p[k]));  // (every stream is idle: quiesce above)
  if (bk_debug("feed"))
    fprintf(stderr, "[feed/gpu] %llu records, %.1f MB file, %llu BGZF blocks in %llu chunks, records across blocks (boundaries guessed and verified per chunk): file -> device table %.3f s (driver thread: before the loop %.3f s, staging calls %.3f s, record phases %.3f s of which waiting for the chunk's totals %.3f s, after the loop %.3f s)\n",
            (unsigned long long) sink.n_rec, file.size() / 1e6, (unsigned long long) nblk_all, (unsigned long long) nchunk, now_s2() - t0, t_loop0 - t0, t_stage, t_records, t_sync, now_s2() - t_loop1);
}

// Does the first block of records end with a record?  (htslib never lets a record leave its block, htsjdk does; the
// chunked decoder checks every block anyway, this only picks the path that is tried first.)
static bool first_block_is_record_aligned(const MappedFile &file)
{
  try
  {
    bk_bam_dev tmp;
    uint32_t n_ref = 0, first_off = 0;
    uint64_t first_in_off = 0;
    parse_bam_header_of_file(file.data(), file.size(), &tmp, n_ref, first_in_off, first_off);
    if (first_in_off == ~0ull) return true;
    // the BGZF block whose deflate stream starts at first_in_off: its header ends there (12 + XLEN bytes: 18 for BGZF)
    std::vector<BgzfBlock> blocks;
    uint64_t total = 0, off = first_in_off - 18;
    std::string why;
    if (first_in_off < 18 || !bgzf_scan_range(file.data(), file.size(), off, 1, blocks, total, why) || blocks.empty() || blocks[0].in_off != first_in_off) return true;
    std::vector<uint8_t> d;
    if (!host_inflate_block(file.data(), blocks[0], d)) return true;
    size_t p = first_off;
    while (p + 4 <= d.size()) p += 4 + (size_t) rd32h(d.data() + p);
    return p == d.size();
  }
  catch (const bk_error &)
  {
    return true;  // the decoder proper reports what is wrong with the file
  }
}

extern "C" int bk_bam_decode_device(const char *path, int device, bk_bam_dev **out, bk_soa *cols, int *n_targets, const char *const **names, const uint32_t **lens,
                                    char *err, size_t errlen)
{
  return bam_decode_device_impl(path, device, out, cols, n_targets, names, lens, err, errlen, nullptr, 0, 1);
}

int bam_decode_device_impl(const char *path, int device, bk_bam_dev **out, bk_soa *cols, int *n_targets, const char *const **names, const uint32_t **lens, char *err,
                           size_t errlen, const FeedConsumer *fc, int part, int parts)
{
  bk_bam_dev *h = nullptr;
  try
  {
    if (!path || !out || !cols) throw bk_error(BK_ERR_ARG, "bk_bam_decode_device: null argument");
    *out = nullptr;
    HIP_CHECK(hipSetDevice(device));
    MappedFile file(path);  // mapped, not read: the header hop touches 18 bytes per block and the H2D copies stream the rest
    h = new bk_bam_dev();
    if (parts < 1 || part < 0 || part >= parts) throw bk_error(BK_ERR_ARG, "bk_bam_decode_device_part: part must lie in [0, parts)");
    bool packed = !first_block_is_record_aligned(file);
    if (!packed)
    {
      try
      {
        decode_chunked(file, device, h, cols, fc, part, parts);
      }
      catch (const not_block_aligned &)
      {
        HIP_CHECK(hipDeviceSynchronize());
        if (fc && fc->on_reset) fc->on_reset(fc->user);
        delete h;
        h = new bk_bam_dev();
        packed = true;
      }
    }
    if (packed && parts > 1)
      decode_packed_part(file, device, h, cols, part, parts);  // (one batch per part: a rank's part of a file fits its GPU)
    else if (packed)
    {
      // in chunks (the faster variant since its inflates run ahead of the record phases: 75-93 ms against 170 for the 1 GB test
      // file); as one batch - file image + inflated stream in HBM - when a record longer than the 8 MiB reserve crosses a chunk
      // (BK_ERR_LIMIT) or BREAKID_FEED_PACKED_BATCH=1 asks for it; BREAKID_FEED_PACKED_CHUNKS=1: in chunks or not at all
      const bool force_chunks = getenv("BREAKID_FEED_PACKED_CHUNKS") != nullptr;
      bool batch = !force_chunks && getenv("BREAKID_FEED_PACKED_BATCH") != nullptr;
      if (!batch)
      {
        try
        {
          decode_packed_chunked(file, device, h, cols);
        }
        catch (const bk_error &e)
        {
          if (e.code != BK_ERR_LIMIT || force_chunks) throw;
          HIP_CHECK(hipDeviceSynchronize());
          delete h;
          h = new bk_bam_dev();
          batch = true;
        }
      }
      if (batch) decode_packed(file, device, h, cols);
    }
    if (packed && fc && fc->on_header) fc->on_header(fc->user, (int) h->names.size(), h->name_ptrs.data(), h->lens.data());
    if (n_targets) *n_targets = (int) h->names.size();
    if (names) *names = h->name_ptrs.data();
    if (lens) *lens = h->lens.data();
    *out = h;
    return BK_OK;
  }
  catch (const bk_error &ex)
  {
    (void) hipDeviceSynchronize();  // nothing of a failed decode is still running when its buffers go
    delete h;
    if (err && errlen) snprintf(err, errlen, "%s", ex.what());
    return ex.code;
  }
}

extern "C" int bk_bam_decode_device_part(const char *path, int device, int part, int parts, bk_bam_dev **out, bk_soa *cols, int *n_targets, const char *const **names,
                                         const uint32_t **lens, char *err, size_t errlen)
{
  return bam_decode_device_impl(path, device, out, cols, n_targets, names, lens, err, errlen, nullptr, part, parts);
}

extern "C" void bk_bam_dev_free(bk_bam_dev *h) { delete h; }
