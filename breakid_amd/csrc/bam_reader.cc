// Host feed: BGZF/BAM decoder -> pinned columnar record table (include/breakid_hip.h: bk_soa).
// Own implementation (zlib inflate only); replaces the htslib reader the reference uses for its two
// sequential passes (BreakID.cc:1414 samread, :1929 sam_read1).  Record layout: SAM spec §4.2 /
// htslib/sam.h:148-181; aux walk as sam.c:1267-1279 (bam_aux_get).
//
// BGZF blocks are independent deflate streams and BAM records are self-delimiting, so both stages run on all host
// cores: (1) block headers are hopped sequentially (18 bytes each), the blocks are inflated in parallel into one
// buffer; (2) record starts are hopped sequentially (4 bytes each) into checkpoints every CHUNK records, the
// chunks are decoded in parallel straight into the fixed-width columns, CIGAR words / aux blobs go through
// chunk-local buffers and are placed by a prefix sum.  Columns live in pinned host memory (hipHostMalloc) when a
// GPU is present, so bk_upload_records(BK_MEM_HOST) runs at PCIe speed.  BREAKID_THREADS overrides the thread count.
#include <hip/hip_runtime_api.h>
#include <zlib.h>

#include <atomic>
#include <memory>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "bk_debug.h"
#include <thread>
#include <vector>

#include "../../include/breakid_hip.h"
#include "bk_hash.h"

namespace
{
struct HostBuf
{
  void *p = nullptr;
  bool pinned = false;
  size_t bytes = 0;
  HostBuf() = default;
  HostBuf(const HostBuf &) = delete;
  HostBuf &operator=(const HostBuf &) = delete;
  ~HostBuf() { release(); }
  void release()
  {
    if (!p) return;
    if (pinned)
      (void) hipHostFree(p);
    else
      free(p);
    p = nullptr;
    bytes = 0;
  }
  // pinned when the HIP runtime has a device; plain memory otherwise (the decoder itself is host code)
  bool alloc(size_t n, bool want_pinned)
  {
    release();
    if (n == 0) n = 16;
    if (want_pinned && hipHostMalloc(&p, n, hipHostMallocDefault) == hipSuccess && p)
    {
      pinned = true;
      bytes = n;
      return true;
    }
    (void) hipGetLastError();
    p = malloc(n);
    pinned = false;
    bytes = p ? n : 0;
    return p != nullptr;
  }
  template <class T> T *as() const { return static_cast<T *>(p); }
};

unsigned n_threads()
{
  if (const char *e = getenv("BREAKID_THREADS"))
  {
    int v = atoi(e);
    if (v > 0) return (unsigned) v;
  }
  unsigned h = std::thread::hardware_concurrency();
  if (h == 0) h = 4;
  return h > 64 ? 64 : h;
}

// dynamic scheduling over [0, njobs): fn(job) on up to nt threads
template <class F> void parallel_for(size_t njobs, unsigned nt, F fn)
{
  if (njobs == 0) return;
  if (nt > njobs) nt = (unsigned) njobs;
  if (nt <= 1)
  {
    for (size_t j = 0; j < njobs; ++j) fn(j);
    return;
  }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  th.reserve(nt);
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&]() {
      for (;;)
      {
        size_t j = next.fetch_add(1);
        if (j >= njobs) break;
        fn(j);
      }
    });
  for (auto &t : th) t.join();
}
}  // namespace

struct bk_bam
{
  std::string path;
  std::vector<uint8_t> data;  // inflated stream
  size_t rec_begin = 0;
  std::vector<std::string> names;
  std::vector<const char *> name_ptrs;
  std::vector<uint32_t> lens;
  // decoded columns
  HostBuf tid, pos, mtid, mpos, isize, flag, mapq, qhash, qcheck, cigar_off, cigar, aux_off, aux;
  double t_inflate_s = 0, t_decode_s = 0;
};

extern "C" uint64_t bk_qname_hash(const char *name, size_t len)
{
  uint64_t h = 0xCBF29CE484222325ull;
  for (size_t i = 0; i < len; ++i)
  {
    h ^= (unsigned char) name[i];
    h *= 0x100000001B3ull;
  }
  h ^= h >> 30;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 27;
  h *= 0x94D049BB133111EBull;
  h ^= h >> 31;
  return h;
}

extern "C" uint32_t bk_qname_check(const char *name, size_t len) { return qname_check32((const uint8_t *) name, (uint32_t) len); }

namespace
{
void set_err(char *err, size_t errlen, const std::string &m)
{
  if (err && errlen) snprintf(err, errlen, "%s", m.c_str());
}
inline uint32_t rd32(const uint8_t *p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24); }
inline uint16_t rd16(const uint8_t *p) { return (uint16_t) (p[0] | (p[1] << 8)); }

struct Block
{
  size_t in_off, clen, out_off;
  uint32_t isize;
};

bool inflate_all(const std::vector<uint8_t> &file, std::vector<uint8_t> &out, std::string &why)
{
  // pass 1: hop over the block headers
  std::vector<Block> blocks;
  size_t off = 0, total = 0;
  while (off < file.size())
  {
    if (off + 18 > file.size())
    {
      why = "truncated BGZF header";
      return false;
    }
    const uint8_t *h = file.data() + off;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4))
    {
      why = "not a BGZF block";
      return false;
    }
    uint16_t xlen = rd16(h + 10);
    if (off + 12 + (size_t) xlen > file.size())
    {
      why = "truncated BGZF header";
      return false;
    }
    const uint8_t *x = h + 12;
    int bsize = -1;
    for (size_t k = 0; k + 4 <= xlen;)
    {
      uint16_t slen = rd16(x + k + 2);
      if (x[k] == 66 && x[k + 1] == 67 && slen == 2 && k + 6 <= xlen) bsize = rd16(x + k + 4);
      k += 4 + (size_t) slen;
    }
    if (bsize < 0 || off + (size_t) bsize + 1 > file.size() || (size_t) bsize + 1 < 12 + (size_t) xlen + 8)
    {
      why = "bad BGZF block size";
      return false;
    }
    Block b;
    b.in_off = off + 12 + xlen;
    b.clen = (size_t) bsize + 1 - (12 + (size_t) xlen) - 8;
    b.isize = rd32(h + bsize + 1 - 4);
    if (b.isize > 65536u)  // BGZF: a block inflates to at most 64 KiB (bgzf.h BGZF_MAX_BLOCK_SIZE); same check as the GPU scanner
    {
      why = "BGZF block claims more than 64 KiB of data";
      return false;
    }
    b.out_off = total;
    total += b.isize;
    blocks.push_back(b);
    off += (size_t) bsize + 1;
  }
  out.assign(total, 0);
  // pass 2: independent deflate streams
  std::atomic<int> bad{0};
  parallel_for(blocks.size(), n_threads(), [&](size_t j) {
    const Block &b = blocks[j];
    if (!b.isize) return;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK)
    {
      bad = 1;
      return;
    }
    zs.next_in = const_cast<Bytef *>(file.data() + b.in_off);
    zs.avail_in = (uInt) b.clen;
    zs.next_out = out.data() + b.out_off;
    zs.avail_out = b.isize;
    int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END) bad = 2;
  });
  if (bad)
  {
    why = bad == 1 ? "inflateInit2 failed" : "inflate failed";
    return false;
  }
  return true;
}

double now_s()
{
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}
}  // namespace

static int bam_open_impl(const char *path, bk_bam **out, char *err, size_t errlen)
{
  if (!path || !out) return BK_ERR_ARG;
  *out = nullptr;
  FILE *f = fopen(path, "rb");
  if (!f)
  {
    set_err(err, errlen, std::string("cannot open ") + path);
    return BK_ERR_IO;
  }
  std::vector<uint8_t> file;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  file.resize(sz > 0 ? (size_t) sz : 0);
  if (sz > 0 && fread(file.data(), 1, (size_t) sz, f) != (size_t) sz)
  {
    fclose(f);
    set_err(err, errlen, "short read");
    return BK_ERR_IO;
  }
  fclose(f);
  std::unique_ptr<bk_bam> guard(new bk_bam());  // freed on every error return and on an exception
  bk_bam *b = guard.get();
  b->path = path;
  std::string why;
  const double t0 = now_s();
  if (!inflate_all(file, b->data, why))
  {
    set_err(err, errlen, why);
    return BK_ERR_IO;
  }
  b->t_inflate_s = now_s() - t0;
  std::vector<uint8_t>().swap(file);
  const std::vector<uint8_t> &d = b->data;
  if (d.size() < 12 || memcmp(d.data(), "BAM\1", 4) != 0)
  {
    set_err(err, errlen, "not a BAM file");
    return BK_ERR_IO;
  }
  size_t p = 4;
  uint32_t l_text = rd32(d.data() + p);
  p += 4 + (size_t) l_text;
  if (p + 4 > d.size())
  {
    set_err(err, errlen, "truncated BAM header");
    return BK_ERR_IO;
  }
  uint32_t n_ref = rd32(d.data() + p);
  p += 4;
  for (uint32_t i = 0; i < n_ref; ++i)
  {
    if (p + 4 > d.size()) { set_err(err, errlen, "truncated BAM header"); return BK_ERR_IO; }
    uint32_t l_name = rd32(d.data() + p);
    p += 4;
    if (p + l_name + 4 > d.size()) { set_err(err, errlen, "truncated BAM header"); return BK_ERR_IO; }
    b->names.emplace_back((const char *) d.data() + p, l_name ? l_name - 1 : 0);
    p += l_name;
    b->lens.push_back(rd32(d.data() + p));
    p += 4;
  }
  for (auto &s : b->names) b->name_ptrs.push_back(s.c_str());
  b->rec_begin = p;
  *out = guard.release();
  return BK_OK;
}

extern "C" int bk_bam_header(const bk_bam *b, int *n_targets, const char *const **names, const uint32_t **lens)
{
  if (!b) return BK_ERR_ARG;
  if (n_targets) *n_targets = (int) b->names.size();
  if (names) *names = b->name_ptrs.data();
  if (lens) *lens = b->lens.data();
  return BK_OK;
}

namespace
{
constexpr size_t CHUNK = 1 << 16;  // records per decode job

struct ChunkOut
{
  std::vector<uint32_t> cigar;
  std::vector<uint8_t> aux;
  uint64_t cigar_base = 0, aux_base = 0;
  int bad = 0;
};

struct Cols
{
  int32_t *tid, *pos, *mtid, *mpos, *isize;
  uint16_t *flag;
  uint8_t *mapq;
  uint64_t *qhash;
  uint32_t *qcheck;
  uint32_t *cigar_off, *aux_off;  // chunk-local offsets first, rebased in the placement pass
};

// decode records [r0, r1) starting at byte offset p of the inflated stream
void decode_chunk(const uint8_t *d, size_t dsize, size_t p, size_t r0, size_t r1, const Cols &c, ChunkOut &o)
{
  for (size_t i = r0; i < r1; ++i)
  {
    uint32_t bs = rd32(d + p);
    p += 4;
    const uint8_t *r = d + p;
    uint8_t l_name = r[8];
    uint16_t n_cig = rd16(r + 12);
    uint32_t l_seq = rd32(r + 16);
    size_t q = 32;
    size_t need = q + l_name + (size_t) n_cig * 4 + ((size_t) l_seq + 1) / 2 + l_seq;
    if (need > bs || p + bs > dsize)
    {
      o.bad = 1;
      return;
    }
    c.tid[i] = (int32_t) rd32(r);
    c.pos[i] = (int32_t) rd32(r + 4);
    c.mapq[i] = r[9];
    c.flag[i] = rd16(r + 14);
    c.mtid[i] = (int32_t) rd32(r + 20);
    c.mpos[i] = (int32_t) rd32(r + 24);
    c.isize[i] = (int32_t) rd32(r + 28);
    size_t qn = l_name ? strnlen((const char *) r + q, l_name) : 0;  // bam_get_qname is a C string
    c.qhash[i] = bk_qname_hash((const char *) r + q, qn);
    c.qcheck[i] = bk_qname_check((const char *) r + q, qn);
    q += l_name;
    c.cigar_off[i] = (uint32_t) o.cigar.size();
    for (uint16_t k = 0; k < n_cig; ++k) o.cigar.push_back(rd32(r + q + 4 * (size_t) k));
    q += (size_t) n_cig * 4 + ((size_t) l_seq + 1) / 2 + l_seq;
    // aux walk: first SA:Z and OC:Z (bam_aux_get returns the first match)
    const uint8_t *sa = nullptr, *oc = nullptr;
    size_t sa_len = 0, oc_len = 0;
    while (q + 3 <= bs)
    {
      const uint8_t *tag = r + q;
      uint8_t type = r[q + 2];
      q += 3;
      size_t len = 0;
      switch (type)
      {
      case 'A': case 'c': case 'C': len = 1; break;
      case 's': case 'S': len = 2; break;
      case 'i': case 'I': case 'f': len = 4; break;
      case 'd': len = 8; break;
      case 'Z': case 'H':
      {
        size_t e = q;
        while (e < bs && r[e]) ++e;
        if (type == 'Z')
        {
          if (!sa && tag[0] == 'S' && tag[1] == 'A') { sa = r + q; sa_len = e - q; }
          if (!oc && tag[0] == 'O' && tag[1] == 'C') { oc = r + q; oc_len = e - q; }
        }
        len = e - q + 1;
        break;
      }
      case 'B':
      {
        if (q + 5 > bs) { q = bs; continue; }
        uint8_t sub = r[q];
        uint32_t cnt = rd32(r + q + 1);
        size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        len = 5 + (size_t) cnt * es;
        break;
      }
      default:
        q = bs;
        continue;
      }
      q += len;
    }
    c.aux_off[i] = (uint32_t) o.aux.size();
    if (sa && sa_len)
    {
      if (oc && oc_len)
      {
        o.aux.insert(o.aux.end(), oc, oc + oc_len);
        o.aux.push_back('\t');
      }
      o.aux.insert(o.aux.end(), sa, sa + sa_len);
    }
    p += bs;
  }
}
}  // namespace

static int bam_decode_impl(bk_bam *b, bk_soa *out, char *err, size_t errlen)
{
  if (!b || !out) return BK_ERR_ARG;
  const double t0 = now_s();
  const std::vector<uint8_t> &dv = b->data;
  const uint8_t *d = dv.data();
  const size_t dsize = dv.size();
  // pass 1: record starts, one checkpoint per CHUNK records
  std::vector<size_t> ckpt;
  size_t n = 0, p = b->rec_begin;
  while (p + 4 <= dsize)
  {
    uint32_t bs = rd32(d + p);
    if (bs < 32 || p + 4 + (size_t) bs > dsize)
    {
      set_err(err, errlen, "truncated BAM record");
      return BK_ERR_IO;
    }
    if (n % CHUNK == 0) ckpt.push_back(p);
    ++n;
    p += 4 + (size_t) bs;
  }
  if (n >= 0xFFFFFFF0ull)
  {
    set_err(err, errlen, "more than 2^32 records in one BAM");
    return BK_ERR_LIMIT;
  }
  int ndev = 0;
  const bool pin = hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
  (void) hipGetLastError();
  bool ok = b->tid.alloc(n * 4, pin) && b->pos.alloc(n * 4, pin) && b->mtid.alloc(n * 4, pin) && b->mpos.alloc(n * 4, pin) && b->isize.alloc(n * 4, pin) &&
            b->flag.alloc(n * 2, pin) && b->mapq.alloc(n, pin) && b->qhash.alloc(n * 8, pin) && b->qcheck.alloc(n * 4, pin) && b->cigar_off.alloc((n + 1) * 4, pin) && b->aux_off.alloc((n + 1) * 4, pin);
  if (!ok)
  {
    set_err(err, errlen, "out of host memory for the record table");
    return BK_ERR_IO;
  }
  Cols c{b->tid.as<int32_t>(), b->pos.as<int32_t>(), b->mtid.as<int32_t>(), b->mpos.as<int32_t>(), b->isize.as<int32_t>(), b->flag.as<uint16_t>(),
         b->mapq.as<uint8_t>(), b->qhash.as<uint64_t>(), b->qcheck.as<uint32_t>(), b->cigar_off.as<uint32_t>(), b->aux_off.as<uint32_t>()};
  // pass 2: chunks in parallel
  const size_t nchunks = ckpt.size();
  std::vector<ChunkOut> co(nchunks);
  const unsigned nt = n_threads();
  parallel_for(nchunks, nt, [&](size_t j) {
    const size_t r0 = j * CHUNK, r1 = (r0 + CHUNK < n) ? r0 + CHUNK : n;
    decode_chunk(d, dsize, ckpt[j], r0, r1, c, co[j]);
  });
  uint64_t ncig = 0, naux = 0;
  for (size_t j = 0; j < nchunks; ++j)
  {
    if (co[j].bad)
    {
      set_err(err, errlen, "corrupt BAM record");
      return BK_ERR_IO;
    }
    co[j].cigar_base = ncig;
    co[j].aux_base = naux;
    ncig += co[j].cigar.size();
    naux += co[j].aux.size();
  }
  if (ncig >= 0xFFFFFFF0ull || naux >= 0xFFFFFFF0ull)
  {
    set_err(err, errlen, "CIGAR / SA columns exceed 32-bit offsets");
    return BK_ERR_LIMIT;
  }
  if (!b->cigar.alloc((ncig ? ncig : 1) * 4, pin) || !b->aux.alloc(naux ? naux : 1, pin))
  {
    set_err(err, errlen, "out of host memory for the record table");
    return BK_ERR_IO;
  }
  uint32_t *cig = b->cigar.as<uint32_t>();
  uint8_t *aux = b->aux.as<uint8_t>();
  if (!ncig) cig[0] = 0;
  if (!naux) aux[0] = 0;
  // pass 3: place the variable-length columns, rebase the offsets
  parallel_for(nchunks, nt, [&](size_t j) {
    const size_t r0 = j * CHUNK, r1 = (r0 + CHUNK < n) ? r0 + CHUNK : n;
    ChunkOut &o = co[j];
    if (!o.cigar.empty()) memcpy(cig + o.cigar_base, o.cigar.data(), o.cigar.size() * 4);
    if (!o.aux.empty()) memcpy(aux + o.aux_base, o.aux.data(), o.aux.size());
    const uint32_t cb = (uint32_t) o.cigar_base, ab = (uint32_t) o.aux_base;
    for (size_t i = r0; i < r1; ++i)
    {
      c.cigar_off[i] += cb;
      c.aux_off[i] += ab;
    }
    std::vector<uint32_t>().swap(o.cigar);
    std::vector<uint8_t>().swap(o.aux);
  });
  c.cigar_off[n] = (uint32_t) ncig;
  c.aux_off[n] = (uint32_t) naux;
  memset(out, 0, sizeof *out);
  out->n = n;
  out->tid = c.tid; out->pos = c.pos; out->mtid = c.mtid; out->mpos = c.mpos; out->isize = c.isize;
  out->flag = c.flag; out->mapq = c.mapq; out->qhash = c.qhash; out->qcheck = c.qcheck;
  out->cigar_off = c.cigar_off; out->cigar = cig; out->aux_off = c.aux_off; out->aux = aux;
  out->n_cigar_words = (uint32_t) ncig;
  out->n_aux_bytes = (uint32_t) naux;
  b->t_decode_s = now_s() - t0;
  if (bk_debug("feed"))
    fprintf(stderr, "[feed] %zu records, %.1f MB inflated: inflate %.3f s, decode %.3f s, %u threads, %s host columns\n", n, dsize / 1e6, b->t_inflate_s,
            b->t_decode_s, nt, b->tid.pinned ? "pinned" : "pageable");
  return BK_OK;
}

// no C++ exception crosses the C boundary: a corrupt file (attacker-chosen ISIZE sums, record counts) must come back as an
// error code, not std::terminate
template <class F> static int no_throw(char *err, size_t errlen, F &&f)
{
  try
  {
    return f();
  }
  catch (const std::bad_alloc &)
  {
    set_err(err, errlen, "out of host memory while decoding the BAM");
    return BK_ERR_LIMIT;
  }
  catch (const std::exception &e)
  {
    set_err(err, errlen, e.what());
    return BK_ERR_IO;
  }
}
extern "C" int bk_bam_open(const char *path, bk_bam **out, char *err, size_t errlen)
{
  return no_throw(err, errlen, [&] { return bam_open_impl(path, out, err, errlen); });
}
extern "C" int bk_bam_decode(bk_bam *b, bk_soa *out, char *err, size_t errlen)
{
  return no_throw(err, errlen, [&] { return bam_decode_impl(b, out, err, errlen); });
}
extern "C" void bk_bam_close(bk_bam *b) { delete b; }
