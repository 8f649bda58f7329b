// Host feed: BGZF/BAM decoder -> columnar record table (include/breakid_hip.h: bk_soa).
// Own implementation (zlib inflate only); replaces the htslib reader the reference uses for its two
// sequential passes (BreakID.cc:1414 samread, :1929 sam_read1).  Record layout: SAM spec §4.2 /
// htslib/sam.h:148-181; aux walk as sam.c:1267-1279 (bam_aux_get).
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/breakid_hip.h"

struct bk_bam
{
  std::string path;
  std::vector<uint8_t> data;  // inflated stream
  size_t rec_begin = 0;
  std::vector<std::string> names;
  std::vector<const char *> name_ptrs;
  std::vector<uint32_t> lens;
  // decoded columns
  std::vector<int32_t> tid, pos, mtid, mpos, isize;
  std::vector<uint16_t> flag;
  std::vector<uint8_t> mapq;
  std::vector<uint64_t> qhash;
  std::vector<uint32_t> cigar_off, cigar, aux_off;
  std::vector<uint8_t> aux;
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

namespace
{
void set_err(char *err, size_t errlen, const std::string &m)
{
  if (err && errlen) snprintf(err, errlen, "%s", m.c_str());
}
inline uint32_t rd32(const uint8_t *p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24); }
inline uint16_t rd16(const uint8_t *p) { return (uint16_t) (p[0] | (p[1] << 8)); }

bool inflate_all(const std::vector<uint8_t> &file, std::vector<uint8_t> &out, std::string &why)
{
  size_t off = 0;
  out.clear();
  out.reserve(file.size() * 4);
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
    const uint8_t *x = h + 12;
    int bsize = -1;
    for (size_t k = 0; k + 4 <= xlen;)
    {
      uint16_t slen = rd16(x + k + 2);
      if (x[k] == 66 && x[k + 1] == 67 && slen == 2) bsize = rd16(x + k + 4);
      k += 4 + slen;
    }
    if (bsize < 0 || off + (size_t) bsize + 1 > file.size())
    {
      why = "bad BGZF block size";
      return false;
    }
    size_t cdata = 12 + xlen, clen = (size_t) bsize + 1 - cdata - 8;
    uint32_t isize = rd32(h + bsize + 1 - 4);
    size_t base = out.size();
    out.resize(base + isize);
    if (isize)
    {
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      if (inflateInit2(&zs, -15) != Z_OK)
      {
        why = "inflateInit2 failed";
        return false;
      }
      zs.next_in = const_cast<Bytef *>(h + cdata);
      zs.avail_in = (uInt) clen;
      zs.next_out = out.data() + base;
      zs.avail_out = isize;
      int rc = inflate(&zs, Z_FINISH);
      inflateEnd(&zs);
      if (rc != Z_STREAM_END)
      {
        why = "inflate failed";
        return false;
      }
    }
    off += (size_t) bsize + 1;
  }
  return true;
}
}  // namespace

extern "C" int bk_bam_open(const char *path, bk_bam **out, char *err, size_t errlen)
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
  bk_bam *b = new bk_bam();
  b->path = path;
  std::string why;
  if (!inflate_all(file, b->data, why))
  {
    set_err(err, errlen, why);
    delete b;
    return BK_ERR_IO;
  }
  const std::vector<uint8_t> &d = b->data;
  if (d.size() < 12 || memcmp(d.data(), "BAM\1", 4) != 0)
  {
    set_err(err, errlen, "not a BAM file");
    delete b;
    return BK_ERR_IO;
  }
  size_t p = 4;
  uint32_t l_text = rd32(d.data() + p);
  p += 4 + l_text;
  if (p + 4 > d.size())
  {
    set_err(err, errlen, "truncated BAM header");
    delete b;
    return BK_ERR_IO;
  }
  uint32_t n_ref = rd32(d.data() + p);
  p += 4;
  for (uint32_t i = 0; i < n_ref; ++i)
  {
    if (p + 4 > d.size()) { set_err(err, errlen, "truncated BAM header"); delete b; return BK_ERR_IO; }
    uint32_t l_name = rd32(d.data() + p);
    p += 4;
    if (p + l_name + 4 > d.size()) { set_err(err, errlen, "truncated BAM header"); delete b; return BK_ERR_IO; }
    b->names.emplace_back((const char *) d.data() + p, l_name ? l_name - 1 : 0);
    p += l_name;
    b->lens.push_back(rd32(d.data() + p));
    p += 4;
  }
  for (auto &s : b->names) b->name_ptrs.push_back(s.c_str());
  b->rec_begin = p;
  *out = b;
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

extern "C" int bk_bam_decode(bk_bam *b, bk_soa *out, char *err, size_t errlen)
{
  if (!b || !out) return BK_ERR_ARG;
  const std::vector<uint8_t> &d = b->data;
  size_t p = b->rec_begin;
  b->cigar_off.assign(1, 0);
  b->aux_off.assign(1, 0);
  while (p + 4 <= d.size())
  {
    uint32_t bs = rd32(d.data() + p);
    p += 4;
    if (bs < 32 || p + bs > d.size())
    {
      set_err(err, errlen, "truncated BAM record");
      return BK_ERR_IO;
    }
    const uint8_t *r = d.data() + p;
    int32_t tid = (int32_t) rd32(r), pos = (int32_t) rd32(r + 4);
    uint8_t l_name = r[8], mq = r[9];
    uint16_t n_cig = rd16(r + 12), fl = rd16(r + 14);
    uint32_t l_seq = rd32(r + 16);
    int32_t mtid = (int32_t) rd32(r + 20), mpos = (int32_t) rd32(r + 24), isz = (int32_t) rd32(r + 28);
    size_t o = 32;
    size_t need = o + l_name + (size_t) n_cig * 4 + (l_seq + 1) / 2 + l_seq;
    if (need > bs)
    {
      set_err(err, errlen, "corrupt BAM record");
      return BK_ERR_IO;
    }
    size_t qn = l_name ? strnlen((const char *) r + o, l_name) : 0;  // bam_get_qname is a C string
    b->qhash.push_back(bk_qname_hash((const char *) r + o, qn));
    o += l_name;
    for (uint16_t k = 0; k < n_cig; ++k) b->cigar.push_back(rd32(r + o + 4 * k));
    o += (size_t) n_cig * 4 + (l_seq + 1) / 2 + l_seq;
    // aux walk: first SA:Z and OC:Z (bam_aux_get returns the first match)
    const uint8_t *sa = nullptr, *oc = nullptr;
    size_t sa_len = 0, oc_len = 0;
    while (o + 3 <= bs)
    {
      const uint8_t *tag = r + o;
      uint8_t type = r[o + 2];
      o += 3;
      size_t len = 0;
      switch (type)
      {
      case 'A': case 'c': case 'C': len = 1; break;
      case 's': case 'S': len = 2; break;
      case 'i': case 'I': case 'f': len = 4; break;
      case 'd': len = 8; break;
      case 'Z': case 'H':
      {
        size_t e = o;
        while (e < bs && r[e]) ++e;
        if (type == 'Z')
        {
          if (!sa && tag[0] == 'S' && tag[1] == 'A') { sa = r + o; sa_len = e - o; }
          if (!oc && tag[0] == 'O' && tag[1] == 'C') { oc = r + o; oc_len = e - o; }
        }
        len = e - o + 1;
        break;
      }
      case 'B':
      {
        if (o + 5 > bs) { o = bs; continue; }
        uint8_t sub = r[o];
        uint32_t cnt = rd32(r + o + 1);
        size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        len = 5 + (size_t) cnt * es;
        break;
      }
      default:
        o = bs;
        continue;
      }
      o += len;
    }
    if (sa && sa_len)
    {
      if (oc && oc_len)
      {
        b->aux.insert(b->aux.end(), oc, oc + oc_len);
        b->aux.push_back('\t');
      }
      b->aux.insert(b->aux.end(), sa, sa + sa_len);
    }
    b->tid.push_back(tid);
    b->pos.push_back(pos);
    b->mtid.push_back(mtid);
    b->mpos.push_back(mpos);
    b->isize.push_back(isz);
    b->flag.push_back(fl);
    b->mapq.push_back(mq);
    b->cigar_off.push_back((uint32_t) b->cigar.size());
    b->aux_off.push_back((uint32_t) b->aux.size());
    p += bs;
  }
  if (b->cigar.empty()) b->cigar.push_back(0);
  if (b->aux.empty()) b->aux.push_back(0);
  memset(out, 0, sizeof *out);
  out->n = b->tid.size();
  out->tid = b->tid.data(); out->pos = b->pos.data(); out->mtid = b->mtid.data(); out->mpos = b->mpos.data(); out->isize = b->isize.data();
  out->flag = b->flag.data(); out->mapq = b->mapq.data(); out->qhash = b->qhash.data();
  out->cigar_off = b->cigar_off.data(); out->cigar = b->cigar.data(); out->aux_off = b->aux_off.data(); out->aux = b->aux.data();
  out->n_cigar_words = b->cigar_off.back();
  out->n_aux_bytes = b->aux_off.back();
  return BK_OK;
}

extern "C" void bk_bam_close(bk_bam *b) { delete b; }
