// GPU feed: BGZF file image -> inflated stream -> columnar record table, all on the device (see bgzf_gpu.hip).
#include "bk_common.h"
#include "bgzf_gpu.h"
#include "../../include/breakid_hip.h"
#include <cstring>
#include <string>
#include <vector>

namespace
{
inline uint32_t rd32h(const uint8_t *p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24); }
inline uint16_t rd16h(const uint8_t *p) { return (uint16_t) (p[0] | (p[1] << 8)); }
}  // namespace

// hops over the BGZF block headers of a file image (18 bytes each); false + why on a malformed file.
// total = bytes of the (256-byte aligned per block) device output buffer
bool bgzf_scan_blocks(const uint8_t *file, uint64_t n, std::vector<BgzfBlock> &blocks, uint64_t &total, std::string &why)
{
  blocks.clear();
  total = 0;
  uint64_t off = 0;
  while (off < n)
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
    total += (b.isize + BGZF_OUT_ALIGN - 1) / BGZF_OUT_ALIGN * BGZF_OUT_ALIGN;
    blocks.push_back(b);
    off += (uint64_t) bsize + 1;
  }
  return true;
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
    DevBuf dfile, dblk, dout, derr;
    uint8_t *f = dfile.as<uint8_t>(n + 8);
    BgzfBlock *b = dblk.as<BgzfBlock>(blocks.size() + 1);
    uint8_t *o = dout.as<uint8_t>(total + 8);
    uint32_t *e = derr.as<uint32_t>(1);
    HIP_CHECK(hipMemcpy(f, file, n, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(b, blocks.data(), blocks.size() * sizeof(BgzfBlock), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(e, 0, 4));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, nullptr));
    launch_bgzf_inflate(f, b, (uint32_t) blocks.size(), o, e, nullptr);
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
