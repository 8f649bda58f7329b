// Name / text codes shared by the host and device decoders and the stage kernels (no HIP dependency: bam_reader.cc is also
// compiled by plain g++ for the CPU / sanitizer build of the host code, oracle/Makefile).
#pragma once
#include <stdint.h>

// second, independent 32-bit hash of a read name (include/breakid_hip.h: bk_qname_check); shared by host and device decoders
#if defined(__HIPCC__)
#define BK_HD __host__ __device__
#else
#define BK_HD
#endif
BK_HD inline uint32_t qname_check32(const uint8_t *name, uint32_t len)
{
  uint32_t h = 0x811C9DC5u ^ (len * 0x9E3779B1u);
  for (uint32_t i = 0; i < len; ++i)
  {
    h = (h ^ name[i]) * 0x01000193u;
    h = (h << 13) | (h >> 19);
    h = h * 5u + 0xE6546B64u;
  }
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h ? h : 1u;
}
// 64-bit code of a CIGAR text (include/breakid_hip.h, bk_split): exact for <n><M|S><n><M|S>, a 63-bit FNV-1a otherwise
BK_HD inline uint64_t cigar_text_code(const uint8_t *s, uint32_t len)
{
  uint32_t i = 0;
  uint64_t code = 1ull << 63;
  bool exact = true;
  for (int k = 0; k < 2 && exact; ++k)
  {
    uint32_t z = 0;
    while (i < len && s[i] == '0') { ++z; ++i; }
    uint64_t v = 0;
    uint32_t d = 0;
    while (i < len && s[i] >= '0' && s[i] <= '9' && d < 10) { v = v * 10 + (uint64_t) (s[i] - '0'); ++i; ++d; }
    if ((z == 0 && d == 0) || z > 3 || v >= (1ull << 28) || i >= len || (s[i] != 'M' && s[i] != 'S')) { exact = false; break; }
    const uint64_t op = s[i] == 'S' ? 1ull : 0ull;
    ++i;
    code |= k == 0 ? ((uint64_t) z << 60) | (op << 57) | (v << 28) : ((uint64_t) z << 58) | (op << 56) | v;
  }
  if (exact && i == len) return code;
  uint64_t h = 0xCBF29CE484222325ull;
  for (uint32_t j = 0; j < len; ++j) { h ^= s[j]; h *= 0x100000001B3ull; }
  return h & ~(1ull << 63);
}
