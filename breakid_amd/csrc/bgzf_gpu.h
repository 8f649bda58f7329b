// BGZF inflate and BAM record decode on the GPU (bgzf_gpu.hip, bam_gpu.hip).
#pragma once
#include "bk_common.h"

struct BgzfBlock
{
  uint64_t in_off;   // first byte of the deflate stream inside the file image
  uint64_t out_off;  // where the block's bytes go in the output buffer: a multiple of 256 (BGZF_OUT_ALIGN)
  uint32_t clen;     // compressed bytes
  uint32_t isize;    // inflated bytes (<= 65536)
};

constexpr uint64_t BGZF_OUT_ALIGN = 256;
// one wavefront per block; *err_dev |= 1 when a block is malformed or does not produce isize bytes
void launch_bgzf_inflate(const uint8_t *file_dev, const BgzfBlock *blk_dev, uint32_t nblk, uint8_t *out_dev, uint32_t *err_dev, hipStream_t st);
