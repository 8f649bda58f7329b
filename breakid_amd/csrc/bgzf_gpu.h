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
constexpr uint32_t BGZF_BATCH_BLOCKS = 16384;          // blocks per decode/resolve launch pair
constexpr uint32_t BGZF_TOKENS_PER_BLOCK = 21846;      // a match produces >= 3 bytes: at most 65536 / 3 per block
// scratch of launch_bgzf_inflate: 8-byte match tokens + one counter per block of a batch
uint32_t bgzf_scratch_blocks(uint32_t nblk);
inline uint64_t bgzf_scratch_bytes(uint32_t nblk) { return (uint64_t) bgzf_scratch_blocks(nblk) * ((uint64_t) BGZF_TOKENS_PER_BLOCK * 8u + 4u) + 64u; }
// decode (one wavefront per block) + resolve (one workgroup per block); scratch_dev = bgzf_scratch_bytes(nblk) bytes;
// file_dev must be 4-byte aligned with >= 4 readable bytes after the last block; *err_dev |= 1 when a block is malformed
// or does not produce isize bytes
// shared_gpu: other chunks' inflate launches run beside this one (the streaming feed) - see k_bgzf_decode_shared
void launch_bgzf_inflate(const uint8_t *file_dev, const BgzfBlock *blk_dev, uint32_t nblk, uint8_t *out_dev, void *scratch_dev, uint32_t *err_dev, hipStream_t st,
                         bool shared_gpu = false);

// A consumer of the chunked GPU feed (bam_gpu.hip: decode_chunked): the stream pass of a context runs on the records of a
// chunk while the next chunks are still being copied and inflated (bk_bam_decode_device_ctx, api.hip).
struct FeedConsumer
{
  void *user = nullptr;
  // the reference list is known (before the first record): the consumer creates its context
  void (*on_header)(void *user, int n_targets, const char *const *names, const uint32_t *lens) = nullptr;
  // records [0, n_ready) of the device columns are final once `ready` (recorded on the producing stream) has completed;
  // n_est_total = the decoder's estimate of the file's record count (sizes the consumer's outputs)
  void (*on_chunk)(void *user, const bk_soa *cols_dev, uint64_t n_ready, uint64_t n_est_total, hipEvent_t ready) = nullptr;
  // the columns are about to move (they grow by copying): the consumer stops reading them before this returns
  void (*before_move)(void *user) = nullptr;
  // the decode starts over with another strategy (records across BGZF blocks): forget everything consumed so far
  void (*on_reset)(void *user) = nullptr;
};
int bam_decode_device_impl(const char *path, int device, bk_bam_dev **out, bk_soa *cols, int *n_targets, const char *const **names, const uint32_t **lens, char *err,
                           size_t errlen, const FeedConsumer *fc, int part = 0, int parts = 1);
