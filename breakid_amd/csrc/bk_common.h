// Internal definitions shared by the HIP translation units of libbreakid_hip.so (gfx950 only).
#pragma once
#include <cstdlib>
#include <execinfo.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <mutex>

#include "../../include/breakid_hip.h"
#include "bk_debug.h"

#define BK_WAVE 64

struct bk_error : std::exception
{
  int code;
  std::string msg;
  bk_error(int c, std::string m) : code(c), msg(std::move(m)) {}
  const char *what() const noexcept override { return msg.c_str(); }
};

#define HIP_CHECK(expr)                                                                                         \
  do                                                                                                            \
  {                                                                                                             \
    hipError_t _e = (expr);                                                                                     \
    if (_e != hipSuccess)                                                                                       \
      throw bk_error(BK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + ":" +   \
                                     std::to_string(__LINE__) + ")");                                           \
  } while (0)

// hipFree waits for the device to go idle: while a persistent kernel runs (the resident sort service) that is "until the service
// gives up", so buffers that are outgrown meanwhile are kept and freed when the last service has stopped.
struct DeferredFrees
{
  static inline std::mutex m;
  static inline int active = 0;
  static inline std::vector<void *> ptrs;
  static void begin()
  {
    std::lock_guard<std::mutex> l(m);
    ++active;
  }
  static void end()
  {
    std::vector<void *> now;
    {
      std::lock_guard<std::mutex> l(m);
      if (--active > 0) return;
      now.swap(ptrs);
    }
    for (void *p : now) (void) hipFree(p);
  }
  static bool defer(void *p)
  {
    std::lock_guard<std::mutex> l(m);
    if (active <= 0) return false;
    ptrs.push_back(p);
    return true;
  }
};

// growable device buffer (never shrinks; bench steps reuse the allocation)
struct DevBuf
{
  void *p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap)
  {
    o.p = nullptr;
    o.cap = 0;
  }
  DevBuf &operator=(DevBuf &&o) noexcept
  {
    if (this != &o)
    {
      release();
      p = o.p;
      cap = o.cap;
      o.p = nullptr;
      o.cap = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void release()
  {
    if (p && !DeferredFrees::defer(p)) (void) hipFree(p);
    p = nullptr;
    cap = 0;
  }
  void *ensure(size_t bytes)
  {
    if (bytes > cap)
    {
      release();
      size_t want = bytes + bytes / 8 + 256;
      if (hipMalloc(&p, want) != hipSuccess)
      {
        p = nullptr;
        (void) hipGetLastError();
        if (bk_debug("alloc"))  // (debugging: who asked - resolve the offsets with llvm-symbolizer -e libbreakid_hip.so)
        {
          void *bt[24];
          const int nb = backtrace(bt, 24);
          backtrace_symbols_fd(bt, nb, 2);
        }
        throw bk_error(BK_ERR_HIP, "hipMalloc of " + std::to_string(want) + " bytes failed (out of device memory, or a size that was computed from a bad count)");
      }
      cap = want;
    }
    return p;
  }
  template <class T> T *as(size_t count) { return (T *) ensure(count * sizeof(T)); }
  template <class T> T *get() const { return (T *) p; }
};

static inline unsigned cdiv(uint64_t a, uint64_t b) { return (unsigned) ((a + b - 1) / b); }

// ---- candidate of the discordant filter (BreakID.cc:1419-1420), 32 bytes ------------------------
struct Cand
{
  uint64_t qhash;
  uint64_t rec;  // index of the record in the whole sample (rec_base of its shard + index in the shard)
  int32_t tid, pos, mtid, mpos;
  uint16_t flag;
  uint8_t mapq, pad;
  uint32_t qcheck;  // second hash of the read name (0 = table without a qcheck column)
};
static_assert(sizeof(Cand) == 40, "Cand must be 40 bytes");

#include "bk_hash.h"

// counters written by the stream kernel
struct StreamCounters
{
  unsigned long long isize_sum;   // sum |isize| over proper pairs
  unsigned long long isize_n;     // count
  unsigned long long n_cand;      // discordant candidates appended
  unsigned long long n_split;     // split tuples appended
  unsigned int max_span;          // max(bam_endpos - pos)
  unsigned int unsorted;          // 1 if (tid,pos) order violated
  unsigned long long n_sa;        // SA-bearing records handed to k_split_records
  unsigned long long pad0;
};

// interned chromosome-name table (device copy): open addressing on FNV-1a of the text
struct NameTableUnused
{
  const uint64_t *hash; // capacity entries, 0 = empty
  const int32_t *id;
  uint32_t mask;        // capacity - 1
  const int32_t *own_id; // per tid: id of chromID2ChrName(tid) (util_bam.cc:128-142)
  int32_t n_targets;
  int32_t empty_id;     // id of ""
};

struct Timing
{
  std::vector<std::string> names;
  std::vector<float> ms;
  std::vector<uint64_t> bytes, touched;
  std::vector<const char *> cnames;
};
