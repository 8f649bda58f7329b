// std::sort (libstdc++ introsort) emulation over all groups at once — see sortemu.hip.
#pragma once
#include "bk_common.h"
#include "prims.h"
#include <vector>
#include <atomic>

// Resident sort service (sortsvc.inc): two persistent kernels that play the introsort replay of every std_sort_groups call made
// through a SortEmuBufs that points at a running service, as tasks.  One per device context; start() .. stop() bracket a stage.
struct SortService
{
  DevBuf ctl, slots[2], seq[2], jobs, dbg, pos[2];
  uint32_t pos_cap[2] = {0, 0};
  uint32_t cap[2] = {0, 0};
  hipStream_t st[2] = {nullptr, nullptr}, quit_stream = nullptr, st_copy = nullptr;
  uint32_t *quit_host = nullptr;  // mapped host memory the workgroups poll
  uint32_t *quit_dev = nullptr;
  uint32_t cap32 = 0;
  uint32_t quit_word = 0, starts = 0;
  size_t wide_lds = 0;
  bool running = false;
  std::atomic<uint32_t> next_slot{0};  // (the lanes' threads draw their slots at the same time)
  uint32_t stats[8] = {};
  // n_bound = elements of all lists that will be sorted while the service runs (sizes the task rings), max_group = the largest
  // group among them
  void start(uint64_t n_bound, uint64_t max_group, hipStream_t after);
  // has a narrow workgroup started within `seconds`?  (No: the narrow kernel's stream shares its hardware queue with the wide
  // kernel's and will wait behind it for ever.)
  bool narrow_running(double seconds) const;
  // ends the kernels; throws when a task reported an error
  void stop();
  uint32_t new_slot() { return next_slot.fetch_add(1u); }
  ~SortService();
  SortService() = default;
  SortService(const SortService &) = delete;
  SortService &operator=(const SortService &) = delete;
};

struct SortEmuBufs
{
  SortService *svc = nullptr;  // set (and running): std_sort_groups submits to it instead of launching the phases itself
  uint32_t svc_slot = 0xFFFFFFFFu, svc_epoch = 0;
  DevBuf cnt, err, segs_a, segs_b, posL, posR, scan_tmp, heap_list, heap_scratch, scratch32, scratch32b, fin_list, lv_tile, lv_segbase, lv_tileseg, lv_bar, chk_key0, chk_cnt, chk_bad, rk_a, rk_b;
  prims::RadixBufs radix;
  // optional observer (host): heavy[g] = largest heapsort segment (elements) any sort through these buffers left to group g's
  // lone-wave heap kernels - what the lanes of api.hip balance on.  Set by the caller around the sorts it wants recorded.
  std::vector<uint32_t> *heavy = nullptr;
  bool heavy_all = false;  // record every segment the level loop left to the heapsort kernels (a group that has one went through all ~2 lg n levels), not only the long ones
  // the three size classes of the heapsort branch run side by side (fork/join around the caller's stream)
  static constexpr int N_AUX = 5;
  hipStream_t aux[N_AUX] = {};
  hipEvent_t fork = nullptr, join[N_AUX] = {};
  SortEmuBufs() = default;
  SortEmuBufs(const SortEmuBufs &) = delete;
  SortEmuBufs &operator=(const SortEmuBufs &) = delete;
  ~SortEmuBufs()
  {
    for (int i = 0; i < N_AUX; ++i)
    {
      if (aux[i]) (void) hipStreamDestroy(aux[i]);
      if (join[i]) (void) hipEventDestroy(join[i]);
    }
    if (fork) (void) hipEventDestroy(fork);
  }
};

// key/idx: n elements, groups are the contiguous ranges goff[g]..goff[g+1]; gof[p] = group of position p.
// On return every group is ordered exactly as std::sort(begin, end, [](a,b){return a.key < b.key;}) leaves it.
void std_sort_groups(uint32_t *key, uint32_t *idx, const uint32_t *gof, const uint64_t *goff, uint32_t ng, uint64_t n, SortEmuBufs &b, hipStream_t st);
