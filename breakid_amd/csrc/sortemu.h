// std::sort (libstdc++ introsort) emulation over all groups at once — see sortemu.hip.
#pragma once
#include "bk_common.h"
#include "prims.h"
#include <vector>

struct SortEmuBufs
{
  DevBuf cnt, err, segs_a, segs_b, lr, segof, posL, posR, ck, scan_tmp, heap_list, heap_scratch, hr_cnt, hr_ck, hr_val, hr_f, hr_ord, rank32, scratch32, scratch32b, fin_list, fin_cnt, lvl, lv_tile, lv_segbase, lv_tileseg, lv_bar, chk_key0, chk_cnt, chk_bad, rk_a, rk_b;
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
