// Interface of join.hip (mate join + chromosome-pair grouping).
#pragma once
#include "bk_common.h"
#include "prims.h"

struct JoinBufs
{
  DevBuf counter, unsorted, okey, oval, key, val, pairs, gof, gflag, gscan, gstart, gkey, scan_tmp;
  prims::RadixBufs radix;
};

struct JoinResult
{
  uint64_t n_pairs = 0;
  uint32_t n_groups = 0;
  bk_pair *pairs = nullptr;      // device, sorted by (numeric chr-pair key, discovery order)
  uint32_t *gof = nullptr;       // device, group index (numeric key order) per pair
  uint64_t *gstart = nullptr;    // device, n_groups+1
  uint32_t *gkey = nullptr;      // device, (p1_tid+1)*(nt+1)+(p2_tid+1) per group
};

void join_candidates(const Cand *cand, uint64_t n_cand, double w, const uint32_t *tprefix, int32_t nt, JoinBufs &b, hipStream_t st, JoinResult &res);
// glex_dev[g] = ordinal of group g in the reference's std::map<string> order of "chrA_chrB"
void join_assign_ids(JoinResult &res, const uint32_t *glex_dev, hipStream_t st);
