// Interface of join.hip (mate join + chromosome-pair grouping).
#pragma once
#include "bk_common.h"
#include "prims.h"
#include <vector>

struct JoinBufs
{
  DevBuf counter, unsorted, okey, oval, key, val, pairs, gof, gflag, gscan, gstart, gkey, scan_tmp, route_cand, route_cnt, route_pairs, route_off, maxrec, key2, val2;
  prims::RadixBufs radix;
  // bits a record index needs (the caller's table: records of the whole sample); the pair sort key is (chr-pair key << rec_bits) |
  // discovery index, so that the sort runs over rec_bits + key bits instead of 32 + key bits
  // < 0: a sharded sample - the indices are rec_base + i of other ranks too (64 bits in Cand / bk_pair): the bits are taken from the
  // largest index among the candidates / pairs at hand
  int rec_bits = 32;
  int rec_bits_eff = 32;  // what the last join / grouping used
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

// ---- one sample over many GPUs: candidates travel to the rank that owns their read-name hash, pairs to the rank that
// owns their chromosome-pair group (all-to-all), so that no rank joins or sorts more than its share ----
// candidates reordered by destination rank ((qhash >> 17) % world); counts[d] = how many go to rank d
Cand *route_candidates(const Cand *cand, uint64_t n, uint32_t world, JoinBufs &b, hipStream_t st, std::vector<uint64_t> &counts);
// the grouped local pair table reordered so that group g starts at off_of_group[g] (groups of one destination adjacent)
bk_pair *route_pairs(const JoinResult &jr, const std::vector<uint64_t> &off_of_group, JoinBufs &b, hipStream_t st);
// received pairs (any order) -> table sorted by (chr-pair key, discovery index)
void group_pairs(const bk_pair *raw, uint64_t np, int32_t nt, JoinBufs &b, hipStream_t st, JoinResult &res);
