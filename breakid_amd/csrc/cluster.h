// Interface of cluster.hip (isolated-pair masking + fast clustering) and ahc.hip.
#pragma once
#include "bk_common.h"
#include "prims.h"
#include "sortemu.h"

// an ordered list of pair-table indices partitioned into groups
struct PairList
{
  uint64_t n = 0;
  uint32_t ng = 0;
  DevBuf idx;   // u32[n]   index into the pair table
  DevBuf gof;   // u32[n]   group (numeric key order) of each element
  DevBuf goff;  // u64[ng+1]
  uint64_t total(hipStream_t st) const;
};

struct ClusterBufs
{
  // observers for the lanes of api.hip (host, may stay empty): the longest heapsort segment of every group in the sorts by x / by y
  std::vector<uint32_t> heavy_x, heavy_y;
  bool observe = false;
  // upper bound on the pairs of any one group (host; 0 = unknown): the anchored-window passes need log2 of it pointer-jumping levels
  uint64_t max_group_bound = 0;
  DevBuf key, perm, tmp, cnt, off, off2, idx2, gof2, goff2, jump, mark, apos, kid, kprev2, k1, k2, pk, knum, clfull, small, scan_tmp;
  SortEmuBufs se;
  prims::RadixBufs radix;
};

// remove_isolated_pairs for every group; L receives the surviving list (x-sorted, duplicates included)
void remove_isolated_all(const bk_pair *pairs, const uint32_t *gof0, const uint64_t *gstart, uint32_t ng, uint64_t n, double w, PairList &L, ClusterBufs &b, hipStream_t st,
                         const uint32_t *drop_group = nullptr);  // drop_group[g] != 0: group g is left to another rank
// the same in two parts, so that a caller may redistribute the groups between them: through the second mask / the third sort
// (gstart_host + keep_host given: the list is built straight from the ranges of the kept groups instead of filtering all n pairs)
void remove_isolated_begin(const bk_pair *pairs, const uint32_t *gof0, const uint64_t *gstart, uint32_t ng, uint64_t n, double w, PairList &L, ClusterBufs &b, hipStream_t st,
                           const uint32_t *drop_group = nullptr, const uint64_t *gstart_host = nullptr, const uint8_t *keep_host = nullptr);
void remove_isolated_end(const bk_pair *pairs, PairList &L, ClusterBufs &b, hipStream_t st);
// dst = src without the groups flagged in drop[] (device, one u32 per group; offsets for all groups are kept)
void list_subset(const PairList &src, const uint32_t *drop, PairList &dst, ClusterBufs &b, hipStream_t st);
// the same from host knowledge: keep_host[g] != 0 selects group g, src_goff_host = src's offsets; one launch over the subset
void list_subset_ranges(const PairList &src, const uint64_t *src_goff_host, const uint8_t *keep_host, PairList &dst, hipStream_t st);
// removes the groups that keep fewer than 2 pairs (they are not clustered, BreakID.cc:125)
void drop_small_groups(PairList &L, ClusterBufs &b, hipStream_t st);
// find_cluster_pairs_enspan_fast for every group with >= 2 pairs; L becomes the clustered list, cluster_out[p] its cluster number
void fast_cluster_all(const bk_pair *pairs, PairList &L, double w, DevBuf &cluster_out, ClusterBufs &b, hipStream_t st);
// two lists over disjoint sets of groups (both with offsets for all ng groups) -> one list in group order; cl_* = the cluster
// numbers that travel with the elements (may be null)
void merge_lists_many(const PairList *const *lists, const uint32_t *const *cls, int K, PairList &out, DevBuf *cl_out, hipStream_t st);
void merge_lists(const PairList &A, const uint32_t *clA, const PairList &B, const uint32_t *clB, PairList &out, DevBuf *cl_out, hipStream_t st);
// test hook: mask_pairs_chr_pos on the list in its current order
void debug_mask_list(const bk_pair *pairs, PairList &L, long dist, ClusterBufs &b, hipStream_t st);
