// Interface of ahc.hip: exact agglomerative clustering (util_cluster.cc semantics) for every group.
#pragma once
#include "bk_common.h"
#include "cluster.h"

struct AhcBufs
{
  DevBuf x, y, comp, csize, keys, vals, rank, cfs, need0, need1, need2, label, rootcomp, cand_t, npts, ent_cnt, pts_off, ent_off, cand_d, cand_o, cnodes_cnt,
      cent_used, cpts_used, cbest_d, cbest_j, cmaxroot, act, cnodes, entries, pts, ptsxy, out_cnt, out_idx, out_cl, err, newoff, scan_tmp;
  prims::RadixBufs radix;
};

// find_cluster_pairs_enspan_ahc (BreakID.cc:1304-1352) for every group with >= 2 pairs.
void ahc_cluster_all(const bk_pair *pairs, PairList &L, double w, DevBuf &cluster_out, AhcBufs &ab, ClusterBufs &cb, hipStream_t st);
