// Interface of ahc.hip: exact agglomerative clustering (util_cluster.cc semantics) for every group.
#pragma once
#include "bk_common.h"
#include "cluster.h"

struct AhcBufs
{
  DevBuf x, y, comp, tmp0, tmp1, tmp2, tmp3, tmp4, tmp5, tmp6, tmp7, scan_tmp;
  prims::RadixBufs radix;
};

// find_cluster_pairs_enspan_ahc (BreakID.cc:1304-1352) for every group with >= 2 pairs.
void ahc_cluster_all(const bk_pair *pairs, PairList &L, double w, DevBuf &cluster_out, AhcBufs &ab, ClusterBufs &cb, hipStream_t st);
