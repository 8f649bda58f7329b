// placeholder until the exact AHC path lands (next commit)
#include "ahc.h"
void ahc_cluster_all(const bk_pair *, PairList &, double, DevBuf &, AhcBufs &, ClusterBufs &, hipStream_t)
{
  throw bk_error(BK_ERR_ARG, "AHC clustering is not available in this build; use fast=1");
}
