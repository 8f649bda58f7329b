// Interface of bp.hip (cluster summary + split-read breakpoint stage).
#pragma once
#include "bk_common.h"
#include "prims.h"

struct ClusterAcc
{
  uint32_t *n;
  unsigned long long *sum1, *sum2;
  uint32_t *min1, *max1, *min2, *max2, *type;
};

// device view of the record columns the region queries touch
struct RecView
{
  uint64_t n;
  const int32_t *tid, *pos;
  const uint16_t *flag;
  const uint8_t *mapq;
  const uint32_t *cigar_off, *cigar;
  const unsigned long long *samp = nullptr;  // search keys of every 1024th record (bp.hip: rec_lower), or null
  uint64_t n_samp = 0;
};

struct BpWork
{
  uint64_t t1lo, t1hi, t2lo, t2hi;
  uint32_t ok, pad;
};

struct BpBufs
{
  DevBuf samp, key, val, kmax, slotbase, an, as1, as2, amin1, amax1, amin2, amax2, atype, keep, off, tmpc, work, nmatch, moff, err, emit, ecount, scan_tmp, cov, depth, voted, nvalid, maxrec;
  prims::RadixBufs radix;
};

// clusters with a voted breakpoint pair (flags bit 1), counted on the device
uint64_t count_valid_clusters(const bk_cluster *cl, uint64_t ncl, BpBufs &b, hipStream_t st);
// rec_bits = bits of the largest record index a tuple may carry (the sort key)
void sort_splits(bk_split *unsorted, uint64_t n, bk_split *sorted, BpBufs &b, hipStream_t st, int rec_bits);
// returns the number of clusters that passed the near-diagonal filter; clusters_out holds them in (group key order, id) order
uint64_t cluster_summary(const bk_pair *pairs, const uint32_t *idx, const uint32_t *gof, const uint32_t *cl, uint64_t n, uint32_t ng, const uint32_t *gkey,
                         const uint32_t *glex, int32_t nt, double w, DevBuf &clusters_out, BpBufs &b, hipStream_t st);
// phases of the breakpoint stage (a sharded run sums `cov` and `depth` over the record shards between them)
uint32_t *bp_cov_partial(const RecView &r, const bk_cluster *cl, uint64_t ncl, double w, int maxspan, BpBufs &b, hipStream_t st);
void bp_vote(const bk_split *sp, uint64_t nsp, bk_cluster *cl, uint64_t ncl, double w, int maxspan, const uint32_t *cov, const int32_t *hdr_id, BpBufs &b, hipStream_t st);
uint32_t *bp_depth_partial(const RecView &r, const bk_cluster *cl, uint64_t ncl, int maxspan, BpBufs &b, hipStream_t st);
void bp_finish(bk_cluster *cl, uint64_t ncl, const uint32_t *depth, BpBufs &b, hipStream_t st);
void split_breakpoints(const RecView &r, const bk_split *sp, uint64_t nsp, bk_cluster *cl, uint64_t ncl, double w, int maxspan, const int32_t *hdr_id, BpBufs &b,
                       hipStream_t st);
// test hook: one raw region through the product's region / verdict / depth device code (res: n tuples, coverage capped at 5, depth, poison)
void debug_region(const RecView &r, const bk_split *sp, uint64_t nsp, int32_t tid, uint32_t start, uint32_t end, int maxspan, unsigned long long depth_pos, bk_split *out,
                  uint32_t cap, uint32_t *res, hipStream_t st);
