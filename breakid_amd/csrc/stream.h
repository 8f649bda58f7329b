// Interface of stream.hip (streaming record pass + bit-exact sd replay).
#pragma once
#include "bk_common.h"

struct SdState
{
  double sumsq;           // sum of v^2 (approximate, only bounds the binade of the running total)
  unsigned int vmax;      // max |isize| over proper pairs
  unsigned int pad;
  long long t_final;      // the reference's insert_size_sd_total after the last add
};

struct SdException
{
  unsigned long long l_before;  // sum of floor(d) over all earlier qualifying records
  double d;
};

struct SdBufs
{
  DevBuf blockL, blockE, scan_tmp, scan_tmp2, exceptions;
  unsigned long long last_exceptions = 0;
};

struct NameTableDev
{
  const uint64_t *hash;
  const int32_t *id;
  uint32_t mask;
  const int32_t *own_id;
  int32_t n_targets;
  int32_t empty_id;
};

// records a lane of k_stream owns per iteration (consecutive ones).  Eight - every column incl. flag as 16-byte loads, mapq as 8-byte
// loads, 94 VGPRs, still five blocks per CU - was measured against four on the 620 M-record table: 4.72 ms against 4.45-4.49 (two
// 16-byte loads per lane and column at a 32-byte lane stride ask for every 128-byte line twice); -DBK_STREAM_V=8 builds that form.
#ifndef BK_STREAM_V
#define BK_STREAM_V 4
#endif
constexpr int STREAM_V = BK_STREAM_V;
static_assert(STREAM_V == 4 || STREAM_V == 8, "k_stream is written for 4 or 8 records per lane");

struct StreamArgs
{
  uint64_t n;        // records [STREAM_V * q_begin, n) are processed by this launch
  uint64_t q_begin;  // first group (STREAM_V records) of this launch; 0 unless the table arrives in pieces
  uint64_t rec_base;  // index of record 0 of this table in the whole sample (0 unless the sample is sharded)
  const int32_t *tid, *pos, *mtid, *mpos, *isize;
  const uint16_t *flag;
  const uint8_t *mapq;
  const uint64_t *qhash;
  const uint32_t *qcheck;  // may be null
  const bk_side *side;     // may be null; when set, qhash / mtid / mpos / qcheck come from it
  const uint32_t *cigar_off, *cigar, *aux_off;
  const uint8_t *aux;
  int mapq_min;
  NameTableDev names;
  StreamCounters *counters;
  SdState *sd;
  Cand *cand;
  unsigned long long cand_cap;
  bk_split *split;
  unsigned long long split_cap;
  uint32_t *sa_list;
  unsigned long long sa_cap;
};

void launch_stream(const StreamArgs &a, hipStream_t st);
// side[i] = {qhash[i], mtid[i], mpos[i], qcheck ? qcheck[i] : 0} (include/breakid_hip.h: bk_side)
void launch_make_side(const uint64_t *qhash, const int32_t *mtid, const int32_t *mpos, const uint32_t *qcheck, uint64_t n, bk_side *side, hipStream_t st);
void launch_split_records(const StreamArgs &a, unsigned long long n_sa, hipStream_t st);
// mean: host-computed (double) sum / (double) n; thr: exception threshold 2^(kmax-53) (or huge = replay all)
void launch_sd_local(const uint16_t *flag, const int32_t *isize, uint64_t n, double mean, double thr, SdBufs &b, hipStream_t st, unsigned long long *l_total,
                     unsigned long long *n_ex_out);
void launch_sd_walk(const SdException *ex, unsigned long long n_ex, unsigned long long l_total, SdState *sd, hipStream_t st);
void launch_sd(const uint16_t *flag, const int32_t *isize, uint64_t n, double mean, double thr, SdState *sd, SdBufs &b, hipStream_t st);
// test hook: rows of the CIGAR table through the device CIGAR model (BAM word rows must start 4-byte aligned in c1)
void debug_cigar(const uint8_t *kind, const uint32_t *c1_off, const uint8_t *c1, const uint32_t *c2_off, const uint8_t *c2, const int32_t *e, uint32_t n, int32_t *out,
                 hipStream_t st);
