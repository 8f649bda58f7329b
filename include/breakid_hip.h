/*
 * breakid_hip.h — C ABI of libbreakid_hip.so, the MI355X (gfx950) implementation of BreakID's hot
 * path: discordant-pair scan + clustering and the split-read (SA-tag / CIGAR) breakpoint scan.
 *
 * The reference (SinOncology/BreakID) has no plugin / FFI layer: its hot path is a chain of free
 * functions called from main() (src/BreakID.cc:93-167).  Each entry point below replaces one link of
 * that chain; the reference call it stands in for is cited next to it.  A reference maintainer binds
 * them from main() as shown in INTEGRATION.md.
 *
 * Conventions: plain C, no exceptions across the boundary; every call returns BK_OK (0) or a negative
 * error code and bk_last_error() gives the text.  The caller owns all inputs; the library owns every
 * output until bk_free().  One host thread per context; the context owns one HIP stream (or uses the
 * one given to bk_set_stream).  Record columns may live in host memory (copied to HBM by
 * bk_upload_records) or already in HBM (BK_MEM_DEVICE: used in place, zero copy).
 */
#ifndef BREAKID_HIP_H
#define BREAKID_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BK_OK 0
#define BK_ERR_ARG (-1)       /* bad argument / call order */
#define BK_ERR_HIP (-2)       /* HIP runtime error (text in bk_last_error) */
#define BK_ERR_NO_DEVICE (-3) /* no gfx950 device / code object missing: the product never falls back to CPU */
#define BK_ERR_UNSORTED (-4)  /* records are not coordinate sorted (the reference needs a .bai, i.e. a sorted BAM) */
#define BK_ERR_CIGAR (-5)     /* the reference's "error cigar" exit(-1) (src/BreakID.cc:954-968) */
#define BK_ERR_IO (-6)        /* BAM decode errors (bk_bam_*) */
#define BK_ERR_LIMIT (-7)     /* an internal capacity was exceeded (text says which) */
#define BK_ERR_COLLISION (-8) /* two different read names (or contig names) share a hash: the call would not be the reference's */

#define BK_MEM_HOST 0
#define BK_MEM_DEVICE 1

typedef struct bk_ctx bk_ctx;

/* Columnar record table (structure of arrays).  One row per BAM alignment record, in file order.
 * Field meaning = bam1_core_t (thirdparty/.../htslib/sam.h:148-181): pos/mpos 0-based.
 *   cigar_off[i]..cigar_off[i+1]  BAM-encoded CIGAR words (len<<4|op) of record i
 *   aux_off[i]..aux_off[i+1]      aux blob of record i: empty when the record has no SA:Z tag,
 *                                 else the SA text, or  OC-text '\t' SA-text  when an OC:Z tag exists
 *   qhash                         64-bit hash of the read name (bk_qname_hash) used for the qname joins
 *   qcheck                        optional (may be NULL): second, independent 32-bit hash of the read name
 *                                 (bk_qname_check, never 0).  The reference compares read names as strings
 *                                 (BreakID.cc:1424, :627-637); with this column every equal-qhash decision of the mate join
 *                                 and of the breakpoint vote is verified, and a mismatch ends the run with BK_ERR_COLLISION
 *                                 instead of a silently different call.  The BAM decoders (bk_bam_*) fill it.
 * A table handed over in device memory (BK_MEM_DEVICE) is used in place: tid, pos, isize, flag, mapq, cigar_off and aux_off
 * must be 16-byte aligned there (the streaming kernels read them as vectors); BK_ERR_ARG otherwise.
 */
typedef struct bk_soa {
  uint64_t n;
  const int32_t *tid, *pos, *mtid, *mpos, *isize;
  const uint16_t *flag;
  const uint8_t *mapq;
  const uint64_t *qhash;
  const uint32_t *cigar_off; /* n+1 */
  const uint32_t *cigar;     /* cigar_off[n] words */
  const uint32_t *aux_off;   /* n+1 */
  const uint8_t *aux;        /* aux_off[n] bytes */
  uint64_t n_cigar_words;    /* = cigar_off[n] (given so that device-resident tables need no read-back) */
  uint64_t n_aux_bytes;      /* = aux_off[n] */
  const uint32_t *qcheck;    /* n entries or NULL */
  const struct bk_side *side; /* n entries or NULL: see bk_side */
} bk_soa;

/* Optional device layout of the four columns only discordant candidates and SA-bearing records need (about 5 % of the records
 * of a WGS sample): qhash, mtid, mpos and qcheck of a record in ONE 32-byte row, so that the streaming pass fetches one sector
 * per candidate instead of one per column (four scattered 4-8-byte reads cost four 64-byte fetches).  A producer that writes a
 * table in device memory may fill it (the synthetic generator does); a host table gets it when it is uploaded; when `side` is
 * given the library reads those four values from it and never touches the qhash / mtid / mpos / qcheck columns. */
typedef struct bk_side {
  uint64_t qhash;
  int32_t mtid, mpos;
  uint32_t qcheck;           /* 0 = the table has no second read-name hash */
  uint32_t reserved[3];      /* 0 */
} bk_side;                   /* 32 bytes */

/* One discordant pair = the numeric content of `discordant_pair` (src/BreakID.h:39-58). */
typedef struct bk_pair {
  uint32_t x, y;             /* p1_chr_pos, p2_chr_pos (genome-wide, util_bam.cc:57-68) */
  uint32_t p1_pos, p2_pos;   /* 1-based */
  int32_t p1_tid, p2_tid;    /* chromosome of each side (-1 = "*") */
  uint16_t p1_flag, p2_flag;
  uint8_t p1_mapq, p2_mapq, p1_rev, p2_rev;
  uint64_t rec;              /* index, in the whole sample, of the record that completed the pair (discovery order); 64 bits like the
                                reference's own counters (long / size_t, BreakID.cc:1379,1911-1913): a shard's records are numbered
                                rec_base + i, and a sample may hold more than 2^32 records even though one context does not */
  uint32_t id;               /* "pair_No_<id>": index inside its group at add_enspan_point_id time */
  int32_t cluster;           /* cluster number inside the group, -1 before clustering */
  uint32_t group;            /* group ordinal, groups ordered like std::map<string> on "chrA_chrB" */
  uint32_t reserved;         /* 0 */
} bk_pair;                   /* 56 bytes */

/* One split-read evidence tuple = numeric content of `split_align_pair` (src/BreakID.h:116-133).
 * Chromosome names are interned (ids < n_targets are header names).  CIGAR strings are only compared for equality by the
 * reference, so they travel as 64-bit codes: a text of the form <n><M|S><n><M|S> (every text that can reach a tuple: both
 * sides pass the ([0-9]+[MS]){2} gate, CigarRoller.cc:326) is encoded EXACTLY (bit 63 set; counts < 2^28, up to 3 leading
 * zeros per count), anything else as a 63-bit hash of the text (bit 63 clear). */
typedef struct bk_split {
  uint64_t rec;              /* index of the record in the whole sample (rec_base + i) */
  int32_t tid, pos, endpos;  /* of the record itself: 0-based pos, bam_endpos (sam.c:344-350) */
  uint32_t reserved;         /* second hash (bk_qname_check of the text) of an SA contig name that is neither in the header nor chr1..22,X,Y
                                (such a name is a 30-bit hash id in prim_chr / sec_chr); 0 otherwise */
  uint64_t qhash;
  int32_t prim_chr, sec_chr;
  uint32_t prim_start, prim_end, prim_bp, sec_start, sec_end, sec_bp;
  uint64_t prim_cigar, sec_cigar;
  uint32_t flags;            /* bit0 = secondary (flag & 0x100); bit1 = "error cigar" record */
  uint32_t qcheck;           /* bk_qname_check of the read name (0 when the table has no qcheck column) */
} bk_split;                  /* 88 bytes */

#define BK_TYPE_DIFF_CHR 1u
#define BK_TYPE_SAME_ORIENT 2u
#define BK_TYPE_ABS_REVERSE 4u
#define BK_TYPE_DEFAULT_ORIENT 8u

/* One cluster = numeric content of `cluster_info` (src/BreakID.h:60-113) that the txt writer needs. */
typedef struct bk_cluster {
  uint32_t group;
  int32_t id;
  int32_t p1_tid, p2_tid;
  uint32_t p1_mean, p2_mean, p1_min, p1_max, p2_min, p2_max;
  uint32_t p1_exact;
  int32_t p2_exact;
  uint32_t n_drp, n_sr;
  uint32_t depth1, depth2;
  uint32_t type_mask;        /* BK_TYPE_* : drp_type_set */
  uint32_t flags;            /* bit0 passed the near-diagonal filter (:348); bit1 valid (:446) */
} bk_cluster;

/* stage ids for bk_fetch */
#define BK_STAGE_SCAN 0      /* bk_pair[]  after scan_discordant_pairs, grouped            */
#define BK_STAGE_ISO 1       /* bk_pair[]  after remove_isolated_pairs                     */
#define BK_STAGE_CLUSTERED 2 /* bk_pair[]  after find_cluster_pairs_enspan_{fast,ahc}      */
#define BK_STAGE_SPLITS 3    /* bk_split[] every accepted split-evidence tuple, record order */
#define BK_STAGE_CLUSTERS 4  /* bk_cluster[] every cluster that passed :348, group/id order */
#define BK_STAGE_GROUP_KEYS 5 /* int32 pairs (p1_tid,p2_tid) per group ordinal               */

/* ---- lifetime ------------------------------------------------------------------------------- */
/* Replaces: samopen + header parsing (BreakID.cc:1391,1410).  target_name[i] NUL-terminated. */
int bk_init(int device, const uint32_t *target_len, const char *const *target_name, int n_targets, bk_ctx **out);
/* Optional, once per process and before its first HIP call: sets GPU_MAX_HW_QUEUES=20 when the environment does not say (the lanes
   of bk_mask_and_cluster, the resident sort service and the GPU feed each want a hardware queue of their own; the runtime reads the
   variable when it starts).  bk_init calls it; a caller with threads of its own calls it before it starts them. */
void bk_prepare_process(void);
void bk_free(bk_ctx *ctx);
const char *bk_last_error(const bk_ctx *ctx); /* ctx may be NULL: error of the failed bk_init */
int bk_set_stream(bk_ctx *ctx, void *hip_stream); /* optional: run on the caller's hipStream_t */
int bk_sync(bk_ctx *ctx);
int bk_get_stream(bk_ctx *ctx, void **hip_stream); /* the hipStream_t the context runs on (collectives of a sharded run are queued on it) */

/* Replaces: the two sequential BAM passes' record access (BreakID.cc:1414, :1929). */
int bk_upload_records(bk_ctx *ctx, const bk_soa *cols, int mem_space);

/* ---- stages (call in this order) ------------------------------------------------------------ */
/* get_mean_insert_size (BreakID.cc:1909-1954): bit-exact mean and sd. */
int bk_isize_stats(bk_ctx *ctx, double *mean, double *sd);
/* scan_discordant_pairs (BreakID.cc:1362-1515): filter, qname mate join, grouping. */
int bk_discordant_pairs(bk_ctx *ctx, int mapq_min, double w, uint64_t *n_pairs, uint32_t *n_groups);
/* remove_isolated_pairs (:1271) + find_cluster_pairs_enspan_fast (:1046) or _ahc (:1304), all groups. */
int bk_mask_and_cluster(bk_ctx *ctx, double w, int fast, uint64_t *n_clustered);
/* per-read SA-tag/CIGAR evidence of find_sa_reads (:892-1030) for every record, once. */
int bk_split_evidence(bk_ctx *ctx, uint64_t *n_tuples);
/* findClusterBreakPointInfoSaTag summary part (:225-352). */
int bk_cluster_summary(bk_ctx *ctx, double w, uint64_t *n_clusters);
/* findEncompassingReadsAndBreakPointInfo (:390-490): region select, find_bp_pair, depth, type. */
int bk_split_breakpoints(bk_ctx *ctx, double w, uint64_t *n_valid);

/* Whole hot path = body of main() between BreakID.cc:98 and :167 (annotation excluded).
 * w_out receives times*sqrt(times)*(mean+3sd) (:103). */
int bk_run(bk_ctx *ctx, int mapq_min, int fast, double *w_out, uint64_t *n_valid);

/* Copy a stage's result to library-owned host memory.  *data stays valid until the next bk_fetch
 * of the same stage or bk_free.  group_off (may be NULL) receives n_groups+1 offsets for pair stages. */
int bk_fetch(bk_ctx *ctx, int stage, const void **data, uint64_t *count, const uint64_t **group_off, uint32_t *n_groups);

/* Per-group counts behind the reference's stage log and `_performance.txt` (BreakID.cc:119-191), groups in the reference's
 * std::map<string> order: pairs after scan_discordant_pairs, after remove_isolated_pairs (:123), after clustering (:129-137), and
 * one past the largest cluster number of the group (-fast numbers from 1: find_cluster_pairs_enspan_fast returned
 * cluster_id_end - 1; AHC from 0: #roots = cluster_id_end + n_isolated_removed - n_clustered).  Valid after bk_cluster_summary. */
typedef struct bk_group_stat {
  int32_t p1_tid, p2_tid;
  uint64_t n_scan, n_isolated_removed, n_clustered;
  uint32_t cluster_id_end;
  uint32_t ordinal;          /* of the group in the reference's std::map<string> order of "chrA_chrB" among ALL groups of the sample
                                (= bk_pair.group; on one GPU simply the row number) */
} bk_group_stat;
int bk_group_stats(bk_ctx *ctx, const bk_group_stat **out, uint32_t *n_groups);

/* per-kernel timing of the last stage calls: name/ms pairs, for bench.py's roofline leg */
int bk_timing(bk_ctx *ctx, const char *const **names, const float **ms, const uint64_t **bytes, int *n);
int bk_timing_enable(bk_ctx *ctx, int on);
/* after bk_timing: bytes each timed stage's own kernels load + store (0 = not modelled), same order and count as bk_timing */
int bk_timing_touched(bk_ctx *ctx, const uint64_t **touched, int *n);

/* ---- one sample sharded over several GPUs -----------------------------------------------------
 * One context per GPU holds a contiguous range of the sample's coordinate-sorted records.  The library does the
 * per-shard work; the caller moves the small tables between the contexts (RCCL all-gather / all-reduce over xGMI
 * from torch.distributed in breakid_amd/sharded.py).  Sequence:
 *   bk_upload_records; bk_shard_begin                       stream pass, record indices are global (rec_base + i)
 *   bk_shard_get_stats -> all-reduce -> bk_shard_set_stats  insert-size sums, spans
 *   bk_shard_sd_local -> all-gather -> bk_shard_sd_finish   bit-exact sd: exceptions replayed in global record order
 *   BK_BUF_CANDIDATES: bk_shard_buffer -> all-gather -> bk_shard_set_buffer; bk_discordant_pairs (replicated join)
 *   bk_shard_group_sizes -> owner per group -> bk_shard_own_groups; bk_mask_and_cluster; bk_cluster_summary
 *   BK_BUF_TUPLES, BK_BUF_CLUSTERS: gather as above
 *   bk_shard_bp_cov -> all-reduce(sum) -> bk_shard_bp_vote -> bk_shard_bp_depth -> all-reduce(sum) -> bk_shard_bp_finish
 * after which every context holds the complete cluster table (bk_fetch).  One CONTEXT holds fewer than 2^32 records; the sample
 * may hold more (record indices are 64-bit across the contexts: rec_base + i). */
typedef struct bk_shard_stats {
  uint64_t isize_sum, isize_n;
  double sumsq;
  uint32_t vmax, max_span;
  uint64_t n_cand, n_split;
} bk_shard_stats;
#define BK_BUF_CANDIDATES 0 /* 40-byte candidates of the discordant filter */
#define BK_BUF_TUPLES 1     /* bk_split, unsorted */
#define BK_BUF_CLUSTERS 2   /* bk_cluster of the groups this rank owns */
int bk_shard_begin(bk_ctx *ctx, uint64_t rec_base, int mapq_min);
int bk_shard_get_stats(bk_ctx *ctx, bk_shard_stats *out);
int bk_shard_set_stats(bk_ctx *ctx, const bk_shard_stats *total);
int bk_shard_sd_local(bk_ctx *ctx, uint64_t *l_total, void **ex_dev, uint64_t *n_ex); /* exceptions: 16 B {u64 l_before, f64 d}, device */
int bk_shard_sd_finish(bk_ctx *ctx, const void *all_ex_dev, uint64_t n_all, uint64_t l_grand, double *mean, double *sd);
int bk_shard_buffer(bk_ctx *ctx, int which, void **dev, uint64_t *count, uint32_t *elem_bytes);
int bk_shard_set_buffer(bk_ctx *ctx, int which, const void *dev, uint64_t count); /* gathered table, device, caller keeps it alive */
int bk_shard_group_sizes(bk_ctx *ctx, const uint64_t **starts, uint32_t *n_groups); /* n_groups+1 pair offsets, numeric key order */
int bk_shard_own_groups(bk_ctx *ctx, const uint8_t *own, uint32_t n_groups);
/* Routed exchange (scales with the number of GPUs: no rank joins or sorts more than its share).  Instead of the
 * replicated join above:
 *   bk_shard_route_candidates -> all-to-all -> bk_shard_set_buffer(BK_BUF_CANDIDATES); bk_discordant_pairs joins the
 *     read names this rank owns ((qhash >> 17) % world)
 *   bk_shard_group_keys + bk_shard_group_sizes -> all-gather of (key, size) -> owner per chr-pair key (LPT on the totals)
 *   bk_shard_route_pairs -> all-to-all -> bk_shard_group_pairs: the table of exactly the groups this rank owns, `group`
 *     ordinals global; then bk_mask_and_cluster, bk_cluster_summary and the gathers / reductions as above. */
int bk_shard_route_candidates(bk_ctx *ctx, uint32_t world, void **dev, const uint64_t **counts); /* 40-byte candidates ordered by destination; counts[world] */
int bk_shard_group_keys(bk_ctx *ctx, const uint32_t **keys, uint32_t *n_groups);                /* (p1_tid+1)*(n_targets+1)+(p2_tid+1) per group */
int bk_shard_route_pairs(bk_ctx *ctx, const uint32_t *dest_of_group, uint32_t n_groups, uint32_t world, void **dev, const uint64_t **counts); /* bk_pair rows ordered by destination */
int bk_shard_group_pairs(bk_ctx *ctx, const void *pairs_dev, uint64_t n, const uint32_t *all_keys, uint32_t n_all_keys);
int bk_shard_bp_cov(bk_ctx *ctx, double w, void **cov_dev, uint64_t *n);   /* u32[2*n_clusters] partial coverage counts */
int bk_shard_bp_vote(bk_ctx *ctx, double w, const void *cov_total_dev);
/* the vote of clusters [lo, hi) only (every rank takes a slice); the voted rows (72 B) and flags (u32) of the slices are
 * then all-gathered in rank order: BK_BUF_CLUSTERS via bk_shard_set_buffer, the flags via bk_shard_bp_set_voted */
int bk_shard_bp_vote_slice(bk_ctx *ctx, double w, const void *cov_total_dev, uint64_t lo, uint64_t hi, void **clusters_dev, void **voted_dev);
int bk_shard_bp_set_voted(bk_ctx *ctx, const void *voted_all_dev);
int bk_shard_bp_depth(bk_ctx *ctx, void **depth_dev, uint64_t *n);         /* u32[2*n_clusters] partial depth counts */
int bk_shard_bp_finish(bk_ctx *ctx, const void *depth_total_dev);

/* Test hook: orders every group [group_off[g], group_off[g+1]) of `key` exactly as
 * std::sort(first, last, [](a, b){ return a.key < b.key; }) of libstdc++ does (the reference's unstable sorts,
 * BreakID.cc:1274-1282,1091,1127) and returns the permutation (perm_out[p] = original index of the element now at p). */
int bk_debug_std_sort(bk_ctx *ctx, const uint32_t *key, const uint64_t *group_off, uint32_t n_groups, uint32_t *perm_out);

/* Test hook: find_cluster_pairs_enspan_ahc (BreakID.cc:1304-1352) on one group of x-sorted points; returns the
 * surviving point indices in output order with their cluster numbers. */
int bk_debug_ahc(bk_ctx *ctx, const uint32_t *x, const uint32_t *y, uint32_t n, double w, uint32_t *idx_out, int32_t *cluster_out, uint32_t *n_out);

/* Test hooks that run the reference's unit vectors (tests/golden/units.json, *.regions.json) through the device code of the
 * product path - the same kernels / device functions the stages use:
 *   bk_debug_points  one group of points in the given order; mode 0 = mask_pairs_chr_pos (BreakID.cc:1813-1877), 1 =
 *                    remove_isolated_pairs (:1271-1285), 2 = find_cluster_pairs_enspan_fast (:1046-1160, x-sorted input):
 *                    surviving point indices in output order (+ cluster numbers for mode 2)
 *   bk_debug_cigar   n rows of the CIGAR model (CigarRoller.cc / Cigar.cc): c1 = text (kind 0) or BAM words (kind 1, 4-byte
 *                    aligned rows), c2 = SA cigar text, e = tolerance; out6 = rolled op count, begin clips, end clips,
 *                    reference length, matches, is_complementary_cigar(c2, e) (CigarRoller.cc:323-346)
 *   bk_debug_vote    find_bp_pair (BreakID.cc:577-857) on two tuple tables: voted (p1_bp, p2_bp, count) or (-1, -1, 0) when the
 *                    best count stays below 2 (:446)
 *   bk_debug_region  find_sa_reads (:868-1037) on a raw region of the uploaded table: evidence tuples that survive the region
 *                    verdict, total_coverage capped at 5, and cal_single_base_depth (util_bed.cc:154-192) at depth_pos */
int bk_debug_points(bk_ctx *ctx, int mode, const uint32_t *x, const uint32_t *y, uint32_t n, double w, uint32_t *idx_out, int32_t *cluster_out, uint32_t *n_out);
int bk_debug_cigar(bk_ctx *ctx, uint32_t n, const uint8_t *kind, const uint32_t *c1_off, const uint8_t *c1, const uint32_t *c2_off, const uint8_t *c2, const int32_t *e,
                   int32_t *out6);
int bk_debug_vote(bk_ctx *ctx, const bk_split *side1, uint32_t n1, const bk_split *side2, uint32_t n2, int32_t p1_tid, int32_t p2_tid, int32_t *out3);
int bk_debug_region(bk_ctx *ctx, int32_t tid, uint32_t start, uint32_t end, uint64_t depth_pos, bk_split *out, uint32_t cap, uint32_t *n_out, uint32_t *cov_out,
                    uint32_t *depth_out);

/* ---- host feed (C++ BGZF/BAM decoder -> pinned SoA); replaces htslib's reader for this path --- */
typedef struct bk_bam bk_bam;
uint64_t bk_qname_hash(const char *name, size_t len);
uint32_t bk_qname_check(const char *name, size_t len); /* the qcheck column: independent of bk_qname_hash, folds the length in, never 0 */
int bk_bam_open(const char *path, bk_bam **out, char *err, size_t errlen);
int bk_bam_header(const bk_bam *b, int *n_targets, const char *const **names, const uint32_t **lens);
/* decode all records into a SoA owned by the bk_bam (pinned when a GPU is present) */
int bk_bam_decode(bk_bam *b, bk_soa *out, char *err, size_t errlen);
void bk_bam_close(bk_bam *b);

/* The same feed on the GPU: BGZF blocks are inflated on the device (bgzf_gpu.hip) and the records are decoded into
 * device-resident columns (cols holds device pointers: bk_upload_records(ctx, cols, BK_MEM_DEVICE)).
 * Files whose records stay inside their BGZF blocks (htslib / samtools writers) are streamed in chunks
 * (BREAKID_FEED_CHUNK_MB, default 32-64 MiB, four to eight in flight): device memory holds the chunks in flight and the
 * columns.  Files whose records run across blocks (htsjdk / Picard / GATK writers, long reads): record boundaries are
 * guessed per block and verified to chain, in chunks that start at the record carried over from the chunk before; a record
 * longer than 8 MiB that crosses a chunk sends the file to a one-batch variant (file image + inflated stream in HBM).
 * BK_ERR_IO when the boundaries cannot be established, BK_ERR_LIMIT when neither variant can take the file - take
 * bk_bam_open / bk_bam_decode then. */
typedef struct bk_bam_dev bk_bam_dev;
int bk_bam_decode_device(const char *path, int device, bk_bam_dev **out, bk_soa *cols, int *n_targets, const char *const **names, const uint32_t **lens,
                         char *err, size_t errlen);
/* One part of a file whose records stay inside their BGZF blocks: the blocks that start in [b(part), b(part + 1)), where b(k)
 * is the first block start at or behind k / parts of the file's bytes (found by hopping over the block headers, so every caller
 * finds the same boundaries and the parts tile the file) - the record range of one rank of a sharded run (bk_shard_*), decoded on
 * that rank's own GPU; the ranks' record counts give their rec_base.  The header is parsed by every part.  BK_ERR_IO for files
 * whose records run across blocks (they have no such cut points: take the host decoder). */
int bk_bam_decode_device_part(const char *path, int device, int part, int parts, bk_bam_dev **out, bk_soa *cols, int *n_targets, const char *const **names,
                              const uint32_t **lens, char *err, size_t errlen);
void bk_bam_dev_free(bk_bam_dev *h);
/* What the GPU feed keeps between files so that the second file of a process starts at full speed: three page-locked staging
 * buffers per chunk size (32 or 64 MiB each), and per device the feed slots of the last decode (streams, events, device buffers:
 * ~0.3 GB per slot, four to eight slots).  This call gives all of it back; no decode may be running.  The next file pays the
 * first-file cost again (~60 ms: registration of the staging buffers, allocations). */
void bk_feed_release_caches(void);
/* Feed and hot path overlapped: the reference reads the BAM twice, one pass after the other (BreakID.cc:1929,
 * :1414); here the file is read once, and the record-level kernel of the hot path (insert-size sums, discordant filter,
 * SA gate: k_stream) runs on the records of a feed chunk while the following chunks are still being copied and inflated.
 * Creates the context itself (the reference list comes out of the file), attaches the device table (BK_MEM_DEVICE) and
 * returns with the stream pass complete: continue with bk_isize_stats / bk_discordant_pairs(mapq_min) / ...  *bam_out owns
 * the columns and must outlive the context.  Same file support and errors as bk_bam_decode_device. */
int bk_bam_decode_device_ctx(const char *path, int device, int mapq_min, bk_bam_dev **bam_out, bk_ctx **ctx_out, int *n_targets, const char *const **names,
                             const uint32_t **lens, char *err, size_t errlen);
/* test / measurement hook: inflates a whole BGZF file image on the GPU, bytes back to the host */
int bk_debug_bgzf_inflate(const void *file, uint64_t n, void *out, uint64_t out_cap, uint64_t *out_len, float *kernel_ms, char *err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
