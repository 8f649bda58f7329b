/*
 * breakid_multi.h - C ABI of libbreakid_rccl.so: ONE sample over the GPUs of a node without Python.
 *
 * The chromosome-pair groups of the reference are independent (src/BreakID.cc:119-167), the per-record stages are local to
 * a record range; only small derived tables cross the links.  bk_multi_run() holds the orchestration that
 * breakid_amd/sharded.py does over torch.distributed, issued from C++: one host thread per GPU, each with its own bk_ctx
 * (include/breakid_hip.h, bk_shard_* entry points), exchanging through
 *   BK_TRANSPORT_RCCL   librccl directly, on each context's own HIP stream (no host synchronisation between a kernel and
 *                       the collective that ships its output): ncclAllReduce for the coverage / depth counts, grouped
 *                       ncclBroadcast as the variable-size all-gather (no padding to the largest rank), grouped
 *                       ncclSend/ncclRecv for the two all-to-alls (candidates to the owner of their read-name hash, pairs to
 *                       the owner of their chr-pair group); one rank per GPU, xGMI on a node;
 *   BK_TRANSPORT_LOCAL  the same sequence between contexts of one process by device-to-device copies (several contexts may
 *                       share one GPU: this is how a single-GPU box exercises the N-rank code path; RCCL refuses two ranks
 *                       on one device).
 * The BreakID command line reaches it through `--gpus N [--comm rccl|local]` (bk_multi_run_bam, falling back to the host
 * decoder + bk_multi_run).
 */
#ifndef BREAKID_MULTI_H
#define BREAKID_MULTI_H

#include "breakid_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BK_TRANSPORT_AUTO 0  /* RCCL when the node has at least n_gpus devices, else LOCAL */
#define BK_TRANSPORT_RCCL 1
#define BK_TRANSPORT_LOCAL 2

/* Splits the coordinate-sorted host table into n_gpus contiguous record ranges, runs the whole hot path (the body of the
 * reference's main() between BreakID.cc:98 and :167) with rank r on device r % device_count, and returns rank 0's context,
 * which holds the complete cluster table (bk_fetch(BK_STAGE_CLUSTERS)); the caller releases it with bk_multi_free.
 * n_clustered_total = pairs that survived clustering over all groups (what decides whether the index / refGene are opened). */
int bk_multi_run(const bk_soa *host_table, const uint32_t *target_len, const char *const *target_name, int n_targets, int n_gpus, int transport, int mapq_min,
                 int fast, double *w_out, uint64_t *n_clustered_total, bk_ctx **ctx0_out, char *err, size_t errlen);

/* The same from the file: rank r decodes part r of n_gpus of the BAM on its own GPU (bk_bam_decode_device_part: BGZF inflate and
 * record decode on the device, no host table) and takes the records in place; the ranks' record counts give their rec_base.
 * n_targets / names / lens receive the file's reference list (owned by the library, valid until bk_multi_free of the context).
 * BK_ERR_IO for files whose records run across BGZF blocks (no cut points): decode on the host and call bk_multi_run then. */
int bk_multi_run_bam(const char *path, int n_gpus, int transport, int mapq_min, int fast, double *w_out, uint64_t *n_clustered_total, bk_ctx **ctx0_out, int *n_targets,
                     const char *const **names, const uint32_t **lens, char *err, size_t errlen);

/* Frees a context returned by bk_multi_run / bk_multi_run_bam AND the device tables it points at (the gathered tuple / cluster
 * tables; for bk_multi_run_bam also rank 0's decoded records and the reference names).  bk_free alone would leave those allocated. */
void bk_multi_free(bk_ctx *ctx);

/* Insert-size statistics (what bk_isize_stats gives on one GPU) and the per-group counters of the whole sample (bk_group_stats:
 * every group once, summed over the ranks) of the run that returned `ctx`; valid until bk_multi_free. */
int bk_multi_stats(bk_ctx *ctx, double *mean, double *sd, const bk_group_stat **groups, uint32_t *n_groups);

#ifdef __cplusplus
}
#endif
#endif
