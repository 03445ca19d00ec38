/* tfgx_dist.h — C ABI of the halo exchange of a destination-range sharded graph (SURVEY.md §8b last table row, §8e).
 *
 * Lives in its own library (tf_geometric_amd/lib/libtfgx_dist.so = these entry points + librccl) so that libtfgx.so,
 * the compute library, has no communication dependency.  Replaces nothing in the reference — tf_geometric's two
 * "distributed" demos replicate the whole graph per GPU and all-reduce gradients (demo/demo_distributed_gcn.py:38-57,
 * :99); this is what a non-torch host (the tf.load_op_library route of INTEGRATION.md) calls to reach the sharded
 * path: one process per GPU, an ncclComm_t the HOST created (ncclCommInitRank), plain device pointers.
 *
 * Exchange = an all-to-all-v of source-feature rows in R rounds.  Round j: rank p packs the rows peer q asked for
 * (send_idx, precomputed by the plan) and posts grouped ncclSend / ncclRecv to every peer on the COMM stream; the
 * compute stream goes on with the own-source edge pass and later waits per round (exchange_finish) before reducing
 * that round's halo edges.  xGMI is point-to-point: one grouped exchange drives all 7 links of a GPU at once.
 * All functions return 0 on success; message via tfgx_last_error() of libtfgx.so is NOT shared — use
 * tfgx_dist_last_error().
 */
#ifndef TFGX_DIST_H
#define TFGX_DIST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tfgx_halo_plan tfgx_halo_plan;   /* host object: counts, offsets, events */

const char* tfgx_dist_last_error(void);

/* Communicator bootstrap for hosts that do not already own an ncclComm_t (the Python host: torch.distributed does not hand
 * out its communicator).  Rank 0 calls tfgx_dist_unique_id (ncclGetUniqueId) and broadcasts the TFGX_DIST_UNIQUE_ID_BYTES
 * bytes over whatever control channel the host has (the torch.distributed store / a gloo group / MPI); every rank then
 * calls tfgx_dist_comm_init (ncclCommInitRank on the CURRENT HIP device).  *comm_out is an ncclComm_t.
 * ENVIRONMENT: the MI355X host driver supports dmabuf IPC only — the host process must have HSA_ENABLE_IPC_MODE_LEGACY=0 in
 * its environment BEFORE its first HIP call (the ROCr runtime reads it at initialisation; the Python host defaults it at
 * import of tf_geometric_amd.dist, bench.py before importing torch); without it a multi-rank ncclCommInitRank fails inside
 * hipIpcGetMemHandle. */
#define TFGX_DIST_UNIQUE_ID_BYTES 128
int tfgx_dist_unique_id(void* id_out /* TFGX_DIST_UNIQUE_ID_BYTES */);
int tfgx_dist_comm_init(int32_t world, int32_t rank, const void* id, void** comm_out);
int tfgx_dist_comm_destroy(void* nccl_comm);
/* ncclCommAbort: tears the communicator down WITHOUT waiting for the peers (ncclCommDestroy may wait for outstanding
 * operations).  For the failure paths of a bring-up — a rank whose peers reported a failed ncclCommInitRank / self-check
 * must not block inside a collective teardown. */
int tfgx_dist_comm_abort(void* nccl_comm);
/* What the communicator itself reports: ncclCommCount / ncclCommUserRank / ncclCommCuDevice.  bench.py prints world_out as
 * `rccl_ranks`, so a scaling line states how many ranks the RCCL communicator that carried the halo rows really had
 * (any pointer may be NULL). */
int tfgx_dist_comm_info(void* nccl_comm, int32_t* world_out, int32_t* rank_out, int32_t* device_out);

/* Plan-time personalised exchange of raw device bytes (edge routing of ShardedGraph.from_partitioned, halo request
 * lists): peer q receives send[send_off_q ...], counts in ELEMENTS of elem_bytes each (host arrays of `world` entries,
 * offsets are the running sums).  One grouped ncclSend / ncclRecv per peer on `stream`. */
int tfgx_alltoallv(const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts,
                   int64_t elem_bytes, int32_t world, void* nccl_comm, void* stream);
/* sum-all-reduce of an int64 device buffer in place (the in-degree histogram the ranks agree their split points on) */
int tfgx_allreduce_sum_i64(int64_t* buf, int64_t count, void* nccl_comm, void* stream);

/* world, rank        : communicator geometry (must match the ncclComm_t passed later)
 * rounds             : R >= 1
 * send_counts[R*world], recv_counts[R*world] (host): rows sent to / received from peer p in round j at [j*world + p];
 *                      a shard never asks itself for rows, so the plan builder leaves the entries of `rank` itself 0
 *                      (equal non-zero self counts are accepted: RCCL matches a self send/recv inside the group)
 * send_dense_start[R*world] (host) or NULL: >= 0 -> the rows for that (round, peer) are the CONTIGUOUS own rows
 *                      [start, start + count) and are sent straight from x_own — no pack, no send-buffer space (a peer
 *                      that asked for (nearly) all of this rank's rows: every peer of a uniform random graph at 8 GPUs);
 *                      -1 -> packed through send_idx
 * send_idx (device)  : int32 local row ids to pack, concatenated round-major then peer-major over the PACKED (round,
 *                      peer) entries only (tfgx_halo_plan_rows_packed entries); the plan keeps the pointer, the caller
 *                      keeps the memory alive
 * The halo table is laid out round-major then peer-major: round j, peer p starts at row
 * sum(recv_counts[0 .. j*world + p)). */
int tfgx_halo_plan_create(int32_t world, int32_t rank, int32_t rounds, const int64_t* send_counts,
                          const int64_t* recv_counts, const int64_t* send_dense_start, const int32_t* send_idx,
                          tfgx_halo_plan** out);
int tfgx_halo_plan_destroy(tfgx_halo_plan* plan);
int64_t tfgx_halo_plan_rows_sent(const tfgx_halo_plan* plan);
int64_t tfgx_halo_plan_rows_packed(const tfgx_halo_plan* plan);
int64_t tfgx_halo_plan_rows_received(const tfgx_halo_plan* plan);

/* Pack + post every round.  x_own [n_own, F] (ld = ldx; ldx == F when the plan has dense entries): this rank's rows;
 * halo [rows_received, F] (ld = ld_halo); send_buf: device scratch of at least rows_packed * F floats, owned by the
 * caller, untouched until finish(all).
 * nccl_comm: an ncclComm_t.  compute_stream orders the packs after whatever produced x_own; comm_stream carries the
 * sends / receives.  Returns immediately (asynchronous). */
int tfgx_halo_exchange_start(tfgx_halo_plan* plan, const float* x_own, int64_t ldx, int64_t F, float* halo,
                             int64_t ld_halo, float* send_buf, size_t send_buf_floats, void* nccl_comm,
                             void* compute_stream, void* comm_stream);

/* Make compute_stream wait for round `round` (0 <= round < R), or for all rounds when round < 0. */
int tfgx_halo_exchange_finish(tfgx_halo_plan* plan, int32_t round, void* compute_stream);

/* Backward of the exchange (training: d(loss)/d(halo rows) belongs to the rows' owners).  d_halo [rows_received, F]
 * dense, laid out as the halo table; back_buf: device scratch of rows_sent * F floats.  start() posts, per round, the
 * grouped sends of this rank's halo-row gradients and the receives of what peers computed for THIS rank's rows, ordered
 * after whatever wrote d_halo on compute_stream — it returns at once, so the caller can compute the own-row part of the
 * transposed pass on compute_stream while the rounds are on the wire.  finish() makes compute_stream wait round by round
 * and adds the returned rows into d_own [n_own, F] (ld = ldd) at the forward send indices (dense entries: at their
 * contiguous row range) — round by round, peer by peer in rank order, one writer per element and launch
 * (tfgx_scatter_add_rows_f32): bit-reproducible, no atomics. */
int tfgx_halo_reverse_start(tfgx_halo_plan* plan, const float* d_halo, int64_t F, float* back_buf,
                            size_t back_buf_floats, void* nccl_comm, void* compute_stream, void* comm_stream);
/* The same round by round: round `round` (0, 1, ..., R - 1, in this order) is posted after whatever wrote ITS rows of d_halo
 * (the halo table is round-major: rows [recv_off[round * world], recv_off[(round + 1) * world])) on compute_stream, so the
 * caller runs the transposed pass window by window and every round travels while the later windows are computed.
 * finish() as above, after all R rounds were started.  Starting round 0 while an earlier sequence is half started abandons
 * that sequence (its posted rounds are waited for) and begins a new one. */
int tfgx_halo_reverse_start_round(tfgx_halo_plan* plan, int32_t round, const float* d_halo, int64_t F, float* back_buf,
                                  size_t back_buf_floats, void* nccl_comm, void* compute_stream, void* comm_stream);
int tfgx_halo_reverse_finish(tfgx_halo_plan* plan, float* d_own, int64_t ldd, int64_t F, const float* back_buf,
                             void* compute_stream);

/* sum-all-reduce of a float buffer in place (weight gradients of replicated layer weights: the one collective the
 * reference's distributed demos perform, demo/demo_distributed_gcn.py:52-57). */
int tfgx_allreduce_sum_f32(float* buf, int64_t count, void* nccl_comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TFGX_DIST_H */
