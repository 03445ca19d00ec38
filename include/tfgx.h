/*
 * tfgx — C ABI of the MI355X (gfx950) message-passing backend for tf_geometric.
 *
 * The reference (tf_geometric 0.1.7) is pure Python: it has no FFI of its own.  Its hot path
 * bottoms out in TensorFlow / tf_sparse ops; each entry point below replaces the ops named in
 * its comment (reference file:line, relative to /root/reference).  A binding for the reference
 * (ctypes today, a tf.load_op_library shim where TensorFlow exists) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless it says "host";
 *   - nothing is allocated here: outputs and workspaces are caller buffers;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), except
 *     tfgx_build_csr_by_dst and tfgx_segment_topk, which synchronise once to report bad indices;
 *   - return value: 0 = ok, otherwise a TFGX_ERR_* code; text via tfgx_last_error();
 *   - results are deterministic (no floating-point atomics anywhere);
 *   - index convention of the reference: edge_index[0] = row = DESTINATION (aggregating node),
 *     edge_index[1] = col = SOURCE (neighbour whose features are gathered)
 *     (tf_geometric/nn/kernel/map_reduce.py:60-70).
 *   - all features float32 row-major, all indices int32 (tf_geometric/data/graph.py:22-23),
 *     E < 2^31; element offsets are 64-bit inside the kernels.
 */
#ifndef TFGX_H
#define TFGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tfgx_stream_t; /* hipStream_t */

enum {
    TFGX_OK = 0,
    TFGX_ERR_INVALID_ARG = 1,  /* null pointer / negative size / unsupported combination */
    TFGX_ERR_INDEX = 2,        /* an edge endpoint is outside [0, n): TF-CPU raises InvalidArgumentError here */
    TFGX_ERR_WORKSPACE = 3,    /* workspace too small */
    TFGX_ERR_HIP = 4           /* a HIP runtime call failed */
};

enum { TFGX_SUM = 0, TFGX_MEAN = 1, TFGX_MAX = 2 };
enum { TFGX_ACT_NONE = 0, TFGX_ACT_RELU = 1 };
enum { TFGX_NORM_BOTH = 0, TFGX_NORM_LEFT = 1, TFGX_NORM_RIGHT = 2 };

/* ABI version of this header: bumped whenever an entry point's signature or a struct's layout changes (a host built
 * against another value must refuse to run: tf_geometric_amd/_lib.py does).  100 = rounds 1-3; 110 = round 4
 * (tfgx_reduce_args.hub_order_slot; tfgx_aggregate_gemm_f32 honours args->out as a side output of the aggregate;
 * tfgx_gat_args / tfgx_gat_backward_args .drop_seed_dev); 111 = + tfgx_column_sum_f32; 112 = round 5 (+ tfgx_split_rows_verify_f32, tfgx_reduce_args.wide_blocks, tfgx_gat_args.state_in_*, tfgx_gat_backward_args.span_*);
 * 113 = round 6 (+ tfgx_gat_backward_args.head_pack / ld_head_pack, tfgx_gat_pack_dst_heads_f32,
 * tfgx_pool_mlp_max_wgrad_*); 114 = + tfgx_gat_args.qgrad_t / qgrad_s / state_t / state_s / state_in_t / state_in_s,
 * tfgx_gat_query_grad_d1_f32. */
#define TFGX_ABI_VERSION 114
int tfgx_version(void);            /* the TFGX_ABI_VERSION the library was built with */
const char* tfgx_last_error(void); /* host string, thread-local, valid until the next failing call */

/* ---------------------------------------------------------------------------------------------
 * Plan: CSR-by-destination.  Replaces the per-call scatter of tf.math.unsorted_segment_* (and the
 * TF1 path's tf.argsort + gathers, tf_geometric/nn/kernel/segment.py:7-11).  Built once per graph
 * and cached like the reference caches its normalised adjacency (nn/conv/gcn.py:125-128).
 *   row_ptr[n_dst+1], col_sorted[E], perm[E]:  CSR position i holds original edge perm[i];
 *   the sort is STABLE, so edges of one destination keep their original relative order.
 * ------------------------------------------------------------------------------------------- */
size_t tfgx_csr_plan_workspace_bytes(int64_t n_dst, int64_t E);
int tfgx_build_csr_by_dst(const int32_t* row, const int32_t* col, int64_t E, int64_t n_dst, int64_t n_src,
                          int32_t* row_ptr, int32_t* col_sorted, int32_t* perm,
                          void* workspace, size_t workspace_bytes, tfgx_stream_t stream);

/* dst[i, :] = src[perm[i], :]  (edge attributes into CSR order); width floats per edge */
int tfgx_permute_rows_f32(const float* src, const int32_t* perm, int64_t E, int64_t width, float* dst,
                          tfgx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Edge preprocessing right before the path: merge_duplicated_edge
 * (tf_geometric/utils/graph_utils.py:67-125) = tf.unique over the hash n*row+col.  Outputs the unique edges in
 * FIRST-OCCURRENCE order (tf.unique's order) and unique_index[E] (edge -> unique edge), with which edge
 * properties are merged by tfgx_segment_reduce_f32 (sum / mean / max; min = -max(-x)).  *n_unique is a device int32.
 * convert_edge_to_upper / convert_edge_to_directed (:126-212) are this call on (min, max) endpoint pairs.
 * ------------------------------------------------------------------------------------------- */
size_t tfgx_merge_edges_workspace_bytes(int64_t E, int64_t n);
int tfgx_merge_duplicated_edges(const int32_t* row, const int32_t* col, int64_t E, int64_t n, int32_t* out_row,
                                int32_t* out_col, int32_t* unique_index, int32_t* n_unique, void* workspace,
                                size_t workspace_bytes, tfgx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Gather - scale - segment reduce.  Replaces, fused and without materialising [E,F]:
 *   tf.gather(x, col)                               nn/kernel/map_reduce.py:63, nn/conv/graph_sage.py:36
 *   gcn_mapper (neighbor_x * w[:,None])             nn/conv/gcn.py:221-222
 *   tf.math.unsorted_segment_sum / _mean / _max     nn/kernel/map_reduce.py:16, :28, :41
 *   tf_sparse SparseMatrix.matmul / @               nn/conv/gcn.py:280, nn/conv/gat.py:89
 *   sum_updater, "+ bias", activation               map_reduce.py:19-20, gcn.py:284-288
 * out[r,:] = act( combine( reduce_{i in [rb[r*s], re[r*s])} w[i]*x[col[i],:]  (+ self_coef[r]*x[r,:]) )
 *                 (+ add_x[r,:]) (+ bias) )
 * Empty destination: 0 (sum, mean), -FLT_MAX (max) — TF semantics.
 * ------------------------------------------------------------------------------------------- */
typedef struct tfgx_reduce_args {
    const int32_t* row_begin; /* row r spans CSR positions [row_begin[r*rp_stride], row_end[r*rp_stride]) */
    const int32_t* row_end;   /* plain CSR: row_begin=row_ptr, row_end=row_ptr+1, rp_stride=1 */
    int64_t rp_stride;
    const int32_t* col;       /* source row of x per CSR position */
    const float* w;           /* per CSR position, or NULL (unweighted: identity_mapper) */
    int64_t n_dst;
    const float* x;           /* [n_src, ldx] */
    int64_t ldx;
    int64_t F;                /* columns reduced */
    float* out;               /* [n_dst, ldo] */
    int64_t ldo;
    int32_t op;               /* TFGX_SUM | TFGX_MEAN | TFGX_MAX */
    int32_t act;              /* TFGX_ACT_* applied last */
    int32_t accumulate;       /* 1: combine with the values already in out (sum/mean: add, max: max) — the
                                 second pass of a local/halo split plan; epilogue terms are applied after */
    int32_t hub_threshold;    /* 0: off. > 0: rows with more than hub_threshold edges ("hubs" of a power-law graph) are
                                 skipped by the main launch and reduced chunk-wise from the lists below */
    const float* self_coef;   /* [n_dst] or NULL: an implicit edge (r, r) of weight self_coef[r] appended after
                                 the row's edges (SparseMatrix.add_diag, nn/conv/gcn.py:72,77,98) */
    const float* bias;        /* [F] or NULL */
    const float* add_x;       /* [n_dst, ld_add] or NULL: sum_updater's "x +" */
    int64_t ld_add;
    const int32_t* mean_count;/* [n_dst] or NULL: divisor for TFGX_MEAN (NULL: row_end-row_begin); <1 -> 1 */
    /* hub rows (only read when hub_threshold > 0 and n_hub_rows > 0); built once per plan by the host */
    const int32_t* hub_rows;        /* [n_hub_rows] destination ids with in-degree > hub_threshold, ascending */
    const int32_t* hub_chunk_ptr;   /* [n_hub_rows+1] chunk range of each hub row */
    const int32_t* hub_chunk_begin; /* [n_hub_chunks] CSR positions: chunk c covers [begin[c], end[c]) */
    const int32_t* hub_chunk_end;
    int64_t n_hub_rows;
    int64_t n_hub_chunks;
    float* hub_scratch;             /* [n_hub_chunks, F] workspace for the per-chunk partial results */
    /* optional split source rows: x holds columns [0, f_main) with leading dimension ldx, x_tail holds columns
       [f_main, F) with leading dimension ld_tail.  With f_main a multiple of 32 and 128-byte aligned rows every
       fetched line of x is fully used (a 400-byte row otherwise straddles four 128-byte lines = 512 bytes) and the
       narrow tail array (4*(F-f_main) bytes per node) stays cache-resident.  NULL = ordinary single array. */
    const float* x_tail;
    int64_t ld_tail;
    int64_t f_main;
    /* optional, with x_tail: edge_tail[i, :] = x_tail[col[i], :] for every edge position i of THIS plan (built once per
       (plan, x) with tfgx_gather_rows_f32).  The tail columns are then streamed next to col / w instead of gathered:
       a 400-byte row costs three line requests + 16 streamed bytes instead of four requests.  For source features
       that do not change between launches (the dataset's input features: layer 0 of every model, every epoch). */
    const float* edge_tail;
    int64_t ld_edge_tail;
    /* optional walk order: lane group i of the launch reduces destination row row_order[i] (NULL: i) — the plan's rows
       sorted by length on skewed graphs, so that the rows sharing a wave are of similar length; a permutation of
       [0, n_dst), results do not depend on it */
    const int32_t* row_order;
    /* optional, TFGX_MAX only (the TRAINING forward of max aggregation): per output element one uint32 =
       (number of edges attaining the row maximum, saturating at 65535) << 16 | (CSR position of the FIRST such edge
       relative to its row's first position; 0xFFFF for an empty row) — what TensorFlow's unsorted_segment_max gradient
       needs (it divides by the tie count, math_grad._UnsortedSegmentMinOrMaxGrad), produced by the tuned forward walk in
       the same pass.  Rows must hold fewer than 65536 edges (longer rows are hub rows: hub_threshold must be 0 here);
       16-byte aligned rows of F <= 256 columns, F % 4 == 0; no self_coef / bias / activation / split rows.
       With accumulate = 1 the launch MERGES into the (out, track) stored by earlier launches over earlier sub-spans of the
       same rows (a larger maximum replaces count and position, an equal one adds its tie count and keeps the earlier
       position): a row reduced span by span — the sharded path's own-source pass, then one pass per halo round — ends
       with the same (out, track) as one launch over the whole row.  track_row_begin (optional, same stride as
       row_begin): positions are stored relative to track_row_begin[row * rp_stride] (the first position of the WHOLE
       row) instead of this launch's row_begin. */
    uint32_t* track;
    int64_t ld_track;
    const int32_t* track_row_begin;
    /* optional, read by tfgx_aggregate_gemm_f32 only, with row_order and hub lists: hub_order_slot[i] = index into hub_rows of
       destination row row_order[i], for i < n_hub_rows (a walk order sorted by descending length puts the hub rows first).
       Checked against hub_rows before use; NULL or a mismatch costs a binary search per hub row instead of one load. */
    const int32_t* hub_order_slot;
    /* Wide rows (F >= 128 made of whole 128-byte lines; tfgx_segment_reduce_f32 only): 0 = the library's policy — column blocks
       of 64 columns on grid.y, every pass gathering one block of every source row (round 5: +7 ... +17 % at F = 128 ... 512 on
       the uniform products-shaped graph) —, 1 = blocks wherever the layout allows, -1 = one burst per source row (what a host
       passes for a power-law plan at widths that are not a power of two: its walk is mostly short rows, whose start-up is
       paid once per pass).  Results do not depend on it. */
    int32_t wide_blocks;
    int32_t reserved_r5;
} tfgx_reduce_args;

int tfgx_segment_reduce_f32(const tfgx_reduce_args* args /* host */, tfgx_stream_t stream);

/* The kernel symbol tfgx_segment_reduce_f32 would launch for `args` (template arguments as rocprofv3 prints them:
   seg_reduce_kernel<VEC, G, CH, IS_MAX, WEIGHTED, SPLIT, TRACK>), written NUL-terminated into buf.  Host-only, launches
   nothing: measurement code (bench.py's roofline.kernel) names the kernel from the dispatch itself. */
int tfgx_segment_reduce_describe(const tfgx_reduce_args* args /* host */, char* buf, size_t buf_bytes);

/* Aggregation -> projection in ONE launch (the aggregate-then-project layers: GCN when units > F evaluated as
 * (A_hat x) W, nn/conv/gcn.py:272-288; the neighbour half of mean / sum GraphSAGE, nn/conv/graph_sage.py:34-58):
 *     C[n_dst, N] = act( reduce(args) @ B[F, N] + bias )
 * where reduce(args) is exactly what tfgx_segment_reduce_f32 would write for `args` (op TFGX_SUM | TFGX_MEAN; w, self_coef,
 * mean_count honoured; args->act / bias are NOT used) — but the [n_dst, F] aggregate is not read back from HBM: 64-row
 * tiles go registers -> LDS -> v_mfma_f32_32x32x2_f32 against B, bias / activation in the epilogue.  B is resident in LDS as
 * far as it fits beside the two tiles (all of it up to F = 100 -> 256; its first 128 columns at F = 128 -> 256); the
 * consumer jobs of the remaining columns read their B operand from global memory (L2-resident: <= 128 KB re-read per tile).
 * args->out: NULL, or [n_dst, ldo] (16-byte aligned rows) that ALSO receives the aggregate itself — the training forward,
 * whose weight gradient needs it; the projection still takes it from LDS.
 * Needs a plain CSR (row_begin = row_ptr, row_end = row_ptr + 1, rp_stride = 1), 16-byte aligned rows, F % 4 == 0,
 * F <= 128, N <= 256: tfgx_aggregate_gemm_fits(F, N) == 1; no accumulate / add_x / track.  Split source rows (x_tail / f_main /
 * edge_tail: the static feature layout) are honoured exactly as by tfgx_segment_reduce_f32.  Hub lists (hub_threshold, hub_rows, hub_chunk_*, hub_scratch) are honoured: the chunks
 * are reduced into hub_scratch by a launch of the ordinary kernel first, and a long row's lane group folds its chunk
 * partials in chunk order instead of walking the edges.  Deterministic.  Callers fall back to the two launches otherwise. */
int tfgx_aggregate_gemm_fits(int64_t F, int64_t N);
int tfgx_aggregate_gemm_f32(const tfgx_reduce_args* args /* host */, const float* B, int64_t ldb,
                            const float* bias /* [N] or NULL */, int32_t act, float* C, int64_t ldc, int64_t N,
                            tfgx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * GCN normalisation (nn/conv/gcn.py:32-130), on the CSR plan.
 *   tfgx_segment_weight_sum_f32 : deg[r] = sum_{i in row r} w[i] (+ diag)     SparseMatrix.segment_sum(axis=-1) :80
 *   tfgx_gcn_norm_edges_f32     : w_out[i], self_coef[r] per norm mode        :62-119
 * The added diagonal (add_diag(fill)) is kept implicit as self_coef (see tfgx_reduce_args).
 *   BOTH : renorm -> deg incl. fill; w' = dis[r]*w*disc[c], self = dis[r]*fill*disc[r];
 *          !renorm -> deg excl. fill; self = fill                            :74-98
 *   LEFT : w' = w/deg[r], self = fill/deg[r]   (deg incl. fill if add_self_loop)   :71-72,101-109
 *   RIGHT: w' = w/deg[c], self = fill/deg[r]   (row degrees, as the reference)     :111-119
 * row_deg must already contain the diagonal where the mode says so (pass diag to weight_sum).
 * col_deg: column-side degrees for sym=False, or NULL to reuse row_deg (sym=True, :85-86).
 * ------------------------------------------------------------------------------------------- */
int tfgx_segment_weight_sum_f32(const int32_t* row_ptr, const float* w /* or NULL = ones */, int64_t n,
                                float diag, float* deg, tfgx_stream_t stream);
int tfgx_gcn_norm_edges_f32(const int32_t* row_ptr, const int32_t* col, const float* w /* or NULL */,
                            int64_t n, const float* row_deg, const float* col_deg /* or NULL */,
                            int32_t norm_mode, float fill, int32_t add_self_loop, int32_t renorm,
                            float* w_out /* [E] */, float* self_coef /* [n] */, tfgx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Edge softmax grouped by destination (nn/kernel/segment.py:26-33; SparseMatrix.segment_softmax(axis=-1),
 * nn/conv/gat.py:83-84).  score/out are [E, H]; if perm != NULL they are in the CALLER's edge order
 * (CSR position i <-> edge perm[i]), else in CSR order.
 * ------------------------------------------------------------------------------------------- */
int tfgx_edge_softmax_f32(const int32_t* row_ptr, const int32_t* perm /* or NULL */, const float* score,
                          int64_t H, int64_t n_dst, float* out, tfgx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused GAT attention (nn/conv/gat.py:56-89 + :112): per destination r and head h, over the row's edges
 * plus (add_self_loop) the appended edge (r,r):
 *   s_e = <Q[r,h,:], K[c_e,h,:]> / scale ; a_e = exp(s_e - max) / (sum exp(s - max) + 1e-8)
 *   out[r, h*dv : (h+1)*dv] = sum_e a_e * V[c_e, h*dv : (h+1)*dv]   (+ bias, act)
 * One pass (online softmax); neither [E,A] gathers nor the [2,H*E] virtual graph are materialised.
 * ------------------------------------------------------------------------------------------- */
typedef struct tfgx_gat_args {
    const int32_t* row_ptr;
    const int32_t* col;
    int64_t n_dst;
    const float* q; int64_t ldq;   /* [n_dst, H*d]  */
    const float* k; int64_t ldk;   /* [n_src, H*d]  */
    const float* v; int64_t ldv;   /* [n_src, H*dv] */
    float* out; int64_t ldo;       /* [n_dst, H*dv] */
    int32_t H, d, dv;
    int32_t add_self_loop;         /* utils/graph_utils.py:350-366 (appended last) */
    float scale;                   /* sqrt(d) (gat.py:78) */
    int32_t act;
    const float* bias;             /* [H*dv] or NULL */
    /* optional: explicit row spans (as in tfgx_reduce_args); NULL row_begin = plain row_ptr */
    const int32_t* row_begin;
    const int32_t* row_end;
    int64_t rp_stride;
    /* optional raw-state output: when non-NULL the kernel writes, per destination row, the un-normalised online
       softmax state acc[H*dv] and (m, l)[H] over the given edge span and does NOT add the self-loop; several such
       passes over disjoint edge spans (local / halo halves of a split plan) are combined by
       tfgx_gat_merge_passes_f32. */
    float* state_acc;              /* [n_dst, H*dv] */
    float* state_ml;               /* [n_dst, 2*H]  */
    /* hub rows (as in tfgx_reduce_args) + the destination of every chunk and two scratch arrays */
    int32_t hub_threshold;
    int32_t reserved;
    const int32_t* hub_rows;
    const int32_t* hub_chunk_ptr;
    const int32_t* hub_chunk_begin;
    const int32_t* hub_chunk_end;
    const int32_t* hub_chunk_row;  /* [n_hub_chunks] destination id of each chunk */
    int64_t n_hub_rows;
    int64_t n_hub_chunks;
    float* hub_scratch_acc;        /* [n_hub_chunks, H*dv] */
    float* hub_scratch_ml;         /* [n_hub_chunks, 2*H]  */
    float* stats_ml;               /* optional [n_dst, 2*H]: final softmax statistics (m, l) per row and head, saved
                                      for tfgx_gat_backward_*; NULL = not written */
    /* training only — dropout of the attention weights AFTER the softmax (SparseMatrix.dropout, gat.py:85; tf.nn.dropout
       semantics: a_e -> a_e * keep_e / (1 - rate)).  keep_e is a pure function of (drop_seed, edge position in this
       plan's CSR order, head) — tfgx_dropout_keep() below — so the backward kernels regenerate it.  The appended
       self-loop edge of row r has position drop_self_base + r (pass the plan's edge count).  0 = no dropout; with
       dropout the hub / raw-state options must be off. */
    float drop_rate;
    int32_t reserved2;
    uint64_t drop_seed;
    int64_t drop_self_base;
    /* optional walk order: lane group i of the launch processes destination row_order[i] (NULL: i).  On a power-law graph
       the host passes the rows sorted by in-degree, so that the rows sharing a wave have similar lengths (13-16 % on the
       whole layer at R-MAT shapes); results do not depend on it.  Ignored by the part / chunk launches. */
    const int32_t* row_order;
    /* optional, with drop_rate > 0: the seed is read from this DEVICE location by the kernel instead of drop_seed — for
       launches captured into a hipGraph, whose arguments are frozen: the captured step advances the value on the device, so
       every replay draws a new mask (forward and backward of one step read the same location). */
    const uint64_t* drop_seed_dev;
    /* optional (round 5): RESUME.  The walk of every launched part starts from the raw state stored here — (acc, m, l) as
       state_acc / state_ml hold them, same indexing — instead of the empty state, and ends as usual: with state_acc set the
       updated raw state is written (it must not alias the state read), without it the row is finished (self-loop, normalise,
       bias, activation, stats_ml).  Chaining KB launches over the SOURCE BLOCKS of a plan whose rows are partitioned by
       source range (row_begin / row_end / rp_stride = KB) makes every launch gather K / V rows of one block only: on a dense
       graph (Reddit shape: 489 in-edges per node, 67 MB of K | V rows) a block fits the L2 of every XCD and the layer's
       attention runs 1.5x faster.  Not combinable with drop_rate > 0 or the hub lists. */
    const float* state_in_acc;     /* [n_parts, H*dv] */
    const float* state_in_ml;      /* [n_parts, 2*H]  */
    /* optional (round 6, ABI 114; d == 1 only — the demo's literal layer GAT(units, num_heads=8, attention_units=8),
       demo/demo_gat.py:22): the sums the QUERY gradient needs, accumulated by the same walk out of the K and V values it
       holds anyway:
         qgrad_t[r, h*dv + j] = sum_e a_e keep_e K[c_e, h] V[c_e, h*dv + j]        qgrad_s[r, h] = sum_e a_e K[c_e, h]
       so that dQ[r, h] = sum_e a_e (keep_e <dO[r,h,:], V[c_e,h,:]> - D[r,h]) K[c_e, h] / scale
                        = (<dO[r,h,:], qgrad_t[r,h,:]> - D[r,h] qgrad_s[r,h]) / scale          (tfgx_gat_query_grad_d1_f32)
       is a per-ROW expression and the backward's destination pass (tfgx_gat_backward_dst_*: a second walk over every edge
       that gathers K and V again) is not run.  A finishing launch writes both (dense rows, 16-byte aligned); a raw-state
       launch (state_acc != NULL) carries the un-normalised sums in state_t / state_s and a resumed one reads
       state_in_t / state_in_s — same indexing as state_acc / [n_parts, H].  All NULL = not computed.  Needs dv % 4 == 0 and
       16-byte aligned V / out rows; not combinable with the hub lists. */
    float* qgrad_t;                /* [n_dst, H*dv] */
    float* qgrad_s;                /* [n_dst, H] */
    float* state_t;                /* [n_parts, H*dv] */
    float* state_s;                /* [n_parts, H] */
    const float* state_in_t;       /* [n_parts, H*dv] */
    const float* state_in_s;       /* [n_parts, H] */
} tfgx_gat_args;

/* 1 if the item survives dropout at `rate`, else 0: the exact decision every kernel of this library makes for
   item = (uint32)(position * H + head) (attention) or the edge position (edge-weight dropout).  Host function. */
int32_t tfgx_dropout_keep(uint64_t seed, uint32_t item, float rate);

int tfgx_gat_fused_f32(const tfgx_gat_args* args /* host */, tfgx_stream_t stream);

/* out[r] = softmax-merge of n_passes raw states (pass t of row r at index t*n_dst + r) + self-loop edge + bias + act;
   q/k/v/out/H/d/dv/scale/add_self_loop/act/bias/n_dst are read from args */
/* The general form: row i of the output merges the raw states part_idx[part_ptr[i] .. part_ptr[i+1]) — any mix of whole
   passes and chunks of long spans.  Raw-state launches of tfgx_gat_fused_f32 (state_acc != NULL) accept two of the hub
   fields for this: hub_chunk_row[p] = destination (Q row) of launched part p (NULL: p itself), and hub_threshold > 0 =
   skip spans longer than that (their chunks are launched separately, with row_begin / row_end = the chunk bounds). */
int tfgx_gat_merge_parts_f32(const tfgx_gat_args* args /* host */, const float* state_acc, const float* state_ml,
                             const int32_t* part_ptr /* [n_dst+1] */, const int32_t* part_idx, tfgx_stream_t stream);
int tfgx_gat_merge_passes_f32(const tfgx_gat_args* args /* host */, const float* state_acc, const float* state_ml,
                              int32_t n_passes, tfgx_stream_t stream);

/* out[r, j] = (1/H) * sum_h in[r, h*U + j]  (+ bias[j], act)   — gat.py:114-120, split_value_heads=False */
int tfgx_head_mean_f32(const float* in, int64_t ld_in, int64_t n, int32_t H, int32_t U, const float* bias,
                       int32_t act, float* out, int64_t ldo, tfgx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Backward pass: what tf.GradientTape differentiates in the reference's training loops
 * (demo/demo_gcn.py:68-77).  d/dx of the sum/mean aggregation is tfgx_segment_reduce_f32 on the transposed
 * (CSR-by-source) plan; the entry points below cover the rest.
 *   tfgx_sddmm_f32                 d/dw of gcn_mapper+sum: out[i] = <a[row(i), :], b[col[i], :]>  (CSR order)
 *   tfgx_segment_max_count_f32     count[r,j] = #{i in row r : w[i]*x[col[i],j] == out[r,j]}
 *   tfgx_segment_max_backward_f32  gx[c,j] = sum over the transposed plan of [tie] * w * g[dst,j] / count[dst,j]
 *                                  (tf.math.unsorted_segment_max's gradient: split evenly among tied maxima)
 *   tfgx_gat_backward_dst_f32      dQ; tfgx_gat_backward_src_f32: dK, dV — attention weights recomputed from
 *                                  stats_ml saved by tfgx_gat_fused_f32; dsum[r,h] = <dO[r,h,:], O[r,h,:]>
 * ------------------------------------------------------------------------------------------- */
int tfgx_sddmm_f32(const int32_t* row_ptr, const int32_t* col, int64_t n_dst, const float* a, int64_t lda,
                   const float* b, int64_t ldb, int64_t F, float* out, tfgx_stream_t stream);
/* training-time forward of the max aggregation: out[r,j] = max_i w[i]*x[col[i],j] (float32 lowest for an empty row) AND
   count[r,j] = number of edges attaining it, in one pass over the edges (the backward then needs no count pass) */
int tfgx_segment_max_with_count_f32(const int32_t* row_ptr, const int32_t* col, const float* w /* or NULL */,
                                    int64_t n_dst, const float* x, int64_t ldx, int64_t F, float* out, int64_t ldo,
                                    float* count, int64_t ldc, tfgx_stream_t stream);
int tfgx_segment_max_count_f32(const int32_t* row_ptr, const int32_t* col, const float* w /* or NULL */, int64_t n_dst,
                               const float* x, int64_t ldx, int64_t F, const float* out, int64_t ldo, float* count,
                               int64_t ldc, tfgx_stream_t stream);
/* d(max aggregate)/d(edge weight), forward-CSR order: grad_w[i] = sum_j [w[i]*x[col[i],j] == out[r,j]] * x[col[i],j] * gn[r,j]
   with gn = grad_out / count (the tie-splitting of tf.math.unsorted_segment_max's gradient), r = row of position i */
int tfgx_segment_max_backward_w_f32(const int32_t* row_ptr, const int32_t* col, const float* w, int64_t n_dst,
                                    const float* x, int64_t ldx, int64_t F, const float* out, int64_t ldo,
                                    const float* gn, int64_t ldgn, float* grad_w, tfgx_stream_t stream);
/* Push form of the same gradient (default for training on graphs without hub rows): the training forward also saves,
   per (row, column), the CSR position of the FIRST maximal edge (argpos, -1 for an empty row); the backward then adds
   w[argpos] * g[r, j] to gx[col[argpos], j] with one float atomic per (row, column) — N*F atomics instead of two gathered
   rows per edge — and walks rows with tied maxima (count > 1) exactly, handing g / count to every tied edge (TensorFlow's
   unsorted_segment_max gradient).  gx is zeroed here.  Float atomics commit in arrival order: sums may differ in the
   last bit between runs (as TensorFlow's GPU kernels do); use tfgx_segment_max_backward_f32 for bit-reproducibility.
   Needs 16-byte aligned rows and F % 4 == 0. */
int tfgx_segment_max_with_arg_f32(const int32_t* row_ptr, const int32_t* col, const float* w /* or NULL */,
                                  int64_t n_dst, const float* x, int64_t ldx, int64_t F, float* out, int64_t ldo,
                                  float* count, int64_t ldc, int32_t* argpos, int64_t lda, tfgx_stream_t stream);
int tfgx_segment_max_backward_push_f32(const int32_t* row_ptr, const int32_t* col, const float* w /* or NULL */,
                                       int64_t n_dst, int64_t n_src, const float* x, int64_t ldx, int64_t F,
                                       const float* out, int64_t ldo, const float* g, int64_t ldg, const float* count,
                                       int64_t ldc, const int32_t* argpos, int64_t lda, float* gx, int64_t ldgx,
                                       tfgx_stream_t stream);
/* Mask form of the same gradient: deterministic AND one gather per edge.  From the arg positions / tie counts of
   tfgx_segment_max_with_arg_f32 a per-edge bit mask over the F columns is built (bit j of edge p: p attains the maximum
   of column j of its row; rows with tied maxima are walked exactly so every tied edge is marked) together with
   gn = g / count; then every SOURCE row walks its out-edges in transposed order (row_ptr_t, dst_t, w_t, and pos_t = the
   forward CSR position of each transposed position, or NULL if identical), reads 4*ceil(F/32) mask bytes per edge and
   gathers gn[dst, j] only where a bit is set (N*F/E columns per edge on average).  One owner per gx element, fixed
   order: bit-reproducible.  workspace: tfgx_segment_max_backward_mask_workspace_bytes(n_dst, E, F) bytes.
   count == NULL: `argpos` holds the PACKED uint32 array of tfgx_reduce_args.track (tie count << 16 | row-relative
   position) written by the tuned training forward — one array read instead of two. */
size_t tfgx_segment_max_backward_mask_workspace_bytes(int64_t n_dst, int64_t E, int64_t F);
int tfgx_segment_max_backward_mask_f32(const int32_t* row_ptr, const int32_t* col, const float* w /* or NULL */,
                                       int64_t n_dst, int64_t E, const float* x, int64_t ldx, int64_t F, const float* out,
                                       int64_t ldo, const float* g, int64_t ldg, const float* count, int64_t ldc,
                                       const int32_t* argpos, int64_t lda, const int32_t* row_ptr_t, const int32_t* dst_t,
                                       const float* w_t /* or NULL */, const int32_t* pos_t /* or NULL */, int64_t n_src,
                                       float* gx, int64_t ldgx, void* workspace, size_t workspace_bytes,
                                       tfgx_stream_t stream);
/* The same in PHASES (the sharded path sends the halo rows' gradients back while the own rows' are still computed):
   phases bit 0 = build gn and the masks into `workspace` (reads the forward arrays; row_ptr_t / gx unused), bit 1 = apply
   them to a WINDOW of source rows: row_ptr_t points at the window's first row, n_src = rows in the window, gx at the
   window's first output row; n_dst, E, F and the workspace must be the ones of the build call.  3 = both (the call above). */
int tfgx_segment_max_backward_mask_phases_f32(const int32_t* row_ptr, const int32_t* col, const float* w /* or NULL */,
                                              int64_t n_dst, int64_t E, const float* x, int64_t ldx, int64_t F,
                                              const float* out, int64_t ldo, const float* g, int64_t ldg,
                                              const float* count, int64_t ldc, const int32_t* argpos, int64_t lda,
                                              const int32_t* row_ptr_t, const int32_t* dst_t, const float* w_t /* or NULL */,
                                              const int32_t* pos_t /* or NULL */, int64_t n_src, float* gx, int64_t ldgx,
                                              void* workspace, size_t workspace_bytes, int32_t phases, tfgx_stream_t stream);
int tfgx_segment_max_backward_f32(const int32_t* row_ptr_t, const int32_t* dst_t, const float* w_t /* or NULL */,
                                  int64_t n_src, const float* x, int64_t ldx, int64_t F, const float* out, int64_t ldo,
                                  const float* g, int64_t ldg, const float* count, int64_t ldc, float* gx, int64_t ldgx,
                                  int64_t n_dst, float* gn_scratch /* [n_dst, F] workspace or NULL (slow path) */,
                                  tfgx_stream_t stream);

/* Max-pool GraphSAGE, layer 0 (round 6, ABI 113): the weight / bias gradient of the pooling MLP h = relu(x W + b) under
   red[r, :] = max over in-edges of h[col, :] (tf_geometric/nn/conv/graph_sage.py:260-269) WITHOUT materialising dh:
       dW[k, j] = sum_r [red[r, j] > 0] g[r, j] / count[r, j] * x[w, k] summed over the sources w that attain the maximum,
       db[j]    = the same with x = 1
   destination-major: the x rows of a destination's in-edges are staged in LDS once (the forward's own gather) and every
   (feature, column) accumulator is a register of one thread for the whole launch — deterministic, no atomics.  `packed` is the
   tracked forward's (count << 16 | position) array (tfgx_reduce_args.track), `h` is read only where a maximum is tied.
   Replaces tfgx_segment_max_backward_mask_f32 + the ReLU-mask pass + tfgx_gemm_tn_f32 when NO gradient w.r.t. x is wanted.
   Shapes: F_in a multiple of 4 in [4, 124], Fp in {128, 256, 512} (tfgx_pool_mlp_max_wgrad_applies), rows shorter than 65536
   edges (the packed format's bound). */
int tfgx_pool_mlp_max_wgrad_applies(int64_t F_in, int64_t Fp);
/* once per graph and (F_in, Fp): the launch's work items (rows cut into chunks of edges that fit the LDS stage, workgroup by
   workgroup) into a caller-owned device buffer; E = row_ptr[n_dst] */
size_t tfgx_pool_mlp_max_wgrad_plan_bytes(int64_t n_dst, int64_t E, int64_t F_in, int64_t Fp);
int tfgx_pool_mlp_max_wgrad_plan(const int32_t* row_ptr, int64_t n_dst, int64_t E, int64_t F_in, int64_t Fp, void* plan_buf,
                                 size_t plan_bytes, tfgx_stream_t stream);
size_t tfgx_pool_mlp_max_wgrad_workspace_bytes(int64_t n_dst, int64_t F_in, int64_t Fp);
int tfgx_pool_mlp_max_wgrad_f32(const int32_t* row_ptr, const int32_t* col, int64_t n_dst, int64_t E, const float* x, int64_t ldx,
                                int64_t F_in, const float* h, int64_t ldh, const float* red, int64_t ldr,
                                const int32_t* packed, int64_t ldp, const float* g, int64_t ldg, int64_t Fp,
                                const void* plan_buf /* tfgx_pool_mlp_max_wgrad_plan of the same row_ptr, F_in, Fp */, float* dW,
                                int64_t lddw, float* db /* [Fp] or NULL */, void* workspace, size_t workspace_bytes,
                                tfgx_stream_t stream);

/* Chunk lists of the rows of a plan that are too long for one lane group ("hubs" of a power-law graph), as the host
   builds them once per plan (the same lists tfgx_reduce_args / tfgx_gat_args carry for the forward): rows with more
   than `threshold` positions; hub row rows[i] owns chunks [chunk_ptr[i], chunk_ptr[i+1]); chunk c covers CSR positions
   [chunk_begin[c], chunk_end[c]) of row chunk_row[c].  The *_hub_f32 backward entry points below take them (NULL: every
   row is walked by one lane group), run the hub rows chunk-wise into a caller-lent scratch and add a row's chunk
   partials in chunk order — deterministic, no atomics. */
typedef struct tfgx_hub_lists {
    int32_t threshold;
    int32_t reserved;
    int64_t n_rows;
    int64_t n_chunks;
    const int32_t* rows;
    const int32_t* chunk_ptr;
    const int32_t* chunk_begin;
    const int32_t* chunk_end;
    const int32_t* chunk_row;
} tfgx_hub_lists;

typedef struct tfgx_gat_backward_args {
    const int32_t* row_ptr;    /* forward plan (by destination) + col: used by the dst pass */
    const int32_t* col;
    int64_t n_dst;
    const int32_t* row_ptr_t;  /* transposed plan (by source) + destination per position: used by the src pass */
    const int32_t* dst_t;
    int64_t n_src;             /* may exceed n_dst (a shard's [own | halo] source table): the appended self-loop of
                                  destination r is source r, so only sources [0, n_dst) take part in one */
    const float* q; int64_t ldq;
    const float* k; int64_t ldk;
    const float* v; int64_t ldv;
    const float* grad_out; int64_t ld_grad_out;   /* dO [n_dst, H*dv] */
    const float* stats_ml;                        /* [n_dst, 2H] from the forward */
    const float* dsum;                            /* [n_dst, H] */
    int32_t H, d, dv, add_self_loop;
    float scale;
    int32_t reserved;
    float* grad_q; int64_t ld_grad_q;             /* [n_dst, H*d]  (dst pass) */
    float* grad_k; int64_t ld_grad_k;             /* [n_src, H*d]  (src pass) */
    float* grad_v; int64_t ld_grad_v;             /* [n_src, H*dv] (src pass) */
    /* attention dropout of the forward (same three values as tfgx_gat_args) + for the src pass the forward-CSR
       position of every transposed-plan position */
    float drop_rate;
    int32_t reserved2;
    uint64_t drop_seed;
    int64_t drop_self_base;
    const int32_t* edge_pos_t;                    /* [E] or NULL when drop_rate == 0 */
    /* row strides of stats_ml / dsum; 0 = dense (2H / H).  The src pass gathers, per edge, the destination's dO, Q,
       (m, l) and D rows: a caller that interleaves them into ONE row per destination (then grad_out / q / stats_ml /
       dsum point into that table with its row stride) turns four gathers into one contiguous burst. */
    int64_t ld_stats_ml;
    int64_t ld_dsum;
    /* optional walk orders (see tfgx_gat_args.row_order): destinations for the dst pass, sources for the src pass */
    const int32_t* row_order;
    const int32_t* row_order_t;
    const uint64_t* drop_seed_dev;                /* see tfgx_gat_args.drop_seed_dev (same location as the forward) */
    /* optional (round 5): ONE BLOCK of a source-blocked pass.  Row p of the pass being called (destination for the dst pass,
       source for the src pass) walks positions [span_begin[p * span_stride], span_end[p * span_stride]) of the col / dst_t
       array handed in — a plan whose rows are partitioned by the range of the OTHER endpoint (CsrPlan.source_blocks) —; with
       accumulate = 1 the gradients of the pass are ADDED to the stored ones (blocks are launched in order, so the sum is
       deterministic); add_self_loop is set by the caller on the last block only.  Every launch then gathers rows of one
       block, which the L2 of every XCD serves (see tfgx_gat_args.state_in_acc).  Fast-kernel head geometries, no attention
       dropout, no hub lists.  NULL span_begin = the plain pass over row_ptr / row_ptr_t. */
    const int32_t* span_begin;
    const int32_t* span_end;
    int64_t span_stride;
    int32_t accumulate;
    int32_t reserved3;
    /* optional (round 6, ABI 113), src pass: the per-head scalars of every DESTINATION in one contiguous block per head,
       head_pack[r * ld_head_pack + h * B + ...] = [ Q[r,h,0..d) | m | 1 / (l + 1e-8) | D | zero pad ], B = roundup4(d + 3)
       floats (tfgx_gat_pack_dst_heads_f32 writes it behind dO in the packed table).  With it the pass issues, per edge and
       lane, the dO load and B / 4 16-byte loads of that block — three line requests where q / stats_ml / dsum as separate
       pointers cost five (the L2s serve ~145 G requests/s: the pass went 3.9 -> 2.9 ms at Reddit shape).  q / stats_ml /
       dsum stay mandatory: the one-lane kernels of odd head geometries read them. */
    const float* head_pack;
    int64_t ld_head_pack;
} tfgx_gat_backward_args;

/* Prepares both backward passes in ONE sweep over the destination rows: dsum[r, h] = <dO[r, h, :], O[r, h, :]> (dense
   [n_dst, H]) and the packed table pack[r] = [ dO (H*dv) | Q (H*d) | (m, l) (2H) | D (H) ] with row stride ld_pack
   (>= H*dv + H*d + 3H; a multiple of 32 floats keeps rows on whole 128-byte lines) that the source pass gathers. */
int tfgx_gat_pack_dst_f32(const float* grad_out, int64_t ld_grad_out, const float* out, int64_t ldo, const float* q,
                          int64_t ldq, const float* stats_ml /* [n_dst, 2H] */, int64_t n_dst, int32_t H, int32_t d,
                          int32_t dv, float* pack, int64_t ld_pack, float* dsum /* [n_dst, H] */, tfgx_stream_t stream);
/* the same sweep, head-block form: pack[r] = [ dO (H*dv) | pad to 4 floats | per head: Q (d), m, 1 / (l + 1e-8), D, pad to
   roundup4(d + 3) ] with ld_pack >= roundup4(H*dv) + H*roundup4(d + 3); grad_out = pack, head_pack = pack + roundup4(H*dv), both
   with row stride ld_pack (a multiple of 4, pack 16-byte aligned), is what tfgx_gat_backward_args.head_pack expects. */
int tfgx_gat_pack_dst_heads_f32(const float* grad_out, int64_t ld_grad_out, const float* out, int64_t ldo, const float* q,
                                int64_t ldq, const float* stats_ml /* [n_dst, 2H] */, int64_t n_dst, int32_t H, int32_t d,
                                int32_t dv, float* pack, int64_t ld_pack, float* dsum /* [n_dst, H] */, tfgx_stream_t stream);
/* d == 1 (ABI 114): dQ out of the forward's sums (tfgx_gat_args.qgrad_t / qgrad_s) instead of the destination pass:
   grad_q[r, h] = (<grad_out[r, h, :], qgrad_t[r, h, :]> - dsum[r, h] * qgrad_s[r, h]) / scale, dsum = <dO, O> per head as
   tfgx_gat_pack_dst*_f32 writes it.  Replaces the dQ the reference's tape derives from gat.py:73-89. */
int tfgx_gat_query_grad_d1_f32(const float* grad_out, int64_t ld_grad_out, const float* qgrad_t, int64_t ld_t,
                               const float* qgrad_s /* [n_dst, H] */, const float* dsum /* [n_dst, H] */, int64_t n_dst,
                               int32_t H, int32_t dv, float scale, float* grad_q, int64_t ld_grad_q, tfgx_stream_t stream);
int tfgx_gat_backward_dst_f32(const tfgx_gat_backward_args* args /* host */, tfgx_stream_t stream);
int tfgx_gat_backward_src_f32(const tfgx_gat_backward_args* args /* host */, tfgx_stream_t stream);
/* the same passes on a graph with hub rows: `hub` = chunk lists of the forward plan (dst pass) / of the transposed plan
   (src pass); hub_scratch: n_chunks * H*d floats (dst) / n_chunks * (H*d + H*dv) floats (src).  NULL lists = plain pass. */
/* tfgx_edge_softmax_f32 with hub rows handled chunk-wise (per-chunk statistics, ordered fold per row, per-chunk
   normalisation); hub_scratch: n_chunks * 2 * Hp floats, Hp = H rounded up to a power of two */
int tfgx_edge_softmax_hub_f32(const int32_t* row_ptr, const int32_t* perm, const float* score, int64_t H, int64_t n_dst,
                              float* out, const tfgx_hub_lists* hub, float* hub_scratch, tfgx_stream_t stream);
/* tfgx_sddmm_f32 with hub rows walked chunk-wise (every edge owns its output element: no scratch) */
int tfgx_sddmm_hub_f32(const int32_t* row_ptr, const int32_t* col, int64_t n_dst, const float* a, int64_t lda,
                       const float* b, int64_t ldb, int64_t F, float* out, const tfgx_hub_lists* hub, tfgx_stream_t stream);
int tfgx_gat_backward_dst_hub_f32(const tfgx_gat_backward_args* args, const tfgx_hub_lists* hub, float* hub_scratch,
                                  tfgx_stream_t stream);
int tfgx_gat_backward_src_hub_f32(const tfgx_gat_backward_args* args, const tfgx_hub_lists* hub_t, float* hub_scratch,
                                  tfgx_stream_t stream);
/* hub-aware forms of tfgx_segment_max_count_f32 / tfgx_segment_max_backward_f32 (hub lists of the forward plan / of the
   transposed plan; hub_scratch: n_chunks * F floats) */
int tfgx_segment_max_count_hub_f32(const int32_t* row_ptr, const int32_t* col, const float* w /* or NULL */, int64_t n_dst,
                                   const float* x, int64_t ldx, int64_t F, const float* out, int64_t ldo, float* count,
                                   int64_t ldc, const tfgx_hub_lists* hub, float* hub_scratch, tfgx_stream_t stream);
int tfgx_segment_max_backward_hub_f32(const int32_t* row_ptr_t, const int32_t* dst_t, const float* w_t /* or NULL */,
                                      int64_t n_src, const float* x, int64_t ldx, int64_t F, const float* out,
                                      int64_t ldo, const float* g, int64_t ldg, const float* count, int64_t ldc, float* gx,
                                      int64_t ldgx, int64_t n_dst, float* gn_scratch, const tfgx_hub_lists* hub_t,
                                      float* hub_scratch, tfgx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Dense GEMM beside the path: C = act(A[M,K] @ B[K,N] + bias) with fp32-input MFMA
 * (x @ kernel: nn/conv/gcn.py:272, nn/conv/gat.py:52,61,70, nn/conv/graph_sage.py:43-44).
 * Bitwise a k-ordered fp32 FMA chain per output element.
 * ------------------------------------------------------------------------------------------- */
int tfgx_gemm_bias_act_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                           int32_t act, float* C, int64_t ldc, int64_t M, int64_t K, int64_t N,
                           tfgx_stream_t stream);

/* same, with the activation applied to columns [0, act_cols) only: one pass over x for several projections that
   share the input (GAT's Q | K | V = x @ [Wq | Wk | W], relu on Q and K, none on V — nn/conv/gat.py:52-70) */
int tfgx_gemm_bias_act_cols_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                                int32_t act, int64_t act_cols, float* C, int64_t ldc, int64_t M, int64_t K, int64_t N,
                                tfgx_stream_t stream);

/* same, with a caller-lent workspace of tfgx_gemm_workspace_bytes(M, K, N) bytes: small-M / long-K products (Cora:
   2708 x 1433 x 16) are cut along K over the idle CUs, the partial products are summed in split order (deterministic)
   together with bias and activation; tall products on the row-streaming kernel (M >= 2^18) keep their tile counters
   there (~2 KB, zeroed in stream order by the call): the kernel's waves then CLAIM their 32-row tiles instead of walking
   a fixed map — same values in every element (a tile's arithmetic does not depend on which wave runs it), 1-4 % less
   time.  The workspace must belong to this call until it has completed on `stream`.  workspace == NULL behaves like
   tfgx_gemm_bias_act_cols_f32. */
size_t tfgx_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N);
int tfgx_gemm_bias_act_cols_ws_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                                   int32_t act, int64_t act_cols, float* C, int64_t ldc, int64_t M, int64_t K, int64_t N,
                                   void* workspace, size_t workspace_bytes, tfgx_stream_t stream);

/* Weight gradient of a dense layer (the backward of tfgx_gemm_bias_act_f32):
   dW[Ka, N] = X[M, Ka]^T @ G[M, N] and, when db != NULL, db[N] = column sums of G (the bias gradient), reduced over
   the M rows on the fp32 matrix cores (operands streamed from global memory in the MFMA's own layout, no LDS).
   Per-workgroup partials live in the caller's workspace (tfgx_gemm_tn_workspace_bytes) and are summed in a fixed
   order: deterministic, no atomics. */
size_t tfgx_gemm_tn_workspace_bytes(int64_t M, int64_t Ka, int64_t N, int32_t want_bias);
int tfgx_gemm_tn_f32(const float* X, int64_t ldx, const float* G, int64_t ldg, int64_t M, int64_t Ka, int64_t N,
                     float* dW, int64_t ldw, float* db /* or NULL */, void* workspace, size_t workspace_bytes,
                     tfgx_stream_t stream);
/* same with G gated by a ReLU output: G[m, n] counts only where gate[m, n] > 0 (gate = the output of the layer whose
   epilogue applied the ReLU) — the masked gradient is consumed in registers and never written out. */
int tfgx_gemm_tn_gated_f32(const float* X, int64_t ldx, const float* G, int64_t ldg, const float* gate, int64_t ld_gate,
                           int64_t M, int64_t Ka, int64_t N, float* dW, int64_t ldw, float* db /* or NULL */,
                           void* workspace, size_t workspace_bytes, tfgx_stream_t stream);

/* out[c, r] = in[r, c] (a layer's [K, N] kernel transposed, so that d/dx = G @ kernel^T runs on the forward GEMM). */
int tfgx_transpose_f32(const float* in, int64_t ldi, int64_t rows, int64_t cols, float* out, int64_t ldo,
                       tfgx_stream_t stream);

/* Neighbour sampling on the CSR plan (RandomNeighborSampler.sample, tf_geometric/utils/graph_utils.py:667-772, a
   pure-Python per-node loop in the reference).  Row r receives out_ptr[r+1]-out_ptr[r] = m neighbours out of its d:
   m >= d: all of them, in order; m < d: m distinct ones, uniformly (Floyd); m > d with replace_when_short: m draws
   with replacement ("padding").  Counter-based generator keyed by (seed, row, draw): reproducible for a seed, but NOT
   numpy's stream — parity with the reference is distributional, not bitwise.  Rows with more than 256 draws out of a
   longer neighbour list use selection sampling (one ordered pass, no scratch); max_per_row is informative. */
int tfgx_sample_neighbors(const int32_t* row_ptr, const int32_t* col, const float* w /* or NULL */, int64_t n_dst,
                          const int32_t* out_ptr, int32_t max_per_row, int32_t replace_when_short, uint64_t seed,
                          int32_t* out_col, float* out_w /* or NULL */, tfgx_stream_t stream);

/* Segmented top-k (topk_pool, tf_geometric/nn/pool/topk_pool.py:6-87): for every segment id s in [0, num_segments)
   keep the node_k(s) highest-scored of its count(s) items, node_k = min(k, count) when k >= 0, otherwise
   min(count, ceil(float32(count) * ratio)).  out_index[0 .. *out_count) receives the kept items' positions in the
   caller's arrays, ordered as the reference returns them: segments ascending, scores descending, equal scores in the
   caller's order (-0.0 == +0.0).  out_index must hold n entries; out_count is one device int32.  Segment ids outside
   [0, num_segments) -> TFGX_ERR_INDEX.  Synchronises the stream once (id validation). */
size_t tfgx_segment_topk_workspace_bytes(int64_t n, int64_t num_segments);
int tfgx_segment_topk(const int32_t* segment, const float* score, int64_t n, int64_t num_segments, int32_t k,
                      float ratio, int32_t* out_index, int32_t* out_count, void* workspace, size_t workspace_bytes,
                      tfgx_stream_t stream);

/* x[n, F] -> x_main[n, f_main] + x_tail[n, F - f_main] in one pass (the split source layout of tfgx_reduce_args) */
int tfgx_split_rows_f32(const float* x, int64_t ldx, int64_t n, int64_t F, int64_t f_main, float* x_main,
                        int64_t ld_main, float* x_tail, int64_t ld_tail, tfgx_stream_t stream);

/* Is the split layout still a copy of x?  Compares `samples` rows of x bit for bit with x_main / x_tail — rows 0 and n - 1
   always, the rest drawn from (seed, i); samples >= n compares every row — and leaves 1 in *mismatch (device int32; zeroed
   first) when any bit differs.  The host runs this before it serves a layout it built on its OWN initiative (automatic
   promotion of a tensor seen twice): a write that bypassed torch's version counter then demotes the layout instead of being
   aggregated from a stale copy.  (The reference never caches feature values: nn/conv/gcn.py:125-128 caches the adjacency.) */
int tfgx_split_rows_verify_f32(const float* x, int64_t ldx, int64_t n, int64_t F, int64_t f_main, const float* x_main,
                               int64_t ld_main, const float* x_tail, int64_t ld_tail, int64_t samples, uint64_t seed,
                               int32_t* mismatch /* device */, tfgx_stream_t stream);

/* h = h * rsqrt(max(sum(h^2), 1e-12)) per row, in place (tf.nn.l2_normalize, graph_sage.py:58) */
int tfgx_l2_normalize_rows_f32(float* h, int64_t ld, int64_t n, int64_t F, tfgx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Destination-range sharding + halo exchange support (no counterpart in the reference).
 *   tfgx_gather_rows_f32   : pack rows x[idx[i], :] -> out[i, :]   (send side of the halo all-to-all-v); idx == NULL:
 *                            rows 0 .. M-1 (a strided row copy, e.g. own rows into the [own | halo] table)
 *   tfgx_halo_mark         : flags[c] = 1 for every source c of the slice outside [own_lo, own_hi)
 *   tfgx_halo_compact      : halo_ids = sorted ids with flags set; pos[c] = its rank; *n_halo (device int32)
 *   tfgx_halo_remap_cols   : col -> local source-table index: own rows first, then halo rows
 *   tfgx_split_by_source_class : stable per-row partition of the edges by source class (own rows | halo round 0 |
 *                            halo round 1 | ...), so each class is one strided view of a single row_ptr_k array
 * ------------------------------------------------------------------------------------------- */
int tfgx_gather_rows_f32(const float* x, int64_t ldx, const int32_t* idx, int64_t M, int64_t F,
                         float* out, int64_t ldo, tfgx_stream_t stream);
/* gout[m, n] = out[m, n] > 0 ? g[m, n] : 0: backward of the ReLU fused into tfgx_gemm_bias_act_f32 /
   tfgx_segment_reduce_f32 epilogues (tf.nn.relu under a GradientTape, demo/demo_gcn.py:68-77).  gout may alias g. */
int tfgx_relu_backward_f32(const float* g, int64_t ldg, const float* out, int64_t ldo, int64_t M, int64_t N,
                           float* gout, int64_t ldgo, tfgx_stream_t stream);
/* out[c] = sum_m g[m, c] (g [M, N], row stride ldg): the bias gradient where the bias rode in an aggregation epilogue
   (tf.GradientTape's reduce_sum over the node axis, demo/demo_gcn.py:68-77).  Deterministic (fixed two-phase order).
   workspace: tfgx_column_sum_workspace_bytes(M, N) bytes of device scratch. */
size_t tfgx_column_sum_workspace_bytes(int64_t M, int64_t N);
int tfgx_column_sum_f32(const float* g, int64_t ldg, int64_t M, int64_t N, float* out, void* workspace,
                        size_t workspace_bytes, tfgx_stream_t stream);
/* dst[idx[i], :] += src[i, :] for i in [0, M): owner-side accumulate of the reverse halo exchange (gradients of halo rows
   returning to their owners during training).  idx must hold UNIQUE ids within one call (one peer's request list does);
   peers are applied by the caller in a fixed order, so the sum is deterministic without atomics.  idx == NULL: the
   identity list (dst[i, :] += src[i, :]: a peer that requested a contiguous block of rows). */
int tfgx_scatter_add_rows_f32(float* dst, int64_t ldd, const int32_t* idx, int64_t M, int64_t F, const float* src,
                              int64_t lds, tfgx_stream_t stream);
/* n_class source classes (own rows + one class per halo exchange ROUND, so the halo pass of round j can run while
   round j+1 is still on the wire): class of source c = first k with
   c < class_bounds[k] (device int32 [n_class-1] used); row r's class-k edges = [rpk[r*n_class+k], rpk[r*n_class+k+1]);
   row_ptr_k has n_dst*n_class + 1 entries; n_class <= 17 */
int tfgx_split_by_source_class(const int32_t* row_ptr, const int32_t* col_local, const float* w /* or NULL */,
                               int64_t n_dst, int64_t E, const int32_t* class_bounds, int32_t n_class,
                               int32_t* row_ptr_k, int32_t* col_out, float* w_out /* or NULL */, tfgx_stream_t stream);
size_t tfgx_halo_workspace_bytes(int64_t n_global);
int tfgx_halo_mark(const int32_t* col, int64_t E, int32_t own_lo, int32_t own_hi, int64_t n_global,
                   int32_t* flags /* [n_global], zeroed here */, tfgx_stream_t stream);
int tfgx_halo_compact(const int32_t* flags, int64_t n_global, int32_t* pos /* [n_global] */,
                      int32_t* halo_ids /* [n_global] capacity */, int32_t* n_halo /* device [1] */,
                      void* workspace, size_t workspace_bytes, tfgx_stream_t stream);
int tfgx_halo_remap_cols(const int32_t* col, int64_t E, int32_t own_lo, int32_t own_hi, const int32_t* pos,
                         int32_t n_own, int32_t* col_local, tfgx_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif /* TFGX_H */
