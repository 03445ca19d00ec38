/*
 * C restatement of the tf_geometric message-passing hot path (CPU, fp32, op-for-op).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ (as a second, independent checker next to
 * oracle/tfg_oracle.py) and by bench.py's `cpu_baseline` leg.  Never linked into or
 * called from the product library (tf_geometric_amd/lib/libtfgx.so).
 *
 * PARITY UNPINNED: the reference ships no golden vectors for this path and its
 * arithmetic lives in un-vendored tensorflow / tf_sparse (see oracle/tfg_oracle.py
 * header and DESIGN.md); this file restates the published op semantics at the
 * reference's call sites:
 *
 *   tfgo_aggregate_coo_f32   = tf.gather(x, col) -> (* edge_weight[:,None]) ->
 *                              tf.math.unsorted_segment_{sum,mean,max}(msg, row, N)
 *                              reference: tf_geometric/nn/kernel/map_reduce.py:60-70,
 *                                         :15-16 (sum) :27-28 (mean) :31-42 (max),
 *                                         tf_geometric/nn/conv/gcn.py:221-222 (gcn_mapper)
 *   tfgo_segment_softmax_f32 = tf_geometric/nn/kernel/segment.py:26-33
 *
 * The scatter loop follows TF-CPU's UnsortedSegmentFunctor: one pass over the E
 * messages in edge order, out[row[e]] (+)= msg[e]; an empty segment keeps the
 * initial value (0 for sum/mean, float lowest for max).  With threads > 1 the
 * destination range is split across threads (each thread scans the whole edge
 * list and keeps only its own rows): same per-row order, no atomics.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { TFGO_SUM = 0, TFGO_MEAN = 1, TFGO_MAX = 2 };

/* returns 0 on success, 1 if an index is out of range (TF-CPU: InvalidArgumentError) */
int tfgo_aggregate_coo_f32(const float* x, int64_t ldx, const int32_t* row, const int32_t* col,
                           const float* w /* may be NULL */, int64_t E, int64_t n_dst, int64_t n_src,
                           int64_t F, int op, float* out, int64_t ldo, int threads)
{
    for (int64_t e = 0; e < E; ++e)
        if (row[e] < 0 || row[e] >= n_dst || col[e] < 0 || col[e] >= n_src) return 1;
    if (threads < 1) threads = 1;
    const float init = (op == TFGO_MAX) ? -FLT_MAX : 0.0f;
    int32_t* cnt = NULL;
    if (op == TFGO_MEAN) cnt = (int32_t*)calloc((size_t)(n_dst > 0 ? n_dst : 1), sizeof(int32_t));

#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
#else
        const int t = 0, T = 1;
#endif
        const int64_t lo = n_dst * t / T, hi = n_dst * (t + 1) / T;
        for (int64_t r = lo; r < hi; ++r)
            for (int64_t f = 0; f < F; ++f) out[r * ldo + f] = init;
        for (int64_t e = 0; e < E; ++e) {
            const int64_t r = row[e];
            if (r < lo || r >= hi) continue;
            const float* xs = x + (int64_t)col[e] * ldx;   /* tf.gather(x, col)[e] */
            float* o = out + r * ldo;
            if (op == TFGO_MAX) {
                if (w) { const float we = w[e]; for (int64_t f = 0; f < F; ++f) { float m = xs[f] * we; o[f] = m > o[f] ? m : o[f]; } }
                else   { for (int64_t f = 0; f < F; ++f) o[f] = xs[f] > o[f] ? xs[f] : o[f]; }
            } else {
                if (w) { const float we = w[e]; for (int64_t f = 0; f < F; ++f) o[f] += xs[f] * we; }
                else   { for (int64_t f = 0; f < F; ++f) o[f] += xs[f]; }
                if (cnt) cnt[r]++;
            }
        }
        if (op == TFGO_MEAN)
            for (int64_t r = lo; r < hi; ++r) {
                const float c = (float)(cnt[r] > 1 ? cnt[r] : 1);
                for (int64_t f = 0; f < F; ++f) out[r * ldo + f] /= c;
            }
    }
    free(cnt);
    return 0;
}

/* segment.py:26-33: score[E,H] grouped by seg[e]; out same layout */
int tfgo_segment_softmax_f32(const float* score, const int32_t* seg, int64_t E, int64_t H,
                             int64_t n_seg, float* out)
{
    float* mx = (float*)malloc(sizeof(float) * (size_t)(n_seg * H + 1));
    float* dn = (float*)malloc(sizeof(float) * (size_t)(n_seg * H + 1));
    for (int64_t i = 0; i < n_seg * H; ++i) { mx[i] = -FLT_MAX; dn[i] = 0.0f; }
    for (int64_t e = 0; e < E; ++e) {
        if (seg[e] < 0 || seg[e] >= n_seg) { free(mx); free(dn); return 1; }
        for (int64_t h = 0; h < H; ++h) {
            float s = score[e * H + h];
            float* m = mx + (int64_t)seg[e] * H + h;
            if (s > *m) *m = s;
        }
    }
    for (int64_t e = 0; e < E; ++e)
        for (int64_t h = 0; h < H; ++h) {
            float ex = expf(score[e * H + h] - mx[(int64_t)seg[e] * H + h]);
            out[e * H + h] = ex;
            dn[(int64_t)seg[e] * H + h] += ex;
        }
    for (int64_t e = 0; e < E; ++e)
        for (int64_t h = 0; h < H; ++h)
            out[e * H + h] = out[e * H + h] / (dn[(int64_t)seg[e] * H + h] + 1e-8f);
    free(mx); free(dn);
    return 0;
}

/* x[M,K] @ w[K,N] (+bias) — the dense GEMM beside the path (nn/conv/gcn.py:272), fp32 k-ordered */
void tfgo_gemm_f32(const float* a, const float* b, const float* bias, int64_t M, int64_t K, int64_t N,
                   int relu, float* c, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        float* ci = c + i * N;
        for (int64_t j = 0; j < N; ++j) ci[j] = 0.0f;
        for (int64_t k = 0; k < K; ++k) {
            const float aik = a[i * K + k];
            const float* bk = b + k * N;
            for (int64_t j = 0; j < N; ++j) ci[j] += aik * bk[j];
        }
        for (int64_t j = 0; j < N; ++j) {
            float v = ci[j] + (bias ? bias[j] : 0.0f);
            ci[j] = (relu && v < 0.0f) ? 0.0f : v;
        }
    }
}

/*
 * Strong multi-core CPU baseline for bench.py: the same arithmetic (gather, * w, segment-sum/mean/max in edge
 * order per destination) over a row-sorted (CSR-by-destination) edge list, destination ranges balanced by edge
 * count across threads.  The sort is done once by the caller and not timed — exactly like the GPU plan.
 */
int tfgo_aggregate_csr_f32(const float* x, int64_t ldx, const int32_t* row_ptr, const int32_t* col,
                           const float* w /* may be NULL */, int64_t n_dst, int64_t F, int op, float* out,
                           int64_t ldo, int threads)
{
    if (threads < 1) threads = 1;
    const int64_t E = row_ptr[n_dst];
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
#else
        const int t = 0, T = 1;
#endif
        /* first row whose start offset >= t*E/T (binary search) */
        int64_t bounds[2];
        for (int k = 0; k < 2; ++k) {
            const int64_t target = E * (t + k) / T;
            int64_t lo = 0, hi = n_dst;
            if (t + k == T) lo = n_dst;
            else while (lo < hi) { int64_t mid = (lo + hi) / 2; if (row_ptr[mid] < target) lo = mid + 1; else hi = mid; }
            bounds[k] = (t + k == 0) ? 0 : lo;
        }
        for (int64_t r = bounds[0]; r < bounds[1]; ++r) {
            float* o = out + r * ldo;
            const float init = (op == TFGO_MAX) ? -FLT_MAX : 0.0f;
            for (int64_t f = 0; f < F; ++f) o[f] = init;
            for (int64_t i = row_ptr[r]; i < row_ptr[r + 1]; ++i) {
                const float* xs = x + (int64_t)col[i] * ldx;
                const float we = w ? w[i] : 1.0f;
                if (op == TFGO_MAX) { for (int64_t f = 0; f < F; ++f) { float m = w ? xs[f] * we : xs[f]; o[f] = m > o[f] ? m : o[f]; } }
                else if (w) { for (int64_t f = 0; f < F; ++f) o[f] += xs[f] * we; }
                else { for (int64_t f = 0; f < F; ++f) o[f] += xs[f]; }
            }
            if (op == TFGO_MEAN) {
                const int64_t c = row_ptr[r + 1] - row_ptr[r];
                const float d = (float)(c > 1 ? c : 1);
                for (int64_t f = 0; f < F; ++f) o[f] /= d;
            }
        }
    }
    return 0;
}

/* first-touch copy: pages of dst are faulted in by the thread that later owns the same index range, so a large
   array ends up spread over the NUMA nodes of the host instead of sitting on the allocating thread's node
   (bench.py's cpu_baseline: random row gathers then draw on every memory controller) */
void tfgo_parallel_copy_f32(float* dst, const float* src, int64_t n, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < n; ++i) dst[i] = src[i];
}
