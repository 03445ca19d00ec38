// GCN normalisation on the CSR-by-destination plan.
// Reference: tf_geometric/nn/conv/gcn.py:32-130 (gcn_norm_adj) — SparseMatrix.add_diag / segment_sum /
// tf.pow(deg, -0.5|-1) with inf/nan -> 0 / diags(D) @ A @ diags(D).  Runs once per graph (cached).
// One group of 8 lanes per destination row; deterministic (fixed shuffle tree, no atomics).  Rows longer than kLongRow edges —
// the hubs of a power-law graph: 329 k in-edges on the products-sized R-MAT graph, where eight lanes walking one row made
// this once-per-graph kernel a 65 ms latency chain — are left to the WHOLE workgroup after its short rows (round 5).
#include "tfgx_common.h"

namespace tfgx {
namespace {

constexpr int NG = 8;  // lanes per row

__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int o = NG / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, NG);
    return v;
}

constexpr int kLongRow = 2048;             // edges: longer rows are walked by the whole workgroup
constexpr int kRowsPerBlock = kBlock / NG;

__global__ __launch_bounds__(kBlock) void weight_sum_kernel(const int32_t* __restrict__ row_ptr,
                                                            const float* __restrict__ w, int64_t n, float diag,
                                                            float* __restrict__ deg)
{
    __shared__ int long_rows[kRowsPerBlock];
    __shared__ int n_long;
    __shared__ float part[kBlock / 64];
    const int lane = threadIdx.x % NG, grp = threadIdx.x / NG;
    for (int64_t base = int64_t(blockIdx.x) * kRowsPerBlock; base < n; base += int64_t(gridDim.x) * kRowsPerBlock) {
        if (threadIdx.x == 0) n_long = 0;
        __syncthreads();
        const int64_t r = base + grp;
        if (r < n) {
            const int s = row_ptr[r], e = row_ptr[r + 1];
            if (w && e - s > kLongRow) {
                if (lane == 0) long_rows[atomicAdd(&n_long, 1)] = grp;
            } else {
                float acc = 0.0f;
                if (w) {
                    for (int i = s + lane; i < e; i += NG) acc += w[i];
                    acc = group_sum(acc);
                } else {
                    acc = float(e - s);
                }
                if (lane == 0) deg[r] = acc + diag;
            }
        }
        __syncthreads();
        const int nl = n_long;
        for (int k = 0; k < nl; ++k) {
            const int slot = long_rows[k];      // (arrival order: a row's sum does not depend on when it is computed)
            const int64_t rl = base + slot;
            const int s = row_ptr[rl], e = row_ptr[rl + 1];
            float acc = 0.0f;
            for (int i = s + int(threadIdx.x); i < e; i += kBlock) acc += w[i];       // fixed per-thread order
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);           // fixed tree inside the wave
            if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
            __syncthreads();
            if (threadIdx.x == 0) {
                float t = 0.0f;
                for (int q = 0; q < kBlock / 64; ++q) t += part[q];                   // waves in order
                deg[rl] = t + diag;
            }
            __syncthreads();
        }
    }
}

// tf.pow(deg, p) then _remove_inf_and_nan (gcn.py:23-29, :81-82, :103-105)
__device__ __forceinline__ float inv_pow(float d, bool half)
{
    const float v = half ? (1.0f / sqrtf(d)) : (1.0f / d);
    return (isinf(v) || isnan(v)) ? 0.0f : v;
}

__global__ __launch_bounds__(kBlock) void gcn_norm_kernel(const int32_t* __restrict__ row_ptr,
                                                          const int32_t* __restrict__ col,
                                                          const float* __restrict__ w, int64_t n,
                                                          const float* __restrict__ row_deg,
                                                          const float* __restrict__ col_deg, int mode, float fill,
                                                          int add_self_loop, int renorm,
                                                          float* __restrict__ w_out, float* __restrict__ self_coef)
{
    __shared__ int long_rows[kRowsPerBlock];
    __shared__ int n_long;
    const int lane = threadIdx.x % NG, grp = threadIdx.x / NG;
    const float* cdeg = col_deg ? col_deg : row_deg;
    // every edge's value is independent: `first` / `step` only say which lanes walk the row (its 8-lane group, or — a long
    // row — the whole workgroup); the self-loop coefficient is written by the caller's lane 0
    auto edges = [&](int64_t r, int first, int step) {
        const int s = row_ptr[r], e = row_ptr[r + 1];
        if (mode == TFGX_NORM_BOTH) {
            const float dr = inv_pow(row_deg[r], true);
            int i = s + first;
            // (four independent col -> degree chains per lane: a long row — the hubs of a power-law graph cluster in a few
            // workgroups — is bound by the latency of that dependent gather)
            for (; i + 3 * step < e; i += 4 * step) {
                int c4[4];
                float d4[4], w4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { c4[u] = col[i + u * step]; w4[u] = w ? w[i + u * step] : 1.0f; }
#pragma unroll
                for (int u = 0; u < 4; ++u) d4[u] = cdeg[c4[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) w_out[i + u * step] = dr * w4[u] * inv_pow(d4[u], true);
            }
            for (; i < e; i += step) {
                const float wi = w ? w[i] : 1.0f;
                w_out[i] = dr * wi * inv_pow(cdeg[col[i]], true);   // (D^-1/2 A) D^-1/2, left product first (:94)
            }
        } else if (mode == TFGX_NORM_LEFT) {
            const float dr = inv_pow(row_deg[r], false);
            for (int i = s + first; i < e; i += step) w_out[i] = dr * (w ? w[i] : 1.0f);           // :109
        } else {  // RIGHT: row degrees applied on the column side (:113, :119)
            for (int i = s + first; i < e; i += step) w_out[i] = (w ? w[i] : 1.0f) * inv_pow(row_deg[col[i]], false);
        }
    };
    for (int64_t base = int64_t(blockIdx.x) * kRowsPerBlock; base < n; base += int64_t(gridDim.x) * kRowsPerBlock) {
        if (threadIdx.x == 0) n_long = 0;
        __syncthreads();
        const int64_t r = base + grp;
        if (r < n) {
            const int len = row_ptr[r + 1] - row_ptr[r];
            if (len > kLongRow) {
                if (lane == 0) long_rows[atomicAdd(&n_long, 1)] = grp;
            } else {
                edges(r, lane, NG);
            }
            if (lane == 0) {
                float sc = 0.0f;
                if (mode == TFGX_NORM_BOTH) {
                    const float dr = inv_pow(row_deg[r], true);
                    if (add_self_loop) sc = renorm ? (dr * fill * inv_pow(cdeg[r], true)) : fill;  // :77 / :98
                } else if (mode == TFGX_NORM_LEFT) {
                    sc = add_self_loop ? inv_pow(row_deg[r], false) * fill : 0.0f;
                } else {
                    sc = add_self_loop ? fill * inv_pow(row_deg[r], false) : 0.0f;
                }
                self_coef[r] = sc;
            }
        }
        __syncthreads();
        const int nl = n_long;
        for (int k = 0; k < nl; ++k) edges(base + long_rows[k], int(threadIdx.x), kBlock);     // (which thread writes an edge's value does not change it)
        __syncthreads();
    }
}

}  // namespace
}  // namespace tfgx

using namespace tfgx;

extern "C" int tfgx_segment_weight_sum_f32(const int32_t* row_ptr, const float* w, int64_t n, float diag,
                                           float* deg, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n >= 0, "negative n");
    if (n == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr && deg, "null pointer");
    weight_sum_kernel<<<grid_for(n, kRowsPerBlock), kBlock, 0, as_stream(stream)>>>(row_ptr, w, n, diag, deg);
    TFGX_LAUNCH_CHECK("weight_sum_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_gcn_norm_edges_f32(const int32_t* row_ptr, const int32_t* col, const float* w, int64_t n,
                                       const float* row_deg, const float* col_deg, int32_t norm_mode, float fill,
                                       int32_t add_self_loop, int32_t renorm, float* w_out, float* self_coef,
                                       tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n >= 0, "negative n");
    TFGX_REQUIRE(norm_mode >= TFGX_NORM_BOTH && norm_mode <= TFGX_NORM_RIGHT, "bad norm mode");
    if (n == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr && row_deg && self_coef, "null pointer");  // col / w_out may be null when E == 0
    gcn_norm_kernel<<<grid_for(n, kRowsPerBlock), kBlock, 0, as_stream(stream)>>>(
        row_ptr, col, w, n, row_deg, col_deg, norm_mode, fill, add_self_loop, renorm, w_out, self_coef);
    TFGX_LAUNCH_CHECK("gcn_norm_kernel");
    return TFGX_OK;
}
