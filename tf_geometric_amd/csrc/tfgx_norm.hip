// GCN normalisation on the CSR-by-destination plan.
// Reference: tf_geometric/nn/conv/gcn.py:32-130 (gcn_norm_adj) — SparseMatrix.add_diag / segment_sum /
// tf.pow(deg, -0.5|-1) with inf/nan -> 0 / diags(D) @ A @ diags(D).  Runs once per graph (cached).
// One group of 8 lanes per destination row; deterministic (fixed shuffle tree, no atomics).
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

__global__ __launch_bounds__(kBlock) void weight_sum_kernel(const int32_t* __restrict__ row_ptr,
                                                            const float* __restrict__ w, int64_t n, float diag,
                                                            float* __restrict__ deg)
{
    const int lane = threadIdx.x % NG;
    int64_t r = (blockIdx.x * int64_t(kBlock) + threadIdx.x) / NG;
    const int64_t stride = int64_t(gridDim.x) * kBlock / NG;
    for (; r < n; r += stride) {
        const int s = row_ptr[r], e = row_ptr[r + 1];
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
    const int lane = threadIdx.x % NG;
    int64_t r = (blockIdx.x * int64_t(kBlock) + threadIdx.x) / NG;
    const int64_t stride = int64_t(gridDim.x) * kBlock / NG;
    const float* cdeg = col_deg ? col_deg : row_deg;
    for (; r < n; r += stride) {
        const int s = row_ptr[r], e = row_ptr[r + 1];
        if (mode == TFGX_NORM_BOTH) {
            const float dr = inv_pow(row_deg[r], true);
            for (int i = s + lane; i < e; i += NG) {
                const float wi = w ? w[i] : 1.0f;
                w_out[i] = dr * wi * inv_pow(cdeg[col[i]], true);   // (D^-1/2 A) D^-1/2, left product first (:94)
            }
            if (lane == 0) {
                float sc = 0.0f;
                if (add_self_loop) sc = renorm ? (dr * fill * inv_pow(cdeg[r], true)) : fill;  // :77 / :98
                self_coef[r] = sc;
            }
        } else if (mode == TFGX_NORM_LEFT) {
            const float dr = inv_pow(row_deg[r], false);
            for (int i = s + lane; i < e; i += NG) w_out[i] = dr * (w ? w[i] : 1.0f);           // :109
            if (lane == 0) self_coef[r] = add_self_loop ? dr * fill : 0.0f;
        } else {  // RIGHT: row degrees applied on the column side (:113, :119)
            for (int i = s + lane; i < e; i += NG) w_out[i] = (w ? w[i] : 1.0f) * inv_pow(row_deg[col[i]], false);
            if (lane == 0) self_coef[r] = add_self_loop ? fill * inv_pow(row_deg[r], false) : 0.0f;
        }
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
    weight_sum_kernel<<<grid_for(n * NG, kBlock), kBlock, 0, as_stream(stream)>>>(row_ptr, w, n, diag, deg);
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
    gcn_norm_kernel<<<grid_for(n * NG, kBlock), kBlock, 0, as_stream(stream)>>>(
        row_ptr, col, w, n, row_deg, col_deg, norm_mode, fill, add_self_loop, renorm, w_out, self_coef);
    TFGX_LAUNCH_CHECK("gcn_norm_kernel");
    return TFGX_OK;
}
