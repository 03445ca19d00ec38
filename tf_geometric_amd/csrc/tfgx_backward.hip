// Backward pass of the hot path — the gradients TensorFlow's autodiff produces for
// tf_geometric's training loops (demo/demo_gcn.py:68-77, tf.GradientTape).
//
//   d/dx of gather-scale-segment-sum/mean  = the SAME forward kernel on the transposed (CSR-by-source) plan
//                                            (tfgx_segment_reduce_f32; no new kernel)
//   d/dw (edge weights)                    = tfgx_sddmm_f32           out[i] = <a[row(i)], b[col[i]]>
//   unsorted_segment_max gradient          = tfgx_segment_max_count_f32 + tfgx_segment_max_backward_f32
//                                            (TF semantics: the gradient is split evenly among tied maxima)
//   GAT attention gradient                 = tfgx_gat_backward_dst_f32 (dQ) + tfgx_gat_backward_src_f32 (dK, dV),
//                                            flash-attention style: alpha is recomputed from the saved (m, l)
// Every accumulation has one owner (a destination row or a source row): deterministic, no atomics.
// These kernels favour clarity over speed (one lane per (row, head) for GAT); the forward path is the tuned one.
#include "tfgx_common.h"
#include <cfloat>

namespace tfgx {
namespace {

// ---------------------------------------------------------------- SDDMM: out[i] = <a[r], b[col[i]]>
template <int G>
__global__ __launch_bounds__(kBlock) void sddmm_kernel(const int32_t* __restrict__ row_ptr,
                                                       const int32_t* __restrict__ col, int64_t n_dst,
                                                       const float* __restrict__ a, int64_t lda,
                                                       const float* __restrict__ b, int64_t ldb, int F,
                                                       float* __restrict__ out)
{
    constexpr int ROWS_PER_BLOCK = kBlock / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    for (int64_t r = int64_t(blockIdx.x) * ROWS_PER_BLOCK + grp; r < n_dst; r += int64_t(gridDim.x) * ROWS_PER_BLOCK) {
        const int s = row_ptr[r], e = row_ptr[r + 1];
        const float* ar = a + r * lda;
        for (int i = s; i < e; ++i) {
            const float* br = b + int64_t(col[i]) * ldb;
            float acc = 0.0f;
            for (int j = lane; j < F; j += G) acc = fmaf(ar[j], br[j], acc);
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
            if (lane == 0) out[i] = acc;
        }
    }
}

// ---------------------------------------------------------------- segment-max gradient
// count[r, j] = #{ i in row r : w[i] * x[col[i], j] == out[r, j] }
__global__ __launch_bounds__(kBlock) void max_count_kernel(const int32_t* __restrict__ row_ptr,
                                                           const int32_t* __restrict__ col,
                                                           const float* __restrict__ w, int64_t n_dst,
                                                           const float* __restrict__ x, int64_t ldx, int F,
                                                           const float* __restrict__ out, int64_t ldo,
                                                           float* __restrict__ count, int64_t ldc)
{
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t total = n_dst * F;
    for (; t < total; t += stride) {
        const int64_t r = t / F;
        const int j = int(t - r * F);
        const float o = out[r * ldo + j];
        int c = 0;
        for (int i = row_ptr[r]; i < row_ptr[r + 1]; ++i) {
            const float m = w ? w[i] * x[int64_t(col[i]) * ldx + j] : x[int64_t(col[i]) * ldx + j];
            c += (m == o);
        }
        count[r * ldc + j] = float(c);
    }
}

// transposed plan (row_ptr_t over SOURCES, dst_t = destination of each position, w_t in that order):
// gx[c, j] = sum_i [w*x[c,j] == out[dst,j]] * w * g[dst, j] / count[dst, j]
__global__ __launch_bounds__(kBlock) void max_backward_kernel(const int32_t* __restrict__ row_ptr_t,
                                                              const int32_t* __restrict__ dst_t,
                                                              const float* __restrict__ w_t, int64_t n_src,
                                                              const float* __restrict__ x, int64_t ldx, int F,
                                                              const float* __restrict__ out, int64_t ldo,
                                                              const float* __restrict__ g, int64_t ldg,
                                                              const float* __restrict__ count, int64_t ldc,
                                                              float* __restrict__ gx, int64_t ldgx)
{
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t total = n_src * F;
    for (; t < total; t += stride) {
        const int64_t c = t / F;
        const int j = int(t - c * F);
        const float xv = x[c * ldx + j];
        float acc = 0.0f;
        for (int i = row_ptr_t[c]; i < row_ptr_t[c + 1]; ++i) {
            const int64_t r = dst_t[i];
            const float wi = w_t ? w_t[i] : 1.0f;
            const float m = w_t ? wi * xv : xv;
            if (m == out[r * ldo + j]) acc += wi * g[r * ldg + j] / count[r * ldc + j];
        }
        gx[c * ldgx + j] = acc;
    }
}

// ---------------------------------------------------------------- GAT backward
struct GB {
    const int32_t* row_ptr;   // forward plan (by destination) or transposed plan (by source)
    const int32_t* other;     // col (sources) for the dst pass; destinations for the src pass
    int64_t n;                // rows of this pass
    const float* q; int64_t ldq;
    const float* k; int64_t ldk;
    const float* v; int64_t ldv;
    const float* go; int64_t ldgo;   // dO [n_dst, H*dv]
    const float* ml;                 // [n_dst, 2H] saved (m, l) of the forward softmax
    const float* dsum;               // [n_dst, H]  D = <dO, O> per head
    int32_t H, d, dv, add_self_loop;
    float scale;
    float* gq; int64_t ldgq;
    float* gk; int64_t ldgk;
    float* gv; int64_t ldgv;
};

__device__ __forceinline__ float edge_alpha(const GB& a, int64_t r, int64_t c, int h, float& s_out)
{
    const float* qp = a.q + r * a.ldq + h * a.d;
    const float* kp = a.k + c * a.ldk + h * a.d;
    float dot = 0.0f;
    for (int u = 0; u < a.d; ++u) dot = fmaf(qp[u], kp[u], dot);
    const float s = dot / a.scale;
    s_out = s;
    return expf(s - a.ml[r * 2 * a.H + 2 * h]) / (a.ml[r * 2 * a.H + 2 * h + 1] + 1e-8f);
}

__device__ __forceinline__ float edge_ds(const GB& a, int64_t r, int64_t c, int h, float alpha)
{
    const float* gop = a.go + r * a.ldgo + h * a.dv;
    const float* vp = a.v + c * a.ldv + h * a.dv;
    float da = 0.0f;
    for (int u = 0; u < a.dv; ++u) da = fmaf(gop[u], vp[u], da);
    return alpha * (da - a.dsum[r * a.H + h]);    // softmax backward: ds = alpha * (dalpha - sum alpha*dalpha)
}

// one lane per (destination r, head h): dQ[r,h,:] = sum_e ds_e * K[c_e,h,:] / scale
__global__ __launch_bounds__(kBlock) void gat_backward_dst_kernel(const GB a)
{
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t total = a.n * a.H;
    for (; t < total; t += stride) {
        const int64_t r = t / a.H;
        const int h = int(t - r * a.H);
        float* gqp = a.gq + r * a.ldgq + h * a.d;
        for (int u = 0; u < a.d; ++u) gqp[u] = 0.0f;
        const int s = a.row_ptr[r], e = a.row_ptr[r + 1];
        for (int i = s; i <= e; ++i) {
            if (i == e && !a.add_self_loop) break;
            const int64_t c = (i == e) ? r : a.other[i];
            float sc;
            const float alpha = edge_alpha(a, r, c, h, sc);
            const float ds = edge_ds(a, r, c, h, alpha) / a.scale;
            const float* kp = a.k + c * a.ldk + h * a.d;
            for (int u = 0; u < a.d; ++u) gqp[u] = fmaf(ds, kp[u], gqp[u]);
        }
    }
}

// one lane per (source c, head h): dV[c,h,:] = sum_e alpha_e dO[r_e,h,:];  dK[c,h,:] = sum_e ds_e Q[r_e,h,:] / scale
__global__ __launch_bounds__(kBlock) void gat_backward_src_kernel(const GB a)
{
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t total = a.n * a.H;
    for (; t < total; t += stride) {
        const int64_t c = t / a.H;
        const int h = int(t - c * a.H);
        float* gkp = a.gk + c * a.ldgk + h * a.d;
        float* gvp = a.gv + c * a.ldgv + h * a.dv;
        for (int u = 0; u < a.d; ++u) gkp[u] = 0.0f;
        for (int u = 0; u < a.dv; ++u) gvp[u] = 0.0f;
        const int s = a.row_ptr[c], e = a.row_ptr[c + 1];
        for (int i = s; i <= e; ++i) {
            if (i == e && !a.add_self_loop) break;
            const int64_t r = (i == e) ? c : a.other[i];
            float sc;
            const float alpha = edge_alpha(a, r, c, h, sc);
            const float ds = edge_ds(a, r, c, h, alpha) / a.scale;
            const float* qp = a.q + r * a.ldq + h * a.d;
            const float* gop = a.go + r * a.ldgo + h * a.dv;
            for (int u = 0; u < a.d; ++u) gkp[u] = fmaf(ds, qp[u], gkp[u]);
            for (int u = 0; u < a.dv; ++u) gvp[u] = fmaf(alpha, gop[u], gvp[u]);
        }
    }
}

}  // namespace
}  // namespace tfgx

using namespace tfgx;

extern "C" int tfgx_sddmm_f32(const int32_t* row_ptr, const int32_t* col, int64_t n_dst, const float* a, int64_t lda,
                              const float* b, int64_t ldb, int64_t F, float* out, tfgx_stream_t stream)
{
    TFGX_REQUIRE(n_dst >= 0 && F >= 1 && lda >= F && ldb >= F, "bad size");
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr && a && b, "null pointer");
    hipStream_t s = as_stream(stream);
    if (F <= 8) sddmm_kernel<8><<<grid_for(n_dst, kBlock / 8, 1 << 20), kBlock, 0, s>>>(row_ptr, col, n_dst, a, lda, b, ldb, int(F), out);
    else if (F <= 32) sddmm_kernel<16><<<grid_for(n_dst, kBlock / 16, 1 << 20), kBlock, 0, s>>>(row_ptr, col, n_dst, a, lda, b, ldb, int(F), out);
    else sddmm_kernel<32><<<grid_for(n_dst, kBlock / 32, 1 << 20), kBlock, 0, s>>>(row_ptr, col, n_dst, a, lda, b, ldb, int(F), out);
    TFGX_LAUNCH_CHECK("sddmm_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_segment_max_count_f32(const int32_t* row_ptr, const int32_t* col, const float* w, int64_t n_dst,
                                          const float* x, int64_t ldx, int64_t F, const float* out, int64_t ldo,
                                          float* count, int64_t ldc, tfgx_stream_t stream)
{
    TFGX_REQUIRE(n_dst >= 0 && F >= 1 && ldx >= F && ldo >= F && ldc >= F, "bad size");
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr && x && out && count, "null pointer");
    max_count_kernel<<<grid_for(n_dst * F, kBlock), kBlock, 0, as_stream(stream)>>>(row_ptr, col, w, n_dst, x, ldx,
                                                                                   int(F), out, ldo, count, ldc);
    TFGX_LAUNCH_CHECK("max_count_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_segment_max_backward_f32(const int32_t* row_ptr_t, const int32_t* dst_t, const float* w_t,
                                             int64_t n_src, const float* x, int64_t ldx, int64_t F, const float* out,
                                             int64_t ldo, const float* g, int64_t ldg, const float* count, int64_t ldc,
                                             float* gx, int64_t ldgx, tfgx_stream_t stream)
{
    TFGX_REQUIRE(n_src >= 0 && F >= 1 && ldx >= F && ldo >= F && ldg >= F && ldc >= F && ldgx >= F, "bad size");
    if (n_src == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr_t && x && out && g && count && gx, "null pointer");
    max_backward_kernel<<<grid_for(n_src * F, kBlock), kBlock, 0, as_stream(stream)>>>(
        row_ptr_t, dst_t, w_t, n_src, x, ldx, int(F), out, ldo, g, ldg, count, ldc, gx, ldgx);
    TFGX_LAUNCH_CHECK("max_backward_kernel");
    return TFGX_OK;
}

static int fill_gb(const tfgx_gat_backward_args* p, GB& a)
{
    TFGX_REQUIRE(p != nullptr, "args is null");
    TFGX_REQUIRE(p->H >= 1 && p->d >= 1 && p->dv >= 1 && p->scale > 0.0f, "bad H / d / dv / scale");
    TFGX_REQUIRE(p->q && p->k && p->v && p->grad_out && p->stats_ml && p->dsum, "null pointer");
    a.q = p->q; a.ldq = p->ldq; a.k = p->k; a.ldk = p->ldk; a.v = p->v; a.ldv = p->ldv;
    a.go = p->grad_out; a.ldgo = p->ld_grad_out; a.ml = p->stats_ml; a.dsum = p->dsum;
    a.H = p->H; a.d = p->d; a.dv = p->dv; a.add_self_loop = p->add_self_loop; a.scale = p->scale;
    a.gq = p->grad_q; a.ldgq = p->ld_grad_q; a.gk = p->grad_k; a.ldgk = p->ld_grad_k;
    a.gv = p->grad_v; a.ldgv = p->ld_grad_v;
    return TFGX_OK;
}

extern "C" int tfgx_gat_backward_dst_f32(const tfgx_gat_backward_args* p, tfgx_stream_t stream)
{
    GB a;
    int rc = fill_gb(p, a);
    if (rc) return rc;
    TFGX_REQUIRE(p->row_ptr && p->grad_q && p->n_dst >= 0, "dst pass needs row_ptr / grad_q");
    if (p->n_dst == 0) return TFGX_OK;
    a.row_ptr = p->row_ptr; a.other = p->col; a.n = p->n_dst;
    gat_backward_dst_kernel<<<grid_for(a.n * a.H, kBlock), kBlock, 0, as_stream(stream)>>>(a);
    TFGX_LAUNCH_CHECK("gat_backward_dst_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_gat_backward_src_f32(const tfgx_gat_backward_args* p, tfgx_stream_t stream)
{
    GB a;
    int rc = fill_gb(p, a);
    if (rc) return rc;
    TFGX_REQUIRE(p->row_ptr_t && p->grad_k && p->grad_v && p->n_src >= 0, "src pass needs row_ptr_t / grad_k / grad_v");
    if (p->n_src == 0) return TFGX_OK;
    a.row_ptr = p->row_ptr_t; a.other = p->dst_t; a.n = p->n_src;
    gat_backward_src_kernel<<<grid_for(a.n * a.H, kBlock), kBlock, 0, as_stream(stream)>>>(a);
    TFGX_LAUNCH_CHECK("gat_backward_src_kernel");
    return TFGX_OK;
}
