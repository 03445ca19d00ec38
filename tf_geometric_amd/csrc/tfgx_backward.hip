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
//   unsorted_segment_max (training forward) = tfgx_segment_max_with_count_f32: maxima and tie counts in one pass
//   d(max)/d(edge weight)                  = tfgx_segment_max_backward_w_f32 (the SDDMM kernel with an arg-max mask)
//   unsorted_segment_max gradient, mask form = tfgx_segment_max_with_arg_f32 + tfgx_segment_max_backward_mask_f32
//                                            (per-edge winner bit masks, one gather per edge; the default)
// Every accumulation has one owner (a destination row or a source row): deterministic, no atomics on global memory —
// except the OPTIONAL push form of the max gradient (tfgx_segment_max_backward_push_f32, N*F float atomics).
// Each entry point has a tuned kernel (lane group per row, float4 columns, the forward kernel's mapping) and a plain
// one-lane-per-output fallback for layouts the tuned one does not cover.
#include "tfgx_common.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <type_traits>
#ifndef TFGX_GAT_BWD_DPP_SUM
#define TFGX_GAT_BWD_DPP_SUM 1        // developer A/B: 0 = the per-head <dO, V> reduction as a loop of ds_bpermute shuffles
#endif
#ifndef TFGX_GAT_BWD_POW2_SCALE
#define TFGX_GAT_BWD_POW2_SCALE 1     // developer A/B: 0 = the attention backward always DIVIDES by scale
#endif

namespace tfgx {
namespace {

// ---------------------------------------------------------------- SDDMM: out[i] = <a[r], b[col[i]]>
template <int G>
__global__ __launch_bounds__(kBlock) void sddmm_kernel(const int32_t* __restrict__ rb, const int32_t* __restrict__ re,
                                                       const int32_t* __restrict__ part_row, int skip,
                                                       const int32_t* __restrict__ col, int64_t n_dst,
                                                       const float* __restrict__ a, int64_t lda,
                                                       const float* __restrict__ b, int64_t ldb, int F,
                                                       float* __restrict__ out, const float* __restrict__ w = nullptr,
                                                       const float* __restrict__ mx = nullptr, int64_t ldmx = 0)
{
    constexpr int ROWS_PER_BLOCK = kBlock / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    // virtual rows (see HubLists below): part p = positions [rb[p], re[p]) of row part_row[p]; every edge has its own output
    for (int64_t p = int64_t(blockIdx.x) * ROWS_PER_BLOCK + grp; p < n_dst; p += int64_t(gridDim.x) * ROWS_PER_BLOCK) {
        const int s = rb[p], e = re[p];
        if (skip > 0 && e - s > skip) continue;
        const int64_t r = part_row ? int64_t(part_row[p]) : p;
        const float* ar = a + r * lda;
        for (int i = s; i < e; ++i) {
            const float* br = b + int64_t(col[i]) * ldb;
            float acc = 0.0f;
            if (mx) {   // d(max)/dw: only the features where this edge attains the row maximum contribute
                const float wi = w[i];
                for (int j = lane; j < F; j += G) acc = (wi * br[j] == mx[r * ldmx + j]) ? fmaf(ar[j], br[j], acc) : acc;
            } else
            for (int j = lane; j < F; j += G) acc = fmaf(ar[j], br[j], acc);
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
            if (lane == 0) out[i] = acc;
        }
    }
}

// Tuned variant (16-byte aligned rows, F % 4 == 0, F <= 4 * G * CH): G lanes per destination row, the a-row lives in
// registers, U = 8 edges in flight (8 independent dwordx4 gathers per lane before the first FMA), the 8 partial dot
// products are reduced across the group and written as one coalesced 8-wide store.  Same traffic as the forward
// segment-sum: one gathered source row per edge.
// MASKED (d(max aggregate)/d(edge weight)): a = g / count, and only features j with w[i] * b[col[i], j] == mx[r, j]
// (the edge attains the row maximum there) enter the dot product.
template <int G, int CH, bool MASKED>
__global__ __launch_bounds__(kBlock) void sddmm_fast_kernel(const int32_t* __restrict__ rb,
                                                            const int32_t* __restrict__ re,
                                                            const int32_t* __restrict__ part_row, int skip,
                                                            const int32_t* __restrict__ col, int64_t n_dst,
                                                            const float* __restrict__ a, int64_t lda,
                                                            const float* __restrict__ b, int64_t ldb, int F,
                                                            float* __restrict__ out, const float* __restrict__ w,
                                                            const float* __restrict__ mx, int64_t ldmx)
{
    constexpr int ROWS_PER_BLOCK = kBlock / G, U = 8, UL = 8 / CH;      // UL edges (8 float4 row pieces) in flight at a time
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    int joff[CH];
    bool jvalid[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int j = 4 * (lane + c * G);
        jvalid[c] = j < F;
        joff[c] = jvalid[c] ? j : F - 4;
    }
    for (int64_t p = int64_t(blockIdx.x) * ROWS_PER_BLOCK + grp; p < n_dst; p += int64_t(gridDim.x) * ROWS_PER_BLOCK) {
        const int s = rb[p], e = re[p];
        if (skip > 0 && e - s > skip) continue;
        const int64_t r = part_row ? int64_t(part_row[p]) : p;
        float4 av[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int j = 4 * (lane + c * G);
            av[c] = j < F ? *reinterpret_cast<const float4*>(a + r * lda + j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 mv[MASKED ? CH : 1];
        if (MASKED) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int j = 4 * (lane + c * G);
                mv[c] = j < F ? *reinterpret_cast<const float4*>(mx + r * ldmx + j) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // (col, w) are read G at a time, one per lane, coalesced, and the NEXT batch is requested while the current one's
        // rows are being gathered (as the forward kernel does): the index load heads every gather's dependency chain, and
        // one 8-lane load per 8 edges left a full memory round trip in front of each sub-batch
        int cj_next = (s + lane < e) ? col[s + lane] : 0;                        // 0: a valid row for the padded slots
        float wj_next = (MASKED && s + lane < e) ? w[s + lane] : 0.0f;
        for (int base = s; base < e; base += G) {
          const int cj = cj_next;
          const float wj = wj_next;
          cj_next = 0;
          wj_next = 0.0f;
          if (base + G + lane < e) {
              cj_next = col[base + G + lane];
              if (MASKED) wj_next = w[base + G + lane];
          }
          const int cnt = min(G, e - base);
          for (int j0 = 0; j0 < cnt; j0 += U) {
            const int i0 = base + j0;
            // all U row loads are issued before the first product is formed, from CLAMPED column offsets with no branch
            // around them: a predicated load makes the compiler drain vmcnt(0) after each one (the previous version of this
            // loop ran its "8 gathers in flight" strictly one after the other); lanes past F re-read the last valid vector
            // against a zero a-row
            float acc[U];
#pragma unroll
            for (int u0 = 0; u0 < U; u0 += UL) {
                float4 bv[UL][CH];
                float wu[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const int cu = __shfl(cj, (j0 + u0 + u) & (G - 1), G);
                    wu[u] = MASKED ? __shfl(wj, (j0 + u0 + u) & (G - 1), G) : 0.0f;
                    const float* br = b + int64_t(cu) * ldb;
#pragma unroll
                    for (int c = 0; c < CH; ++c) bv[u][c] = *reinterpret_cast<const float4*>(br + joff[c]);
                }
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    float t = 0.0f;
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const float4 b4 = bv[u][c];
                        if (MASKED) {
                            const float4 m4 = mv[c];
                            t = (jvalid[c] && wu[u] * b4.x == m4.x) ? fmaf(av[c].x, b4.x, t) : t;
                            t = (jvalid[c] && wu[u] * b4.y == m4.y) ? fmaf(av[c].y, b4.y, t) : t;
                            t = (jvalid[c] && wu[u] * b4.z == m4.z) ? fmaf(av[c].z, b4.z, t) : t;
                            t = (jvalid[c] && wu[u] * b4.w == m4.w) ? fmaf(av[c].w, b4.w, t) : t;
                        } else {
                            t = fmaf(av[c].x, b4.x, t);
                            t = fmaf(av[c].y, b4.y, t);
                            t = fmaf(av[c].z, b4.z, t);
                            t = fmaf(av[c].w, b4.w, t);
                        }
                    }
                    acc[u0 + u] = t;
                }
            }
            if constexpr (G >= 8) {
                // Reduce the 8 partial dot products across the G lanes with a VALUE-HALVING butterfly: at each of the first
                // three steps a lane hands half of its values to its partner and keeps (and completes) the other half, so
                // 4 + 2 + 1 exchanges replace 3 x 8; the remaining log2(G) - 3 steps carry one value.  9 cross-lane
                // exchanges per 8 edges instead of 40 at G = 32 (each is an LDS-crossbar ds_bpermute: 5 per edge next to the
                // forward kernel's 2 was what kept this kernel 0.8 ms behind the forward pass at products shape).
                // Afterwards the lanes whose bits (G/2, G/4, G/8) spell u hold edge u's dot product.
                constexpr int o1 = G / 2, o2 = G / 4, o3 = G / 8;
                float a4[4], a2[2];
                const bool hi1 = (lane & o1) != 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float keep = hi1 ? acc[u + 4] : acc[u], send = hi1 ? acc[u] : acc[u + 4];
                    a4[u] = keep + __shfl_xor(send, o1, G);
                }
                const bool hi2 = (lane & o2) != 0;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float keep = hi2 ? a4[u + 2] : a4[u], send = hi2 ? a4[u] : a4[u + 2];
                    a2[u] = keep + __shfl_xor(send, o2, G);
                }
                const bool hi3 = (lane & o3) != 0;
                float v = (hi3 ? a2[1] : a2[0]) + __shfl_xor(hi3 ? a2[0] : a2[1], o3, G);
#pragma unroll
                for (int o = G / 16; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
                const int u_mine = (hi1 ? 4 : 0) + (hi2 ? 2 : 0) + (hi3 ? 1 : 0);
                if ((lane & (o3 - 1)) == 0 && i0 + u_mine < e) out[i0 + u_mine] = v;
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int o = G / 2; o > 0; o >>= 1) acc[u] += __shfl_xor(acc[u], o, G);
                }
                float v = acc[0];
#pragma unroll
                for (int u = 1; u < U; ++u) v = (lane == u) ? acc[u] : v;
                if (lane < U && i0 + lane < e) out[i0 + lane] = v;
            }
          }
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

// fast variants of the two kernels above: a lane group per row, 4 columns per lane (rows 16-byte aligned, F % 4 == 0)
// MODE 0: count[r, :] = #ties;  MODE 1 (transposed plan): gx[c, :] = sum [tie] * w * gn[dst, :], gn = g / count
// Long rows ("hubs" of a power-law graph) must not be walked by one lane group: the launches below take VIRTUAL rows —
// part p spans CSR positions [rb[p], re[p]) of row part_row[p] (NULL: p itself) and writes its (partial) sums to row p of
// `res`; parts longer than `skip` (> 0) are left out.  A hub-aware caller runs the plan's rows with skip = threshold,
// then the hub rows' chunks into a scratch, then hub_sum_finalize_kernel adds a row's chunk partials in chunk order
// (deterministic) — the backward twin of the forward's chunked hub path.
struct HubLists {
    int thr;
    int64_t n_rows, n_chunks;
    const int32_t *rows, *chunk_ptr, *chunk_begin, *chunk_end, *chunk_row;
};

__global__ __launch_bounds__(kBlock) void hub_sum_finalize_kernel(const float* __restrict__ scratch, int64_t lds,
                                                                  const int32_t* __restrict__ chunk_ptr,
                                                                  const int32_t* __restrict__ hub_rows, int64_t n_hub_rows,
                                                                  int F, float* __restrict__ out, int64_t ldo)
{
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (; t < n_hub_rows * F; t += stride) {
        const int64_t i = t / F;
        const int j = int(t - i * F);
        float acc = 0.0f;
        for (int c = chunk_ptr[i]; c < chunk_ptr[i + 1]; ++c) acc += scratch[int64_t(c) * lds + j];
        out[int64_t(hub_rows[i]) * ldo + j] = acc;
    }
}

inline bool fill_hub(const tfgx_hub_lists* h, HubLists& o)
{
    if (h == nullptr || h->threshold <= 0 || h->n_rows <= 0 || h->n_chunks <= 0) return false;
    o.thr = h->threshold; o.n_rows = h->n_rows; o.n_chunks = h->n_chunks;
    o.rows = h->rows; o.chunk_ptr = h->chunk_ptr; o.chunk_begin = h->chunk_begin; o.chunk_end = h->chunk_end;
    o.chunk_row = h->chunk_row;
    return o.rows && o.chunk_ptr && o.chunk_begin && o.chunk_end && o.chunk_row;
}

template <int G, int MODE>
__global__ __launch_bounds__(kBlock) void max_grad_fast_kernel(const int32_t* __restrict__ rb,
                                                               const int32_t* __restrict__ re,
                                                               const int32_t* __restrict__ part_row, int skip,
                                                               const int32_t* __restrict__ other,
                                                               const float* __restrict__ w, int64_t n,
                                                               const float* __restrict__ x, int64_t ldx, int F,
                                                               const float* __restrict__ out, int64_t ldo,
                                                               const float* __restrict__ gn, int64_t ldg,
                                                               float* __restrict__ res, int64_t ldr,
                                                               float* __restrict__ out_w = nullptr,
                                                               int32_t* __restrict__ argpos = nullptr, int64_t lda = 0)
{
    // MODE 0: res = tie count given the row maxima `out`;  MODE 1: res = gradient wrt x (transposed plan);
    // MODE 2: the FORWARD of training: row maximum -> out_w and tie count -> res in ONE pass (online: a value above the
    //         running maximum restarts the count at 1, an equal one increments it), so the backward needs no count pass
    constexpr int VEC = 4;
    constexpr int ROWS_PER_BLOCK = kBlock / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int c_raw = (blockIdx.y * G + lane) * VEC;
    const bool cvalid = c_raw < F;
    const int coff = cvalid ? c_raw : (F - VEC);
    for (int64_t part = int64_t(blockIdx.x) * ROWS_PER_BLOCK + grp; part < n; part += int64_t(gridDim.x) * ROWS_PER_BLOCK) {
        const int s = rb[part], e = re[part];
        if (skip > 0 && e - s > skip) continue;                  // a hub row: its chunks are launched separately
        const int64_t row = part_row ? int64_t(part_row[part]) : part;
        float mine[VEC], acc[VEC];
        int apos[VEC];            // MODE 2: CSR position of the FIRST edge attaining the running maximum (-1: empty row)
#pragma unroll
        for (int i = 0; i < VEC; ++i) apos[i] = -1;
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) mine[i] = -FLT_MAX;                        // running maximum
        } else {
            load_vec<VEC>((MODE == 0 ? out + row * ldo : x + row * ldx) + coff, mine);   // out[r,:] | x[c,:]
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
        for (int base = s; base < e; base += G) {
            const int idx = base + lane;
            const int oj = (idx < e) ? other[idx] : 0;
            const float wj = (w && idx < e) ? w[idx] : 1.0f;
            const int cnt = min(G, e - base);
            for (int j = 0; j < cnt; ++j) {
                const int64_t o = __shfl(oj, j, G);
                const float wi = __shfl(wj, j, G);
                if (MODE == 0) {
                    float xv[VEC];
                    load_vec<VEC>(x + o * ldx + coff, xv);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[i] += ((w ? wi * xv[i] : xv[i]) == mine[i]) ? 1.0f : 0.0f;
                } else if (MODE == 2) {
                    float xv[VEC];
                    load_vec<VEC>(x + o * ldx + coff, xv);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float m = w ? wi * xv[i] : xv[i];
                        apos[i] = (m > mine[i] || apos[i] < 0) ? (base + j) : apos[i];
                        acc[i] = m > mine[i] ? 1.0f : (m == mine[i] ? acc[i] + 1.0f : acc[i]);
                        mine[i] = fmaxf(mine[i], m);
                    }
                } else {
                    float ov[VEC], gv[VEC];
                    load_vec<VEC>(out + o * ldo + coff, ov);
                    load_vec<VEC>(gn + o * ldg + coff, gv);
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        acc[i] += ((w ? wi * mine[i] : mine[i]) == ov[i]) ? wi * gv[i] : 0.0f;
                }
            }
        }
        if (cvalid) {
            store_vec<VEC>(res + part * ldr + coff, acc);
            if (MODE == 2) store_vec<VEC>(out_w + part * ldo + coff, mine);
            if (MODE == 2 && argpos != nullptr) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) argpos[part * lda + coff + i] = apos[i];
            }
        }
    }
}

template <int MODE>
int launch_max_grad_parts(const int32_t* rb, const int32_t* re, const int32_t* part_row, int skip, const int32_t* other,
                          const float* w, int64_t n, const float* x, int64_t ldx, int F, const float* out, int64_t ldo,
                          const float* gn, int64_t ldg, float* res, int64_t ldr, hipStream_t stream,
                          float* out_w = nullptr, int32_t* argpos = nullptr, int64_t lda = 0)
{
    const int lanes = (F + 3) / 4;
#define TFGX_MG(GG)                                                                                            \
    {                                                                                                          \
        dim3 grid(grid_for(n, kBlock / GG, 1 << 20), (lanes + GG - 1) / GG, 1);                                \
        max_grad_fast_kernel<GG, MODE><<<grid, kBlock, 0, stream>>>(rb, re, part_row, skip, other, w, n, x, ldx, F, out,  \
                                                                      ldo, gn, ldg, res, ldr, out_w, argpos, lda);     \
    }
    if (lanes <= 8) TFGX_MG(8)
    else if (lanes <= 16) TFGX_MG(16)
    else if (lanes <= 32) TFGX_MG(32)
    else TFGX_MG(64)
#undef TFGX_MG
    TFGX_LAUNCH_CHECK("max_grad_fast_kernel");
    return TFGX_OK;
}

// rows of a plan; with hub lists: rows up to the threshold, then the hub rows' chunks + the ordered sum of their partials
template <int MODE>
int launch_max_grad(const int32_t* row_ptr, const int32_t* other, const float* w, int64_t n, const float* x,
                    int64_t ldx, int F, const float* out, int64_t ldo, const float* gn, int64_t ldg, float* res,
                    int64_t ldr, hipStream_t stream, float* out_w = nullptr, int32_t* argpos = nullptr, int64_t lda = 0,
                    const HubLists* hub = nullptr, float* scratch = nullptr)
{
    const bool chunked = hub != nullptr && scratch != nullptr && MODE != 2;
    int rc = launch_max_grad_parts<MODE>(row_ptr, row_ptr + 1, nullptr, chunked ? hub->thr : 0, other, w, n, x, ldx, F, out,
                                         ldo, gn, ldg, res, ldr, stream, out_w, argpos, lda);
    if (rc != TFGX_OK || !chunked) return rc;
    rc = launch_max_grad_parts<MODE>(hub->chunk_begin, hub->chunk_end, hub->chunk_row, 0, other, w, hub->n_chunks, x, ldx, F,
                                     out, ldo, gn, ldg, scratch, F, stream);
    if (rc != TFGX_OK) return rc;
    hub_sum_finalize_kernel<<<grid_for(hub->n_rows * F, kBlock), kBlock, 0, stream>>>(scratch, F, hub->chunk_ptr, hub->rows,
                                                                                     hub->n_rows, F, res, ldr);
    TFGX_LAUNCH_CHECK("hub_sum_finalize_kernel");
    return TFGX_OK;
}

// Push form of the segment-max gradient (training on near-regular graphs): the training forward saved, per (row, column),
// the CSR position of the first maximal edge and the number of tied maxima.  A destination row then hands each of its F
// gradient values straight to the ONE source element that produced the maximum:
//     gx[col[argpos[r, j]], j] += w[argpos[r, j]] * g[r, j]
// — N*F scattered float atomics instead of E*F gathered elements (two row gathers per edge in the pull form above: 19 ms
// at products shape vs ~3 ms).  Rows where some column has TIED maxima (count > 1) are walked exactly like the forward
// and every tied edge receives g / count, so TensorFlow's unsorted_segment_max gradient is reproduced for ties too.
// The price: float atomics commit in arrival order, so sums of several contributions to one gx element may differ in
// the last bit from run to run (TensorFlow's own GPU kernels behave the same).  The pull kernel stays available for
// bit-reproducible training (tfgx_segment_max_backward_f32).
template <int G>
__global__ __launch_bounds__(kBlock) void max_backward_push_kernel(const int32_t* __restrict__ row_ptr,
                                                                   const int32_t* __restrict__ col,
                                                                   const float* __restrict__ w, int64_t n_dst,
                                                                   const float* __restrict__ x, int64_t ldx, int F,
                                                                   const float* __restrict__ out, int64_t ldo,
                                                                   const float* __restrict__ g, int64_t ldg,
                                                                   const float* __restrict__ count, int64_t ldc,
                                                                   const int32_t* __restrict__ argpos, int64_t lda,
                                                                   float* __restrict__ gx, int64_t ldgx)
{
    constexpr int VEC = 4;
    constexpr int ROWS_PER_BLOCK = kBlock / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int c_raw = (blockIdx.y * G + lane) * VEC;
    const bool cvalid = c_raw < F;
    const int coff = cvalid ? c_raw : (F - VEC);
    for (int64_t r0 = int64_t(blockIdx.x) * ROWS_PER_BLOCK; r0 < n_dst; r0 += int64_t(gridDim.x) * ROWS_PER_BLOCK) {
        const int64_t r = r0 + grp;
        const bool rvalid = r < n_dst;
        const int s = rvalid ? row_ptr[r] : 0, e = rvalid ? row_ptr[r + 1] : 0;
        float gv[VEC] = {0.f, 0.f, 0.f, 0.f}, cv[VEC] = {0.f, 0.f, 0.f, 0.f};
        int ap[VEC] = {-1, -1, -1, -1};
        if (rvalid && cvalid && e > s) {
            load_vec<VEC>(g + r * ldg + coff, gv);
            load_vec<VEC>(count + r * ldc + coff, cv);
#pragma unroll
            for (int i = 0; i < VEC; ++i) ap[i] = argpos[r * lda + coff + i];
        }
        bool tie = false;
#pragma unroll
        for (int i = 0; i < VEC; ++i) tie |= cv[i] > 1.0f;
        // does any lane of this row's group see a tie?  (groups are aligned sub-ranges of the wave)
        const unsigned long long b = __ballot(tie);
        const int sh = (threadIdx.x % 64) / G * G;
        const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << sh);
        const bool row_has_tie = (b & gmask) != 0ull;
        if (!row_has_tie) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                if (ap[i] >= 0) {
                    const int p = ap[i];
                    const float wi = w ? w[p] : 1.0f;
                    atomicAdd(gx + int64_t(col[p]) * ldgx + coff + i, wi * gv[i]);
                }
            }
        } else {   // exact walk: every edge attaining the maximum of a column receives g / count
            float ov[VEC] = {0.f, 0.f, 0.f, 0.f};
            if (rvalid && cvalid && e > s) load_vec<VEC>(out + r * ldo + coff, ov);
            for (int i0 = s; i0 < e; ++i0) {
                const int c = col[i0];
                const float wi = w ? w[i0] : 1.0f;
                float xv[VEC];
                load_vec<VEC>(x + int64_t(c) * ldx + coff, xv);
                if (cvalid) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float m = w ? wi * xv[i] : xv[i];
                        if (m == ov[i] && cv[i] > 0.0f) atomicAdd(gx + int64_t(c) * ldgx + coff + i, wi * gv[i] / cv[i]);
                    }
                }
            }
        }
    }
}

// Deterministic single-gather form of the segment-max gradient ("mask" form).
//   A (by destination, this kernel): from the arg positions / tie counts the training forward saved, write per EDGE a
//     bit mask over the F columns — bit j of edge p is set iff edge p attains the maximum of column j of its row (rows
//     with tied maxima are walked exactly so that EVERY tied edge is marked) — and gn = g / count per destination row.
//     A row's mask is assembled in LDS (atomicOr on a window of 256 edges) and written out coalesced.
//   B (by source, max_backward_mask_apply_kernel): a source row walks its out-edges in transposed order, reads the
//     16 bytes of mask of each edge's forward position (one line request) and gathers gn[dst, j] only for the columns
//     whose bit is set — on average N*F/E (about 2 of 100) per edge — instead of two whole rows per edge.
// Every gx element has one owner and a fixed summation order: bit-reproducible, no atomics on global memory.
constexpr int kMaskWindow = 256;      // edges of a row whose masks are assembled in LDS at a time

template <int G>
__global__ __launch_bounds__(kBlock) void max_mask_build_kernel(const int32_t* __restrict__ row_ptr,
                                                                const int32_t* __restrict__ col,
                                                                const float* __restrict__ w, int64_t n_dst,
                                                                const float* __restrict__ x, int64_t ldx, int F,
                                                                const float* __restrict__ out, int64_t ldo,
                                                                const float* __restrict__ g, int64_t ldg,
                                                                const float* __restrict__ count, int64_t ldc,
                                                                const int32_t* __restrict__ argpos, int64_t lda,
                                                                float* __restrict__ gn, int64_t ldgn, uint32_t* __restrict__ mask,
                                                                int MW)
{
    constexpr int VEC = 4;
    constexpr int ROWS_PER_BLOCK = kBlock / G;
    constexpr int WPB = G / 8;                                   // mask words covered by one lane group (4*G columns)
    __shared__ uint32_t lds[ROWS_PER_BLOCK][kMaskWindow * WPB];
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int c_raw = (blockIdx.y * G + lane) * VEC;
    const bool cvalid = c_raw < F;
    const int coff = cvalid ? c_raw : (F - VEC);
    const int word0 = blockIdx.y * WPB;                         // first mask word of this column block
    uint32_t* my = lds[grp];
    for (int64_t r0 = int64_t(blockIdx.x) * ROWS_PER_BLOCK; r0 < n_dst; r0 += int64_t(gridDim.x) * ROWS_PER_BLOCK) {
        const int64_t r = r0 + grp;
        const bool rvalid = r < n_dst;
        const int s = rvalid ? row_ptr[r] : 0, e = rvalid ? row_ptr[r + 1] : 0;
        float gv[VEC] = {0.f, 0.f, 0.f, 0.f}, cv[VEC] = {0.f, 0.f, 0.f, 0.f}, ov[VEC] = {0.f, 0.f, 0.f, 0.f};
        int ap[VEC] = {-1, -1, -1, -1};
        if (rvalid && cvalid) {
            load_vec<VEC>(g + r * ldg + coff, gv);
            load_vec<VEC>(out + r * ldo + coff, ov);
            if (count != nullptr) {
                load_vec<VEC>(count + r * ldc + coff, cv);
#pragma unroll
                for (int i = 0; i < VEC; ++i) ap[i] = argpos[r * lda + coff + i];
            } else {      // packed form (tfgx_reduce_args.track): count << 16 | position relative to the row start
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const uint32_t pk = uint32_t(argpos[r * lda + coff + i]);
                    cv[i] = float(pk >> 16);
                    ap[i] = (pk >> 16) ? s + int(pk & 0xFFFFu) : -1;
                }
            }
            float gnv[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) gnv[i] = cv[i] > 0.0f ? gv[i] / cv[i] : 0.0f;
            store_vec<VEC>(gn + r * ldgn + coff, gnv);
        }
        bool tie = false;
#pragma unroll
        for (int i = 0; i < VEC; ++i) tie |= cvalid && cv[i] > 1.0f;
        const unsigned long long bal = __ballot(tie);
        const int sh = (threadIdx.x % 64) / G * G;
        const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << sh);
        const bool row_has_tie = (bal & gmask) != 0ull;
        for (int wb = s; wb < e; wb += kMaskWindow) {
            const int wlen = min(kMaskWindow, e - wb);
            for (int i = lane; i < wlen * WPB; i += G) my[i] = 0u;
            __builtin_amdgcn_wave_barrier();
            if (!row_has_tie) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const int p = ap[i] - wb;
                    if (cvalid && p >= 0 && p < wlen) atomicOr(&my[p * WPB + lane / 8], 1u << ((lane % 8) * 4 + i));
                }
            } else {      // tied maxima somewhere in this row: mark every edge that attains a column's maximum
                for (int k = 0; k < wlen; ++k) {
                    const int c = col[wb + k];
                    const float wi = w ? w[wb + k] : 1.0f;
                    float xv[VEC];
                    load_vec<VEC>(x + int64_t(c) * ldx + coff, xv);
                    uint32_t nib = 0u;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) nib |= ((w ? wi * xv[i] : xv[i]) == ov[i] && cv[i] > 0.0f) ? (1u << i) : 0u;
                    if (cvalid && nib) atomicOr(&my[k * WPB + lane / 8], nib << ((lane % 8) * 4));
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < wlen * WPB; i += G) {
                const int k = i / WPB, ww = i % WPB;
                if (word0 + ww < MW) mask[int64_t(wb + k) * MW + word0 + ww] = my[i];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int G>
__global__ __launch_bounds__(kBlock) void max_backward_mask_apply_kernel(const int32_t* __restrict__ row_ptr_t,
                                                                         const int32_t* __restrict__ dst_t,
                                                                         const float* __restrict__ w_t,
                                                                         const int32_t* __restrict__ pos_t,
                                                                         int64_t n_src, int F,
                                                                         const float* __restrict__ gn, int64_t ldgn,
                                                                         const uint32_t* __restrict__ mask, int MW,
                                                                         float* __restrict__ gx, int64_t ldgx)
{
    constexpr int VEC = 4;
    constexpr int ROWS_PER_BLOCK = kBlock / G;
    constexpr int WPB = G / 8;
    constexpr int U = 8;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int c_raw = (blockIdx.y * G + lane) * VEC;
    const bool cvalid = c_raw < F;
    const int coff = cvalid ? c_raw : (F - VEC);
    const int word = blockIdx.y * WPB + lane / 8;
    const bool wvalid = word < MW;
    const int shift = (lane % 8) * 4;
    for (int64_t c = int64_t(blockIdx.x) * ROWS_PER_BLOCK + grp; c < n_src; c += int64_t(gridDim.x) * ROWS_PER_BLOCK) {
        const int s = row_ptr_t[c], e = row_ptr_t[c + 1];
        float acc[VEC] = {0.f, 0.f, 0.f, 0.f};
        for (int base = s; base < e; base += G) {
            const int idx = base + lane;
            const int rj = idx < e ? dst_t[idx] : 0;
            const int pj = idx < e ? (pos_t ? pos_t[idx] : idx) : 0;
            const float wj = (w_t && idx < e) ? w_t[idx] : 1.0f;
            const int cnt = min(G, e - base);
            for (int j0 = 0; j0 < cnt; j0 += U) {
                uint32_t nib[U];
                int64_t rr[U];
                float ww[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {          // U mask words in flight before the first is looked at
                    const int j = j0 + u < cnt ? j0 + u : cnt - 1;
                    const int64_t p = __shfl(pj, j, G);
                    rr[u] = __shfl(rj, j, G);
                    ww[u] = __shfl(wj, j, G);
                    const uint32_t mw = wvalid ? mask[p * MW + word] : 0u;
                    nib[u] = (j0 + u < cnt) ? ((mw >> shift) & 15u) : 0u;
                }
                // gather gn[dst, j] for the set bits only — as PREDICATED loads issued back to back (no branch between
                // them, so all of a batch's gathers are in flight together), then the FMAs in edge order
                float val[U][VEC];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float* gp = gn + rr[u] * ldgn + coff;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        val[u][i] = 0.0f;
                        if (nib[u] & (1u << i)) val[u][i] = gp[i];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[i] = fmaf(ww[u], val[u][i], acc[i]);
            }
        }
        if (cvalid) store_vec<VEC>(gx + c * ldgx + coff, acc);
    }
}

__global__ void divide_kernel(const float* __restrict__ g, int64_t ldg, const float* __restrict__ cnt, int64_t ldc,
                              int64_t n, int F, float* __restrict__ gn)
{
    int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; t < n * F; t += stride) {
        const int64_t r = t / F;
        const int j = int(t - r * F);
        const float c = cnt[r * ldc + j];
        gn[t] = c > 0.0f ? g[r * ldg + j] / c : 0.0f;
    }
}

// ---------------------------------------------------------------- GAT backward
struct GB {
    const int32_t* row_ptr;   // forward plan (by destination) or transposed plan (by source)
    const int32_t* other;     // col (sources) for the dst pass; destinations for the src pass
    int64_t n;                // rows (or parts) of this launch
    const int32_t* row_begin; // part p spans [row_begin[p * rp_stride], row_end[p * rp_stride]) of row part_row[p] (NULL: p); plain launch: row_ptr, row_ptr+1
    const int32_t* row_end;
    int64_t rp_stride;        // 1, or KB for one block of a source-blocked plan (tfgx_gat_backward_args.span_*)
    int32_t accumulate;       // 1: add to the gradients already stored (later blocks of a source-blocked pass)
    const int32_t* part_row;
    int32_t skip;             // > 0: parts longer than this are left to the chunk launch (hub rows)
    const int32_t* row_order; // walk order of a plain launch (NULL: identity): rows of similar length share a wave
    int64_t n_self;           // rows [0, n_self) of this pass own the appended self-loop (src pass of a rectangular
                              // operator: only sources that are also destinations, i.e. a shard's own rows)
    const float* q; int64_t ldq;
    const float* k; int64_t ldk;
    const float* v; int64_t ldv;
    const float* go; int64_t ldgo;   // dO [n_dst, H*dv]
    const float* ml;                 // [n_dst, 2H] saved (m, l) of the forward softmax
    const float* dsum;               // [n_dst, H]  D = <dO, O> per head
    int64_t ldml, lddsum;            // row strides of ml / dsum (2H / H unless the caller packed the per-row data)
    int32_t H, d, dv, add_self_loop;
    float scale;
    float inv_scale;                 // 1 / scale when scale is a power of two (x * inv_scale == x / scale bit for bit), else 0
    float* gq; int64_t ldgq;
    float* gk; int64_t ldgk;
    float* gv; int64_t ldgv;
    DropCfg drop;                    // the forward's attention dropout (keep mask regenerated from the edge position)
    const int32_t* pos;              // src pass: forward-CSR position of each transposed position (NULL: identity)
    const float* hp; int64_t ldhp;   // src pass, optional: per destination r and head h ONE block of roundup4(d + 3) floats
                                     // [Q[r,h,0..d) | m | 1 / (l + 1e-8) | D | pad] at hp[r * ldhp + h * block] (tfgx_gat_pack_dst_heads_f32)
};

constexpr int head_block(int d) { return (d + 3 + 3) / 4 * 4; }

// keep_scale_or_0 of edge (position p of this pass, destination r when it is the appended self-loop) for head h
__device__ __forceinline__ float edge_keep(const GB& a, int64_t p, bool self_loop, int64_t r, int h)
{
    const int64_t fwd = self_loop ? a.drop.self_base + r : (a.pos ? int64_t(a.pos[p]) : p);
    return drop_scale(a.drop, uint32_t(fwd * a.H + h));
}

__device__ __forceinline__ float edge_alpha(const GB& a, int64_t r, int64_t c, int h, float& s_out)
{
    const float* qp = a.q + r * a.ldq + h * a.d;
    const float* kp = a.k + c * a.ldk + h * a.d;
    float dot = 0.0f;
    for (int u = 0; u < a.d; ++u) dot = fmaf(qp[u], kp[u], dot);
    const float s = dot / a.scale;
    s_out = s;
    return expf(s - a.ml[r * a.ldml + 2 * h]) / (a.ml[r * a.ldml + 2 * h + 1] + 1e-8f);
}

__device__ __forceinline__ float edge_ds(const GB& a, int64_t r, int64_t c, int h, float alpha, float keep)
{
    const float* gop = a.go + r * a.ldgo + h * a.dv;
    const float* vp = a.v + c * a.ldv + h * a.dv;
    float da = 0.0f;
    for (int u = 0; u < a.dv; ++u) da = fmaf(gop[u], vp[u], da);
    // softmax backward: ds = alpha * (dalpha - sum alpha*dalpha), dalpha = keep * <dO, V> under attention dropout
    return alpha * (keep * da - a.dsum[r * a.lddsum + h]);
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
            const float ds = edge_ds(a, r, c, h, alpha, edge_keep(a, i, i == e, r, h)) / a.scale;
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
            if (i == e && (!a.add_self_loop || c >= a.n_self)) break;
            const int64_t r = (i == e) ? c : a.other[i];
            float sc;
            const float alpha = edge_alpha(a, r, c, h, sc);
            const float keep = edge_keep(a, i, i == e, r, h);
            const float ds = edge_ds(a, r, c, h, alpha, keep) / a.scale;
            const float* qp = a.q + r * a.ldq + h * a.d;
            const float* gop = a.go + r * a.ldgo + h * a.dv;
            for (int u = 0; u < a.d; ++u) gkp[u] = fmaf(ds, qp[u], gkp[u]);
            for (int u = 0; u < a.dv; ++u) gvp[u] = fmaf(alpha * keep, gop[u], gvp[u]);
        }
    }
}

// ---- fast GAT backward: a group of G lanes per row, lane -> 4 value columns of one head (as the forward kernel).
// Needs dv % 4 == 0, LH = dv/4 a power of two (the head's lanes form an aligned butterfly) or ONE head of any width,
// d in {1,2,4,8,16,32}; other per-head widths reach these kernels zero-padded by the host (nn/conv/gat.py).
// Per edge the bytes moved are the forward's (K row + V row) for the dst pass and Q row + dO row + 3 scalars for the
// src pass; everything else lives in registers.
template <int D>
__device__ __forceinline__ float dot_d(const float (&a)[D], const float* __restrict__ b, float (&bout)[D])
{
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < D; ++t) {
        bout[t] = b[t];
        s = fmaf(a[t], bout[t], s);
    }
    return s;
}

// Sum of v over the lh lanes of a head (lh a power of two, the lanes aligned at a multiple of lh; lh == G for a single head).
// Inside one DPP row (16 lanes) the steps are data-parallel-primitive adds — row_mirror, row_half_mirror, quad_perm [2,3,0,1],
// quad_perm [1,0,3,2]: each pairs every lane with one of the other half of its 16 / 8 / 4 / 2 lanes — behind wave-uniform
// branches instead of a loop: the loop form (ds_bpermute + s_waitcnt lgkmcnt(0) + branch per step, per edge) put an LDS round
// trip into every edge's dependent chain and cut the four unrolled edges of a batch into separate basic blocks.
template <int CTRL>
__device__ __forceinline__ float dpp_lane_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

template <int G>
__device__ __forceinline__ float head_sum(float v, int lh)
{
#if TFGX_GAT_BWD_DPP_SUM
    if constexpr (G > 16) {
        for (int o = lh >> 1; o >= 16; o >>= 1) v += __shfl_xor(v, o, G);     // steps that cross DPP rows (one wide head)
    }
    // (wave-uniform branches, not selects: the kernel is bound by vector-ALU issue, and masked steps that add 0 cost three
    //  instructions each — measured 10.45 against 10.10 ms for the shuffle loop at lh = 2)
    if constexpr (G >= 16) { if (lh >= 16) v += dpp_lane_f32<0x140>(v); }
    if constexpr (G >= 8) { if (lh >= 8) v += dpp_lane_f32<0x141>(v); }
    if (lh >= 4) v += dpp_lane_f32<0x4E>(v);
    if (lh >= 2) v += dpp_lane_f32<0xB1>(v);
    return v;
#else
    for (int o = lh >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
#endif
}

// the same sum for U values at once: the wave-uniform branches are taken once per BATCH of edges instead of once per edge
// (ISA of the source pass, round 6: four branch blocks per edge, every one a v_cndmask / v_cmp / s_andn2 / s_cbranch group —
// a quarter of the walk's instructions)
template <int G, int U>
__device__ __forceinline__ void head_sum_batch(float (&v)[U], int lh)
{
#if TFGX_GAT_BWD_DPP_SUM
    if constexpr (G > 16) {
        for (int o = lh >> 1; o >= 16; o >>= 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] += __shfl_xor(v[u], o, G);
        }
    }
    if constexpr (G >= 16) {
        if (lh >= 16) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] += dpp_lane_f32<0x140>(v[u]);
        }
    }
    if constexpr (G >= 8) {
        if (lh >= 8) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] += dpp_lane_f32<0x141>(v[u]);
        }
    }
    if (lh >= 4) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] += dpp_lane_f32<0x4E>(v[u]);
    }
    if (lh >= 2) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] += dpp_lane_f32<0xB1>(v[u]);
    }
#else
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = head_sum<G>(v[u], lh);
#endif
}

// One pass that prepares the GAT source pass's packed destination rows (see tfgx_gat_backward_args.ld_stats_ml):
//   pack[r] = [ dO[r, 0..W) | Q[r, 0..A) | (m, l)[r, 0..2H) | D[r, 0..H) ],  D[r, h] = <dO[r, h, :], O[r, h, :]>
// and the dense dsum[r, h] the destination pass reads.  Replaces a reduction and four strided copies.
__global__ __launch_bounds__(kBlock) void gat_pack_dst_kernel(const float* __restrict__ go, int64_t ldgo,
                                                              const float* __restrict__ o, int64_t ldo,
                                                              const float* __restrict__ q, int64_t ldq,
                                                              const float* __restrict__ ml, int64_t n, int H, int dv,
                                                              int A, float* __restrict__ pack, int64_t P,
                                                              float* __restrict__ dsum)
{
    const int W = H * dv;
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int per_row = H + (A + 2 * H + 3) / 4;       // H "head" work items + the small copies, per destination row
    for (; t < n * per_row; t += stride) {
        const int64_t r = t / per_row;
        const int k = int(t - r * per_row);
        float* pr = pack + r * P;
        if (k < H) {                                    // one head: copy dO and reduce <dO, O>
            const float* gp = go + r * ldgo + k * dv;
            const float* op = o + r * ldo + k * dv;
            float acc = 0.0f;
            for (int i = 0; i < dv; ++i) {
                const float gvv = gp[i];
                acc = fmaf(gvv, op[i], acc);
                pr[k * dv + i] = gvv;
            }
            dsum[r * H + k] = acc;
            pr[W + A + 2 * H + k] = acc;
        } else {                                        // four floats of [Q | (m, l)]
            const int c0 = (k - H) * 4;
            for (int c = c0; c < c0 + 4 && c < A + 2 * H; ++c)
                pr[W + c] = c < A ? q[r * ldq + c] : ml[r * 2 * H + (c - A)];
        }
    }
}

// Same result, one lane per FOUR value columns (16-byte loads / stores; the lanes of a head reduce <dO, O> with a
// butterfly) plus ceil((A + 2H) / 4) lanes per row for the [Q | (m, l)] copies.  Needs dv % 4 == 0, LH = dv / 4 a power of
// two that divides the lanes per row, 16-byte aligned dO / O / pack rows.
__global__ __launch_bounds__(kBlock) void gat_pack_dst_vec4_kernel(const float* __restrict__ go, int64_t ldgo,
                                                                   const float* __restrict__ o, int64_t ldo,
                                                                   const float* __restrict__ q, int64_t ldq,
                                                                   const float* __restrict__ ml, int64_t n, int H, int dv,
                                                                   int A, float* __restrict__ pack, int64_t P,
                                                                   float* __restrict__ dsum)
{
    const int W = H * dv, w4 = W / 4, lh = dv / 4;
    const int per_row = w4 + (A + 2 * H + 3) / 4;
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;      // a multiple of 64: lane groups stay aligned
    for (; t < n * per_row; t += stride) {
        const int64_t r = t / per_row;
        const int k = int(t - r * per_row);
        float* pr = pack + r * P;
        if (k < w4) {
            float g4[4], o4[4];
            load_vec<4>(go + r * ldgo + 4 * k, g4);
            load_vec<4>(o + r * ldo + 4 * k, o4);
            store_vec<4>(pr + 4 * k, g4);
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = fmaf(g4[i], o4[i], acc);
            for (int off = lh >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (k % lh == 0) {
                const int head = k / lh;
                dsum[r * H + head] = acc;
                pr[W + A + 2 * H + head] = acc;
            }
        } else {                                        // four floats of [Q | (m, l)]
            const int c0 = (k - w4) * 4;
            for (int c = c0; c < c0 + 4 && c < A + 2 * H; ++c)
                pr[W + c] = c < A ? q[r * ldq + c] : ml[r * 2 * H + (c - A)];
        }
    }
}

// The head-block form of the packed table (round 6): pack[r] = [ dO[r, 0..W) | pad to 4 floats | per head h: Q[r,h,0..d), m, 1/(l + 1e-8), D, pad ].
// The source pass of round 5 read, per edge and lane group, dO (one 16-byte load per lane) and Q / (m, l) / D as THREE more
// load instructions that all fall into the row's last 128-byte line: five line requests per edge where three lines are
// touched — and the pass ran at 143 G requests/s, the rate at which the L2s serve requests, not lines.  With a head's scalars
// contiguous, every lane of the head loads the same 16 (d = 1) ... 48 bytes with dwordx4 loads: three requests per edge.
// One thread per 4 floats of the row; D = <dO, O> is summed per head first (one lane per head, dv sequential terms: the pass is
// 0.1 ms of a 10 ms step).
__global__ __launch_bounds__(kBlock) void gat_pack_dst_heads_kernel(const float* __restrict__ go, int64_t ldgo,
                                                                    const float* __restrict__ o, int64_t ldo,
                                                                    const float* __restrict__ q, int64_t ldq,
                                                                    const float* __restrict__ ml, int64_t n, int H, int d, int dv,
                                                                    float* __restrict__ pack, int64_t P,
                                                                    float* __restrict__ dsum)
{
    const int W = H * dv, hb = head_block(d);
    const bool vec = (dv % 4 == 0) && (ldgo % 4 == 0) && (ldo % 4 == 0) && (P % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(go) | reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(pack)) & 15) == 0;
    const int per_row = (W + 3) / 4 + H;                 // copies of dO in fours, then one work item per head
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (; t < n * per_row; t += stride) {
        const int64_t r = t / per_row;
        const int k = int(t - r * per_row);
        float* pr = pack + r * P;
        if (k < (W + 3) / 4) {
            if (vec) {                                   // whole 16-byte pieces (W % 4 == 0, aligned rows)
                float t4[4];
                load_vec<4>(go + r * ldgo + 4 * k, t4);
                store_vec<4>(pr + 4 * k, t4);
            } else {
                for (int c = 4 * k; c < 4 * k + 4 && c < W; ++c) pr[c] = go[r * ldgo + c];
            }
        } else {
            const int h = k - (W + 3) / 4;
            const float* gp = go + r * ldgo + h * dv;
            const float* op = o + r * ldo + h * dv;
            float acc = 0.0f;
            if (vec) {                                   // same terms in the same order, 16-byte loads
                for (int i = 0; i < dv; i += 4) {
                    float g4[4], o4[4];
                    load_vec<4>(gp + i, g4);
                    load_vec<4>(op + i, o4);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc = fmaf(g4[t], o4[t], acc);
                }
            } else {
                for (int i = 0; i < dv; ++i) acc = fmaf(gp[i], op[i], acc);
            }
            dsum[r * H + h] = acc;
            float* hp = pr + (W + 3) / 4 * 4 + h * hb;        // the head blocks start on a 16-byte boundary of the row
            for (int u = 0; u < d; ++u) hp[u] = q[r * ldq + h * d + u];
            hp[d] = ml[r * 2 * H + 2 * h];
            hp[d + 1] = 1.0f / (ml[r * 2 * H + 2 * h + 1] + 1e-8f);
            hp[d + 2] = acc;
            for (int u = d + 3; u < hb; ++u) hp[u] = 0.0f;
        }
    }
}

// d == 1: dQ[r, h] = (<dO[r, h, :], T[r, h, :]> - D[r, h] S[r, h]) / scale from the sums the forward walk accumulated
// (gat_fused_kernel's QG mode) — one thread per (row, head), sequential dv-term dot as D itself is formed.
__global__ __launch_bounds__(kBlock) void gat_query_grad_d1_kernel(const float* __restrict__ go, int64_t ldgo,
                                                                   const float* __restrict__ t, int64_t ldt,
                                                                   const float* __restrict__ s, const float* __restrict__ dsum,
                                                                   int64_t n, int H, int dv, float scale,
                                                                   float* __restrict__ gq, int64_t ldgq)
{
    int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const bool vec = (dv % 4 == 0) && (ldgo % 4 == 0) && (ldt % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(go) | reinterpret_cast<uintptr_t>(t)) & 15) == 0;
    for (; i < n * H; i += stride) {
        const int64_t r = i / H;
        const int h = int(i - r * H);
        const float* gp = go + r * ldgo + h * dv;
        const float* tp = t + r * ldt + h * dv;
        float acc = 0.0f;
        if (vec) {
            for (int j = 0; j < dv; j += 4) {
                float g4[4], t4[4];
                load_vec<4>(gp + j, g4);
                load_vec<4>(tp + j, t4);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = fmaf(g4[u], t4[u], acc);
            }
        } else {
            for (int j = 0; j < dv; ++j) acc = fmaf(gp[j], tp[j], acc);
        }
        gq[r * ldgq + h] = fmaf(-dsum[i], s[i], acc) / scale;
    }
}

// POW2: a.inv_scale is the exact inverse of a power-of-two scale (instantiated for d = 1, 4, 16, where sqrt(d) is one)
// HP (src pass only): the per-head scalars of the gathered destination come from a.hp (head blocks, see above)
#ifndef TFGX_GAT_BWD_SRC_WAVES
#define TFGX_GAT_BWD_SRC_WAVES 1      // developer A/B: waves per SIMD the head-block source pass of narrow heads is compiled for
#endif
template <int G, int D, bool SRC, bool POW2 = false, bool HP = false>
__global__ __launch_bounds__(kBlock, (SRC && HP && D <= 4) ? TFGX_GAT_BWD_SRC_WAVES : 1) void gat_backward_fast_kernel(const GB a)
{
    constexpr int VEC = 4;
    constexpr int ROWS_PER_BLOCK = kBlock / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int W = a.H * a.dv;
    const int c_raw = (blockIdx.y * G + lane) * VEC;
    const bool cvalid = c_raw < W;
    const int coff = cvalid ? c_raw : (W - VEC);
    const int head = coff / a.dv;
    // lanes that reduce <dO, V> together: the lanes of one head (a power of two, aligned) — or, for a single head of any
    // width (dv / 4 lanes, e.g. 11 for the 44-wide output layer), the whole group with the idle lanes contributing zero
    const int lh = (a.H == 1) ? G : a.dv / VEC;
    const bool head_first = cvalid && (coff % a.dv == 0);

    for (int64_t pi = int64_t(blockIdx.x) * ROWS_PER_BLOCK + grp; pi < a.n; pi += int64_t(gridDim.x) * ROWS_PER_BLOCK) {
        const int64_t part = a.row_order ? int64_t(a.row_order[pi]) : pi;
        const int s0 = a.row_begin[part * a.rp_stride], e0 = a.row_end[part * a.rp_stride];
        if (a.skip > 0 && e0 - s0 > a.skip) continue;             // a hub row: walked chunk-wise by a second launch
        const int64_t row = a.part_row ? int64_t(a.part_row[part]) : part;
        // "mine": the row this group owns (destination r for the dst pass, source c for the src pass)
        float mine_qk[D];     // dst pass: Q[r,h,:]   src pass: K[c,h,:]
        float mine_v[VEC];    // dst pass: dO[r,cols] src pass: V[c,cols]
        float acc_qk[D];      // dst pass: dQ[r,h,:]  src pass: dK[c,h,:]
        float acc_v[VEC];     // src pass only: dV[c,cols]
        const float* own_qk = (SRC ? a.k + row * a.ldk : a.q + row * a.ldq) + head * a.d;
#pragma unroll
        for (int t = 0; t < D; ++t) { mine_qk[t] = own_qk[t]; acc_qk[t] = 0.0f; }
        load_vec<VEC>((SRC ? a.v + row * a.ldv : a.go + row * a.ldgo) + coff, mine_v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc_v[i] = 0.0f;
        float m_r = 0.0f, linv_r = 0.0f, d_r = 0.0f;
        if (!SRC) {
            m_r = a.ml[row * a.ldml + 2 * head];
            linv_r = 1.0f / (a.ml[row * a.ldml + 2 * head + 1] + 1e-8f);
            d_r = a.dsum[row * a.lddsum + head];
        }
        // an edge in two halves, so that U edges can have their loads in flight before the first one is consumed
        // (the one-edge-at-a-time loop left the source pass at 42 G lines/s; the forward's 8-deep walk sustains 55)
        struct EdgeIn {
            float qk[D];
            float v[VEC];
            float m, l, dd;
        };
        // element offsets of the gathered rows as ONE 32 x 32 -> 64-bit multiply each (v_mad_u64_u32): ids are non-negative int32
        // and the leading dimensions fit 32 bits (checked at the entry point); an int64 id times an int64 stride made the
        // compiler emulate a 64 x 64-bit product per address (see gat_fused_kernel)
        const uint32_t ld_qk32 = uint32_t(SRC ? a.ldq : a.ldk), ld_v32 = uint32_t(SRC ? a.ldgo : a.ldv);
        const uint32_t ldml32 = uint32_t(a.ldml), lddsum32 = uint32_t(a.lddsum);
        const uint32_t ldhp32 = uint32_t(a.ldhp);
        auto edge_load = [&](int o_, EdgeIn& in) {   // o: the other endpoint (source c / destination r)
            const uint64_t o = uint64_t(uint32_t(o_));
            if constexpr (SRC && HP) {
                // the head's block [Q | m | 1 / (l + 1e-8) | D | pad]: 16-byte loads, the same bytes for every lane of the head
                constexpr int HB = head_block(D);
                float hb[HB];
                const float* ph = a.hp + o * ldhp32 + head * HB;
#pragma unroll
                for (int t = 0; t < HB; t += 4) load_vec<4>(ph + t, *reinterpret_cast<float (*)[4]>(&hb[t]));
#pragma unroll
                for (int t = 0; t < D; ++t) in.qk[t] = hb[t];
                in.m = hb[D];
                in.l = hb[D + 1];          // already the reciprocal
                in.dd = hb[D + 2];
                load_vec<VEC>(a.go + o * ld_v32 + coff, in.v);
                return;
            }
            const float* pq = (SRC ? a.q : a.k) + o * ld_qk32 + head * a.d;
#pragma unroll
            for (int t = 0; t < D; ++t) in.qk[t] = pq[t];
            load_vec<VEC>((SRC ? a.go : a.v) + o * ld_v32 + coff, in.v);
            if (SRC) {
                in.m = a.ml[o * ldml32 + 2 * head];
                in.l = a.ml[o * ldml32 + 2 * head + 1];
                in.dd = a.dsum[o * lddsum32 + head];
            }
        };
        // pow2 (std::bool_constant<POW2>): the score and ds are divided by the scale with a multiply by its exact inverse — a
        // template parameter of the kernel, so that no branch sits between a batch's gathers and no second copy of the walk
        // costs registers
        // an edge's arithmetic in three steps, so that a batch can run the step in the middle — the sum of <dO, V> over the lanes
        // of a head, and the dropout mask — once for all its edges (head_sum_batch)
        auto edge_scores = [&](const EdgeIn& in, float& sc, float& part, auto pow2) {
            sc = 0.0f;
#pragma unroll
            for (int t = 0; t < D; ++t) sc = fmaf(mine_qk[t], in.qk[t], sc);
            if constexpr (decltype(pow2)::value) sc = sc * a.inv_scale;
            else sc = sc / a.scale;
            part = 0.0f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) part = fmaf(mine_v[i], in.v[i], part);   // <dO, V> over this lane's columns
            part = cvalid ? part : 0.0f;
        };
        auto edge_finish = [&](const EdgeIn& in, float sc, float da, float keep, auto pow2) {
            const float m = SRC ? in.m : m_r;
            // 1 / (l + 1e-8): the destination pass divides once per row; per EDGE (source pass) the correctly rounded division
            // is ten instructions of a walk bound by vector-ALU issue — v_rcp_f32 (1 ulp) there
            const float linv = SRC ? (HP ? in.l : __builtin_amdgcn_rcpf(in.l + 1e-8f)) : linv_r;
            const float dd = SRC ? in.dd : d_r;
            const float alpha = expf(sc - m) * linv;
            float ds = alpha * (keep * da - dd);
            if constexpr (decltype(pow2)::value) ds = ds * a.inv_scale;
            else ds = ds / a.scale;
#pragma unroll
            for (int t = 0; t < D; ++t) acc_qk[t] = fmaf(ds, in.qk[t], acc_qk[t]);
            if (SRC) {
                const float ak = alpha * keep;
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc_v[i] = fmaf(ak, in.v[i], acc_v[i]);
            }
        };
        auto edge = [&](int o, float keep, auto pow2) {
            EdgeIn in;
            edge_load(o, in);
            float sc, part;
            edge_scores(in, sc, part, pow2);
            edge_finish(in, sc, head_sum<G>(part, lh), keep, pow2);
        };
#ifndef TFGX_GAT_BWD_UNROLL_NARROW
#define TFGX_GAT_BWD_UNROLL_NARROW 4  // developer A/B
#endif
#ifndef TFGX_GAT_BWD_UNROLL_MID
#define TFGX_GAT_BWD_UNROLL_MID 2     // developer A/B: edges in flight per lane group for d_head 8 / 16
#endif
        constexpr int U = (D <= 4) ? TFGX_GAT_BWD_UNROLL_NARROW : (D <= 16 ? TFGX_GAT_BWD_UNROLL_MID : 1);
#ifndef TFGX_GAT_BWD_COL_AHEAD
#define TFGX_GAT_BWD_COL_AHEAD 1      // developer A/B: 0 = every batch loads its own neighbour ids right before its gathers
#endif
        constexpr std::bool_constant<POW2> pow2{};
        {
#if TFGX_GAT_BWD_COL_AHEAD
            // as in gat_fused_kernel: the neighbour ids of the NEXT batch are loaded before this batch's gathers are issued
            int oj_next = (s0 + lane < e0) ? a.other[s0 + lane] : 0;
#endif
            for (int base = s0; base < e0; base += G) {
                const int idx = base + lane;
#if TFGX_GAT_BWD_COL_AHEAD
                const int oj = oj_next;
                oj_next = (idx + G < e0) ? a.other[idx + G] : 0;
#else
                const int oj = (idx < e0) ? a.other[idx] : 0;
#endif
                const int pj = (a.drop.thr != 0u && a.pos && idx < e0) ? a.pos[idx] : idx;   // forward-CSR position
                const int cnt = min(G, e0 - base);
                int j = 0;
                for (; j + U <= cnt; j += U) {
                    EdgeIn in[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) edge_load(__shfl(oj, j + u, G), in[u]);
                    float sc[U], da[U], keep[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) edge_scores(in[u], sc[u], da[u], pow2);
                    head_sum_batch<G, U>(da, lh);
                    if (a.drop.thr != 0u) {        // wave-uniform: ONE branch per batch, nothing of the mask otherwise
#pragma unroll
                        for (int u = 0; u < U; ++u) keep[u] = drop_scale(a.drop, uint32_t(int64_t(__shfl(pj, j + u, G)) * a.H + head));
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) keep[u] = 1.0f;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) edge_finish(in[u], sc[u], da[u], keep[u], pow2);
                }
                for (; j < cnt; ++j)
                    edge(__shfl(oj, j, G), drop_scale(a.drop, uint32_t(int64_t(__shfl(pj, j, G)) * a.H + head)), pow2);
            }
            // the appended self-loop belongs to the row's LAST part (chunk launches: the chunk that ends where the row ends)
            if (a.add_self_loop && row < a.n_self && (a.part_row == nullptr || e0 == a.row_ptr[row + 1]))
                edge(int(row), drop_scale(a.drop, uint32_t((a.drop.self_base + row) * a.H + head)), pow2);
        }
        if (SRC) {
            if (cvalid) {
                float* gvp = a.gv + part * a.ldgv + coff;
#ifndef TFGX_GAT_BWD_ACC_NT
#define TFGX_GAT_BWD_ACC_NT 0         // developer A/B: 0 = the block-by-block gradient accumulation uses plain loads / stores
#endif
                if (a.accumulate) {                                 // blocks are applied in order: previous blocks + this one
                    float prev[VEC];
#if TFGX_GAT_BWD_ACC_NT
                    load_vec_nt<VEC>(gvp, prev);                    // streamed once per launch: kept out of the gathered rows' way
#else
                    load_vec<VEC>(gvp, prev);
#endif
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc_v[i] = prev[i] + acc_v[i];
                }
#if TFGX_GAT_BWD_ACC_NT
                if (a.rp_stride > 1) store_vec_nt<VEC>(gvp, acc_v);
                else store_vec<VEC>(gvp, acc_v);
#else
                store_vec<VEC>(gvp, acc_v);
#endif
            }
        }
        if (head_first) {
            float* dst = (SRC ? a.gk + part * a.ldgk : a.gq + part * a.ldgq) + head * a.d;
#pragma unroll
            for (int t = 0; t < D; ++t) dst[t] = a.accumulate ? dst[t] + acc_qk[t] : acc_qk[t];
        }
    }
}

template <int G, bool SRC>
int launch_gat_bwd_d(const GB& a, hipStream_t stream)
{
    constexpr int ROWS_PER_BLOCK = kBlock / G;
    const int W = a.H * a.dv;
    dim3 grid(grid_for(a.n, ROWS_PER_BLOCK, 1 << 20), (W + G * 4 - 1) / (G * 4), 1), block(kBlock, 1, 1);
    const bool pow2 = a.inv_scale != 0.0f;      // only d = 1, 4, 16 have the multiply instantiated; every other d divides
    if constexpr (SRC) {
        if (a.hp != nullptr) {      // head blocks (tfgx_gat_pack_dst_heads_f32): 16-byte aligned by the entry point's checks
            switch (a.d) {
#define TFGX_GAT_BWD_HP_CASE(D_)                                                                       \
    case D_:                                                                                         \
        if (pow2) gat_backward_fast_kernel<G, D_, true, true, true><<<grid, block, 0, stream>>>(a);   \
        else gat_backward_fast_kernel<G, D_, true, false, true><<<grid, block, 0, stream>>>(a);       \
        break
                TFGX_GAT_BWD_HP_CASE(1);
                case 2: gat_backward_fast_kernel<G, 2, true, false, true><<<grid, block, 0, stream>>>(a); break;
                TFGX_GAT_BWD_HP_CASE(4);
                case 8: gat_backward_fast_kernel<G, 8, true, false, true><<<grid, block, 0, stream>>>(a); break;
                TFGX_GAT_BWD_HP_CASE(16);
#undef TFGX_GAT_BWD_HP_CASE
                default: gat_backward_fast_kernel<G, 32, true, false, true><<<grid, block, 0, stream>>>(a); break;
            }
            TFGX_LAUNCH_CHECK("gat_backward_fast_kernel (head blocks)");
            return TFGX_OK;
        }
    }
    switch (a.d) {
#define TFGX_GAT_BWD_POW2_CASE(D_)                                                                   \
    case D_:                                                                                         \
        if (pow2) gat_backward_fast_kernel<G, D_, SRC, true><<<grid, block, 0, stream>>>(a);          \
        else gat_backward_fast_kernel<G, D_, SRC, false><<<grid, block, 0, stream>>>(a);              \
        break
        TFGX_GAT_BWD_POW2_CASE(1);
        case 2: gat_backward_fast_kernel<G, 2, SRC><<<grid, block, 0, stream>>>(a); break;
        TFGX_GAT_BWD_POW2_CASE(4);
        case 8: gat_backward_fast_kernel<G, 8, SRC><<<grid, block, 0, stream>>>(a); break;
        TFGX_GAT_BWD_POW2_CASE(16);
#undef TFGX_GAT_BWD_POW2_CASE
        default: gat_backward_fast_kernel<G, 32, SRC><<<grid, block, 0, stream>>>(a); break;
    }
    TFGX_LAUNCH_CHECK("gat_backward_fast_kernel");
    return TFGX_OK;
}

template <bool SRC>
int launch_gat_bwd(const GB& a, hipStream_t stream)
{
    const int lanes = (a.H * a.dv + 3) / 4;
    if (lanes <= 4) return launch_gat_bwd_d<4, SRC>(a, stream);
    if (lanes <= 8) return launch_gat_bwd_d<8, SRC>(a, stream);
    if (lanes <= 16) return launch_gat_bwd_d<16, SRC>(a, stream);
    if (lanes <= 32) return launch_gat_bwd_d<32, SRC>(a, stream);
    return launch_gat_bwd_d<64, SRC>(a, stream);
}

inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// layout conditions of the fast kernels (else the one-lane-per-(row, head) kernels above are used)
bool gat_bwd_fast_ok(const tfgx_gat_backward_args* p)
{
    const bool d_ok = p->d == 1 || p->d == 2 || p->d == 4 || p->d == 8 || p->d == 16 || p->d == 32;
    // one head of any width (dv / 4 <= 64 lanes: the group reduces as a whole), or several heads of a power-of-two width
    const bool v_ok = p->dv % 4 == 0 && p->dv / 4 <= 64 && (pow2(p->dv / 4) || p->H == 1);
    const bool al = p->ldv % 4 == 0 && p->ld_grad_out % 4 == 0 && aligned_to(p->v, 16) && aligned_to(p->grad_out, 16) &&
                    (p->grad_v == nullptr || (p->ld_grad_v % 4 == 0 && aligned_to(p->grad_v, 16)));
    return d_ok && v_ok && al;
}

}  // namespace
}  // namespace tfgx

using namespace tfgx;

static int sddmm_launch(const char* who, const int32_t* rb, const int32_t* re, const int32_t* part_row, int skip,
                        const int32_t* col, int64_t n_parts, const float* a, int64_t lda, const float* b, int64_t ldb,
                        int64_t F, float* out, const float* w, const float* mx, int64_t ldmx, hipStream_t s)
{
    const bool masked = mx != nullptr;
    const bool al = F % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && aligned_to(a, 16) && aligned_to(b, 16) &&
                    (!masked || (ldmx % 4 == 0 && aligned_to(mx, 16)));
    if (al && F >= 16 && F <= 512) {
#define TFGX_SDDMM_GO(G, CH)                                                                                          \
    do {                                                                                                              \
        if (masked)                                                                                                   \
            sddmm_fast_kernel<G, CH, true><<<grid_for(n_parts, kBlock / G, 1 << 20), kBlock, 0, s>>>(                 \
                rb, re, part_row, skip, col, n_parts, a, lda, b, ldb, int(F), out, w, mx, ldmx);                      \
        else                                                                                                          \
            sddmm_fast_kernel<G, CH, false><<<grid_for(n_parts, kBlock / G, 1 << 20), kBlock, 0, s>>>(                \
                rb, re, part_row, skip, col, n_parts, a, lda, b, ldb, int(F), out, nullptr, nullptr, 0);              \
    } while (0)
        if (F <= 32) TFGX_SDDMM_GO(8, 1);
        else if (F <= 64) TFGX_SDDMM_GO(16, 1);
        else if (F <= 128) TFGX_SDDMM_GO(32, 1);
        else if (F <= 256) TFGX_SDDMM_GO(32, 2);
        else TFGX_SDDMM_GO(32, 4);
#undef TFGX_SDDMM_GO
        TFGX_LAUNCH_CHECK(who);
        return TFGX_OK;
    }
    if (F <= 8) sddmm_kernel<8><<<grid_for(n_parts, kBlock / 8, 1 << 20), kBlock, 0, s>>>(rb, re, part_row, skip, col, n_parts, a, lda, b, ldb, int(F), out, w, mx, ldmx);
    else if (F <= 32) sddmm_kernel<16><<<grid_for(n_parts, kBlock / 16, 1 << 20), kBlock, 0, s>>>(rb, re, part_row, skip, col, n_parts, a, lda, b, ldb, int(F), out, w, mx, ldmx);
    else sddmm_kernel<32><<<grid_for(n_parts, kBlock / 32, 1 << 20), kBlock, 0, s>>>(rb, re, part_row, skip, col, n_parts, a, lda, b, ldb, int(F), out, w, mx, ldmx);
    TFGX_LAUNCH_CHECK(who);
    return TFGX_OK;
}

// every edge owns its output element, so hub rows need no scratch: the plan's rows up to the threshold, then the hub
// rows' chunks as rows of their own
static int sddmm_dispatch(const char* who, const int32_t* row_ptr, const int32_t* col, int64_t n_dst, const float* a,
                          int64_t lda, const float* b, int64_t ldb, int64_t F, float* out, const float* w,
                          const float* mx, int64_t ldmx, hipStream_t s, const tfgx_hub_lists* hub = nullptr)
{
    HubLists hl;
    const bool chunked = fill_hub(hub, hl);
    int rc = sddmm_launch(who, row_ptr, row_ptr + 1, nullptr, chunked ? hl.thr : 0, col, n_dst, a, lda, b, ldb, F, out, w, mx,
                          ldmx, s);
    if (rc != TFGX_OK || !chunked) return rc;
    return sddmm_launch(who, hl.chunk_begin, hl.chunk_end, hl.chunk_row, 0, col, hl.n_chunks, a, lda, b, ldb, F, out, w, mx,
                        ldmx, s);
}

extern "C" int tfgx_sddmm_f32(const int32_t* row_ptr, const int32_t* col, int64_t n_dst, const float* a, int64_t lda,
                              const float* b, int64_t ldb, int64_t F, float* out, tfgx_stream_t stream)
{
    return tfgx_sddmm_hub_f32(row_ptr, col, n_dst, a, lda, b, ldb, F, out, nullptr, stream);
}

extern "C" int tfgx_sddmm_hub_f32(const int32_t* row_ptr, const int32_t* col, int64_t n_dst, const float* a, int64_t lda,
                                  const float* b, int64_t ldb, int64_t F, float* out, const tfgx_hub_lists* hub,
                                  tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n_dst >= 0 && F >= 1 && lda >= F && ldb >= F, "bad size");
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr && a && b, "null pointer");
    return sddmm_dispatch("sddmm_kernel", row_ptr, col, n_dst, a, lda, b, ldb, F, out, nullptr, nullptr, 0,
                          as_stream(stream), hub);
}

extern "C" int tfgx_segment_max_backward_w_f32(const int32_t* row_ptr, const int32_t* col, const float* w, int64_t n_dst,
                                               const float* x, int64_t ldx, int64_t F, const float* out, int64_t ldo,
                                               const float* gn, int64_t ldgn, float* grad_w, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n_dst >= 0 && F >= 1 && ldx >= F && ldo >= F && ldgn >= F, "bad size");
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr && col && w && x && out && gn && grad_w, "null pointer");
    return sddmm_dispatch("sddmm_kernel<masked>", row_ptr, col, n_dst, gn, ldgn, x, ldx, F, grad_w, w, out, ldo,
                          as_stream(stream));
}

extern "C" int tfgx_segment_max_count_f32(const int32_t* row_ptr, const int32_t* col, const float* w, int64_t n_dst,
                                          const float* x, int64_t ldx, int64_t F, const float* out, int64_t ldo,
                                          float* count, int64_t ldc, tfgx_stream_t stream)
{
    return tfgx_segment_max_count_hub_f32(row_ptr, col, w, n_dst, x, ldx, F, out, ldo, count, ldc, nullptr, nullptr, stream);
}

extern "C" int tfgx_segment_max_count_hub_f32(const int32_t* row_ptr, const int32_t* col, const float* w, int64_t n_dst,
                                              const float* x, int64_t ldx, int64_t F, const float* out, int64_t ldo,
                                              float* count, int64_t ldc, const tfgx_hub_lists* hub, float* hub_scratch,
                                              tfgx_stream_t stream)
{
    TFGX_RANGE();
    HubLists hl;
    const bool chunked = fill_hub(hub, hl) && hub_scratch != nullptr;
    TFGX_REQUIRE(n_dst >= 0 && F >= 1 && ldx >= F && ldo >= F && ldc >= F, "bad size");
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr && x && out && count, "null pointer");
    if (F % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldc % 4 == 0 && aligned_to(x, 16) && aligned_to(out, 16) &&
        aligned_to(count, 16))
        return launch_max_grad<0>(row_ptr, col, w, n_dst, x, ldx, int(F), out, ldo, nullptr, 0, count, ldc,
                                  as_stream(stream), nullptr, nullptr, 0, chunked ? &hl : nullptr, hub_scratch);
    max_count_kernel<<<grid_for(n_dst * F, kBlock), kBlock, 0, as_stream(stream)>>>(row_ptr, col, w, n_dst, x, ldx,
                                                                                   int(F), out, ldo, count, ldc);
    TFGX_LAUNCH_CHECK("max_count_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_segment_max_with_count_f32(const int32_t* row_ptr, const int32_t* col, const float* w,
                                              int64_t n_dst, const float* x, int64_t ldx, int64_t F, float* out,
                                              int64_t ldo, float* count, int64_t ldc, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n_dst >= 0 && F >= 1 && ldx >= F && ldo >= F && ldc >= F, "bad size");
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr && x && out && count, "null pointer");
    if (F % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldc % 4 == 0 && aligned_to(x, 16) && aligned_to(out, 16) &&
        aligned_to(count, 16))
        return launch_max_grad<2>(row_ptr, col, w, n_dst, x, ldx, int(F), nullptr, ldo, nullptr, 0, count, ldc,
                                  as_stream(stream), out);
    // layouts the one-pass kernel does not cover: plain forward, then the count pass
    tfgx_reduce_args a;
    memset(&a, 0, sizeof(a));
    a.row_begin = row_ptr; a.row_end = row_ptr + 1; a.rp_stride = 1; a.col = col; a.w = w; a.n_dst = n_dst;
    a.x = x; a.ldx = ldx; a.F = F; a.out = out; a.ldo = ldo; a.op = TFGX_MAX; a.act = TFGX_ACT_NONE;
    const int rc = tfgx_segment_reduce_f32(&a, stream);
    if (rc != TFGX_OK) return rc;
    return tfgx_segment_max_count_f32(row_ptr, col, w, n_dst, x, ldx, F, out, ldo, count, ldc, stream);
}

extern "C" int tfgx_segment_max_with_arg_f32(const int32_t* row_ptr, const int32_t* col, const float* w, int64_t n_dst,
                                             const float* x, int64_t ldx, int64_t F, float* out, int64_t ldo,
                                             float* count, int64_t ldc, int32_t* argpos, int64_t lda,
                                             tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n_dst >= 0 && F >= 1 && ldx >= F && ldo >= F && ldc >= F && lda >= F, "bad size");
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr && x && out && count && argpos, "null pointer");
    TFGX_REQUIRE(F % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldc % 4 == 0 && aligned_to(x, 16) && aligned_to(out, 16) &&
                     aligned_to(count, 16),
                 "needs 16-byte aligned rows and F % 4 == 0 (use tfgx_segment_max_with_count_f32 otherwise)");
    return launch_max_grad<2>(row_ptr, col, w, n_dst, x, ldx, int(F), nullptr, ldo, nullptr, 0, count, ldc,
                              as_stream(stream), out, argpos, lda);
}

extern "C" int tfgx_segment_max_backward_push_f32(const int32_t* row_ptr, const int32_t* col, const float* w,
                                                  int64_t n_dst, int64_t n_src, const float* x, int64_t ldx, int64_t F,
                                                  const float* out, int64_t ldo, const float* g, int64_t ldg,
                                                  const float* count, int64_t ldc, const int32_t* argpos, int64_t lda,
                                                  float* gx, int64_t ldgx, tfgx_stream_t stream_)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n_dst >= 0 && n_src >= 0 && F >= 1 && ldx >= F && ldo >= F && ldg >= F && ldc >= F && lda >= F &&
                     ldgx >= F, "bad size");
    hipStream_t stream = as_stream(stream_);
    TFGX_REQUIRE(n_src == 0 || gx != nullptr, "gx is null");
    for (int64_t r = 0; ldgx != F && r < n_src; ++r) TFGX_HIP_CHECK(hipMemsetAsync(gx + r * ldgx, 0, sizeof(float) * F, stream));
    if (ldgx == F && n_src > 0) TFGX_HIP_CHECK(hipMemsetAsync(gx, 0, sizeof(float) * size_t(n_src) * size_t(F), stream));
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr && col && x && out && g && count && argpos, "null pointer");
    TFGX_REQUIRE(F % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldg % 4 == 0 && ldc % 4 == 0 && aligned_to(x, 16) &&
                     aligned_to(out, 16) && aligned_to(g, 16) && aligned_to(count, 16),
                 "needs 16-byte aligned rows and F % 4 == 0");
    const int lanes = int((F + 3) / 4);
#define TFGX_PUSH(GG)                                                                                               \
    {                                                                                                               \
        dim3 grid(grid_for(n_dst, kBlock / GG, 1 << 20), (lanes + GG - 1) / GG, 1);                                 \
        max_backward_push_kernel<GG><<<grid, kBlock, 0, stream>>>(row_ptr, col, w, n_dst, x, ldx, int(F), out, ldo, \
                                                                 g, ldg, count, ldc, argpos, lda, gx, ldgx);       \
    }
    if (lanes <= 8) TFGX_PUSH(8)
    else if (lanes <= 16) TFGX_PUSH(16)
    else if (lanes <= 32) TFGX_PUSH(32)
    else TFGX_PUSH(64)
#undef TFGX_PUSH
    TFGX_LAUNCH_CHECK("max_backward_push_kernel");
    return TFGX_OK;
}

// row stride of the gn table inside the mask workspace: never a power of two of 512 bytes or more (the apply pass gathers
// single elements of gn rows: the same column of every row would sit on the same few memory channels — plan.pow2_row_stride)
static inline int64_t mask_gn_ld(int64_t F) { return (F >= 128 && (F & (F - 1)) == 0) ? F + 32 : F; }

extern "C" size_t tfgx_segment_max_backward_mask_workspace_bytes(int64_t n_dst, int64_t E, int64_t F)
{
    if (n_dst < 0 || E < 0 || F < 1) return 0;
    const size_t gn = sizeof(float) * size_t(n_dst) * size_t(mask_gn_ld(F));
    const size_t mk = sizeof(uint32_t) * size_t(E) * size_t((F + 31) / 32);
    return (gn + 255) / 256 * 256 + mk + 256;
}

extern "C" int tfgx_segment_max_backward_mask_f32(const int32_t* row_ptr, const int32_t* col, const float* w,
                                                  int64_t n_dst, int64_t E, const float* x, int64_t ldx, int64_t F,
                                                  const float* out, int64_t ldo, const float* g, int64_t ldg,
                                                  const float* count, int64_t ldc, const int32_t* argpos, int64_t lda,
                                                  const int32_t* row_ptr_t, const int32_t* dst_t, const float* w_t,
                                                  const int32_t* pos_t, int64_t n_src, float* gx, int64_t ldgx,
                                                  void* workspace, size_t workspace_bytes, tfgx_stream_t stream_)
{
    return tfgx_segment_max_backward_mask_phases_f32(row_ptr, col, w, n_dst, E, x, ldx, F, out, ldo, g, ldg, count, ldc, argpos,
                                                     lda, row_ptr_t, dst_t, w_t, pos_t, n_src, gx, ldgx, workspace,
                                                     workspace_bytes, 3, stream_);
}

extern "C" int tfgx_segment_max_backward_mask_phases_f32(const int32_t* row_ptr, const int32_t* col, const float* w,
                                                         int64_t n_dst, int64_t E, const float* x, int64_t ldx, int64_t F,
                                                         const float* out, int64_t ldo, const float* g, int64_t ldg,
                                                         const float* count, int64_t ldc, const int32_t* argpos, int64_t lda,
                                                         const int32_t* row_ptr_t, const int32_t* dst_t, const float* w_t,
                                                         const int32_t* pos_t, int64_t n_src, float* gx, int64_t ldgx,
                                                         void* workspace, size_t workspace_bytes, int32_t phases,
                                                         tfgx_stream_t stream_)
{
    TFGX_RANGE();
    TFGX_REQUIRE(phases >= 1 && phases <= 3, "phases: bit 0 = build the masks, bit 1 = apply them to a window of source rows");
    const bool do_build = (phases & 1) != 0, do_apply = (phases & 2) != 0;
    // count == NULL: `argpos` holds the PACKED uint32 of tfgx_reduce_args.track (tie count << 16 | row-relative position)
    if (count == nullptr) ldc = F;
    TFGX_REQUIRE(n_dst >= 0 && n_src >= 0 && E >= 0 && F >= 1 && ldx >= F && ldo >= F && ldg >= F && ldc >= F &&
                     lda >= F && ldgx >= F, "bad size");
    if (n_src == 0 && !do_build) return TFGX_OK;
    TFGX_REQUIRE(!do_apply || n_src == 0 || (row_ptr_t && gx), "null pointer");
    TFGX_REQUIRE(!do_build || n_dst == 0 || (row_ptr && x && out && g && argpos), "null pointer");
    TFGX_REQUIRE(E == 0 || ((!do_build || col) && (!do_apply || dst_t)), "null pointer");
    TFGX_REQUIRE((w == nullptr) == (w_t == nullptr), "w and w_t go together");
    TFGX_REQUIRE(F % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldg % 4 == 0 && ldc % 4 == 0 && ldgx % 4 == 0 &&
                     aligned_to(x, 16) && aligned_to(out, 16) && aligned_to(g, 16) && aligned_to(count, 16) &&
                     aligned_to(gx, 16),
                 "needs 16-byte aligned rows and F % 4 == 0");
    TFGX_REQUIRE(workspace != nullptr &&
                     workspace_bytes >= tfgx_segment_max_backward_mask_workspace_bytes(n_dst, E, F),
                 "workspace too small (tfgx_segment_max_backward_mask_workspace_bytes)");
    hipStream_t stream = as_stream(stream_);
    const int MW = int((F + 31) / 32);
    float* gn = static_cast<float*>(workspace);
    const int64_t ldgn = mask_gn_ld(F);
    const size_t gn_bytes = (sizeof(float) * size_t(n_dst) * size_t(ldgn) + 255) / 256 * 256;
    uint32_t* mask = reinterpret_cast<uint32_t*>(static_cast<char*>(workspace) + gn_bytes);
    const int lanes = int((F + 3) / 4);
#define TFGX_MASK(GG)                                                                                                  \
    {                                                                                                                  \
        const int ny = (lanes + GG - 1) / GG;                                                                          \
        if (do_build && n_dst > 0) {                                                                                   \
            dim3 ga(grid_for(n_dst, kBlock / GG, 1 << 20), ny, 1);                                                     \
            max_mask_build_kernel<GG><<<ga, kBlock, 0, stream>>>(row_ptr, col, w, n_dst, x, ldx, int(F), out, ldo, g,  \
                                                                 ldg, count, ldc, argpos, lda, gn, ldgn, mask, MW);   \
        }                                                                                                              \
        if (do_apply && n_src > 0) {                                                                                   \
            dim3 gb(grid_for(n_src, kBlock / GG, 1 << 20), ny, 1);                                                     \
            max_backward_mask_apply_kernel<GG><<<gb, kBlock, 0, stream>>>(row_ptr_t, dst_t, w_t, pos_t, n_src, int(F), \
                                                                          gn, ldgn, mask, MW, gx, ldgx);              \
        }                                                                                                              \
    }
    if (lanes <= 8) TFGX_MASK(8)
    else if (lanes <= 16) TFGX_MASK(16)
    else if (lanes <= 32) TFGX_MASK(32)
    else TFGX_MASK(64)
#undef TFGX_MASK
    TFGX_LAUNCH_CHECK("max_backward_mask kernels");
    return TFGX_OK;
}

extern "C" int tfgx_segment_max_backward_f32(const int32_t* row_ptr_t, const int32_t* dst_t, const float* w_t,
                                             int64_t n_src, const float* x, int64_t ldx, int64_t F, const float* out,
                                             int64_t ldo, const float* g, int64_t ldg, const float* count, int64_t ldc,
                                             float* gx, int64_t ldgx, int64_t n_dst, float* gn_scratch,
                                             tfgx_stream_t stream)
{
    return tfgx_segment_max_backward_hub_f32(row_ptr_t, dst_t, w_t, n_src, x, ldx, F, out, ldo, g, ldg, count, ldc, gx, ldgx,
                                             n_dst, gn_scratch, nullptr, nullptr, stream);
}

extern "C" int tfgx_segment_max_backward_hub_f32(const int32_t* row_ptr_t, const int32_t* dst_t, const float* w_t,
                                                 int64_t n_src, const float* x, int64_t ldx, int64_t F, const float* out,
                                                 int64_t ldo, const float* g, int64_t ldg, const float* count,
                                                 int64_t ldc, float* gx, int64_t ldgx, int64_t n_dst, float* gn_scratch,
                                                 const tfgx_hub_lists* hub_t, float* hub_scratch, tfgx_stream_t stream)
{
    TFGX_RANGE();
    HubLists hl;
    const bool chunked = fill_hub(hub_t, hl) && hub_scratch != nullptr;
    TFGX_REQUIRE(n_src >= 0 && F >= 1 && ldx >= F && ldo >= F && ldg >= F && ldc >= F && ldgx >= F, "bad size");
    if (n_src == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr_t && x && out && g && count && gx, "null pointer");
    if (gn_scratch != nullptr && F % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldgx % 4 == 0 && aligned_to(x, 16) &&
        aligned_to(out, 16) && aligned_to(gx, 16) && aligned_to(gn_scratch, 16)) {
        // gn = g / count once per destination row, then one pass over the transposed plan with two row gathers per edge
        divide_kernel<<<grid_for(n_dst * F, kBlock), kBlock, 0, as_stream(stream)>>>(g, ldg, count, ldc, n_dst, int(F),
                                                                                   gn_scratch);
        TFGX_LAUNCH_CHECK("divide_kernel");
        return launch_max_grad<1>(row_ptr_t, dst_t, w_t, n_src, x, ldx, int(F), out, ldo, gn_scratch, F, gx, ldgx,
                                  as_stream(stream), nullptr, nullptr, 0, chunked ? &hl : nullptr, hub_scratch);
    }
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
    a.ldml = p->ld_stats_ml > 0 ? p->ld_stats_ml : 2 * int64_t(p->H);
    a.lddsum = p->ld_dsum > 0 ? p->ld_dsum : int64_t(p->H);
    TFGX_REQUIRE(a.ldml >= 2 * p->H && a.lddsum >= p->H, "ld_stats_ml / ld_dsum too small");
    {
        const int64_t lim = int64_t(1) << 31;     // the kernels form row offsets from 32-bit strides
        TFGX_REQUIRE(a.ldq < lim && a.ldk < lim && a.ldv < lim && a.ldgo < lim && a.ldml < lim && a.lddsum < lim,
                     "leading dimensions must be below 2^31 elements");
    }
    a.H = p->H; a.d = p->d; a.dv = p->dv; a.add_self_loop = p->add_self_loop; a.scale = p->scale;
    {   // a power-of-two scale (sqrt(d) for d = 1, 4, 16, 64) divides exactly by a multiply (see tfgx_attn.hip)
        int e2 = 0;
        const float inv = 1.0f / p->scale;
        a.inv_scale = (TFGX_GAT_BWD_POW2_SCALE && std::frexp(p->scale, &e2) == 0.5f && inv >= FLT_MIN && inv <= FLT_MAX) ? inv : 0.0f;
    }
    a.gq = p->grad_q; a.ldgq = p->ld_grad_q; a.gk = p->grad_k; a.ldgk = p->ld_grad_k;
    a.gv = p->grad_v; a.ldgv = p->ld_grad_v;
    TFGX_REQUIRE(p->drop_rate >= 0.0f && p->drop_rate < 1.0f, "drop_rate outside [0, 1)");
    a.drop = make_drop(p->drop_rate, p->drop_seed, p->drop_self_base, p->drop_seed_dev);
    a.pos = nullptr;
    a.hp = nullptr; a.ldhp = 0;
    if (p->head_pack != nullptr) {
        const int64_t need = int64_t(p->H) * head_block(p->d);
        TFGX_REQUIRE(p->ld_head_pack >= need && p->ld_head_pack % 4 == 0 && aligned_to(p->head_pack, 16) &&
                     p->ld_head_pack < (int64_t(1) << 31),
                     "head_pack: rows of H * roundup4(d + 3) floats, 16-byte aligned, row stride a multiple of 4 below 2^31");
        a.hp = p->head_pack; a.ldhp = p->ld_head_pack;
    }
    return TFGX_OK;
}

extern "C" int tfgx_gat_query_grad_d1_f32(const float* grad_out, int64_t ld_grad_out, const float* qgrad_t, int64_t ld_t,
                                          const float* qgrad_s, const float* dsum, int64_t n_dst, int32_t H, int32_t dv,
                                          float scale, float* grad_q, int64_t ld_grad_q, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n_dst >= 0 && H >= 1 && dv >= 1 && scale > 0.0f, "bad n_dst / H / dv / scale");
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(grad_out && qgrad_t && qgrad_s && dsum && grad_q, "null pointer");
    TFGX_REQUIRE(ld_grad_out >= int64_t(H) * dv && ld_t >= int64_t(H) * dv && ld_grad_q >= H, "leading dimension too small");
    gat_query_grad_d1_kernel<<<grid_for(n_dst * H, kBlock), kBlock, 0, as_stream(stream)>>>(
        grad_out, ld_grad_out, qgrad_t, ld_t, qgrad_s, dsum, n_dst, H, dv, scale, grad_q, ld_grad_q);
    TFGX_LAUNCH_CHECK("gat_query_grad_d1_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_gat_pack_dst_heads_f32(const float* grad_out, int64_t ld_grad_out, const float* out, int64_t ldo,
                                           const float* q, int64_t ldq, const float* stats_ml, int64_t n_dst, int32_t H,
                                           int32_t d, int32_t dv, float* pack, int64_t ld_pack, float* dsum,
                                           tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n_dst >= 0 && H >= 1 && d >= 1 && dv >= 1, "bad size");
    const int64_t W = int64_t(H) * dv, A = int64_t(H) * d;
    TFGX_REQUIRE(ld_grad_out >= W && ldo >= W && ldq >= A && ld_pack >= (W + 3) / 4 * 4 + int64_t(H) * head_block(d),
                 "leading dimension too small (ld_pack >= roundup4(H * dv) + H * roundup4(d + 3))");
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(grad_out && out && q && stats_ml && pack && dsum, "null pointer");
    const int per_row = int((W + 3) / 4) + H;
    gat_pack_dst_heads_kernel<<<grid_for(n_dst * per_row, kBlock), kBlock, 0, as_stream(stream)>>>(
        grad_out, ld_grad_out, out, ldo, q, ldq, stats_ml, n_dst, H, d, dv, pack, ld_pack, dsum);
    TFGX_LAUNCH_CHECK("gat_pack_dst_heads_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_gat_pack_dst_f32(const float* grad_out, int64_t ld_grad_out, const float* out, int64_t ldo,
                                     const float* q, int64_t ldq, const float* stats_ml, int64_t n_dst, int32_t H,
                                     int32_t d, int32_t dv, float* pack, int64_t ld_pack, float* dsum,
                                     tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n_dst >= 0 && H >= 1 && d >= 1 && dv >= 1, "bad size");
    const int64_t W = int64_t(H) * dv, A = int64_t(H) * d;
    TFGX_REQUIRE(ld_grad_out >= W && ldo >= W && ldq >= A && ld_pack >= W + A + 3 * int64_t(H), "leading dimension too small");
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(grad_out && out && q && stats_ml && pack && dsum, "null pointer");
    const int lh = dv / 4;
    const int per_row4 = int(W / 4 + (A + 2 * H + 3) / 4);
    const bool vec4 = dv % 4 == 0 && pow2(lh) && lh <= 64 && per_row4 % lh == 0 && ld_grad_out % 4 == 0 && ldo % 4 == 0 &&
                      ld_pack % 4 == 0 && aligned_to(grad_out, 16) && aligned_to(out, 16) && aligned_to(pack, 16);
    if (vec4) {
        gat_pack_dst_vec4_kernel<<<grid_for(n_dst * per_row4, kBlock), kBlock, 0, as_stream(stream)>>>(
            grad_out, ld_grad_out, out, ldo, q, ldq, stats_ml, n_dst, H, dv, int(A), pack, ld_pack, dsum);
        TFGX_LAUNCH_CHECK("gat_pack_dst_vec4_kernel");
        return TFGX_OK;
    }
    const int per_row = H + int((A + 2 * H + 3) / 4);
    gat_pack_dst_kernel<<<grid_for(n_dst * per_row, kBlock), kBlock, 0, as_stream(stream)>>>(
        grad_out, ld_grad_out, out, ldo, q, ldq, stats_ml, n_dst, H, dv, int(A), pack, ld_pack, dsum);
    TFGX_LAUNCH_CHECK("gat_pack_dst_kernel");
    return TFGX_OK;
}

// one pass of the GAT backward over a plan's rows; with hub lists (fast kernels only): rows up to the threshold, then the
// hub rows' chunks into the scratch ([n_chunks, H*d] for the dst pass, [n_chunks, H*d + H*dv] for the src pass), then
// the ordered sum of every hub row's chunk partials
template <bool SRC>
static int gat_backward_pass(const tfgx_gat_backward_args* p, GB& a, const tfgx_hub_lists* hub, float* scratch,
                             hipStream_t stream)
{
    a.row_begin = a.row_ptr; a.row_end = a.row_ptr + 1; a.part_row = nullptr; a.skip = 0; a.rp_stride = 1; a.accumulate = 0;
    a.row_order = SRC ? p->row_order_t : p->row_order;
    if (p->span_begin != nullptr) {
        // one block of a source-blocked pass (tfgx.h): explicit spans, gradients accumulated over the blocks in launch order
        TFGX_REQUIRE(p->span_end != nullptr && p->span_stride >= 1, "span_begin needs span_end and span_stride >= 1");
        TFGX_REQUIRE(gat_bwd_fast_ok(p) && p->drop_rate == 0.0f && hub == nullptr,
                     "span passes: fast-kernel head geometry only, no attention dropout, no hub lists");
        a.row_begin = p->span_begin; a.row_end = p->span_end; a.rp_stride = p->span_stride; a.accumulate = p->accumulate ? 1 : 0;
        a.row_order = nullptr;
        return launch_gat_bwd<SRC>(a, stream);
    }
    if (!gat_bwd_fast_ok(p)) {
        if (SRC) gat_backward_src_kernel<<<grid_for(a.n * a.H, kBlock), kBlock, 0, stream>>>(a);
        else gat_backward_dst_kernel<<<grid_for(a.n * a.H, kBlock), kBlock, 0, stream>>>(a);
        TFGX_LAUNCH_CHECK("gat_backward_{dst,src}_kernel");
        return TFGX_OK;
    }
    HubLists hl;
    const bool chunked = fill_hub(hub, hl) && scratch != nullptr;
    if (chunked) a.skip = hl.thr;
    int rc = launch_gat_bwd<SRC>(a, stream);
    if (rc != TFGX_OK || !chunked) return rc;
    const int A = a.H * a.d, W = a.H * a.dv;
    GB c = a;
    c.n = hl.n_chunks; c.row_begin = hl.chunk_begin; c.row_end = hl.chunk_end; c.part_row = hl.chunk_row; c.skip = 0;
    c.row_order = nullptr;
    float* sq = scratch;                                   // [n_chunks, A]: dQ (dst pass) / dK (src pass) partials
    float* sv = scratch + size_t(hl.n_chunks) * size_t(A); // [n_chunks, W]: dV partials (src pass)
    if (SRC) { c.gk = sq; c.ldgk = A; c.gv = sv; c.ldgv = W; }
    else { c.gq = sq; c.ldgq = A; }
    rc = launch_gat_bwd<SRC>(c, stream);
    if (rc != TFGX_OK) return rc;
    hub_sum_finalize_kernel<<<grid_for(hl.n_rows * A, kBlock), kBlock, 0, stream>>>(sq, A, hl.chunk_ptr, hl.rows, hl.n_rows, A,
                                                                                  SRC ? a.gk : a.gq, SRC ? a.ldgk : a.ldgq);
    TFGX_LAUNCH_CHECK("hub_sum_finalize_kernel");
    if (SRC) {
        hub_sum_finalize_kernel<<<grid_for(hl.n_rows * W, kBlock), kBlock, 0, stream>>>(sv, W, hl.chunk_ptr, hl.rows, hl.n_rows,
                                                                                      W, a.gv, a.ldgv);
        TFGX_LAUNCH_CHECK("hub_sum_finalize_kernel");
    }
    return TFGX_OK;
}

extern "C" int tfgx_gat_backward_dst_f32(const tfgx_gat_backward_args* p, tfgx_stream_t stream)
{
    return tfgx_gat_backward_dst_hub_f32(p, nullptr, nullptr, stream);
}

extern "C" int tfgx_gat_backward_dst_hub_f32(const tfgx_gat_backward_args* p, const tfgx_hub_lists* hub, float* hub_scratch,
                                             tfgx_stream_t stream)
{
    TFGX_RANGE();
    GB a;
    int rc = fill_gb(p, a);
    if (rc) return rc;
    TFGX_REQUIRE(p->row_ptr && p->grad_q && p->n_dst >= 0, "dst pass needs row_ptr / grad_q");
    if (p->n_dst == 0) return TFGX_OK;
    a.row_ptr = p->row_ptr; a.other = p->col; a.n = p->n_dst; a.n_self = p->n_dst;
    return gat_backward_pass<false>(p, a, hub, hub_scratch, as_stream(stream));
}

extern "C" int tfgx_gat_backward_src_f32(const tfgx_gat_backward_args* p, tfgx_stream_t stream)
{
    return tfgx_gat_backward_src_hub_f32(p, nullptr, nullptr, stream);
}

extern "C" int tfgx_gat_backward_src_hub_f32(const tfgx_gat_backward_args* p, const tfgx_hub_lists* hub_t, float* hub_scratch,
                                             tfgx_stream_t stream)
{
    TFGX_RANGE();
    GB a;
    int rc = fill_gb(p, a);
    if (rc) return rc;
    TFGX_REQUIRE(p->row_ptr_t && p->grad_k && p->grad_v && p->n_src >= 0, "src pass needs row_ptr_t / grad_k / grad_v");
    if (p->n_src == 0) return TFGX_OK;
    a.row_ptr = p->row_ptr_t; a.other = p->dst_t; a.n = p->n_src;
    a.n_self = p->n_src < p->n_dst ? p->n_src : p->n_dst;   // source c has a self-loop only if it is destination c too
    TFGX_REQUIRE(p->drop_rate == 0.0f || p->edge_pos_t, "the src pass needs edge_pos_t to regenerate the dropout mask");
    a.pos = p->drop_rate > 0.0f ? p->edge_pos_t : nullptr;
    return gat_backward_pass<true>(p, a, hub_t, hub_scratch, as_stream(stream));
}
