// Row utilities and destination-range sharding helpers.
//   tfgx_l2_normalize_rows_f32 : tf.nn.l2_normalize(h, axis=-1) (tf_geometric/nn/conv/graph_sage.py:58)
//   tfgx_gather_rows_f32       : send-side pack of the halo all-to-all-v (no counterpart in the reference,
//                                which replicates the whole graph per GPU: demo/demo_distributed_gcn.py:38-57)
//   tfgx_halo_* / tfgx_split_by_source_class : per-rank plan for a destination-range shard
#include "tfgx_common.h"
#include <hipcub/hipcub.hpp>

namespace tfgx {
namespace {

__global__ __launch_bounds__(kBlock) void l2_normalize_kernel(float* __restrict__ h, int64_t ld, int64_t n, int F)
{
    const int lane = threadIdx.x & 63;
    int64_t r = (blockIdx.x * int64_t(kBlock) + threadIdx.x) >> 6;
    const int64_t stride = (int64_t(gridDim.x) * kBlock) >> 6;
    for (; r < n; r += stride) {
        float* row = h + r * ld;
        float ss = 0.0f;
        for (int j = lane; j < F; j += 64) ss = fmaf(row[j], row[j], ss);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        for (int j = lane; j < F; j += 64) row[j] *= inv;
    }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                             const int32_t* __restrict__ idx, int64_t M, int F,
                                                             float* __restrict__ out, int64_t ldo)
{
    const int per_row = F / VEC;
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t total = M * per_row;
    for (; t < total; t += stride) {
        const int64_t i = t / per_row;
        const int j = int(t - i * per_row) * VEC;
        const float* src = x + (idx ? int64_t(idx[i]) : i) * ldx + j;      // idx == NULL: rows 0 .. M-1 (strided copy)
        float* dst = out + i * ldo + j;
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
        else *dst = *src;
    }
}

// dst[idx[i], :] += src[i, :] — the owner-side accumulate of the REVERSE halo exchange (gradients of halo rows coming
// home).  The ids of one call must be unique (one peer's request list is): then every destination element has exactly
// one writer and the result is deterministic without atomics; the caller applies the peers in a fixed order.
template <int VEC>
__global__ __launch_bounds__(kBlock) void scatter_add_rows_kernel(float* __restrict__ dst, int64_t ldd,
                                                                  const int32_t* __restrict__ idx, int64_t M, int F,
                                                                  const float* __restrict__ src, int64_t lds)
{
    const int per_row = F / VEC;
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t total = M * per_row;
    for (; t < total; t += stride) {
        const int64_t i = t / per_row;
        const int j = int(t - i * per_row) * VEC;
        float* d = dst + (idx ? int64_t(idx[i]) : i) * ldd + j;      // idx == NULL: rows 0 .. M-1 (a contiguous block)
        const float* sp = src + i * lds + j;
        if constexpr (VEC == 4) {
            float4 a = *reinterpret_cast<float4*>(d);
            const float4 b = *reinterpret_cast<const float4*>(sp);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            *reinterpret_cast<float4*>(d) = a;
        } else {
            *d += *sp;
        }
    }
}

__global__ void halo_mark_kernel(const int32_t* __restrict__ col, int64_t E, int32_t lo, int32_t hi,
                                 int32_t* __restrict__ flags)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; i < E; i += stride) {
        const int32_t c = col[i];
        if (c < lo || c >= hi) flags[c] = 1;   // benign race: every writer stores the same value
    }
}

__global__ void halo_scatter_kernel(const int32_t* __restrict__ flags, const int32_t* __restrict__ pos, int64_t n,
                                    int32_t* __restrict__ halo_ids, int32_t* __restrict__ n_halo)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) {
        if (flags[i]) halo_ids[pos[i]] = static_cast<int32_t>(i);
        if (i == n - 1) *n_halo = pos[i] + (flags[i] ? 1 : 0);
    }
}

__global__ void halo_remap_kernel(const int32_t* __restrict__ col, int64_t E, int32_t lo, int32_t hi,
                                  const int32_t* __restrict__ pos, int32_t n_own, int32_t* __restrict__ col_local)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; i < E; i += stride) {
        const int32_t c = col[i];
        col_local[i] = (c >= lo && c < hi) ? (c - lo) : (n_own + pos[c]);
    }
}

// stable per-row partition into n_class source classes: class of a source c = first k with c < bounds[k]
// (class 0 = own rows, classes 1.. = exchange rounds of the halo); one thread per destination row (plan time only)
constexpr int kMaxClasses = 17;
__global__ void split_by_class_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_local,
                                      const float* __restrict__ w, int64_t n_dst, const int32_t* __restrict__ bounds,
                                      int n_class, int32_t* __restrict__ rpk, int32_t* __restrict__ col_out,
                                      float* __restrict__ w_out)
{
    int64_t r = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; r < n_dst; r += stride) {
        const int s = row_ptr[r], e = row_ptr[r + 1];
        int cnt[kMaxClasses];
        for (int k = 0; k < n_class; ++k) cnt[k] = 0;
        for (int i = s; i < e; ++i) {
            const int32_t c = col_local[i];
            int k = 0;
            while (k < n_class - 1 && c >= bounds[k]) ++k;
            cnt[k]++;
        }
        int pos[kMaxClasses];
        int acc = s;
        for (int k = 0; k < n_class; ++k) {
            rpk[r * n_class + k] = acc;
            pos[k] = acc;
            acc += cnt[k];
        }
        for (int i = s; i < e; ++i) {
            const int32_t c = col_local[i];
            int k = 0;
            while (k < n_class - 1 && c >= bounds[k]) ++k;
            const int dst = pos[k]++;
            col_out[dst] = c;
            if (w_out) w_out[dst] = w[i];
        }
        if (r == n_dst - 1) rpk[n_dst * n_class] = e;
    }
}

// ---- neighbour sampling (RandomNeighborSampler.sample, tf_geometric/utils/graph_utils.py:667-772) ----------
// counter-based generator: a 64-bit mix of (seed, row, draw) — reproducible, order-independent, no state
__device__ __forceinline__ uint32_t draw_u32(uint64_t seed, uint64_t row, uint32_t i)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (row * 0x100000001B3ull + i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return static_cast<uint32_t>((z ^ (z >> 31)) >> 32);
}
__device__ __forceinline__ int draw_below(uint64_t seed, uint64_t row, uint32_t i, int n)   // uniform in [0, n)
{
    return static_cast<int>((uint64_t(draw_u32(seed, row, i)) * uint64_t(n)) >> 32);
}

constexpr int kSampleLongRow = 64;       // rows with more neighbours / draws than this get a whole wave (coalesced walks)

constexpr int kFloydMax = 32;            // Floyd's O(m^2) duplicate check only for few draws; otherwise one O(d) selection pass

__device__ __forceinline__ bool sample_row_is_long(int d, int m)
{
    return (d > kSampleLongRow && m > kFloydMax) || m > kSampleLongRow;
}

// one thread per destination row; out row r holds cnt[r] = out_ptr[r+1]-out_ptr[r] sampled CSR positions' (col, w)
__global__ void sample_neighbors_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                        const float* __restrict__ w, int64_t n_dst,
                                        const int32_t* __restrict__ out_ptr, int replace_when_short, uint64_t seed,
                                        int32_t* __restrict__ out_col, float* __restrict__ out_w)
{
    int64_t r = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; r < n_dst; r += stride) {
        const int s = row_ptr[r], d = row_ptr[r + 1] - s;
        const int o = out_ptr[r], m = out_ptr[r + 1] - o;
        if (m == 0) continue;
        if (sample_row_is_long(d, m)) continue;                   // hub rows: sample_neighbors_long_kernel, a wave each
        if (m >= d && !(replace_when_short && m > d)) {          // keep every neighbour, in order (:742-744)
            for (int i = 0; i < d; ++i) {
                out_col[o + i] = col[s + i];
                if (out_w) out_w[o + i] = w ? w[s + i] : 1.0f;
            }
            continue;
        }
        if (m > d) {                                             // padding: m draws WITH replacement (:749)
            for (int i = 0; i < m; ++i) {
                const int t = draw_below(seed, uint64_t(r), uint32_t(i), d);
                out_col[o + i] = col[s + t];
                if (out_w) out_w[o + i] = w ? w[s + t] : 1.0f;
            }
            continue;
        }
        if (m > kFloydMax) {
            // m < d, more than a few draws (ratio sampling): Knuth's selection sampling (Algorithm S) — one pass over
            // the d neighbours, position j is taken with probability (m - taken) / (d - j): uniform without replacement,
            // no scratch, neighbour order kept
            int taken = 0;
            for (int j = 0; j < d && taken < m; ++j) {
                if (draw_below(seed, uint64_t(r), uint32_t(j), d - j) < m - taken) {
                    out_col[o + taken] = col[s + j];
                    if (out_w) out_w[o + taken] = w ? w[s + j] : 1.0f;
                    ++taken;
                }
            }
            continue;
        }
        // m < d: Floyd's algorithm — m distinct positions of [0, d) with uniform probability, no replacement
        int chosen[kFloydMax];
        int c = 0;
        for (int j = d - m; j < d; ++j) {
            int t = draw_below(seed, uint64_t(r), uint32_t(j - (d - m)), j + 1);
            bool dup = false;
            for (int q = 0; q < c; ++q) dup |= (chosen[q] == t);
            chosen[c++] = dup ? j : t;
        }
        for (int i = 0; i < m; ++i) {
            out_col[o + i] = col[s + chosen[i]];
            if (out_w) out_w[o + i] = w ? w[s + chosen[i]] : 1.0f;
        }
    }
}

// Rows with more than 64 neighbours or draws (all of a power-law graph's mass; hubs of 10^4 .. 10^5 neighbours): one WAVE
// per row instead of one lane — coalesced walks, no divergence between a hub and its wave-mates (R-MAT, ratio = 0.5:
// 43 ms -> see profiles/r02_skew_cliff_scan.jsonl).  keep-all and with-replacement rows are copied / drawn lane-parallel.
// m < d: the row's d positions are cut into 64 contiguous strata; with ONE uniform integer U in [0, d) per row, stratum
// [b0, b1) receives (m*b1 + U)/d - (m*b0 + U)/d of the m samples (integer divisions: never more than it holds, exactly m
// in total, output offsets in closed form, expectation m * len / d) and draws them by selection sampling (Algorithm S).
// Neighbour order is kept and EVERY neighbour is included with probability exactly m / d, as with the reference's
// np.random.choice; the joint draw is systematic across strata rather than a uniform m-subset (lower variance).
__global__ __launch_bounds__(kBlock) void sample_neighbors_long_kernel(const int32_t* __restrict__ row_ptr,
                                                                       const int32_t* __restrict__ col,
                                                                       const float* __restrict__ w, int64_t n_dst,
                                                                       const int32_t* __restrict__ out_ptr,
                                                                       int replace_when_short, uint64_t seed,
                                                                       int32_t* __restrict__ out_col,
                                                                       float* __restrict__ out_w)
{
    constexpr int WV = 64;
    const int lane = threadIdx.x % WV;
    int64_t r = (blockIdx.x * int64_t(kBlock) + threadIdx.x) / WV;
    const int64_t stride = int64_t(gridDim.x) * kBlock / WV;
    for (; r < n_dst; r += stride) {
        const int s = row_ptr[r], d = row_ptr[r + 1] - s;
        const int o = out_ptr[r], m = out_ptr[r + 1] - o;
        if (m == 0 || !sample_row_is_long(d, m)) continue;
        if (m >= d && !(replace_when_short && m > d)) {          // keep every neighbour, in order
            for (int i = lane; i < d; i += WV) {
                out_col[o + i] = col[s + i];
                if (out_w) out_w[o + i] = w ? w[s + i] : 1.0f;
            }
            continue;
        }
        if (m > d) {                                             // padding: m draws WITH replacement (same draws as one lane)
            for (int i = lane; i < m; i += WV) {
                const int t = draw_below(seed, uint64_t(r), uint32_t(i), d);
                out_col[o + i] = col[s + t];
                if (out_w) out_w[o + i] = w ? w[s + t] : 1.0f;
            }
            continue;
        }
        const int b0 = int(int64_t(d) * lane / WV), b1 = int(int64_t(d) * (lane + 1) / WV);        // stratum [b0, b1)
        const int64_t U = draw_below(seed, uint64_t(r), 0xFFFFFFFFu, d);                           // the row's random offset
        const int o0 = int((int64_t(m) * b0 + U) / d), ml = int((int64_t(m) * b1 + U) / d) - o0;   // its samples
        int taken = 0;
        for (int j = b0; j < b1 && taken < ml; ++j) {
            if (draw_below(seed, uint64_t(r), uint32_t(j), b1 - j) < ml - taken) {
                out_col[o + o0 + taken] = col[s + j];
                if (out_w) out_w[o + o0 + taken] = w ? w[s + j] : 1.0f;
                ++taken;
            }
        }
    }
}

// one pass: x[n, F] -> main[n, f_main] (128-byte aligned rows) + tail[n, F - f_main]
__global__ __launch_bounds__(kBlock) void split_rows_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int F,
                                                            int f_main, float* __restrict__ xm, int64_t ldm,
                                                            float* __restrict__ xt, int64_t ldt)
{
    const int per_row = F / 4;
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t total = n * per_row;
    for (; t < total; t += stride) {
        const int64_t i = t / per_row;
        const int j = int(t - i * per_row) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + i * ldx + j);
        if (j < f_main) *reinterpret_cast<float4*>(xm + i * ldm + j) = v;
        else *reinterpret_cast<float4*>(xt + i * ldt + (j - f_main)) = v;
    }
}

// the split layout against the table it was built from, on a SAMPLE of rows (rows 0 and n - 1 always, the others drawn from
// (seed, i)): any bit that differs raises *mismatch.  One thread per (sampled row, 4 columns).
__global__ __launch_bounds__(kBlock) void split_rows_verify_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int F,
                                                                   int f_main, const float* __restrict__ xm, int64_t ldm,
                                                                   const float* __restrict__ xt, int64_t ldt, int64_t samples,
                                                                   uint64_t seed, int32_t* __restrict__ mismatch)
{
    const int per_row = F / 4;
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t total = samples * per_row;
    bool bad = false;
    for (; t < total; t += stride) {
        const int64_t k = t / per_row;
        const int j = int(t - k * per_row) * 4;
        int64_t i;
        if (samples >= n) i = k;                                   // full comparison
        else if (k == 0) i = 0;
        else if (k == 1) i = n - 1;
        else i = int64_t((uint64_t(drop_hash(seed, uint32_t(k))) << 16 ^ drop_hash(seed ^ 0x9E3779B97F4A7C15ull, uint32_t(k))) % uint64_t(n));
        const uint4 a = *reinterpret_cast<const uint4*>(x + i * ldx + j);
        const uint4 b = j < f_main ? *reinterpret_cast<const uint4*>(xm + i * ldm + j)
                                   : *reinterpret_cast<const uint4*>(xt + i * ldt + (j - f_main));
        bad = bad || a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w;
    }
    if (bad) atomicOr(mismatch, 1);
}

}  // namespace
}  // namespace tfgx

using namespace tfgx;

extern "C" int tfgx_l2_normalize_rows_f32(float* h, int64_t ld, int64_t n, int64_t F, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n >= 0 && F >= 1 && ld >= F, "bad size");
    if (n == 0) return TFGX_OK;
    TFGX_REQUIRE(h != nullptr, "null pointer");
    l2_normalize_kernel<<<grid_for(n * 64, kBlock), kBlock, 0, as_stream(stream)>>>(h, ld, n, int(F));
    TFGX_LAUNCH_CHECK("l2_normalize_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_gather_rows_f32(const float* x, int64_t ldx, const int32_t* idx, int64_t M, int64_t F,
                                    float* out, int64_t ldo, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(M >= 0 && F >= 1 && ldx >= F && ldo >= F, "bad size");
    if (M == 0) return TFGX_OK;
    TFGX_REQUIRE(x && out, "null pointer");
    const bool v4 = (F % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && aligned_to(x, 16) && aligned_to(out, 16);
    if (v4)
        gather_rows_kernel<4><<<grid_for(M * (F / 4), kBlock), kBlock, 0, as_stream(stream)>>>(x, ldx, idx, M, int(F),
                                                                                              out, ldo);
    else
        gather_rows_kernel<1><<<grid_for(M * F, kBlock), kBlock, 0, as_stream(stream)>>>(x, ldx, idx, M, int(F), out,
                                                                                        ldo);
    TFGX_LAUNCH_CHECK("gather_rows_kernel");
    return TFGX_OK;
}

// gout[m, n] = out[m, n] > 0 ? g[m, n] : 0 — the backward of the ReLU that rides in a GEMM / aggregation epilogue
// (one pass instead of compare + cast + multiply); row strides allow column slices of wider matrices
namespace tfgx {
namespace {
template <int VEC>
__global__ __launch_bounds__(kBlock) void relu_backward_kernel(const float* __restrict__ g, int64_t ldg,
                                                               const float* __restrict__ out, int64_t ldo, int64_t M,
                                                               int N, float* __restrict__ gout, int64_t ldgo)
{
    const int per_row = N / VEC;
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (; t < M * per_row; t += stride) {
        const int64_t m = t / per_row;
        const int c = int(t - m * per_row) * VEC;
        float gv[VEC], ov[VEC];
        load_vec<VEC>(g + m * ldg + c, gv);
        load_vec<VEC>(out + m * ldo + c, ov);
#pragma unroll
        for (int i = 0; i < VEC; ++i) gv[i] = ov[i] > 0.0f ? gv[i] : 0.0f;
        store_vec<VEC>(gout + m * ldgo + c, gv);
    }
}
}  // namespace
}  // namespace tfgx

extern "C" int tfgx_relu_backward_f32(const float* g, int64_t ldg, const float* out, int64_t ldo, int64_t M, int64_t N,
                                      float* gout, int64_t ldgo, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(M >= 0 && N >= 1 && N < (int64_t(1) << 30), "bad M / N");
    TFGX_REQUIRE(ldg >= N && ldo >= N && ldgo >= N, "leading dimension too small");
    if (M == 0) return TFGX_OK;
    TFGX_REQUIRE(g && out && gout, "null pointer");
    const bool v4 = N % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0 && ldgo % 4 == 0 && tfgx::aligned_to(g, 16) &&
                    tfgx::aligned_to(out, 16) && tfgx::aligned_to(gout, 16);
    if (v4)
        tfgx::relu_backward_kernel<4><<<tfgx::grid_for(M * (N / 4), tfgx::kBlock), tfgx::kBlock, 0, tfgx::as_stream(stream)>>>(
            g, ldg, out, ldo, M, int(N), gout, ldgo);
    else
        tfgx::relu_backward_kernel<1><<<tfgx::grid_for(M * N, tfgx::kBlock), tfgx::kBlock, 0, tfgx::as_stream(stream)>>>(
            g, ldg, out, ldo, M, int(N), gout, ldgo);
    TFGX_LAUNCH_CHECK("relu_backward_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_scatter_add_rows_f32(float* dst, int64_t ldd, const int32_t* idx, int64_t M, int64_t F,
                                        const float* src, int64_t lds, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(M >= 0 && F >= 1 && ldd >= F && lds >= F, "bad shape");
    if (M == 0) return TFGX_OK;
    TFGX_REQUIRE(dst && src, "null pointer");
    const bool v4 = (F % 4 == 0) && (ldd % 4 == 0) && (lds % 4 == 0) && aligned_to(dst, 16) && aligned_to(src, 16);
    if (v4)
        scatter_add_rows_kernel<4><<<grid_for(M * (F / 4), kBlock), kBlock, 0, as_stream(stream)>>>(dst, ldd, idx, M,
                                                                                                   int(F), src, lds);
    else
        scatter_add_rows_kernel<1><<<grid_for(M * F, kBlock), kBlock, 0, as_stream(stream)>>>(dst, ldd, idx, M, int(F),
                                                                                             src, lds);
    TFGX_LAUNCH_CHECK("scatter_add_rows_kernel");
    return TFGX_OK;
}

extern "C" size_t tfgx_halo_workspace_bytes(int64_t n_global)
{
    size_t temp = 0;
    const int32_t* in = nullptr;
    int32_t* out = nullptr;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, temp, in, out, static_cast<int>(n_global > 0 ? n_global : 1));
    return temp + 256;
}

extern "C" int tfgx_halo_mark(const int32_t* col, int64_t E, int32_t own_lo, int32_t own_hi, int64_t n_global,
                              int32_t* flags, tfgx_stream_t stream_)
{
    TFGX_RANGE();
    hipStream_t stream = as_stream(stream_);
    TFGX_REQUIRE(E >= 0 && n_global >= 0 && flags, "bad argument");
    TFGX_HIP_CHECK(hipMemsetAsync(flags, 0, sizeof(int32_t) * size_t(n_global), stream));
    if (E == 0) return TFGX_OK;
    halo_mark_kernel<<<grid_for(E, kBlock), kBlock, 0, stream>>>(col, E, own_lo, own_hi, flags);
    TFGX_LAUNCH_CHECK("halo_mark_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_halo_compact(const int32_t* flags, int64_t n_global, int32_t* pos, int32_t* halo_ids,
                                 int32_t* n_halo, void* workspace, size_t workspace_bytes, tfgx_stream_t stream_)
{
    TFGX_RANGE();
    hipStream_t stream = as_stream(stream_);
    TFGX_REQUIRE(n_global >= 0 && n_halo, "bad argument");
    if (n_global == 0) {
        TFGX_HIP_CHECK(hipMemsetAsync(n_halo, 0, sizeof(int32_t), stream));
        return TFGX_OK;
    }
    TFGX_REQUIRE(flags && pos && halo_ids && workspace, "null pointer");
    if (workspace_bytes < tfgx_halo_workspace_bytes(n_global)) {
        set_error("tfgx_halo_compact: workspace too small");
        return TFGX_ERR_WORKSPACE;
    }
    size_t temp = workspace_bytes;
    TFGX_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(workspace, temp, flags, pos, static_cast<int>(n_global), stream));
    halo_scatter_kernel<<<grid_for(n_global, kBlock), kBlock, 0, stream>>>(flags, pos, n_global, halo_ids, n_halo);
    TFGX_LAUNCH_CHECK("halo_scatter_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_halo_remap_cols(const int32_t* col, int64_t E, int32_t own_lo, int32_t own_hi,
                                    const int32_t* pos, int32_t n_own, int32_t* col_local, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(E >= 0, "bad argument");
    if (E == 0) return TFGX_OK;
    TFGX_REQUIRE(col && pos && col_local, "null pointer");
    halo_remap_kernel<<<grid_for(E, kBlock), kBlock, 0, as_stream(stream)>>>(col, E, own_lo, own_hi, pos, n_own,
                                                                            col_local);
    TFGX_LAUNCH_CHECK("halo_remap_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_split_rows_f32(const float* x, int64_t ldx, int64_t n, int64_t F, int64_t f_main, float* x_main,
                                   int64_t ld_main, float* x_tail, int64_t ld_tail, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n >= 0 && F >= 8 && F % 4 == 0 && f_main > 0 && f_main < F && f_main % 4 == 0, "bad F / f_main");
    if (n == 0) return TFGX_OK;
    TFGX_REQUIRE(x && x_main && x_tail, "null pointer");
    TFGX_REQUIRE(ldx >= F && ldx % 4 == 0 && ld_main >= f_main && ld_main % 4 == 0 && ld_tail >= F - f_main &&
                     ld_tail % 4 == 0 && aligned_to(x, 16) && aligned_to(x_main, 16) && aligned_to(x_tail, 16),
                 "rows must be 16-byte aligned");
    split_rows_kernel<<<grid_for(n * (F / 4), kBlock), kBlock, 0, as_stream(stream)>>>(x, ldx, n, int(F), int(f_main),
                                                                                     x_main, ld_main, x_tail, ld_tail);
    TFGX_LAUNCH_CHECK("split_rows_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_split_rows_verify_f32(const float* x, int64_t ldx, int64_t n, int64_t F, int64_t f_main, const float* x_main,
                                          int64_t ld_main, const float* x_tail, int64_t ld_tail, int64_t samples, uint64_t seed,
                                          int32_t* mismatch, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n >= 0 && F >= 8 && F % 4 == 0 && f_main > 0 && f_main < F && f_main % 4 == 0, "bad F / f_main");
    TFGX_REQUIRE(mismatch != nullptr, "null pointer");
    TFGX_HIP_CHECK(hipMemsetAsync(mismatch, 0, sizeof(int32_t), as_stream(stream)));
    if (n == 0 || samples <= 0) return TFGX_OK;
    TFGX_REQUIRE(x && x_main && x_tail, "null pointer");
    TFGX_REQUIRE(ldx >= F && ldx % 4 == 0 && ld_main >= f_main && ld_main % 4 == 0 && ld_tail >= F - f_main &&
                     ld_tail % 4 == 0 && aligned_to(x, 16) && aligned_to(x_main, 16) && aligned_to(x_tail, 16),
                 "rows must be 16-byte aligned");
    if (samples > n) samples = n;
    if (samples < n && samples < 2) samples = 2;
    split_rows_verify_kernel<<<grid_for(samples * (F / 4), kBlock), kBlock, 0, as_stream(stream)>>>(
        x, ldx, n, int(F), int(f_main), x_main, ld_main, x_tail, ld_tail, samples, seed, mismatch);
    TFGX_LAUNCH_CHECK("split_rows_verify_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_sample_neighbors(const int32_t* row_ptr, const int32_t* col, const float* w, int64_t n_dst,
                                     const int32_t* out_ptr, int32_t max_per_row, int32_t replace_when_short,
                                     uint64_t seed, int32_t* out_col, float* out_w, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n_dst >= 0 && max_per_row >= 0, "bad size");
    // max_per_row is informative only: rows with more than kMaxSampleK draws take the scratch-free selection-sampling
    // branch, keep-all rows need no scratch at all
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr && out_ptr, "null pointer");
    sample_neighbors_kernel<<<grid_for(n_dst, kBlock), kBlock, 0, as_stream(stream)>>>(
        row_ptr, col, w, n_dst, out_ptr, replace_when_short, seed, out_col, out_w);
    TFGX_LAUNCH_CHECK("sample_neighbors_kernel");
    // rows the lane-per-row kernel skipped (hubs): a wave each; a pass over row_ptr / out_ptr when there are none
    // (an ODD grid: with a power-of-two row stride the hubs of an R-MAT graph — ids k * 2^16 — would all land in one wave)
    sample_neighbors_long_kernel<<<grid_for(n_dst * 64, kBlock, (1 << 14) - 3), kBlock, 0, as_stream(stream)>>>(
        row_ptr, col, w, n_dst, out_ptr, replace_when_short, seed, out_col, out_w);
    TFGX_LAUNCH_CHECK("sample_neighbors_long_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_split_by_source_class(const int32_t* row_ptr, const int32_t* col_local, const float* w,
                                          int64_t n_dst, int64_t E, const int32_t* class_bounds, int32_t n_class,
                                          int32_t* row_ptr_k, int32_t* col_out, float* w_out, tfgx_stream_t stream_)
{
    TFGX_RANGE();
    hipStream_t stream = as_stream(stream_);
    TFGX_REQUIRE(n_dst >= 0 && E >= 0 && row_ptr_k && n_class >= 1 && n_class <= kMaxClasses, "bad argument");
    if (n_dst == 0) {
        TFGX_HIP_CHECK(hipMemsetAsync(row_ptr_k, 0, sizeof(int32_t), stream));
        return TFGX_OK;
    }
    TFGX_REQUIRE(row_ptr && class_bounds, "null pointer");
    TFGX_REQUIRE((w == nullptr) == (w_out == nullptr), "w and w_out must both be given or both be null");
    split_by_class_kernel<<<grid_for(n_dst, kBlock), kBlock, 0, stream>>>(row_ptr, col_local, w, n_dst, class_bounds,
                                                                          n_class, row_ptr_k, col_out, w_out);
    TFGX_LAUNCH_CHECK("split_by_class_kernel");
    return TFGX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// out[c] = sum_m g[m, c]: the bias gradient of a layer whose bias rides in an AGGREGATION epilogue (the column sums of the
// masked output gradient; where the bias rides in a GEMM epilogue tfgx_gemm_tn_f32 produces it beside dW).  Deterministic
// two-phase reduction: every workgroup sums a strided set of 4-row slabs into one partial row of the workspace (threads laid
// out [rows][columns], the row copies folded through LDS in order), then one thread per column folds the partial rows in
// workgroup order.  Replaces torch's g.sum(0), which takes 18.7 ms on a [2.4 M, 47] gradient (ogbn-products has 47 classes:
// an odd width sends torch's reduction down a 24 GB/s path) against 0.15 ms here.
namespace tfgx {
namespace {
constexpr int kColSumMaxBlocks = 1024;

__global__ __launch_bounds__(kBlock) void column_sum_partial_kernel(const float* __restrict__ g, int64_t ldg, int64_t M, int N,
                                                                    int cw, float* __restrict__ parts)
{
    __shared__ float red[kBlock];
    const int rows = kBlock / cw;                       // row copies per workgroup pass
    const int tx = threadIdx.x % cw, ty = threadIdx.x / cw;
    const int64_t stride = int64_t(gridDim.x) * rows;
    for (int c0 = 0; c0 < N; c0 += cw) {
        const int c = c0 + tx;
        // four independent chains, eight loads in flight per thread: with one load per iteration a [2.4 M, 40] gradient ran
        // at 1.4 TB/s (round 4 trace of the products-shape training step); the four partial sums are folded in a fixed order
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (c < N && ty < rows) {
            const float* p = g + c;
            int64_t m = int64_t(blockIdx.x) * rows + ty;
            for (; m + 7 * stride < M; m += 8 * stride) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[(m + u * stride) * ldg];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u & 3] += v[u];
            }
            for (; m < M; m += stride) acc[0] += p[m * ldg];
        }
        red[threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        __syncthreads();
        if (ty == 0 && c < N) {
            float s = red[tx];
            for (int r = 1; r < rows; ++r) s += red[r * cw + tx];
            parts[int64_t(blockIdx.x) * N + c] = s;
        }
        __syncthreads();
    }
}

// out[c] = sum over the workgroups' partials, in a FIXED order: 16 slices of the partial list per column (threadIdx.y), each
// summed in list order, then the 16 slice sums in slice order (one thread per column looping over 1024 partials took 234 us)
constexpr int kFoldSlices = 16;
__global__ __launch_bounds__(64 * kFoldSlices) void column_sum_fold_kernel(const float* __restrict__ parts, int blocks, int N,
                                                                          float* __restrict__ out)
{
    __shared__ float red[kFoldSlices][64];
    const int tx = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int per = (blocks + kFoldSlices - 1) / kFoldSlices;
    float s = 0.0f;
    if (c < N) {
        const int b1 = min(blocks, (sl + 1) * per);
        for (int b = sl * per; b < b1; ++b) s += parts[int64_t(b) * N + c];
    }
    red[sl][tx] = s;
    __syncthreads();
    if (sl == 0 && c < N) {
        float t = red[0][tx];
        for (int k = 1; k < kFoldSlices; ++k) t += red[k][tx];
        out[c] = t;
    }
}

inline int column_sum_blocks(int64_t M, int cw)
{
    const int64_t rows = kBlock / cw;
    const int64_t want = (M + rows * 8 - 1) / (rows * 8);          // >= 8 passes of rows per workgroup
    return int(want < 1 ? 1 : (want > kColSumMaxBlocks ? kColSumMaxBlocks : want));
}
inline int column_sum_cw(int64_t N)
{
    int cw = 1;
    while (cw < N && cw < kBlock) cw <<= 1;
    return cw;
}
}  // namespace
}  // namespace tfgx

extern "C" size_t tfgx_column_sum_workspace_bytes(int64_t M, int64_t N)
{
    if (M <= 0 || N <= 0) return 0;
    return sizeof(float) * size_t(tfgx::column_sum_blocks(M, tfgx::column_sum_cw(N))) * size_t(N);
}

extern "C" int tfgx_column_sum_f32(const float* g, int64_t ldg, int64_t M, int64_t N, float* out, void* workspace,
                                   size_t workspace_bytes, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(M >= 0 && N >= 1 && N < (int64_t(1) << 30), "bad M / N");
    TFGX_REQUIRE(out != nullptr, "out is null");
    hipStream_t st = tfgx::as_stream(stream);
    if (M == 0) {
        TFGX_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(float) * size_t(N), st));
        return TFGX_OK;
    }
    TFGX_REQUIRE(g != nullptr && ldg >= N, "null pointer / leading dimension too small");
    const int cw = tfgx::column_sum_cw(N), blocks = tfgx::column_sum_blocks(M, cw);
    TFGX_REQUIRE(workspace != nullptr && workspace_bytes >= sizeof(float) * size_t(blocks) * size_t(N),
                 "workspace too small (tfgx_column_sum_workspace_bytes)");
    float* parts = static_cast<float*>(workspace);
    tfgx::column_sum_partial_kernel<<<blocks, tfgx::kBlock, 0, st>>>(g, ldg, M, int(N), cw, parts);
    TFGX_LAUNCH_CHECK("column_sum_partial_kernel");
    tfgx::column_sum_fold_kernel<<<int((N + 63) / 64), 64 * tfgx::kFoldSlices, 0, st>>>(parts, blocks, int(N), out);
    TFGX_LAUNCH_CHECK("column_sum_fold_kernel");
    return TFGX_OK;
}
