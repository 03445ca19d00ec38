// GAT edge attention on the CSR-by-destination plan.
//
//   tfgx_edge_softmax_f32 : tf_geometric/nn/kernel/segment.py:26-33 (segment_softmax) and
//                           SparseMatrix.segment_softmax(axis=-1) at nn/conv/gat.py:83-84.
//   tfgx_gat_fused_f32    : nn/conv/gat.py:56 (gather Q by row), :65 (gather K by col), :73-79 (per-head
//                           scores), :83-84 (softmax), :87-89 (att @ V), :112 (head concat), :116-120.
//                           One pass per destination row with an online softmax; the reference's [E',A]
//                           gathers, the [2, H*E'] virtual edge index and the [H*E'] score vector are
//                           never materialised.
//   tfgx_head_mean_f32    : nn/conv/gat.py:114 (split_value_heads=False).
//
// Fused kernel mapping: a group of G lanes owns one destination row; lane -> VEC consecutive columns of
// the H*dv-wide value row, hence one head.  Every lane of a head computes that head's score redundantly
// (the K loads of a head's lanes hit the same address and coalesce), so the online-softmax state
// (m, l, acc[VEC]) is lane-private and no cross-lane traffic is needed beyond the col broadcast.
#include "tfgx_common.h"
#include <cfloat>
#include <cmath>
#include <type_traits>
#ifndef TFGX_GAT_POW2_SCALE
#define TFGX_GAT_POW2_SCALE 1         // developer A/B: 0 = the attention kernels always DIVIDE the score by scale
#endif

namespace tfgx {
namespace {

// ---------------------------------------------------------------- edge softmax (standalone)
// one thread per (row, head): any H; rows are walked serially (kept for H > 64)
__global__ __launch_bounds__(kBlock) void edge_softmax_serial_kernel(const int32_t* __restrict__ row_ptr,
                                                                     const int32_t* __restrict__ perm,
                                                                     const float* __restrict__ score, int H,
                                                                     int64_t n_dst, float* __restrict__ out)
{
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t total = n_dst * H;
    for (; t < total; t += stride) {
        const int64_t r = t / H;
        const int h = int(t - r * H);
        const int s = row_ptr[r], e = row_ptr[r + 1];
        float m = -FLT_MAX;
        for (int i = s; i < e; ++i) m = fmaxf(m, score[int64_t(perm ? perm[i] : i) * H + h]);
        float d = 0.0f;
        for (int i = s; i < e; ++i) d += expf(score[int64_t(perm ? perm[i] : i) * H + h] - m);
        d += 1e-8f;
        for (int i = s; i < e; ++i) {
            const int64_t eid = perm ? perm[i] : i;
            out[eid * H + h] = expf(score[eid * H + h] - m) / d;
        }
    }
}

// A group of G lanes per destination row, arranged as (G / Hp edge slots) x (Hp = next power of two >= H heads): adjacent
// lanes read adjacent heads of one edge (coalesced), the slots stride over the row's edges, per-head reductions are
// butterflies across the slots.  Short rows run with the smallest G that holds Hp (H = 8: exactly the one-thread-per-
// (row, head) mapping); rows longer than kSoftmaxWide edges (hubs of a power-law graph) are left to a second launch with
// a whole wave per row — R-MAT, 30 M edges, H = 8: 340 ms (serial rows) -> see profiles/r02_skew_cliff_scan.jsonl.
constexpr int kSoftmaxWide = 1024;

template <int G>
__device__ __forceinline__ float softmax_group_reduce(float v, int Hp, bool is_max, float* lds)
{
    constexpr int W = G < 64 ? G : 64;
    for (int o = W / 2; o >= Hp; o >>= 1) {
        const float t = __shfl_xor(v, o, W);
        v = is_max ? fmaxf(v, t) : v + t;
    }
    if constexpr (G > 64) {          // a whole workgroup per row: combine the waves' per-head values through LDS
        const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
        __syncthreads();
        if (lane < Hp) lds[wave * 64 + lane] = v;
        __syncthreads();
        float acc = lds[lane % Hp];
        for (int k = 1; k < G / 64; ++k) {
            const float t = lds[k * 64 + lane % Hp];
            acc = is_max ? fmaxf(acc, t) : acc + t;
        }
        v = acc;
    }
    return v;
}

template <int G, bool WIDE>
__global__ __launch_bounds__(kBlock) void edge_softmax_kernel(const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ perm,
                                                              const float* __restrict__ score, int H, int Hp,
                                                              int64_t n_dst, float* __restrict__ out, int wide_limit)
{
    constexpr int ROWS_PER_BLOCK = kBlock / G;
    __shared__ float lds[G > 64 ? (G / 64) * 64 : 1];
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int h = lane % Hp, slot = lane / Hp, slots = G / Hp;
    const bool hv = h < H;
    for (int64_t r = int64_t(blockIdx.x) * ROWS_PER_BLOCK + grp; r < n_dst; r += int64_t(gridDim.x) * ROWS_PER_BLOCK) {
        const int s = row_ptr[r], e = row_ptr[r + 1];
        if (WIDE != (e - s > wide_limit)) continue;              // (uniform over the group / workgroup: it owns ONE row)
        float m = -FLT_MAX;
        if (hv) {
            int i = s + slot;
            for (; i + 3 * slots < e; i += 4 * slots) {          // four independent index -> score chains in flight
                const float a0 = score[int64_t(perm ? perm[i] : i) * H + h];
                const float a1 = score[int64_t(perm ? perm[i + slots] : i + slots) * H + h];
                const float a2 = score[int64_t(perm ? perm[i + 2 * slots] : i + 2 * slots) * H + h];
                const float a3 = score[int64_t(perm ? perm[i + 3 * slots] : i + 3 * slots) * H + h];
                m = fmaxf(fmaxf(m, fmaxf(a0, a1)), fmaxf(a2, a3));
            }
            for (; i < e; i += slots) m = fmaxf(m, score[int64_t(perm ? perm[i] : i) * H + h]);
        }
        m = softmax_group_reduce<G>(m, Hp, true, lds);
        float d = 0.0f;
        if (hv) {
            int i = s + slot;
            for (; i + 3 * slots < e; i += 4 * slots) {
                const float a0 = score[int64_t(perm ? perm[i] : i) * H + h];
                const float a1 = score[int64_t(perm ? perm[i + slots] : i + slots) * H + h];
                const float a2 = score[int64_t(perm ? perm[i + 2 * slots] : i + 2 * slots) * H + h];
                const float a3 = score[int64_t(perm ? perm[i + 3 * slots] : i + 3 * slots) * H + h];
                d += (expf(a0 - m) + expf(a1 - m)) + (expf(a2 - m) + expf(a3 - m));
            }
            for (; i < e; i += slots) d += expf(score[int64_t(perm ? perm[i] : i) * H + h] - m);
        }
        d = softmax_group_reduce<G>(d, Hp, false, lds) + 1e-8f;
        if (hv)
            for (int i = s + slot; i < e; i += slots) {
                const int64_t eid = perm ? perm[i] : i;
                out[eid * H + h] = expf(score[eid * H + h] - m) / d;
            }
    }
}

// Hub rows with the plan's chunk lists (tfgx_edge_softmax_hub_f32): a wave per CHUNK instead of a workgroup per row —
// (1) per-chunk (max, sum of exp) of every head, (2) per hub row: fold its chunks' pairs in chunk order into the row's
// (M, D) and hand them back to every chunk, (3) per chunk: normalise.  A 10^5-edge hub becomes ~10^2 independent waves.
template <int MODE>       // 0: chunk statistics -> ml[p, 0..Hp) = max, ml[p, Hp..2Hp) = sum;  1: normalise with ml[p]
__global__ __launch_bounds__(kBlock) void edge_softmax_chunk_kernel(const int32_t* __restrict__ cb,
                                                                    const int32_t* __restrict__ ce, int64_t n_chunks,
                                                                    const int32_t* __restrict__ perm,
                                                                    const float* __restrict__ score, int H, int Hp,
                                                                    float* __restrict__ ml, float* __restrict__ out)
{
    constexpr int G = 64;
    const int lane = threadIdx.x % G;
    const int h = lane % Hp, slot = lane / Hp, slots = G / Hp;
    const bool hv = h < H;
    for (int64_t p = (blockIdx.x * int64_t(kBlock) + threadIdx.x) / G; p < n_chunks; p += int64_t(gridDim.x) * kBlock / G) {
        const int s = cb[p], e = ce[p];
        if (MODE == 0) {
            float m = -FLT_MAX;
            if (hv)
                for (int i = s + slot; i < e; i += slots) m = fmaxf(m, score[int64_t(perm ? perm[i] : i) * H + h]);
            for (int o = G / 2; o >= Hp; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, G));
            float d = 0.0f;
            if (hv)
                for (int i = s + slot; i < e; i += slots) d += expf(score[int64_t(perm ? perm[i] : i) * H + h] - m);
            for (int o = G / 2; o >= Hp; o >>= 1) d += __shfl_xor(d, o, G);
            if (lane < Hp) {
                ml[p * 2 * Hp + lane] = m;
                ml[p * 2 * Hp + Hp + lane] = d;
            }
        } else {
            const float m = ml[p * 2 * Hp + h], d = ml[p * 2 * Hp + Hp + h];
            if (hv)
                for (int i = s + slot; i < e; i += slots) {
                    const int64_t eid = perm ? perm[i] : i;
                    out[eid * H + h] = expf(score[eid * H + h] - m) / d;
                }
        }
    }
}

__global__ __launch_bounds__(kBlock) void edge_softmax_hub_combine_kernel(const int32_t* __restrict__ chunk_ptr,
                                                                          int64_t n_hub_rows, int Hp,
                                                                          float* __restrict__ ml)
{
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    for (; t < n_hub_rows * Hp; t += int64_t(gridDim.x) * kBlock) {
        const int64_t i = t / Hp;
        const int h = int(t - i * Hp);
        const int c0 = chunk_ptr[i], c1 = chunk_ptr[i + 1];
        float M = -FLT_MAX;
        for (int c = c0; c < c1; ++c) M = fmaxf(M, ml[int64_t(c) * 2 * Hp + h]);
        float D = 0.0f;
        for (int c = c0; c < c1; ++c) D += ml[int64_t(c) * 2 * Hp + Hp + h] * expf(ml[int64_t(c) * 2 * Hp + h] - M);
        D += 1e-8f;                                                      // segment.py:30
        for (int c = c0; c < c1; ++c) {
            ml[int64_t(c) * 2 * Hp + h] = M;
            ml[int64_t(c) * 2 * Hp + Hp + h] = D;
        }
    }
}

// ---------------------------------------------------------------- fused GAT
struct GArgs {
    const int32_t* row_ptr;
    const int32_t* col;
    int64_t n_dst;
    const float* q; int64_t ldq;
    const float* k; int64_t ldk;
    const float* v; int64_t ldv;
    float* out; int64_t ldo;
    int32_t H, d, dv, W;  // W = H*dv
    int32_t add_self_loop;
    float scale;
    float inv_scale;      // 1 / scale when scale is a power of two (then dot * inv_scale == dot / scale bit for bit), else 0
    int32_t act;
    const float* bias;
    int32_t kvec;         // 1: K/Q head slices are 16-byte aligned and d % 4 == 0
    // parts: "row" p of the launch spans CSR positions [row_begin[p*rp_stride], row_end[p*rp_stride]) and belongs to
    // destination part_row[p] (NULL: p). state_acc != NULL: write the raw online-softmax state (acc, m, l) of the
    // part instead of the normalised output — chunks of hub rows and the two passes of a split plan are merged by
    // gat_merge_kernel.
    const int32_t* row_begin;
    const int32_t* row_end;
    int64_t rp_stride;
    const int32_t* part_row;
    const int32_t* row_order;   // walk order of the main launch (NULL: identity)
    float* state_acc;     // [n_parts, W]
    float* state_ml;      // [n_parts, 2H]  (m, l) per head
    const float* state_in_acc;   // optional: the walk of part p RESUMES from this stored state instead of (0, -FLT_MAX, 0) —
    const float* state_in_ml;    // source-blocked passes chained through the state (tfgx_gat_args.state_in_acc)
    int32_t hub_threshold;
    float* stats_ml;      // [n_dst, 2H] final (m, l) for the backward pass, or NULL
    DropCfg drop;         // attention dropout (training): acc += p * keep_scale_or_0 * V; the denominator is untouched
    // QG kernels only (tfgx_gat_args.qgrad_t): the query gradient's sums, final (normalised) or as raw state
    float* qg_t;          // [n_dst, W]
    float* qg_s;          // [n_dst, H]
    float* state_t;       // [n_parts, W]
    float* state_s;       // [n_parts, H]
    const float* state_in_t;
    const float* state_in_s;
};

// 1 / s when s is a power of two with a normal reciprocal: x * (1 / s) and x / s are then the same correctly rounded value
inline float exact_inverse_or_zero(float s)
{
#if TFGX_GAT_POW2_SCALE
    int e = 0;
    if (!(s > 0.0f) || std::frexp(s, &e) != 0.5f) return 0.0f;
    const float inv = 1.0f / s;
    return (inv >= FLT_MIN && inv <= FLT_MAX) ? inv : 0.0f;
#else
    (void)s;
    return 0.0f;
#endif
}

// D > 0: compile-time head width (Q slice lives in registers); D == 0: runtime d, Q re-read (cache-hot)
template <int D>
__device__ __forceinline__ float head_dot(const float (&qreg)[D > 0 ? D : 1], const float* __restrict__ qp,
                                          const float* __restrict__ kp, int d, int kvec)
{
    float s = 0.0f;
    if constexpr (D == 0) {
        for (int t = 0; t < d; ++t) s = fmaf(qp[t], kp[t], s);
    } else if constexpr (D % 4 == 0) {
        if (kvec) {
#pragma unroll
            for (int t = 0; t < D; t += 4) {
                const float4 kk = *reinterpret_cast<const float4*>(kp + t);
                s = fmaf(qreg[t], kk.x, s);
                s = fmaf(qreg[t + 1], kk.y, s);
                s = fmaf(qreg[t + 2], kk.z, s);
                s = fmaf(qreg[t + 3], kk.w, s);
            }
        } else {
#pragma unroll
            for (int t = 0; t < D; ++t) s = fmaf(qreg[t], kp[t], s);
        }
    } else {
#pragma unroll
        for (int t = 0; t < D; ++t) s = fmaf(qreg[t], kp[t], s);
    }
    return s;
}

// POW2: a.inv_scale is the exact inverse of a power-of-two scale (instantiated for d = 1, 4, 16, where sqrt(d) is one)
#ifndef TFGX_GAT_D8_WAVES
#define TFGX_GAT_D8_WAVES 5           // waves per SIMD the d_head = 8 walk is compiled for.  At 5 (96 VGPRs) the compiler spills ONE
#endif                                // 64-bit kernel-invariant pointer: stored before the row loop, reloaded once per ROW in the state /
                                      // output epilogue (ISA read: no scratch access inside the edge loops).  At 4 (98 VGPRs, no scratch) the
                                      // Reddit-shape A = 64 attention runs 6.45 instead of 5.77 ms, the one-pass walk at products density
                                      // 10.27 instead of 9.89 ms (same box, profiles/r06_gat_d8_waves.jsonl): the fifth wave is worth more
// KS ("K split", round 6; d == dv in {8, 16}, 16-byte aligned K rows): the d floats of a head's K / Q slice are DISTRIBUTED over
// the head's d / 4 lanes — lane `sub` of the head holds floats [4 sub, 4 sub + 4) — instead of every lane of the head loading
// the whole slice.  One dwordx4 per lane then covers the K row contiguously: at A = 64 (8 heads x 8) ONE load instruction
// touches the row's 2 lines, where two instructions each touched both (4 line REQUESTS for 2 lines; the L2s serve requests,
// not lines — see tfgx_backward.hip's head blocks).  The score is the sum of the lanes' partial dots (one or two DPP adds).
template <int CTRL>
__device__ __forceinline__ float attn_dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

// QG ("query gradient", round 6; d == 1 — the demo's literal layer, attention_units == num_heads): the walk also accumulates
//   T[r, cols] = sum_e a_e keep_e K[c_e, h] V[c_e, cols]        S[r, h] = sum_e a_e K[c_e, h]
// from the K and V values it has in registers anyway.  dQ[r, h] = sum_e ds_e K[c_e, h] / scale with ds_e = a_e (keep_e <dO, V_e> - D)
// is then (<dO[r, h, :], T[r, h, :]> - D[r, h] S[r, h]) / scale, a per-ROW expression: the backward's destination pass — a second
// walk over every edge that gathers K and V again, 2.55 of the Reddit-shaped layer's 9.7 ms — is not run (tfgx_gat_query_grad_d1_f32).
#ifndef TFGX_GAT_STATE_NT
#define TFGX_GAT_STATE_NT 0           // developer A/B: 0 = the chained raw state is loaded / stored like any other row
#endif
#if TFGX_GAT_STATE_NT
#define TFGX_STATE_LOAD load_vec_nt
#define TFGX_STATE_STORE store_vec_nt
#else
#define TFGX_STATE_LOAD load_vec
#define TFGX_STATE_STORE store_vec
#endif
#ifndef TFGX_GAT_QG_EXPERIMENT
#define TFGX_GAT_QG_EXPERIMENT 0      // TIMING ONLY (wrong sums): 1 = the sums are not carried through the chained state, 2 = not accumulated
#endif
#ifndef TFGX_GAT_QG_WAVES
#define TFGX_GAT_QG_WAVES 4           // developer A/B: waves per SIMD the QG kernels are compiled for (110 VGPRs as written)
#endif
template <int VEC, int G, int D, bool POW2 = false, bool KS = false, bool QG = false>
__global__ __launch_bounds__(kBlock, (D > 0 && D <= 4) ? (QG ? TFGX_GAT_QG_WAVES : 5) : (D == 8 ? TFGX_GAT_D8_WAVES : 1)) void gat_fused_kernel(const GArgs a)
{
    static_assert(!KS || (VEC == 4 && (D == 8 || D == 16)), "KS: d == dv in {8, 16}, 4 columns per lane");
    static_assert(!QG || (D == 1 && VEC == 4), "QG: one attention unit per head, 4 columns per lane");
    constexpr int DL = KS ? 4 : D;                 // floats of the head's slice this lane holds
    constexpr int ROWS_PER_BLOCK = kBlock / G;
#ifndef TFGX_GAT_ONE_EXP
#define TFGX_GAT_ONE_EXP 1            // developer A/B: 0 = two exponentials per edge (exp(m - mn), exp(sc - mn))
#endif
#ifndef TFGX_GAT_COL_AHEAD
#define TFGX_GAT_COL_AHEAD 1          // developer A/B: 0 = every batch loads its own source ids right before its gathers
#endif
#ifndef TFGX_GAT_UNROLL_NARROW
#define TFGX_GAT_UNROLL_NARROW 4      // developer A/B: edges in flight per lane group when a head's K slice is <= 4 floats
#endif
#ifndef TFGX_GAT_QG_UNROLL
#define TFGX_GAT_QG_UNROLL 4          // developer A/B: edges in flight per lane group in the QG kernels
#endif
    constexpr int UNROLL = QG ? TFGX_GAT_QG_UNROLL : ((D > 0 && D <= 4) ? TFGX_GAT_UNROLL_NARROW : 4);
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int c_raw = (blockIdx.y * G + lane) * VEC;
    const bool cvalid = c_raw < a.W;
    const int coff = cvalid ? c_raw : (a.W - VEC);
    const int head = coff / a.dv;
    const int hoff = head * a.d + (KS ? (coff - head * a.dv) : 0);        // KS: this lane's 4 floats of the slice (d == dv)
    // the score of an edge from this lane's (partial) dot product
    auto lane_sum = [](float s) {
        if constexpr (KS) {
            s += attn_dpp_f32<0xB1>(s);                                     // quad_perm [1,0,3,2]: the other lane of the pair
            if constexpr (D == 16) s += attn_dpp_f32<0x4E>(s);             // quad_perm [2,3,0,1]: the other pair of the quad
        }
        return s;
    };

    for (int64_t r = int64_t(blockIdx.x) * ROWS_PER_BLOCK + grp; r < a.n_dst;
         r += int64_t(gridDim.x) * ROWS_PER_BLOCK) {
        const int64_t idx = r;                                           // loop position; restored before the next turn
        const int64_t part = a.row_order ? int64_t(a.row_order[idx]) : idx;   // degree-ordered walk on skewed graphs
        const int s = a.row_begin[part * a.rp_stride], e = a.row_end[part * a.rp_stride];
        if (a.hub_threshold > 0 && e - s > a.hub_threshold) continue;   // chunked + merged separately
        r = a.part_row ? int64_t(a.part_row[part]) : part;
        const float* qp = a.q + r * a.ldq + hoff;
        float qreg[DL > 0 ? DL : 1];
        if constexpr (D > 0) {
#pragma unroll
            for (int t = 0; t < DL; ++t) qreg[t] = qp[t];
        }
        // `l` is the softmax denominator WITHOUT the running maximum's own term exp(0) = 1: sum over the OTHER edges of
        // exp(score - m).  A peaked row (one score 15 above the rest) has l_full = 1.025: kept whole, every later term of
        // ~5e-5 is rounded to the 1.2e-7 grid of [1, 2) IN THE SAME DIRECTION (a small positive term added to a value that
        // sits on the grid) — 500 such additions left l_full 1.3e-5 off (measured, Reddit shape, d_head = 1; float32
        // evaluated op for op: 2e-7), which is exactly the forward's 3.9e-5 tail and, through sum_e alpha_e != 1, 20 x
        // that in dQ.  Without the 1 the sum lives on the grid of its own (small) magnitude; the 1 is added where the
        // denominator is used.  l_full = l + (m > -FLT_MAX): m leaves -FLT_MAX exactly when a running maximum exists.
        // Second level: a row whose runner-up is close to its maximum still holds l ~ 1 (measured: 6.6e-6 off), so the
        // terms of one batch of G edges are summed in `ls` first and folded into l once per batch — the big accumulator
        // takes deg / G additions instead of deg (one multiply per edge more: both levels shrink when the maximum moves).
        float m = -FLT_MAX, l = 0.0f, ls = 0.0f;
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
        float acc_t[QG ? VEC : 1], s_k = 0.0f;       // QG: the raw T columns of this lane and the head's raw S
#pragma unroll
        for (int i = 0; i < (QG ? VEC : 1); ++i) acc_t[i] = 0.0f;
        if (a.state_in_acc) {   // resume the online softmax where the previous pass over this part left it
            TFGX_STATE_LOAD<VEC>(a.state_in_acc + part * a.W + coff, acc);
            m = a.state_in_ml[part * 2 * a.H + 2 * head];
            l = a.state_in_ml[part * 2 * a.H + 2 * head + 1];
            l -= (m > -FLT_MAX) ? 1.0f : 0.0f;                        // stored whole (exact: 1 is a multiple of ulp(l_full))
            if constexpr (QG && TFGX_GAT_QG_EXPERIMENT != 1) {
                TFGX_STATE_LOAD<VEC>(a.state_in_t + part * a.W + coff, acc_t);
                s_k = a.state_in_s[part * a.H + head];
            }
        }
        auto l_full = [&]() { return (l + ls) + ((m > -FLT_MAX) ? 1.0f : 0.0f); };

        auto step = [&](float sc, const float (&vv)[VEC], int64_t pos, float kj = 0.0f) {
#if TFGX_GAT_ONE_EXP
            // mn = max(m, sc): one of exp(m - mn), exp(sc - mn) is exp(0) = 1, the other exp(-|sc - m|) — ONE exponential per
            // edge (the loop is bound by vector-ALU issue, not by its gathers: ~55 instructions per edge, 25 of them the two
            // expf).  Same bits as the two-exponential form: m - sc == -(sc - m) exactly; the `x - x` terms keep its NaN for an
            // infinite score or running maximum (exp(inf - inf)), and are +0 otherwise.
            const float dlt = sc - m;
            const bool up = dlt > 0.0f;                  // sc is the new running maximum (false for NaN, as fmaxf keeps m)
            const float ex = expf(up ? -dlt : dlt);
            const float corr = up ? ex : 1.0f + (m - m);
            const float p = up ? 1.0f + (sc - sc) : ex;
            const float mn = up ? sc : m;
            // the denominator without the maximum's 1 (see above).  sc takes over: the old maximum becomes an ordinary term
            // exp(m - sc) = ex and the others shrink by it, (l + 1) * ex = fmaf(l, ex, ex); otherwise l + ex — ONE expression
            ls = fmaf(ls, corr, ex);
            l *= corr;
#else
            const float mn = fmaxf(m, sc);
            const float corr = expf(m - mn);
            const float p = expf(sc - mn);
            ls = fmaf(ls, corr, sc > m ? corr : p);
            l *= corr;
#endif
            const float pk = p * drop_scale(a.drop, uint32_t(pos * a.H + head));
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(acc[i], corr, pk * vv[i]);
            if constexpr (QG && TFGX_GAT_QG_EXPERIMENT != 2) {
                const float pkk = pk * kj;
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc_t[i] = fmaf(acc_t[i], corr, pkk * vv[i]);
                s_k = fmaf(s_k, corr, p * kj);
            }
            m = mn;
        };
        // one edge's score; QG (d == 1): the K value itself is kept for the sums above (q * k, the same product head_dot forms)
        auto score_of = [&](const float* kp, float& kj) {
            if constexpr (QG) {
                kj = kp[0];
                return qreg[0] * kj;
            } else {
                return lane_sum(head_dot<DL>(qreg, qp, kp, a.d, a.kvec));
            }
        };
        // element offset of a gathered row as ONE 32 x 32 -> 64-bit multiply (v_mad_u64_u32): ids are non-negative int32 and
        // the leading dimensions fit 32 bits (checked at the entry point).  `int64_t(c) * a.ldk` made the compiler emulate a
        // 64 x 64-bit product — v_ashrrev, 2 v_mul_lo_u32, v_mad_u64_u32, v_add3 per address, three of them quarter-rate:
        // 40 % of this loop's vector-ALU time for its two addresses per edge.
        const uint32_t ldk32 = uint32_t(a.ldk), ldv32 = uint32_t(a.ldv);
        auto row_off = [](int c, uint32_t ld) { return uint64_t(uint32_t(c)) * ld; };
        // score = <q, k> / scale (gat.py:79); a power-of-two scale (sqrt(d) for d = 1, 4, 16, 64) divides exactly by a multiply
        auto scaled = [&](auto pow2, float dot) {
            if constexpr (decltype(pow2)::value) return dot * a.inv_scale;
            else return dot / a.scale;
        };

        // the scale's kind is a template parameter: a run-time branch on it would sit between the gathers of a batch and split
        // them into one basic block each (measured: attention 3.45 -> 4.06 ms in 8 source blocks), and two copies of the walk
        // inside one kernel cost 9 registers and with them the fifth wave per SIMD (one-pass walk on sparse graphs: + 3-5 %)
        constexpr std::bool_constant<POW2> pow2{};
        {
#if TFGX_GAT_COL_AHEAD
            // the source ids of the NEXT batch of G edges are loaded before this batch's gathers are issued: in a source-blocked
            // pass a row has ~60 edges per block (4 batches), and each batch's id load sat in front of its gathers
            int cj_next = (s + lane < e) ? a.col[s + lane] : 0;
#endif
            for (int base = s; base < e; base += G) {
#if TFGX_GAT_COL_AHEAD
                const int cj = cj_next;
                cj_next = (base + G + lane < e) ? a.col[base + G + lane] : 0;
#else
                const int mine = base + lane;
                const int cj = (mine < e) ? a.col[mine] : 0;
#endif
                const int cnt = min(G, e - base);
                int j = 0;
                for (; j + UNROLL <= cnt; j += UNROLL) {
                    float sc[UNROLL], kj[QG ? UNROLL : 1];
                    float vv[UNROLL][VEC];
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        const int c = __shfl(cj, j + u, G);
                        if constexpr (QG) kj[u] = a.k[row_off(c, ldk32) + hoff];    // the score is formed where it is used: one
                        else sc[u] = scaled(pow2, score_of(a.k + row_off(c, ldk32) + hoff, kj[0]));   // live register per edge, not two
                        load_vec<VEC>(a.v + row_off(c, ldv32) + coff, vv[u]);
                    }
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        if constexpr (QG) step(scaled(pow2, qreg[0] * kj[u]), vv[u], base + j + u, kj[u]);
                        else step(sc[u], vv[u], base + j + u);
                    }
                }
                for (; j < cnt; ++j) {
                    const int c = __shfl(cj, j, G);
                    float kj;
                    const float sc = scaled(pow2, score_of(a.k + row_off(c, ldk32) + hoff, kj));
                    float vv[VEC];
                    load_vec<VEC>(a.v + row_off(c, ldv32) + coff, vv);
                    step(sc, vv, base + j, kj);
                }
                l += ls;                 // fold the batch (see the declaration of l)
                ls = 0.0f;
            }
        }
        if (a.state_acc) {      // raw state of this part; the self-loop edge is added by the merge
            if (cvalid) {
                TFGX_STATE_STORE<VEC>(a.state_acc + part * a.W + coff, acc);
                if (coff % a.dv == 0) {
                    a.state_ml[part * 2 * a.H + 2 * head] = m;
                    a.state_ml[part * 2 * a.H + 2 * head + 1] = l_full();
                }
                if constexpr (QG && TFGX_GAT_QG_EXPERIMENT != 1) {
                    TFGX_STATE_STORE<VEC>(a.state_t + part * a.W + coff, acc_t);
                    if (coff % a.dv == 0) a.state_s[part * a.H + head] = s_k;
                }
            }
            r = idx;            // restore the loop variable
            continue;
        }
        if (a.add_self_loop) {  // the appended (r, r) edge comes last (graph_utils.py:350-366)
            float kj;
            const float dot = score_of(a.k + r * a.ldk + hoff, kj);
            const float sc = scaled(pow2, dot);
            float vv[VEC];
            load_vec<VEC>(a.v + r * a.ldv + coff, vv);
            step(sc, vv, a.drop.self_base + r, kj);
        }
        if (a.stats_ml && cvalid && (coff % a.dv == 0)) {
            a.stats_ml[r * 2 * a.H + 2 * head] = m;
            a.stats_ml[r * 2 * a.H + 2 * head + 1] = l_full();
        }
        if (cvalid) {
            const float den = l_full() + 1e-8f;   // segment.py:30
            float res[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float o = acc[i] / den;
                if (a.bias) o += a.bias[coff + i];
                res[i] = apply_act(o, a.act);
            }
            store_vec<VEC>(a.out + r * a.ldo + coff, res);
            if constexpr (QG) {
                float t_[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) t_[i] = acc_t[i] / den;
                store_vec<VEC>(a.qg_t + r * a.W + coff, t_);
                if (coff % a.dv == 0) a.qg_s[r * a.H + head] = s_k / den;
            }
        }
        r = idx;
    }
}

// Merge the raw states of a destination's parts (hub chunks: parts [part_ptr[i], part_ptr[i+1]); split-plan passes:
// parts {t * n_rows + i}), append the self-loop edge, normalise, bias, activation.  One thread per (row, column).
struct GMerge {
    const float* state_acc;
    const float* state_ml;
    const int32_t* rows;       // [n_merge] destination ids, or NULL (identity)
    const int32_t* part_ptr;   // [n_merge+1] or NULL (-> fixed_parts passes with stride n_rows)
    const int32_t* part_idx;   // optional with part_ptr: state index of the k-th part = part_idx[part_ptr[i] + k]
    int32_t fixed_parts;
    int64_t n_rows;            // stride between passes when part_ptr == NULL
    int64_t n_merge;
    const float* q; int64_t ldq;
    const float* k; int64_t ldk;
    const float* v; int64_t ldv;
    float* out; int64_t ldo;
    int32_t H, d, dv, W;
    int32_t add_self_loop;
    float scale;
    int32_t act;
    const float* bias;
    float* stats_ml;
};

__global__ __launch_bounds__(kBlock) void gat_merge_kernel(const GMerge g)
{
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t total = g.n_merge * g.W;
    for (; t < total; t += stride) {
        const int64_t i = t / g.W;
        const int j = int(t - i * g.W);
        const int h = j / g.dv;
        const int64_t r = g.rows ? g.rows[i] : i;
        const int np = g.part_ptr ? (g.part_ptr[i + 1] - g.part_ptr[i]) : g.fixed_parts;
        float s_self = -FLT_MAX;
        if (g.add_self_loop) {
            const float* qp = g.q + r * g.ldq + h * g.d;
            const float* kp = g.k + r * g.ldk + h * g.d;
            float dot = 0.0f;
            for (int u = 0; u < g.d; ++u) dot = fmaf(qp[u], kp[u], dot);
            s_self = dot / g.scale;
        }
        float M = s_self;
        for (int p = 0; p < np; ++p) {
            int64_t part = g.part_ptr ? (g.part_ptr[i] + p) : (int64_t(p) * g.n_rows + i);
            if (g.part_idx) part = g.part_idx[part];
            M = fmaxf(M, g.state_ml[part * 2 * g.H + 2 * h]);
        }
        float Lsum = 0.0f, O = 0.0f;
        for (int p = 0; p < np; ++p) {
            int64_t part = g.part_ptr ? (g.part_ptr[i] + p) : (int64_t(p) * g.n_rows + i);
            if (g.part_idx) part = g.part_idx[part];
            const float mp = g.state_ml[part * 2 * g.H + 2 * h];
            const float lp = g.state_ml[part * 2 * g.H + 2 * h + 1];
            const float c = (lp > 0.0f) ? expf(mp - M) : 0.0f;     // an empty part holds (m, l) = (-FLT_MAX, 0)
            Lsum = fmaf(lp, c, Lsum);
            O = fmaf(g.state_acc[part * g.W + j], c, O);
        }
        if (g.add_self_loop) {
            const float c = expf(s_self - M);
            Lsum += c;
            O = fmaf(c, g.v[r * g.ldv + j], O);
        }
        if (g.stats_ml && (j % g.dv == 0)) {
            g.stats_ml[r * 2 * g.H + 2 * h] = M;
            g.stats_ml[r * 2 * g.H + 2 * h + 1] = Lsum;
        }
        float o = O / (Lsum + 1e-8f);
        if (g.bias) o += g.bias[j];
        g.out[r * g.ldo + j] = apply_act(o, g.act);
    }
}

inline bool gat_k_split()           // developer A/B: TFGX_GAT_K_SPLIT=0 -> every lane of a head loads the head's whole K slice
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("TFGX_GAT_K_SPLIT");
        v = (e != nullptr && atoi(e) == 0) ? 0 : 1;
    }
    return v != 0;
}

template <int VEC, int G>
int launch_gat_d(const GArgs& a, hipStream_t stream)
{
    constexpr int ROWS_PER_BLOCK = kBlock / G;
    dim3 grid(grid_for(a.n_dst, ROWS_PER_BLOCK, 1 << 20), (a.W + G * VEC - 1) / (G * VEC), 1);
    dim3 block(kBlock, 1, 1);
    const bool pow2 = a.inv_scale != 0.0f;      // only d = 1, 4, 16 have the multiply instantiated; every other d divides
    const bool ks = gat_k_split() && a.d == a.dv && a.kvec && (a.d == 8 || a.d == 16) && (a.W / 4) % (a.d / 4) == 0;
    if (a.qg_t != nullptr || a.state_t != nullptr) {         // the entry point has checked d == 1 and VEC == 4
        if constexpr (VEC == 4) {
            if (pow2) gat_fused_kernel<VEC, G, 1, true, false, true><<<grid, block, 0, stream>>>(a);
            else gat_fused_kernel<VEC, G, 1, false, false, true><<<grid, block, 0, stream>>>(a);
            TFGX_LAUNCH_CHECK("gat_fused_kernel (query-gradient sums)");
            return TFGX_OK;
        } else {
            return TFGX_ERR_INVALID_ARG;
        }
    }
    switch (a.d) {
#define TFGX_GAT_POW2_CASE(D_)                                                                   \
    case D_:                                                                                     \
        if (pow2) gat_fused_kernel<VEC, G, D_, true><<<grid, block, 0, stream>>>(a);              \
        else gat_fused_kernel<VEC, G, D_, false><<<grid, block, 0, stream>>>(a);                  \
        break
        TFGX_GAT_POW2_CASE(1);
        case 2: gat_fused_kernel<VEC, G, 2><<<grid, block, 0, stream>>>(a); break;
        TFGX_GAT_POW2_CASE(4);
        case 8:
            if constexpr (VEC == 4) {
                if (ks) { gat_fused_kernel<VEC, G, 8, false, true><<<grid, block, 0, stream>>>(a); break; }
            }
            gat_fused_kernel<VEC, G, 8><<<grid, block, 0, stream>>>(a);
            break;
        case 16:
            if constexpr (VEC == 4) {
                if (ks && pow2) { gat_fused_kernel<VEC, G, 16, true, true><<<grid, block, 0, stream>>>(a); break; }
            }
            if (pow2) gat_fused_kernel<VEC, G, 16, true><<<grid, block, 0, stream>>>(a);
            else gat_fused_kernel<VEC, G, 16, false><<<grid, block, 0, stream>>>(a);
            break;
#undef TFGX_GAT_POW2_CASE
        case 32: gat_fused_kernel<VEC, G, 32><<<grid, block, 0, stream>>>(a); break;
        default: gat_fused_kernel<VEC, G, 0><<<grid, block, 0, stream>>>(a); break;
    }
    TFGX_LAUNCH_CHECK("gat_fused_kernel");
    return TFGX_OK;
}

template <int VEC>
int launch_gat(const GArgs& a, hipStream_t stream)
{
    const int lanes = (a.W + VEC - 1) / VEC;
    if (lanes <= 8) return launch_gat_d<VEC, 8>(a, stream);
    if (lanes <= 16) return launch_gat_d<VEC, 16>(a, stream);
    if (lanes <= 32) return launch_gat_d<VEC, 32>(a, stream);
    return launch_gat_d<VEC, 64>(a, stream);  // wider rows tile over grid.y
}

__global__ __launch_bounds__(kBlock) void head_mean_kernel(const float* __restrict__ in, int64_t ld_in, int64_t n,
                                                           int H, int U, const float* __restrict__ bias, int act,
                                                           float* __restrict__ out, int64_t ldo)
{
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t total = n * U;
    for (; t < total; t += stride) {
        const int64_t r = t / U;
        const int j = int(t - r * U);
        float s = 0.0f;
        for (int h = 0; h < H; ++h) s += in[r * ld_in + int64_t(h) * U + j];  // tf.add_n order (gat.py:114)
        s = s / float(H);
        if (bias) s += bias[j];
        out[r * ldo + j] = apply_act(s, act);
    }
}

}  // namespace
}  // namespace tfgx

using namespace tfgx;

extern "C" int tfgx_edge_softmax_f32(const int32_t* row_ptr, const int32_t* perm, const float* score, int64_t H,
                                     int64_t n_dst, float* out, tfgx_stream_t stream)
{
    return tfgx_edge_softmax_hub_f32(row_ptr, perm, score, H, n_dst, out, nullptr, nullptr, stream);
}

extern "C" int tfgx_edge_softmax_hub_f32(const int32_t* row_ptr, const int32_t* perm, const float* score, int64_t H,
                                         int64_t n_dst, float* out, const tfgx_hub_lists* hub, float* hub_scratch,
                                         tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(H >= 1 && n_dst >= 0, "bad H / n_dst");
    if (n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(row_ptr != nullptr, "row_ptr is null");
    hipStream_t st = as_stream(stream);
    if (H > 64) {
        edge_softmax_serial_kernel<<<grid_for(n_dst * H, kBlock), kBlock, 0, st>>>(row_ptr, perm, score, int(H), n_dst, out);
    } else {
        int Hp = 1;
        while (Hp < H) Hp <<= 1;
        const bool chunked = hub != nullptr && hub_scratch != nullptr && hub->threshold > 0 && hub->n_rows > 0 &&
                             hub->n_chunks > 0 && hub->chunk_ptr && hub->chunk_begin && hub->chunk_end;
        const int limit = chunked ? hub->threshold : kSoftmaxWide;
#define TFGX_SM(GG) edge_softmax_kernel<GG, false><<<grid_for(n_dst, kBlock / GG, 1 << 20), kBlock, 0, st>>>(row_ptr, perm, score, int(H), Hp, n_dst, out, limit)
        if (Hp <= 8) TFGX_SM(8);
        else if (Hp <= 16) TFGX_SM(16);
        else if (Hp <= 32) TFGX_SM(32);
        else TFGX_SM(64);
#undef TFGX_SM
        if (chunked) {       // hub_scratch: n_chunks * 2 * Hp floats
            edge_softmax_chunk_kernel<0><<<grid_for(hub->n_chunks * 64, kBlock), kBlock, 0, st>>>(
                hub->chunk_begin, hub->chunk_end, hub->n_chunks, perm, score, int(H), Hp, hub_scratch, out);
            edge_softmax_hub_combine_kernel<<<grid_for(hub->n_rows * Hp, kBlock), kBlock, 0, st>>>(hub->chunk_ptr, hub->n_rows,
                                                                                               Hp, hub_scratch);
            edge_softmax_chunk_kernel<1><<<grid_for(hub->n_chunks * 64, kBlock), kBlock, 0, st>>>(
                hub->chunk_begin, hub->chunk_end, hub->n_chunks, perm, score, int(H), Hp, hub_scratch, out);
        } else {
            // no chunk lists: long rows get a whole 256-thread workgroup each (a pass over row_ptr when there are none)
            edge_softmax_kernel<kBlock, true><<<grid_for(n_dst, 1, (1 << 14) - 3), kBlock, 0, st>>>(row_ptr, perm, score, int(H), Hp,
                                                                                             n_dst, out, limit);
        }
    }
    TFGX_LAUNCH_CHECK("edge_softmax_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_gat_fused_f32(const tfgx_gat_args* p, tfgx_stream_t stream_)
{
    TFGX_RANGE();
    TFGX_REQUIRE(p != nullptr, "args is null");
    TFGX_REQUIRE(p->H >= 1 && p->d >= 1 && p->dv >= 1 && p->n_dst >= 0, "bad H / d / dv / n_dst");
    TFGX_REQUIRE(p->scale > 0.0f, "scale must be positive");
    if (p->n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE((p->row_ptr || p->row_begin) && p->q && p->k && p->v && (p->out || p->state_acc), "null pointer");
    const int64_t W = int64_t(p->H) * p->dv, A = int64_t(p->H) * p->d;
    TFGX_REQUIRE(p->ldq >= A && p->ldk >= A && p->ldv >= W && (p->state_acc || p->ldo >= W),
                 "leading dimension too small");
    TFGX_REQUIRE(p->ldk < (int64_t(1) << 31) && p->ldv < (int64_t(1) << 31), "ldk / ldv must be below 2^31 elements");
    GArgs a;
    a.row_ptr = p->row_ptr; a.col = p->col; a.n_dst = p->n_dst;
    a.q = p->q; a.ldq = p->ldq; a.k = p->k; a.ldk = p->ldk; a.v = p->v; a.ldv = p->ldv;
    a.out = p->out; a.ldo = p->ldo; a.H = p->H; a.d = p->d; a.dv = p->dv; a.W = int32_t(W);
    a.add_self_loop = p->add_self_loop; a.scale = p->scale; a.act = p->act; a.bias = p->bias;
    a.inv_scale = exact_inverse_or_zero(p->scale);
    a.kvec = (p->d % 4 == 0) && (p->ldk % 4 == 0) && aligned_to(p->k, 16);
    a.row_begin = p->row_begin ? p->row_begin : p->row_ptr;
    a.row_end = p->row_begin ? p->row_end : p->row_ptr + 1;
    a.rp_stride = p->row_begin ? p->rp_stride : 1;
    TFGX_REQUIRE(a.row_end != nullptr && a.rp_stride >= 1, "bad row_begin / row_end / rp_stride");
    a.part_row = nullptr; a.state_acc = p->state_acc; a.state_ml = p->state_ml; a.hub_threshold = 0;
    a.state_in_acc = p->state_in_acc; a.state_in_ml = p->state_in_ml;
    TFGX_REQUIRE((p->state_in_acc == nullptr) == (p->state_in_ml == nullptr), "state_in_acc and state_in_ml go together");
    TFGX_REQUIRE(p->state_in_acc == nullptr || (p->state_in_acc != p->state_acc && p->state_in_ml != p->state_ml),
                 "the resumed state must not alias the state being written");
    TFGX_REQUIRE(p->state_in_acc == nullptr || (p->drop_rate == 0.0f && !(p->hub_threshold > 0 && p->n_hub_rows > 0 && p->state_acc == nullptr)),
                 "state_in cannot be combined with attention dropout or the hub lists");
    a.stats_ml = p->stats_ml;
    a.qg_t = p->qgrad_t; a.qg_s = p->qgrad_s; a.state_t = p->state_t; a.state_s = p->state_s;
    a.state_in_t = p->state_in_t; a.state_in_s = p->state_in_s;
    const bool qg = p->qgrad_t || p->qgrad_s || p->state_t || p->state_s || p->state_in_t || p->state_in_s;
    if (qg) {
        TFGX_REQUIRE(p->d == 1, "the query-gradient sums (qgrad_t / state_t) exist for d == 1 only");
        TFGX_REQUIRE(!(p->hub_threshold > 0 && p->n_hub_rows > 0 && p->state_acc == nullptr),
                     "the query-gradient sums cannot be combined with the hub lists");
        if (p->state_acc) TFGX_REQUIRE(p->state_t && p->state_s && !p->qgrad_t && !p->qgrad_s,
                                       "a raw-state launch carries the sums in state_t / state_s");
        else TFGX_REQUIRE(p->qgrad_t && p->qgrad_s && !p->state_t && !p->state_s, "a finishing launch writes qgrad_t and qgrad_s");
        TFGX_REQUIRE((p->state_in_acc != nullptr) == (p->state_in_t != nullptr) && (p->state_in_t != nullptr) == (p->state_in_s != nullptr),
                     "state_in_t / state_in_s go with state_in_acc");
        TFGX_REQUIRE(p->state_in_t == nullptr || (p->state_in_t != p->state_t && p->state_in_s != p->state_s),
                     "the resumed sums must not alias the sums being written");
        TFGX_REQUIRE(aligned_to(p->qgrad_t, 16) && aligned_to(p->state_t, 16) && aligned_to(p->state_in_t, 16),
                     "qgrad_t / state_t must be 16-byte aligned");
    }
    a.row_order = (p->state_acc == nullptr && p->row_begin == nullptr) ? p->row_order : nullptr;
    if (p->state_acc) {
        // raw-state launches over arbitrary PARTS (tfgx.h): hub_chunk_row, when given, names the destination (Q row) of
        // every launched part; hub_threshold > 0 skips spans longer than that (their chunks are launched separately)
        a.part_row = p->hub_chunk_row;
        a.hub_threshold = p->hub_threshold > 0 ? p->hub_threshold : 0;
    }
    TFGX_REQUIRE(p->drop_rate >= 0.0f && p->drop_rate < 1.0f, "drop_rate outside [0, 1)");
    TFGX_REQUIRE(p->drop_rate == 0.0f || (p->state_acc == nullptr && !(p->hub_threshold > 0 && p->n_hub_rows > 0)),
                 "attention dropout cannot be combined with the raw-state / hub options");
    a.drop = make_drop(p->drop_rate, p->drop_seed, p->drop_self_base, p->drop_seed_dev);
    TFGX_REQUIRE((p->state_acc == nullptr) == (p->state_ml == nullptr), "state_acc and state_ml go together");
    const bool use_hub = p->hub_threshold > 0 && p->n_hub_rows > 0 && p->state_acc == nullptr;
    if (use_hub) {
        TFGX_REQUIRE(p->hub_rows && p->hub_chunk_ptr && p->hub_chunk_begin && p->hub_chunk_end && p->hub_chunk_row &&
                         p->hub_scratch_acc && p->hub_scratch_ml && p->n_hub_chunks > 0,
                     "hub rows given without chunk lists / scratch");
        a.hub_threshold = p->hub_threshold;
    }
    hipStream_t stream = as_stream(stream_);
    auto vec_for = [&](const float* outp, int64_t ldo_) {
        auto ok = [&](int vec) {
            const size_t al = sizeof(float) * vec;
            return (p->dv % vec == 0) && (p->ldv % vec == 0) && (ldo_ % vec == 0) && aligned_to(p->v, al) &&
                   aligned_to(outp, al);
        };
        return ok(4) ? 4 : (ok(2) ? 2 : 1);
    };
    auto launch = [&](const GArgs& g, int vec) {
        if (vec == 4) return launch_gat<4>(g, stream);
        if (vec == 2) return launch_gat<2>(g, stream);
        return launch_gat<1>(g, stream);
    };
    const int vec_main = p->state_acc ? vec_for(p->state_acc, W) : vec_for(p->out, p->ldo);
    TFGX_REQUIRE(!qg || vec_main == 4, "the query-gradient sums need dv % 4 == 0 and 16-byte aligned V / out rows");
    int rc = launch(a, vec_main);
    if (rc != TFGX_OK || !use_hub) return rc;

    // hub rows: raw state per chunk, then ordered merge (+ self loop, bias, activation)
    GArgs c = a;
    c.row_begin = p->hub_chunk_begin; c.row_end = p->hub_chunk_end; c.rp_stride = 1; c.n_dst = p->n_hub_chunks;
    c.part_row = p->hub_chunk_row; c.state_acc = p->hub_scratch_acc; c.state_ml = p->hub_scratch_ml;
    c.hub_threshold = 0; c.stats_ml = nullptr; c.row_order = nullptr;
    rc = launch(c, vec_for(p->hub_scratch_acc, W));
    if (rc != TFGX_OK) return rc;
    GMerge g;
    g.state_acc = p->hub_scratch_acc; g.state_ml = p->hub_scratch_ml; g.rows = p->hub_rows;
    g.part_ptr = p->hub_chunk_ptr; g.part_idx = nullptr; g.fixed_parts = 0; g.n_rows = 0; g.n_merge = p->n_hub_rows;
    g.q = p->q; g.ldq = p->ldq; g.k = p->k; g.ldk = p->ldk; g.v = p->v; g.ldv = p->ldv;
    g.out = p->out; g.ldo = p->ldo; g.H = p->H; g.d = p->d; g.dv = p->dv; g.W = int32_t(W);
    g.add_self_loop = p->add_self_loop; g.scale = p->scale; g.act = p->act; g.bias = p->bias;
    g.stats_ml = p->stats_ml;
    gat_merge_kernel<<<grid_for(g.n_merge * W, kBlock), kBlock, 0, stream>>>(g);
    TFGX_LAUNCH_CHECK("gat_merge_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_gat_merge_passes_f32(const tfgx_gat_args* p, const float* state_acc, const float* state_ml,
                                         int32_t n_passes, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(p != nullptr && state_acc && state_ml && n_passes >= 1, "bad argument");
    TFGX_REQUIRE(p->H >= 1 && p->d >= 1 && p->dv >= 1 && p->n_dst >= 0 && p->scale > 0.0f, "bad H / d / dv / n_dst");
    TFGX_REQUIRE(p->drop_rate == 0.0f, "attention dropout is not available on merged passes");
    if (p->n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(p->q && p->k && p->v && p->out, "null pointer");
    const int64_t W = int64_t(p->H) * p->dv;
    GMerge g;
    g.state_acc = state_acc; g.state_ml = state_ml; g.rows = nullptr; g.part_ptr = nullptr; g.part_idx = nullptr;
    g.fixed_parts = n_passes; g.n_rows = p->n_dst; g.n_merge = p->n_dst;
    g.q = p->q; g.ldq = p->ldq; g.k = p->k; g.ldk = p->ldk; g.v = p->v; g.ldv = p->ldv;
    g.out = p->out; g.ldo = p->ldo; g.H = p->H; g.d = p->d; g.dv = p->dv; g.W = int32_t(W);
    g.add_self_loop = p->add_self_loop; g.scale = p->scale; g.act = p->act; g.bias = p->bias;
    g.stats_ml = p->stats_ml;
    gat_merge_kernel<<<grid_for(g.n_merge * W, kBlock), kBlock, 0, as_stream(stream)>>>(g);
    TFGX_LAUNCH_CHECK("gat_merge_kernel");
    return TFGX_OK;
}

extern "C" int tfgx_gat_merge_parts_f32(const tfgx_gat_args* p, const float* state_acc, const float* state_ml,
                                        const int32_t* part_ptr, const int32_t* part_idx, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(p != nullptr && state_acc && state_ml && part_ptr && part_idx, "bad argument");
    TFGX_REQUIRE(p->H >= 1 && p->d >= 1 && p->dv >= 1 && p->n_dst >= 0 && p->scale > 0.0f, "bad H / d / dv / n_dst");
    TFGX_REQUIRE(p->drop_rate == 0.0f, "attention dropout is not available on merged parts");
    if (p->n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(p->q && p->k && p->v && p->out, "null pointer");
    const int64_t W = int64_t(p->H) * p->dv;
    GMerge g;
    g.state_acc = state_acc; g.state_ml = state_ml; g.rows = nullptr; g.part_ptr = part_ptr; g.part_idx = part_idx;
    g.fixed_parts = 0; g.n_rows = p->n_dst; g.n_merge = p->n_dst;
    g.q = p->q; g.ldq = p->ldq; g.k = p->k; g.ldk = p->ldk; g.v = p->v; g.ldv = p->ldv;
    g.out = p->out; g.ldo = p->ldo; g.H = p->H; g.d = p->d; g.dv = p->dv; g.W = int32_t(W);
    g.add_self_loop = p->add_self_loop; g.scale = p->scale; g.act = p->act; g.bias = p->bias;
    g.stats_ml = p->stats_ml;
    gat_merge_kernel<<<grid_for(g.n_merge * W, kBlock), kBlock, 0, as_stream(stream)>>>(g);
    TFGX_LAUNCH_CHECK("gat_merge_kernel");
    return TFGX_OK;
}

extern "C" int32_t tfgx_dropout_keep(uint64_t seed, uint32_t item, float rate)
{
    return drop_scale(make_drop(rate, seed, 0), item) != 0.0f ? 1 : 0;
}

extern "C" int tfgx_head_mean_f32(const float* in, int64_t ld_in, int64_t n, int32_t H, int32_t U,
                                  const float* bias, int32_t act, float* out, int64_t ldo, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n >= 0 && H >= 1 && U >= 1, "bad size");
    if (n == 0) return TFGX_OK;
    TFGX_REQUIRE(in && out && ld_in >= int64_t(H) * U && ldo >= U, "bad pointer / leading dimension");
    head_mean_kernel<<<grid_for(n * U, kBlock), kBlock, 0, as_stream(stream)>>>(in, ld_in, n, H, U, bias, act, out,
                                                                               ldo);
    TFGX_LAUNCH_CHECK("head_mean_kernel");
    return TFGX_OK;
}
