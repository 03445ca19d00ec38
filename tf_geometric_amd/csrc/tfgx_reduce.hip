// Gather - scale - segment reduce over a CSR-by-destination plan (the hot kernel).
//
// Replaces tf.gather(x, col) -> gcn_mapper -> tf.math.unsorted_segment_{sum,mean,max}
// (reference: tf_geometric/nn/kernel/map_reduce.py:60-70, :15-42; nn/conv/gcn.py:221-222, :280)
// without ever materialising the [E, F] message tensor.
//
// Mapping (wave = 64 lanes): a destination row is owned by a GROUP of G lanes (G = 4..64, a power of
// two, picked from F), so a wave owns 64/G consecutive destination rows and a 256-thread workgroup
// 256/G of them.  Each lane owns VEC consecutive feature columns per chunk (16-byte loads when the
// layout allows).  Per row the group reads G (col, w) pairs at a time, coalesced, one per lane, then
// walks them: the pair is broadcast inside the group with a lane shuffle (G = 64: v_readlane, which
// also makes the row base address scalar) and every lane issues one VEC-wide load of x[col] — the
// source row is read as one contiguous 4*F-byte burst.  The walk is unrolled by UNROLL edges so that
// UNROLL independent loads are in flight per lane before the first FMA.  Accumulation is a single
// in-order FMA chain per output element: deterministic, no atomics, original edge order per row.
//
// WIDE rows (round 5, F >= 128 with 16-byte rows): the row is NOT read as one burst.  It is cut into column blocks of
// 64 columns (256 bytes; 128 columns at F = 256) on blockIdx.y: every launch pass gathers one block of every source row, the
// blocks of a row are requested far apart in time.  Same-box A/B over every placement (tools/wide_row_ab.cpp,
// profiles/r05_wide_row_ab*.jsonl): one 1 KB / 2 KB burst per row 20.0 / 43.5 ms at F = 256 / 512, the same blocks
// requested at the same time by sibling waves (same workgroup, or workgroups on the same XCD) no better — the blocks on
// grid.y 18.5 / 37.1 ms; EA read latency 2175 -> 1640 cycles, DRAM-credit stalls per request 0.38 -> 0.11 (pmc passes in
// profiles/r05_wide_pmc.md).  The (col, w) stream is re-read per block (8 bytes against 256).
#include "tfgx_common.h"
#include <cfloat>
#include <cstdio>
#include <cstdlib>

namespace tfgx {
namespace {

template <int G>
__device__ __forceinline__ int bcast_i(int v, int j)
{
    if constexpr (G == 64) return __builtin_amdgcn_readlane(v, j);
    else return __shfl(v, j, G);
}
template <int G>
__device__ __forceinline__ float bcast_f(float v, int j)
{
    if constexpr (G == 64) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
    else return __shfl(v, j, G);
}

struct KArgs {
    const int32_t* row_begin;
    const int32_t* row_end;
    int64_t rp_stride;
    const int32_t* col;
    const float* w;
    int64_t n_dst;
    const float* x;
    int64_t ldx;
    int32_t F;
    int32_t col0;     // first column handled by this launch's blockIdx.y == 0
    float* out;
    int64_t ldo;
    int32_t op, act, accumulate;
    const float* self_coef;
    const float* bias;
    const float* add_x;
    int64_t ld_add;
    const int32_t* mean_count;
    int32_t hub_threshold;   // > 0: rows with more edges than this are left to the hub path
    const float* x_tail;     // split source rows: columns >= f_main live in x_tail[n_src, ld_tail] (see tfgx.h)
    int64_t ld_tail;
    int32_t f_main;
    const int32_t* row_order; // optional walk order: lane group i reduces row row_order[i] (skewed plans: degree order)
    uint32_t* track;          // TFGX_MAX training forward: (tie count << 16 | position of the first maximal edge in its row)
    int64_t ld_track;
    const int32_t* track_row_begin;   // optional: positions are stored relative to track_row_begin[row * rp_stride]
    const float* edge_tail;  // optional with SPLIT: the tail columns of every edge's SOURCE row, in this plan's edge order
    int64_t ld_edge_tail;    // (streamed next to col / w instead of gathered: one line request fewer per edge)
    int32_t wide_blocks;     // tfgx_reduce_args.wide_blocks: 0 = library policy, 1 = column blocks, -1 = one burst per row
};

#ifndef TFGX_REDUCE_GRID_CAP_DEFAULT
#define TFGX_REDUCE_GRID_CAP_DEFAULT (1 << 20)
#endif
#ifndef TFGX_REDUCE_MASKED_TAIL
#define TFGX_REDUCE_MASKED_TAIL 1
#endif
#ifndef TFGX_UNROLL_CH2
#define TFGX_UNROLL_CH2 4
#endif
#ifndef TFGX_UNROLL_CH4
#define TFGX_UNROLL_CH4 2
#endif

template <int VEC, int G, int CH, bool IS_MAX, bool WEIGHTED, bool SPLIT, bool TRACK, int U>
__device__ __forceinline__ void seg_reduce_row(const KArgs& a, int64_t r, int s, int e, int cj_next, float wj_next, int lane,
                                               const int (&coff)[CH], const bool (&cvalid)[CH], const float* const (&xb)[CH],
                                               const int64_t (&xl)[CH], const float* const (&xs)[CH],
                                               const int64_t (&xsl)[CH], const bool (&by_edge)[CH], float init);

// TRACK (IS_MAX only; the TRAINING forward of max aggregation): next to the running maximum every lane keeps, per column,
// how many edges attain it (TF's unsorted_segment_max gradient divides by that count) and the position of the FIRST one
// inside its row, and the epilogue writes them packed into ONE uint32 per element (count << 16 | position): the mask-form
// backward reads that instead of a float count array and an int32 position array (rows up to 65535 edges: longer rows are
// hub rows and take the chunked path, which does not track).
// U: edges per batch (0 = the default for the group shape: 8, 4 / 2 with several column chunks per lane).
template <int VEC, int G, int CH, bool IS_MAX, bool WEIGHTED, bool SPLIT = false, bool TRACK = false, int U = 0>
__global__ __launch_bounds__(kBlock) void seg_reduce_kernel(const KArgs a)
{
    constexpr int ROWS_PER_BLOCK = kBlock / G;
    constexpr int COLS_PER_PASS = G * VEC * CH;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    // column blocks on grid.y (wide rows): the launch works through the passes in order, so at any moment every workgroup
    // gathers the SAME column block of its source rows — the table a pass touches is 1 / gridDim.y of the whole (a 614 MB stripe
    // of a 4.9 GB table at F = 512).  That is what the gain comes from: the same blocks ROTATED over the workgroups
    // ((blockIdx.x + blockIdx.y) mod gridDim.y, all stripes live at once) run no faster than one burst per row
    // (profiles/r05_ab_wide_blocks_modes_uniform.jsonl: 40.0 vs 47.8 vs 46.1 ms at F = 512).
    const int colbase = a.col0 + blockIdx.y * COLS_PER_PASS;

    // column offsets of this lane; lanes past F re-read the last valid vector (branch-free, discarded)
    int coff[CH];
    bool cvalid[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int c = colbase + (k * G + lane) * VEC;
        cvalid[k] = c < a.F;
        coff[k] = cvalid[k] ? c : (a.F - VEC);
    }
    // per-lane source base / stride: one array normally; with SPLIT the lanes owning columns >= f_main read the
    // narrow tail array (whole 128-byte lines of the main array carry no unused bytes then)
    const float* xb[CH];     // base / stride of the rows this lane gathers (indexed by source id, or by edge position
    int64_t xl[CH];          // when by_edge[k])
    const float* xs[CH];     // base / stride of the lane's slice of a NODE's row (self-loop term of the epilogue)
    int64_t xsl[CH];
    bool by_edge[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        xb[k] = a.x + coff[k];
        xl[k] = a.ldx;
        by_edge[k] = false;
        if constexpr (SPLIT) {
            if (coff[k] >= a.f_main) {
                xb[k] = a.x_tail + (coff[k] - a.f_main);
                xl[k] = a.ld_tail;
            }
        }
        xs[k] = xb[k];
        xsl[k] = xl[k];
        if constexpr (SPLIT) {
            if (coff[k] >= a.f_main && a.edge_tail != nullptr) {
                xb[k] = a.edge_tail + (coff[k] - a.f_main);
                xl[k] = a.ld_edge_tail;
                by_edge[k] = true;
            }
        }
    }
    const float init = IS_MAX ? -FLT_MAX : 0.0f;

    // A lane group walks SEVERAL rows (the launch caps the grid, launch_cfg): the header chain of a row — row_ptr ->
    // first (col, w) batch -> first gathered rows — is software-pipelined across rows.  While row r is reduced, the
    // first (col, w) batch of row r + stride and the (begin, end) pair of row r + 2*stride are already in flight, so a
    // new row starts with its indices in registers instead of two dependent memory round trips.
    const int64_t rstride = int64_t(gridDim.x) * ROWS_PER_BLOCK;
    int64_t r = int64_t(blockIdx.x) * ROWS_PER_BLOCK + grp;
    // loop position -> destination row: the identity, or the plan's degree order on skewed graphs (rows of similar length
    // then share a wave; results do not depend on it)
    auto row_of = [&](int64_t i) -> int64_t { return a.row_order ? int64_t(a.row_order[i]) : i; };
    int s = 0, e = 0, s1 = 0, e1 = 0;
    if (r < a.n_dst) {
        const int64_t q = row_of(r);
        s = a.row_begin[q * a.rp_stride];
        e = a.row_end[q * a.rp_stride];
    }
    if (r + rstride < a.n_dst) {
        const int64_t q = row_of(r + rstride);
        s1 = a.row_begin[q * a.rp_stride];
        e1 = a.row_end[q * a.rp_stride];
    }
    int cj_first = 0;
    float wj_first = 0.0f;
    if (s + lane < e) {
        cj_first = a.col[s + lane];
        if constexpr (WEIGHTED) wj_first = a.w[s + lane];
    }
    for (; r < a.n_dst; r += rstride) {
        int s2 = 0, e2 = 0;                                       // header of the row after next
        if (r + 2 * rstride < a.n_dst) {
            const int64_t q = row_of(r + 2 * rstride);
            s2 = a.row_begin[q * a.rp_stride];
            e2 = a.row_end[q * a.rp_stride];
        }
        int cj_first1 = 0;                                        // first (col, w) batch of the next row
        float wj_first1 = 0.0f;
        if (s1 + lane < e1) {
            cj_first1 = a.col[s1 + lane];
            if constexpr (WEIGHTED) wj_first1 = a.w[s1 + lane];
        }
        const int s_cur = G == 64 ? __builtin_amdgcn_readfirstlane(s) : s;
        const int e_cur = G == 64 ? __builtin_amdgcn_readfirstlane(e) : e;
        int cj_next = cj_first;
        float wj_next = wj_first;
        s = s1; e = e1; s1 = s2; e1 = e2; cj_first = cj_first1; wj_first = wj_first1;      // rotate the pipeline
        if (a.hub_threshold > 0 && e_cur - s_cur > a.hub_threshold) continue;   // handled by the chunked hub path
        seg_reduce_row<VEC, G, CH, IS_MAX, WEIGHTED, SPLIT, TRACK, U>(a, row_of(r), s_cur, e_cur, cj_next, wj_next, lane, coff,
                                                                      cvalid, xb, xl, xs, xsl, by_edge, init);
    }
}

// One destination row [s, e) of the plan, reduced by a group of G lanes (body of seg_reduce_kernel).
template <int VEC, int G, int CH, bool IS_MAX, bool WEIGHTED, bool SPLIT, bool TRACK, int U>
__device__ __forceinline__ void seg_reduce_row(const KArgs& a, int64_t r, int s, int e, int cj_next, float wj_next, int lane,
                                               const int (&coff)[CH], const bool (&cvalid)[CH], const float* const (&xb)[CH],
                                               const int64_t (&xl)[CH], const float* const (&xs)[CH],
                                               const int64_t (&xsl)[CH], const bool (&by_edge)[CH], float init)
{
    constexpr int UNROLL_W = U > 0 ? U : ((CH >= 4) ? TFGX_UNROLL_CH4 : (CH == 2 ? TFGX_UNROLL_CH2 : 8));
    constexpr int UNROLL = UNROLL_W < G ? UNROLL_W : G;
    {
        float acc[CH][VEC];
        int tcnt[TRACK ? CH : 1][TRACK ? VEC : 1], tpos[TRACK ? CH : 1][TRACK ? VEC : 1];
#pragma unroll
        for (int k = 0; k < CH; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[k][v] = init;
                if constexpr (TRACK) { tcnt[k][v] = 0; tpos[k][v] = -1; }
            }
        // online tie count / first position (identical to the MODE 2 walk of tfgx_backward.hip): a value above the running
        // maximum restarts the count at 1 and moves the position, an equal one increments the count
        auto track_step = [&](int k, int v, float m, int p) {
            if constexpr (TRACK) {
                const bool gt = m > acc[k][v];
                tpos[k][v] = (gt || tpos[k][v] < 0) ? p : tpos[k][v];
                tcnt[k][v] = gt ? 1 : (m == acc[k][v] ? tcnt[k][v] + 1 : tcnt[k][v]);
            }
        };

        // (col, w) of the NEXT batch are loaded while the current batch's rows are in flight: the index load heads every
        // gather's dependency chain (same-box A/B: 4-6 % at F <= 64, 2 % on a 57 GB table, neutral at F = 100)
        for (int base = s; base < e; base += G) {
            const int cj = cj_next;
            const float wj = wj_next;
            const int nxt = base + G + lane;
            if (nxt < e) {
                cj_next = a.col[nxt];
                if constexpr (WEIGHTED) wj_next = a.w[nxt];
            }
            const int cnt = min(G, e - base);
            int j = 0;
            for (; j + UNROLL <= cnt; j += UNROLL) {
                float xv[UNROLL][CH][VEC];
                float ww[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int c = bcast_i<G>(cj, j + u);
                    if constexpr (WEIGHTED) ww[u] = bcast_f<G>(wj, j + u);
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        if constexpr (SPLIT) load_vec<VEC>(xb[k] + (by_edge[k] ? int64_t(base + j + u) : int64_t(c)) * xl[k], xv[u][k]);
                        else load_vec<VEC>(a.x + int64_t(c) * a.ldx + coff[k], xv[u][k]);
                    }
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                    for (int k = 0; k < CH; ++k)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            if constexpr (IS_MAX) {
                                const float m = WEIGHTED ? xv[u][k][v] * ww[u] : xv[u][k][v];
                                track_step(k, v, m, base + j + u);
                                acc[k][v] = fmaxf(acc[k][v], m);
                            } else {
                                acc[k][v] = WEIGHTED ? fmaf(ww[u], xv[u][k][v], acc[k][v]) : acc[k][v] + xv[u][k][v];
                            }
                        }
            }
            // (narrow rows, G < 16, keep the one-load-per-edge remainder: they are bound by instruction issue, and the
            // repeated loads of the masked batch cost 4 % at F = 32 — profiles/r05_ab_masked_tail.jsonl)
            constexpr bool kMaskedTail = TFGX_REDUCE_MASKED_TAIL && G >= 16;
            if constexpr (!kMaskedTail) {
            for (; j < cnt; ++j) {          // developer A/B (-DTFGX_REDUCE_MASKED_TAIL=0): the round 1-4 remainder, one dependent load per edge
                const int c = bcast_i<G>(cj, j);
                float wv = 1.0f;
                if constexpr (WEIGHTED) wv = bcast_f<G>(wj, j);
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    float xv[VEC];
                    if constexpr (SPLIT) load_vec<VEC>(xb[k] + (by_edge[k] ? int64_t(base + j) : int64_t(c)) * xl[k], xv);
                    else load_vec<VEC>(a.x + int64_t(c) * a.ldx + coff[k], xv);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        if constexpr (IS_MAX) {
                            const float m = WEIGHTED ? xv[v] * wv : xv[v];
                            track_step(k, v, m, base + j);
                            acc[k][v] = fmaxf(acc[k][v], m);
                        } else {
                            acc[k][v] = WEIGHTED ? fmaf(wv, xv[v], acc[k][v]) : acc[k][v] + xv[v];
                        }
                    }
                }
            }
            } else {
            if (j < cnt) {
                // the last, partial batch as ONE batch (round 5): the loads of the missing slots repeat the last edge's
                // (clamped index: the same lines, an L1 hit), the arithmetic of a missing slot is dropped by a SELECT — a
                // branch there lets the compiler sink the slot's load into it and wait for it alone.  Until round 4 the
                // remainder ran one dependent load per edge: up to UNROLL - 1 round trips per row, all of a short row's.
                float xv[UNROLL][CH][VEC];
                float ww[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int idx = min(j + u, cnt - 1);
                    const int c = bcast_i<G>(cj, idx);
                    if constexpr (WEIGHTED) ww[u] = bcast_f<G>(wj, idx);
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        if constexpr (SPLIT) load_vec<VEC>(xb[k] + (by_edge[k] ? int64_t(base + idx) : int64_t(c)) * xl[k], xv[u][k]);
                        else load_vec<VEC>(a.x + int64_t(c) * a.ldx + coff[k], xv[u][k]);
                    }
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const bool live = j + u < cnt;
#pragma unroll
                    for (int k = 0; k < CH; ++k)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            if constexpr (IS_MAX) {
                                const float m = WEIGHTED ? xv[u][k][v] * ww[u] : xv[u][k][v];
                                if constexpr (TRACK) {
                                    if (live) track_step(k, v, m, base + j + u);
                                }
                                const float t = fmaxf(acc[k][v], m);
                                acc[k][v] = live ? t : acc[k][v];
                            } else {
                                const float t = WEIGHTED ? fmaf(ww[u], xv[u][k][v], acc[k][v]) : acc[k][v] + xv[u][k][v];
                                acc[k][v] = live ? t : acc[k][v];
                            }
                        }
                }
            }
            }
        }

        // ---- epilogue (per destination row) ----
        const float sc = a.self_coef ? a.self_coef[r] : 0.0f;
        float divisor = 1.0f;
        if (a.op == TFGX_MEAN) {
            const int cnt = a.mean_count ? a.mean_count[r] : (e - s);
            divisor = float(cnt > 1 ? cnt : 1);
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (!cvalid[k]) continue;
            float res[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) res[v] = acc[k][v];
            float* op = a.out + r * a.ldo + coff[k];
            float mine[TRACK ? VEC : 1], before[TRACK ? VEC : 1];      // TRACK + accumulate: this pass's maxima / the stored ones
            if (a.accumulate) {
                float prev[VEC];
                load_vec<VEC>(op, prev);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    if constexpr (TRACK) { mine[v] = res[v]; before[v] = prev[v]; }
                    res[v] = IS_MAX ? fmaxf(prev[v], res[v]) : prev[v] + res[v];
                }
            }
            if (a.self_coef) {
                float xself[VEC];
                if constexpr (SPLIT) load_vec<VEC>(xs[k] + r * xsl[k], xself);
                else load_vec<VEC>(a.x + r * a.ldx + coff[k], xself);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    if constexpr (IS_MAX) res[v] = fmaxf(res[v], sc * xself[v]);
                    else res[v] = fmaf(sc, xself[v], res[v]);
                }
            }
            if (a.op == TFGX_MEAN) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) res[v] = res[v] / divisor;
            }
            if (a.add_x) {
                float xa[VEC];
                load_vec<VEC>(a.add_x + r * a.ld_add + coff[k], xa);
#pragma unroll
                for (int v = 0; v < VEC; ++v) res[v] = xa[v] + res[v];
            }
            if (a.bias) {
                float b[VEC];
                load_vec<VEC>(a.bias + coff[k], b);
#pragma unroll
                for (int v = 0; v < VEC; ++v) res[v] += b[v];
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) res[v] = apply_act(res[v], a.act);
            store_vec<VEC>(op, res);
            if constexpr (TRACK) {
                // positions are relative to the row's first position: of THIS launch's span, or — when the row is reduced in
                // several launches over consecutive sub-spans (the sharded path: own-source edges, then one sub-span per
                // halo round) — of the whole row (track_row_begin)
                const int s_rel = a.track_row_begin ? a.track_row_begin[r * a.rp_stride] : s;
                uint32_t pk[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const uint32_t c = uint32_t(tcnt[k][v] < 65535 ? tcnt[k][v] : 65535);
                    pk[v] = (c << 16) | (tpos[k][v] < 0 ? 0xFFFFu : uint32_t(tpos[k][v] - s_rel) & 0xFFFFu);
                }
                uint32_t* tp = a.track + r * a.ld_track + coff[k];
                if (a.accumulate) {
                    // merge with what earlier launches stored for this row: a larger maximum replaces (count, position), an
                    // equal one adds its tie count and keeps the EARLIER position (sub-spans come in position order)
                    uint32_t old[VEC];
                    if constexpr (VEC == 4) {
                        const uint4 o4 = *reinterpret_cast<const uint4*>(tp);
                        old[0] = o4.x; old[1] = o4.y; old[2] = o4.z; old[3] = o4.w;
                    } else {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) old[v] = tp[v];
                    }
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const uint32_t oc = old[v] >> 16, nc = pk[v] >> 16;
                        const uint32_t sum = oc + nc < 65535u ? oc + nc : 65535u;
                        const bool none_before = (old[v] & 0xFFFFu) == 0xFFFFu && oc == 0u;
                        if (mine[v] > before[v] || none_before) { /* keep pk[v] */ }
                        else if (mine[v] == before[v] && nc > 0u) pk[v] = (sum << 16) | (old[v] & 0xFFFFu);
                        else pk[v] = old[v];
                    }
                }
                if constexpr (VEC == 4) *reinterpret_cast<uint4*>(tp) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) tp[v] = pk[v];
                }
            }
        }
    }
}

// Workgroups per launch: enough to fill 256 CUs x 8 XCDs at full occupancy several times over, few enough that every
// lane group walks several rows and the row-header pipeline of seg_reduce_kernel has something to overlap.
// TFGX_REDUCE_GRID_CAP overrides (developer A/B, profiles/r02_ab_grid_cap.jsonl).
// developer experiment: TFGX_REDUCE_DUMMY_LDS=<bytes> makes every launch reserve that much (unused) dynamic LDS, which caps
// the workgroups per CU — how does the gather rate depend on the waves in flight? (profiles/r03_occupancy_probe.jsonl)
inline size_t dummy_lds_bytes()
{
    static long v = -1;
    if (v < 0) {
        const char* e = getenv("TFGX_REDUCE_DUMMY_LDS");
        v = (e != nullptr && atol(e) > 0) ? atol(e) : 0;
    }
    return size_t(v);
}

inline int reduce_grid_cap()
{
    static int cap = 0;
    if (cap == 0) {
        const char* e = getenv("TFGX_REDUCE_GRID_CAP");
        cap = (e != nullptr && atoi(e) > 0) ? atoi(e) : TFGX_REDUCE_GRID_CAP_DEFAULT;
    }
    return cap;
}

// developer A/B: TFGX_REDUCE_WIDE_BLOCKS=0 runs wide rows as one burst per row again (the round 1-4 shapes)
inline bool wide_blocks_enabled()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("TFGX_REDUCE_WIDE_BLOCKS");
        v = (e != nullptr && atoi(e) == 0) ? 0 : 1;
    }
    return v != 0;
}
inline int wide_g256()              // developer A/B: lanes per row at F = 256 (two 128-column blocks or four 64-column ones)
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("TFGX_REDUCE_WIDE_G256");
        v = (e != nullptr && atoi(e) == 16) ? 16 : 32;
    }
    return v;
}

template <int VEC, int G, int CH, int U = 0>
int launch_cfg(const KArgs& a, bool is_max, bool weighted, int ny, hipStream_t stream)
{
    constexpr int ROWS_PER_BLOCK = kBlock / G;
    // very wide rows (several column chunks per lane): fewer, longer-lived workgroups (same-box A/B at F = 512: 42.4 vs
    // 46.0 ms, profiles/r02_ab_grid_cap.txt); everything narrower runs best with one row per lane group
    dim3 grid(grid_for(a.n_dst, ROWS_PER_BLOCK, CH >= 2 ? (reduce_grid_cap() < 4096 ? reduce_grid_cap() : 4096) : reduce_grid_cap()), ny, 1);
    dim3 block(kBlock, 1, 1);
    if constexpr (VEC == 4 && CH == 1) {
        if (a.x_tail != nullptr) {   // split rows: sum / mean only (the case that matters: 400-byte rows)
            if (is_max) {
                if (weighted) seg_reduce_kernel<VEC, G, CH, true, true, true, false, U><<<grid, block, dummy_lds_bytes(), stream>>>(a);
                else seg_reduce_kernel<VEC, G, CH, true, false, true, false, U><<<grid, block, dummy_lds_bytes(), stream>>>(a);
            } else {
                if (weighted) seg_reduce_kernel<VEC, G, CH, false, true, true, false, U><<<grid, block, dummy_lds_bytes(), stream>>>(a);
                else seg_reduce_kernel<VEC, G, CH, false, false, true, false, U><<<grid, block, dummy_lds_bytes(), stream>>>(a);
            }
            TFGX_LAUNCH_CHECK("seg_reduce_kernel<split>");
            return TFGX_OK;
        }
    }
    if (a.x_tail != nullptr) {
        set_error("tfgx_segment_reduce_f32: x_tail needs 16-byte aligned rows and F <= 256");
        return TFGX_ERR_INVALID_ARG;
    }
    if (a.track != nullptr) {
        if constexpr (VEC == 4 && CH == 1) {
            if (weighted) seg_reduce_kernel<VEC, G, CH, true, true, false, true, U><<<grid, block, dummy_lds_bytes(), stream>>>(a);
            else seg_reduce_kernel<VEC, G, CH, true, false, false, true, U><<<grid, block, dummy_lds_bytes(), stream>>>(a);
            TFGX_LAUNCH_CHECK("seg_reduce_kernel<track>");
            return TFGX_OK;
        }
        set_error("tfgx_segment_reduce_f32: track needs 16-byte aligned rows, F % 4 == 0, of F <= 256 columns or of whole 128-byte lines");
        return TFGX_ERR_INVALID_ARG;
    }
    if (is_max) {
        if (weighted) seg_reduce_kernel<VEC, G, CH, true, true, false, false, U><<<grid, block, dummy_lds_bytes(), stream>>>(a);
        else seg_reduce_kernel<VEC, G, CH, true, false, false, false, U><<<grid, block, dummy_lds_bytes(), stream>>>(a);
    } else {
        if (weighted) seg_reduce_kernel<VEC, G, CH, false, true, false, false, U><<<grid, block, dummy_lds_bytes(), stream>>>(a);
        else seg_reduce_kernel<VEC, G, CH, false, false, false, false, U><<<grid, block, dummy_lds_bytes(), stream>>>(a);
    }
    TFGX_LAUNCH_CHECK("seg_reduce_kernel");
    return TFGX_OK;
}

// Group shape of a launch: lanes per row G, column chunks per lane CH, column blocks on grid.y NY, edges per batch U
// (0 = default) — the ONE place it is decided (launch_vec dispatches on it, tfgx_segment_reduce_describe reports it).
struct GroupShape { int G, CH, NY, U; };

inline GroupShape group_shape(const KArgs& a, int vec)
{
    const int lanes = (a.F + vec - 1) / vec;
    // wide rows made of whole 128-byte lines (F >= 128, F % 32 == 0, line-aligned table, no split layout): column blocks of
    // 64 columns (two lines per gathered piece) on grid.y, 16 pieces in flight per lane — see the head of this file.  Rows
    // that are NOT whole lines keep the single burst: every block boundary inside a line would cost one more line request.
    // F = 256 runs as two 128-column blocks (same-box A/B: 18.5 ms against 19.3 for four blocks, 20.0 for one burst).
    // The caller's hint (tfgx_reduce_args.wide_blocks) can switch the blocks off (-1): on a power-law plan the walk is mostly
    // short rows, whose per-row start-up is paid once per pass, and the hot source rows already hit in the caches with one
    // burst per row (R-MAT at products size, F = 192 / 224: 13.4 / 15.7 ms with bursts against 14.6 / 18.0 with blocks).
    if (vec == 4 && a.F >= 128 && a.F % 32 == 0 && a.ldx % 32 == 0 && aligned_to(a.x, 128) && a.x_tail == nullptr &&
        a.wide_blocks >= 0 && (a.wide_blocks > 0 || wide_blocks_enabled())) {
        const int g = a.F == 256 ? wide_g256() : 16;
        return GroupShape{g, 1, (lanes + g - 1) / g, 16};
    }
    if (lanes <= 4) return GroupShape{4, 1, 1, 0};
    if (lanes <= 8) return GroupShape{8, 1, 1, 0};
    if (lanes <= 16) return GroupShape{16, 1, 1, 0};
    if (lanes <= 32) return GroupShape{32, 1, 1, 0};
    if (lanes <= 64) return GroupShape{64, 1, 1, 0};
    if (lanes <= 128) return GroupShape{64, 2, 1, 0};
    const int per = 64 * 4;      // wider rows: column blocks of 64 lanes x 4 chunks on grid.y (col / w re-read per block)
    return GroupShape{64, 4, (lanes + per - 1) / per, 0};
}

template <int VEC>
int launch_vec(const KArgs& a, bool is_max, bool weighted, hipStream_t stream)
{
    const GroupShape g = group_shape(a, VEC);
    if constexpr (VEC == 4) {
        if (g.U == 16 && g.G == 16) return launch_cfg<4, 16, 1, 16>(a, is_max, weighted, g.NY, stream);
        if (g.U == 16 && g.G == 32) return launch_cfg<4, 32, 1, 16>(a, is_max, weighted, g.NY, stream);
    }
    if (g.CH == 1) {
        switch (g.G) {
            case 4: return launch_cfg<VEC, 4, 1>(a, is_max, weighted, 1, stream);
            case 8: return launch_cfg<VEC, 8, 1>(a, is_max, weighted, 1, stream);
            case 16: return launch_cfg<VEC, 16, 1>(a, is_max, weighted, 1, stream);
            case 32: return launch_cfg<VEC, 32, 1>(a, is_max, weighted, 1, stream);
            default: return launch_cfg<VEC, 64, 1>(a, is_max, weighted, 1, stream);
        }
    }
    if (g.CH == 2) return launch_cfg<VEC, 64, 2>(a, is_max, weighted, 1, stream);
    return launch_cfg<VEC, 64, 4>(a, is_max, weighted, g.NY, stream);
}

// Hub rows (in-degree > hub_threshold): the row is cut into chunks of consecutive edges, every chunk is reduced like
// an ordinary row into scratch[chunk, :] (second launch of seg_reduce_kernel over the chunk list), and this kernel
// folds a row's chunk partials IN CHUNK ORDER and applies the epilogue: deterministic, no atomics, and the long
// fp32 sum becomes a sum of short partial sums.
struct HubArgs {
    const int32_t* hub_rows;
    const int32_t* hub_chunk_ptr;
    const float* scratch;
    int64_t n_hub;
    KArgs k;
};

__global__ __launch_bounds__(kBlock) void hub_finalize_kernel(const HubArgs h)
{
    const KArgs& a = h.k;
    const bool is_max = a.op == TFGX_MAX;
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t total = h.n_hub * a.F;
    for (; t < total; t += stride) {
        const int64_t i = t / a.F;
        const int j = int(t - i * a.F);
        const int64_t r = h.hub_rows[i];
        float res = is_max ? -FLT_MAX : 0.0f;
        // chunk partials folded IN CHUNK ORDER (same bits as a plain loop), but a batch of loads is issued before the first
        // add of the batch: the longest row of a power-law graph has thousands of chunks (R-MAT at products size: 2572),
        // and one dependent load per add made this kernel a 1 ms latency chain behind a 0.06 ms transfer
        constexpr int HB = 16;
        int c = h.hub_chunk_ptr[i];
        const int c_end = h.hub_chunk_ptr[i + 1];
        const float* sp = h.scratch + int64_t(c) * a.F + j;
        for (; c + HB <= c_end; c += HB, sp += int64_t(HB) * a.F) {
            float v[HB];
#pragma unroll
            for (int u = 0; u < HB; ++u) v[u] = __builtin_nontemporal_load(sp + int64_t(u) * a.F);
#pragma unroll
            for (int u = 0; u < HB; ++u) res = is_max ? fmaxf(res, v[u]) : res + v[u];
        }
        for (; c < c_end; ++c, sp += a.F) {
            const float v = *sp;
            res = is_max ? fmaxf(res, v) : res + v;
        }
        float* op = a.out + r * a.ldo + j;
        if (a.accumulate) res = is_max ? fmaxf(*op, res) : *op + res;
        if (a.self_coef) {
            // split source rows: columns >= f_main of a NODE's row live in x_tail (ldx is then f_main-wide)
            const float xv = (a.x_tail != nullptr && j >= a.f_main) ? a.x_tail[r * a.ld_tail + (j - a.f_main)]
                                                                    : a.x[r * a.ldx + j];
            // sum / mean: ONE fused multiply-add, like the implicit edge of a row that is walked whole (and like the fused
            // launch's fold of a hub row) — a hub row's bits do not depend on which launch reduced it
            res = is_max ? fmaxf(res, a.self_coef[r] * xv) : fmaf(a.self_coef[r], xv, res);
        }
        if (a.op == TFGX_MEAN) {
            const int cnt = a.mean_count ? a.mean_count[r]
                                         : (a.row_end[r * a.rp_stride] - a.row_begin[r * a.rp_stride]);
            res = res / float(cnt > 1 ? cnt : 1);
        }
        if (a.add_x) res = a.add_x[r * a.ld_add + j] + res;
        if (a.bias) res += a.bias[j];
        *op = apply_act(res, a.act);
    }
}

int launch_any(const KArgs& a, int vec, bool is_max, bool weighted, hipStream_t stream)
{
    if (vec == 4) return launch_vec<4>(a, is_max, weighted, stream);
    if (vec == 2) return launch_vec<2>(a, is_max, weighted, stream);
    return launch_vec<1>(a, is_max, weighted, stream);
}

}  // namespace
}  // namespace tfgx

using namespace tfgx;

// widest vector the layout allows for every row pointer that is touched
static int vector_width(const tfgx_reduce_args* p)
{
    auto ok = [&](int vec) {
        const size_t al = sizeof(float) * vec;
        bool good = (p->F % vec == 0) && (p->ldx % vec == 0) && (p->ldo % vec == 0) && aligned_to(p->x, al) &&
                    aligned_to(p->out, al);
        if (p->add_x) good = good && (p->ld_add % vec == 0) && aligned_to(p->add_x, al);
        if (p->bias) good = good && aligned_to(p->bias, al);
        return good;
    };
    int vec = ok(4) ? 4 : (ok(2) ? 2 : 1);
    // narrow rows: prefer 8 lanes with narrower loads over 4 lanes x dwordx4 — 8 rows per wave instead of 16 (less
    // degree divergence inside a wave) and 8 edges in flight per row instead of 4
    while (vec > 1 && p->F / vec < 8) vec /= 2;
    return vec;
}

extern "C" int tfgx_segment_reduce_describe(const tfgx_reduce_args* p, char* buf, size_t buf_bytes)
{
    TFGX_REQUIRE(p != nullptr && buf != nullptr && buf_bytes > 0, "null argument");
    const int vec = vector_width(p);
    KArgs a;
    a.F = int32_t(p->F); a.ldx = p->ldx; a.x = p->x; a.x_tail = p->x_tail; a.wide_blocks = p->wide_blocks;
    const GroupShape g = group_shape(a, vec);
    const bool split = p->x_tail != nullptr && vec == 4 && g.CH == 1;
    snprintf(buf, buf_bytes, "seg_reduce_kernel<%d, %d, %d, %s, %s, %s, %s, %d>", vec, g.G, g.CH, p->op == TFGX_MAX ? "true" : "false",
             p->w ? "true" : "false", split ? "true" : "false", p->track ? "true" : "false", g.U);
    return TFGX_OK;
}

extern "C" int tfgx_segment_reduce_f32(const tfgx_reduce_args* p, tfgx_stream_t stream_)
{
    TFGX_RANGE();
    TFGX_REQUIRE(p != nullptr, "args is null");
    TFGX_REQUIRE(p->n_dst >= 0 && p->F >= 1 && p->F < (int64_t(1) << 30), "bad n_dst / F");
    TFGX_REQUIRE(p->op == TFGX_SUM || p->op == TFGX_MEAN || p->op == TFGX_MAX, "bad op");
    TFGX_REQUIRE(p->act == TFGX_ACT_NONE || p->act == TFGX_ACT_RELU, "bad act");
    if (p->n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(p->row_begin && p->row_end && p->out && p->x, "null pointer");
    TFGX_REQUIRE((p->ldx >= p->F || p->x_tail) && p->ldo >= p->F, "leading dimension < F");
    TFGX_REQUIRE(p->rp_stride >= 1, "rp_stride < 1");
    TFGX_REQUIRE(!(p->add_x) || p->ld_add >= p->F, "ld_add < F");

    KArgs a;
    a.row_begin = p->row_begin; a.row_end = p->row_end; a.rp_stride = p->rp_stride;
    a.col = p->col; a.w = p->w; a.n_dst = p->n_dst; a.x = p->x; a.ldx = p->ldx; a.F = int32_t(p->F);
    a.col0 = 0; a.out = p->out; a.ldo = p->ldo; a.op = p->op; a.act = p->act; a.accumulate = p->accumulate;
    a.self_coef = p->self_coef; a.bias = p->bias; a.add_x = p->add_x; a.ld_add = p->ld_add;
    a.mean_count = p->mean_count;
    a.hub_threshold = 0;
    a.x_tail = p->x_tail; a.ld_tail = p->ld_tail; a.f_main = int32_t(p->f_main);
    a.edge_tail = p->edge_tail; a.ld_edge_tail = p->ld_edge_tail;
    a.wide_blocks = p->wide_blocks;
    a.row_order = p->row_order;
    a.track = p->track; a.ld_track = p->ld_track; a.track_row_begin = p->track_row_begin;
    if (p->track) {
        TFGX_REQUIRE(p->op == TFGX_MAX && !p->self_coef && !p->x_tail && p->hub_threshold == 0 && !p->bias && !p->add_x &&
                         p->act == TFGX_ACT_NONE,
                     "track: plain TFGX_MAX launches only (no self_coef / bias / activation / split rows / hub lists)");
        TFGX_REQUIRE(p->ld_track >= p->F && p->ld_track % 4 == 0 && aligned_to(p->track, 16), "track: bad leading dimension / alignment");
    }
    TFGX_REQUIRE(p->edge_tail == nullptr ||
                     (p->x_tail != nullptr && p->ld_edge_tail >= p->F - p->f_main && p->ld_edge_tail % 4 == 0 &&
                      aligned_to(p->edge_tail, 16)),
                 "edge_tail needs the split-row layout (x_tail) and 16-byte aligned rows");
    if (p->x_tail) {
        TFGX_REQUIRE(p->f_main > 0 && p->f_main < p->F && p->f_main % 4 == 0 && (p->F - p->f_main) % 4 == 0 &&
                         p->ld_tail >= p->F - p->f_main && p->ld_tail % 4 == 0 && p->ldx >= p->f_main &&
                         aligned_to(p->x_tail, 16),
                     "bad split-row layout (f_main / ld_tail / alignment)");
    }
    const bool use_hub = p->hub_threshold > 0 && p->n_hub_rows > 0;
    if (use_hub) {
        TFGX_REQUIRE(p->hub_rows && p->hub_chunk_ptr && p->hub_chunk_begin && p->hub_chunk_end && p->hub_scratch &&
                         p->n_hub_chunks > 0,
                     "hub rows given without chunk lists / scratch");
        a.hub_threshold = p->hub_threshold;
    }

    const bool is_max = p->op == TFGX_MAX;
    const bool weighted = p->w != nullptr;
    hipStream_t stream = as_stream(stream_);

    int vec = vector_width(p);
    int rc = launch_any(a, vec, is_max, weighted, stream);
    if (rc != TFGX_OK || !use_hub) return rc;

    // hub path: (2) chunk partials -> scratch, (3) ordered fold + epilogue
    KArgs c = a;
    c.row_begin = p->hub_chunk_begin; c.row_end = p->hub_chunk_end; c.rp_stride = 1;
    c.n_dst = p->n_hub_chunks; c.out = p->hub_scratch; c.ldo = p->F;
    c.op = is_max ? TFGX_MAX : TFGX_SUM; c.act = TFGX_ACT_NONE; c.accumulate = 0;
    c.self_coef = nullptr; c.bias = nullptr; c.add_x = nullptr; c.mean_count = nullptr; c.hub_threshold = 0;
    c.row_order = nullptr; c.track = nullptr; c.track_row_begin = nullptr;
    const bool sok = (p->F % 4 == 0) && aligned_to(p->hub_scratch, 16) && (p->ldx % 4 == 0) && aligned_to(p->x, 16);
    const bool sok2 = (p->F % 2 == 0) && aligned_to(p->hub_scratch, 8) && (p->ldx % 2 == 0) && aligned_to(p->x, 8);
    rc = launch_any(c, sok ? 4 : (sok2 ? 2 : 1), is_max, weighted, stream);
    if (rc != TFGX_OK) return rc;
    HubArgs h;
    h.hub_rows = p->hub_rows; h.hub_chunk_ptr = p->hub_chunk_ptr; h.scratch = p->hub_scratch;
    h.n_hub = p->n_hub_rows; h.k = a;
    hub_finalize_kernel<<<grid_for(p->n_hub_rows * p->F, kBlock), kBlock, 0, stream>>>(h);
    TFGX_LAUNCH_CHECK("hub_finalize_kernel");
    return TFGX_OK;
}
