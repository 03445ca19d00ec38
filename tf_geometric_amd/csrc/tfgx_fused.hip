// Aggregation -> projection in ONE launch:  C = act( (reduce_{edges of r} w * x[col] (+ self_coef[r] * x[r]) [/ deg]) @ B + bias )
// for the aggregate-then-project layers — GCN when units > F (nn/conv/gcn.py:272-288 evaluated as (A_hat x) W), mean / sum
// GraphSAGE's neighbour half (nn/conv/graph_sage.py:34-58) — instead of tfgx_segment_reduce_f32 writing the [N, F] aggregate
// to HBM and tfgx_gemm_bias_act_f32 reading it back (1.9 GB of round trip at products shape, F = 100).
//
// One persistent 1024-thread workgroup per CU (16 waves; the gather walk keeps its rate down to 16 waves per CU:
// profiles/r03_occupancy_probe.jsonl).  LDS holds B ([KP][LDW] floats, loaded once) and two tiles of aggregated
// rows, TRANSPOSED (At[k][m], m = destination row inside the 64-row tile) so that the MFMA A operand is a conflict-free
// ds_read over consecutive m.  When all of B does not fit beside the two tiles (F = 128 -> 256: 135 KB + 66 KB > 160 KB)
// only its first NL columns (a multiple of 64) are resident; the consumer jobs of the remaining columns read their B
// operand straight from global memory — 128-byte row segments of a <= 128 KB matrix that every workgroup re-reads for every
// tile, i.e. L2 hits — with 8 k-pairs (16 loads) in flight per wave.
//   * producers: every wave repeatedly takes the next UNIT (64 / G consecutive destination rows of the current tile: one per
//     lane group) from an LDS counter, reduces it exactly like seg_reduce_kernel (G lanes per row, 4 columns per lane,
//     (col, w) batches prefetched, 8 gathered rows in flight, one in-order FMA chain per output element), and stores the row
//     into the tile.  Units are handed out dynamically, so a wave that drew short rows simply takes more of them;
//   * consumers: the last waves to arrive at a tile (per-tile arrival counter) each multiply one 32-row x 64-column block of
//     it — v_mfma_f32_32x32x2_f32 against B from LDS, bias / activation in the epilogue, rows written straight to C — while
//     the other waves are already reducing the next tile into the other buffer.  No workgroup barrier after the B load: a
//     buffer is re-used only after its `done` sequence number says the previous occupant has been multiplied.
// Deterministic (fixed per-row edge order, fixed k order), fp32 throughout.  Roofline: HBM (the gather), as the unfused
// aggregation; the MFMA work (2 N F U flops = 0.8 ms of one wave per CU at products shape) rides on otherwise idle pipes.
#include "tfgx_common.h"
#include "tfgx_mfma.h"
#include <cstdlib>

namespace tfgx {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileRows = 64;
constexpr int kLda = kTileRows + 1;
constexpr int kBufs = 2;
constexpr int kFusedThreads = 1024;
#ifndef TFGX_FUSED_VEC_STORE
#define TFGX_FUSED_VEC_STORE 1
#endif
#ifndef TFGX_FUSED_JB
#define TFGX_FUSED_JB 2
#endif

struct FArgs {
    const int32_t* row_ptr;
    const int32_t* col;
    const float* w;
    int64_t n_dst;
    const float* x;
    int64_t ldx;
    int32_t F;
    int32_t op;
    const float* self_coef;
    const int32_t* mean_count;
    const float* B;
    int64_t ldb;
    const float* bias;
    int32_t act;
    int32_t N;
    float* C;
    int64_t ldc;
    int32_t KP;        // F rounded up to even (the MFMA consumes two k per step)
    int32_t n_blocks;  // 32-column output blocks, rounded up to a multiple of 4
    int32_t LDW;       // 32 * n_blocks + 8
    int64_t n_tiles;
    int32_t NL;        // columns of B resident in LDS (multiple of 64; LDW = NL + 8); columns >= NL: B operand from global
    float* agg;        // optional side output (training forward): the aggregated rows themselves, [n_dst, ld_agg], or NULL
    int64_t ld_agg;
    int32_t dbg;       // developer experiment (-DTFGX_FUSED_DEBUG builds only): 1 = skip the multiplication and the stores, 2 = skip the stores only
    // power-law graphs: rows longer than hub_threshold were cut into chunks and reduced chunk by chunk into hub_scratch by a
    // launch of the ordinary kernel BEFORE this one; their lane group folds the chunk partials in chunk order instead of
    // walking the edges (what hub_finalize_kernel does in the unfused path)
    int32_t hub_threshold;
    int32_t n_hub;
    const int32_t* hub_rows;        // ascending
    const int32_t* hub_chunk_ptr;
    const float* hub_scratch;       // [chunks, F]
    // skewed plans: slot i of the launch reduces destination row row_order[i] (the plan's rows sorted by length), so the 64
    // rows of a tile are of similar length and no lane group holds a tile back; NULL: i
    const int32_t* row_order;
    const int32_t* hub_order_slot;  // optional: slot of row row_order[i] in hub_rows, i < n_hub (rows sorted by length: hubs first)
    // optional split source rows (the static feature layout, tfgx_reduce_args.x_tail / edge_tail): x holds columns [0, f_main),
    // x_tail columns [f_main, F) per NODE, edge_tail the same tail columns per EDGE of this plan (streamed next to col / w)
    const float* x_tail;
    int64_t ld_tail;
    int32_t f_main;
    const float* edge_tail;
    int64_t ld_edge_tail;
};

#ifndef TFGX_FUSED_PRIO
#define TFGX_FUSED_PRIO 0
#endif
#ifdef TFGX_FUSED_DEBUG
#define TFGX_FUSED_DBG_IS(v) (a.dbg == (v))
#else
#define TFGX_FUSED_DBG_IS(v) false
#endif

template <int G>
__device__ __forceinline__ int bcast_i(int v, int j) { return __shfl(v, j, G); }
template <int G>
__device__ __forceinline__ float bcast_f(float v, int j) { return __shfl(v, j, G); }

#ifdef TFGX_FUSED_DEBUG
__device__ uint64_t g_fused_wave_end[4096 * 2];     // per wave of the last launch: entry tick, exit tick (100 MHz)
#endif
template <int G, bool WEIGHTED>
__global__ __launch_bounds__(kFusedThreads) void agg_gemm_kernel(const FArgs a)
{
#ifdef TFGX_FUSED_DEBUG
    const uint64_t dbg_t0 = wall_clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ws = lds;                                        // [KP][LDW]
    float* At = Ws + a.KP * a.LDW;                          // [kBufs][KP][kLda]
    int* ctrl = reinterpret_cast<int*>(At + kBufs * a.KP * kLda);   // [0] next unit | [1 + b] arrivals of buffer b | [1 + kBufs + b] done seq
    int* rowid = ctrl + 16;                                 // [kBufs][kTileRows]: the destination row of each tile slot (row_order)
    constexpr int RPW = 64 / G;                             // destination rows per wave step (one per lane group)
    constexpr int UNITS = kTileRows / RPW;
    constexpr int UNROLL = 8;
    const int tid = threadIdx.x, lane64 = tid & 63;
    const int lane = lane64 % G, grp = lane64 / G;

    for (int i = tid; i < a.KP * a.LDW; i += kFusedThreads) {
        const int k = i / a.LDW, n = i - k * a.LDW;
        Ws[i] = (k < a.F && n < a.N && n < a.NL) ? a.B[int64_t(k) * a.ldb + n] : 0.0f;
    }
    for (int i = tid; i < kBufs * a.KP * kLda; i += kFusedThreads) At[i] = 0.0f;      // rows k >= F stay zero for good
    if (tid < 16) ctrl[tid] = 0;                            // ([6 + b]: finished halves of buffer b's tile)
    __syncthreads();

#if TFGX_FUSED_PRIO
    __builtin_amdgcn_s_setprio(TFGX_FUSED_PRIO);    // developer A/B: producers (gather + FMA) above the consumers' MFMA stream
#endif
    const int64_t my_tiles = a.n_tiles > int64_t(blockIdx.x) ? (a.n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int64_t total_units = my_tiles * UNITS;
    const int c_raw = lane * 4;
    const bool cvalid = c_raw < a.F;
    const int coff = cvalid ? c_raw : a.F - 4;              // lanes past F re-read the last valid vector (discarded)
    const int l31 = lane64 & 31, kh = lane64 >> 5;
    const bool vec_store = TFGX_FUSED_VEC_STORE && (a.N % 4 == 0) && (a.ldc % 4 == 0) &&
                           ((reinterpret_cast<uintptr_t>(a.C) & 15) == 0);                                      // wave-uniform
    // per-lane source base / stride (seg_reduce_kernel's SPLIT scheme): one array normally; with split rows the lanes that own
    // columns >= f_main read the node-tail array — or, for gathered rows, the per-EDGE tail stream (indexed by CSR position)
    const float* xb = a.x + coff;           // gathered rows
    int64_t xl = a.ldx;
    const float* xsb = xb;                  // a NODE's own row (the self-loop term)
    int64_t xsl = xl;
    bool by_edge = false;
    if (a.x_tail != nullptr && coff >= a.f_main) {
        xb = xsb = a.x_tail + (coff - a.f_main);
        xl = xsl = a.ld_tail;
        if (a.edge_tail != nullptr) {
            xb = a.edge_tail + (coff - a.f_main);
            xl = a.ld_edge_tail;
            by_edge = true;
        }
    }
    // element offset of a gathered row as ONE 32 x 32 -> 64-bit multiply (v_mad_u64_u32): ids / CSR positions are non-negative
    // int32 and the strides fit 32 bits (checked at the entry point); int64 x int64 is emulated with three quarter-rate
    // multiplies per address — vector-ALU work the MFMA half of this kernel competes with
    const uint32_t xl32 = uint32_t(xl);
    auto row_off = [&](int i) { return uint64_t(uint32_t(i)) * xl32; };

    while (true) {
        int u = 0;
        if (lane64 == 0) u = atomicAdd(&ctrl[0], 1);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= total_units) break;
        const int q = u / UNITS, slot = u - q * UNITS;      // tile sequence number inside this workgroup, unit inside the tile
        const int buf = q % kBufs;
        const int64_t tile = int64_t(blockIdx.x) + int64_t(q) * gridDim.x;
        const int m = slot * RPW + grp;                     // row inside the tile
        const int64_t ri = tile * kTileRows + m;
        const int64_t r = (a.row_order != nullptr && ri < a.n_dst) ? int64_t(a.row_order[ri]) : ri;

        // ---- producer: reduce destination row r (seg_reduce_kernel's walk) -------------------------------------------
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (r < a.n_dst) {
            const int s = a.row_ptr[r], e = a.row_ptr[r + 1];
            const bool hub = a.hub_threshold > 0 && e - s > a.hub_threshold;      // uniform inside the lane group
            int cj_next = 0;
            float wj_next = 0.0f;
            if (!hub && s + lane < e) {
                cj_next = a.col[s + lane];
                if constexpr (WEIGHTED) wj_next = a.w[s + lane];
            }
            if (hub) {
                // r is in the list: its slot.  Walk order by length puts the hub rows first, and the plan then hands over
                // their slots (one load, checked); otherwise a binary search
                int lo = (a.hub_order_slot != nullptr && a.row_order != nullptr && ri < a.n_hub) ? a.hub_order_slot[ri] : -1;
                if (lo < 0 || lo >= a.n_hub || a.hub_rows[lo] != int32_t(r)) {
                    int hi = a.n_hub - 1;
                    lo = 0;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (a.hub_rows[mid] < int32_t(r)) lo = mid + 1;
                        else hi = mid;
                    }
                }
                // chunk partials in chunk order, eight loads in flight (thousands of chunks on the longest rows)
                int c = a.hub_chunk_ptr[lo];
                const int c_end = a.hub_chunk_ptr[lo + 1];
                for (; c + UNROLL <= c_end; c += UNROLL) {
                    float pv[UNROLL][4];
#pragma unroll
                    for (int t = 0; t < UNROLL; ++t) load_vec<4>(a.hub_scratch + int64_t(c + t) * a.F + coff, pv[t]);
#pragma unroll
                    for (int t = 0; t < UNROLL; ++t)
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[v] += pv[t][v];
                }
                for (; c < c_end; ++c) {
                    float pv[4];
                    load_vec<4>(a.hub_scratch + int64_t(c) * a.F + coff, pv);
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[v] += pv[v];
                }
            }
            for (int base = s; base < (hub ? s : e); base += G) {
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
                    float xv[UNROLL][4];
                    float ww[UNROLL];
#pragma unroll
                    for (int t = 0; t < UNROLL; ++t) {
                        const int c = bcast_i<G>(cj, j + t);
                        if constexpr (WEIGHTED) ww[t] = bcast_f<G>(wj, j + t);
                        load_vec<4>(xb + row_off(by_edge ? base + j + t : c), xv[t]);
                    }
#pragma unroll
                    for (int t = 0; t < UNROLL; ++t)
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[v] = WEIGHTED ? fmaf(ww[t], xv[t][v], acc[v]) : acc[v] + xv[t][v];
                }
                for (; j < cnt; ++j) {
                    const int c = bcast_i<G>(cj, j);
                    float wv = 1.0f;
                    if constexpr (WEIGHTED) wv = bcast_f<G>(wj, j);
                    float xv[4];
                    load_vec<4>(xb + row_off(by_edge ? base + j : c), xv);
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[v] = WEIGHTED ? fmaf(wv, xv[v], acc[v]) : acc[v] + xv[v];
                }
            }
            if (a.self_coef) {                               // the implicit (r, r) edge appended after the row's edges
                const float sc = a.self_coef[r];
                float xs[4];
                load_vec<4>(xsb + r * xsl, xs);
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] = fmaf(sc, xs[v], acc[v]);
            }
            if (a.op == TFGX_MEAN) {
                const int cnt = a.mean_count ? a.mean_count[r] : (e - s);
                const float divisor = float(cnt > 1 ? cnt : 1);
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] = acc[v] / divisor;
            }
        }
        if (a.agg != nullptr && cvalid && r < a.n_dst)        // training forward: the weight gradient needs the aggregate
            store_vec<4>(a.agg + r * a.ld_agg + coff, acc);
        // ---- hand the row over: wait until the buffer's previous tile (q - kBufs) has been multiplied, store transposed
        if (q >= kBufs) {
            volatile int* done = ctrl + 1 + kBufs + buf;
            while (*done < q - kBufs + 1) __builtin_amdgcn_s_sleep(1);
            __threadfence_block();
        }
        float* ab = At + buf * a.KP * kLda;
        if (cvalid) {
#pragma unroll
            for (int v = 0; v < 4; ++v) ab[(coff + v) * kLda + m] = acc[v];
        }
        if (a.row_order != nullptr && lane == 0) rowid[buf * kTileRows + m] = ri < a.n_dst ? int(r) : -1;
        __threadfence_block();
        int arrived = 0;
        if (lane64 == 0) arrived = atomicAdd(&ctrl[1 + buf], 1);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        constexpr int JB = TFGX_FUSED_JB;                    // 32-column blocks per consumer job
        const int njobs = 2 * (a.n_blocks / JB);
        if (arrived < UNITS - njobs) continue;

        // ---- consumers: the last arrivers of tile q each multiply ONE 32-row x (32 JB)-column block of it (the last one at
        // once, the ones before it as soon as the tile is complete) -> C[tile rows, :] = act(At^T @ Ws + bias).  Up to eight
        // waves per tile (JB = 2 measured best: 4 halves the jobs, 1 loses the shared A operand): on tiles of short rows (the tail of a power-law graph in walk order) the producers are done in less than
        // one wave's multiplication time, and a single consumer wave was what the launch waited for
        const int job = UNITS - 1 - arrived;                 // the last arriver takes block 0, the one before it block 1, ...
        if (job > 0) {
            volatile int* arr = ctrl + 1 + buf;
            while (*arr < UNITS) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence_block();
#if TFGX_FUSED_PRIO
        __builtin_amdgcn_s_setprio(0);       // multiply at LOW priority: the producers' vector-ALU work on this SIMD goes first
#endif
        const int half = job & 1, nb_first = (job >> 1) * JB;
        for (int mb = half; mb < (TFGX_FUSED_DBG_IS(1) ? 0 : half + 1); ++mb) {
            for (int nb0 = nb_first; nb0 < nb_first + JB; nb0 += JB) {     // n_blocks is a multiple of 4 (zero-padded columns of Ws)
                f32x16 c4[JB];
#pragma unroll
                for (int jb = 0; jb < JB; ++jb)
#pragma unroll
                    for (int t = 0; t < 16; ++t) c4[jb][t] = 0.0f;
                const float* ap = ab + kh * kLda + mb * 32 + l31;
                const int pairs = a.KP / 2;
                int pr = 0;
                if (nb0 * 32 < a.NL) {
                    // B resident in LDS.  KU k-pairs per step: all (1 + JB) * KU LDS reads of a step are issued before its
                    // JB * KU MFMAs (a read-wait-multiply chain per MFMA left the single consumer wave at ~2.5x the MFMA time
                    // and the producers waiting for buffers)
                    const float* bp = Ws + kh * a.LDW + nb0 * 32 + l31;
                    constexpr int KU = 4;
                    for (; pr + KU <= pairs; pr += KU) {
                        float av[KU], bv[KU][JB];
#pragma unroll
                        for (int t = 0; t < KU; ++t) {
                            av[t] = ap[(2 * (pr + t)) * kLda];
#pragma unroll
                            for (int jb = 0; jb < JB; ++jb) bv[t][jb] = bp[(2 * (pr + t)) * a.LDW + jb * 32];
                        }
                        __builtin_amdgcn_sched_barrier(0);      // left alone the scheduler sinks every read next to its MFMA
#pragma unroll
                        for (int t = 0; t < KU; ++t)
#pragma unroll
                            for (int jb = 0; jb < JB; ++jb)
                                c4[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t][jb], c4[jb], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    for (; pr < pairs; ++pr) {
                        const float av = ap[(2 * pr) * kLda];
                        float bv[JB];
#pragma unroll
                        for (int jb = 0; jb < JB; ++jb) bv[jb] = bp[(2 * pr) * a.LDW + jb * 32];
#pragma unroll
                        for (int jb = 0; jb < JB; ++jb) c4[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[jb], c4[jb], 0, 0, 0);
                    }
                } else {
                    // B columns past the resident ones: the operand comes from global memory (L2: every workgroup re-reads the
                    // same <= 128 KB for every tile).  Lane (l31, kh) reads B[2 pr + kh][col]: two 128-byte row segments per
                    // load; KG k-pairs = JB * KG loads in flight per wave before the first MFMA of the step.  Columns >= N read
                    // column N - 1 (never stored).
                    const float* gp[JB];
#pragma unroll
                    for (int jb = 0; jb < JB; ++jb) {
                        const int gn = (nb0 + jb) * 32 + l31;
                        gp[jb] = a.B + int64_t(kh) * a.ldb + (gn < a.N ? gn : a.N - 1);
                    }
                    constexpr int KG = 8;
                    for (; pr + KG <= pairs; pr += KG) {
                        float av[KG], bv[KG][JB];
#pragma unroll
                        for (int t = 0; t < KG; ++t)
#pragma unroll
                            for (int jb = 0; jb < JB; ++jb) bv[t][jb] = gp[jb][int64_t(2 * (pr + t)) * a.ldb];
#pragma unroll
                        for (int t = 0; t < KG; ++t) av[t] = ap[(2 * (pr + t)) * kLda];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < KG; ++t)
#pragma unroll
                            for (int jb = 0; jb < JB; ++jb)
                                c4[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t][jb], c4[jb], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    for (; pr < pairs; ++pr) {
                        const float av = ap[(2 * pr) * kLda];
                        float bv[JB];
#pragma unroll
                        for (int jb = 0; jb < JB; ++jb) bv[jb] = (2 * pr + kh < a.F) ? gp[jb][int64_t(2 * pr) * a.ldb] : 0.0f;
#pragma unroll
                        for (int jb = 0; jb < JB; ++jb) c4[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[jb], c4[jb], 0, 0, 0);
                    }
                }
                // D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  Two phases on purpose (as in
                // tfgx_gemm.hip): bias + activation IN PLACE first, then every store reads its own accumulator register —
                // results computed into a temporary right before each store make the compiler drain vmcnt(0) between
                // consecutive stores (measured here: 2.0 ms of the 10.8 ms launch went into 128 serialised stores per tile)
#pragma unroll
                for (int jb = 0; jb < JB; ++jb) {
                    const int gn = (nb0 + jb) * 32 + l31;
                    const float bv = (a.bias && gn < a.N) ? a.bias[gn] : 0.0f;
#pragma unroll
                    for (int t = 0; t < 16; ++t) c4[jb][t] = apply_act(c4[jb][t] + bv, a.act);
                }
                const int64_t row0 = (TFGX_FUSED_DBG_IS(3) ? (tile & 63) : tile) * kTileRows + mb * 32 + 4 * kh;   // dbg 3: stores land in 4096 rows
                const bool full = tile * kTileRows + kTileRows <= a.n_dst;
                if (a.row_order != nullptr && !vec_store) {     // walk order without 16-byte stores (odd N / unaligned C)
                    // walk order: tile slot -> destination row through the ids the producers left in LDS
                    const int* rid = rowid + buf * kTileRows + mb * 32 + 4 * kh;
#pragma unroll
                    for (int jb = 0; jb < JB; ++jb) {
                        const int gn = (nb0 + jb) * 32 + l31;
                        if (gn >= a.N || TFGX_FUSED_DBG_IS(2)) continue;
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            const int rr = rid[(t & 3) + 8 * (t >> 2)];
                            if (rr >= 0) __builtin_nontemporal_store(c4[jb][t], a.C + int64_t(rr) * a.ldc + gn);
                        }
                    }
                    continue;
                }
#if TFGX_FUSED_VEC_STORE
                if (vec_store && (full || a.row_order != nullptr)) {
                    // 16-byte stores through a quad transpose of the accumulators (tfgx_mfma.h): 8 instead of 32 store
                    // instructions per job; lane i of a quad ends with row 8 g + i + 4 kh, columns 4 q .. 4 q + 3
                    const int qi = lane64 & 3, qc = (l31 >> 2) * 4;
                    typedef float f32x4s __attribute__((ext_vector_type(4)));
                    // destination rows of this lane's four register groups (g): natural order = ONE multiply + adds; walk
                    // order = the ids the producers left in LDS (-1: past the end)
                    int64_t roff[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int slot = mb * 32 + 8 * g + qi + 4 * kh;
                        const int64_t rr = a.row_order != nullptr ? int64_t(rowid[buf * kTileRows + slot]) : tile * kTileRows + slot;
                        roff[g] = rr >= 0 ? rr * a.ldc : int64_t(-1);
                    }
#pragma unroll
                    for (int jb = 0; jb < JB; ++jb) {
                        float r4[4][4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {          // (every lane takes part in the quad permutes)
#pragma unroll
                            for (int k = 0; k < 4; ++k) r4[g][k] = c4[jb][4 * g + k];
                            quad_transpose4(r4[g], lane64);
                        }
                        const int gn = (nb0 + jb) * 32 + qc;
                        if (gn < a.N) {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                if (roff[g] >= 0)
                                    __builtin_nontemporal_store(f32x4s{r4[g][0], r4[g][1], r4[g][2], r4[g][3]},
                                                                reinterpret_cast<f32x4s*>(a.C + roff[g] + gn));
                        }
                    }
                    continue;
                }
#endif
#pragma unroll
                for (int jb = 0; jb < JB; ++jb) {
                    const int gn = (nb0 + jb) * 32 + l31;
                    if (gn >= a.N || TFGX_FUSED_DBG_IS(2)) continue;
                    float* cp = a.C + row0 * a.ldc + gn;
                    if (full) {
                        // streaming (non-temporal) stores: the output is written once and not read by this launch; kept out
                        // of the caches it does not evict source rows the gather still hits there
                        if (!TFGX_FUSED_DBG_IS(4)) {
#pragma unroll
                            for (int t = 0; t < 16; ++t)
                                __builtin_nontemporal_store(c4[jb][t], cp + int64_t((t & 3) + 8 * (t >> 2)) * a.ldc);
                        } else {
#pragma unroll
                            for (int t = 0; t < 16; ++t) cp[int64_t((t & 3) + 8 * (t >> 2)) * a.ldc] = c4[jb][t];
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            const int dr = (t & 3) + 8 * (t >> 2);
                            if (row0 + dr < a.n_dst) cp[int64_t(dr) * a.ldc] = c4[jb][t];
                        }
                    }
                }
            }
        }
        __threadfence_block();
#if TFGX_FUSED_PRIO
        __builtin_amdgcn_s_setprio(TFGX_FUSED_PRIO);
#endif
        int fin = 0;
        if (lane64 == 0) fin = atomicAdd(&ctrl[6 + buf], 1);
        fin = __builtin_amdgcn_readfirstlane(fin);
        if (fin == njobs - 1) {
            // the last block to finish hands the buffer back (wave-uniform branch; every lane stores the same values: no
            // single-lane branch around the hand-over)
            ctrl[1 + buf] = 0;
            ctrl[6 + buf] = 0;
            __threadfence_block();
            *reinterpret_cast<volatile int*>(ctrl + 1 + kBufs + buf) = q + 1;
        }
    }
#ifdef TFGX_FUSED_DEBUG
    if ((threadIdx.x & 63) == 0) {
        const int gw = int(blockIdx.x) * (kFusedThreads / 64) + int(threadIdx.x >> 6);
        if (gw < 4096) {
            g_fused_wave_end[2 * gw] = dbg_t0;
            g_fused_wave_end[2 * gw + 1] = wall_clock64();
        }
    }
#endif
}

inline size_t fused_lds_bytes(int kp, int ldw)
{
    return sizeof(float) * (size_t(kp) * ldw + size_t(kBufs) * kp * kLda) + sizeof(int) * (16 + kBufs * kTileRows);
}

constexpr size_t kFusedLdsLimit = 160 * 1024;

// Columns of B kept in LDS for a [kp, np] projection (np = N rounded up to 128): all of them when B and the two tiles fit,
// else the largest multiple of 64 (one consumer job = 64 columns) that does; 0 = not even 64 columns fit.
inline int fused_resident_cols(int kp, int np)
{
    for (int nl = np; nl >= 64; nl -= 64)
        if (fused_lds_bytes(kp, nl + 8) <= kFusedLdsLimit) return nl;
    return 0;
}

}  // namespace
}  // namespace tfgx

using namespace tfgx;

// 1 if tfgx_aggregate_gemm_f32 takes rows of F columns projected to N columns: the two tiles and at least 64 columns of B
// resident in 160 KB of LDS (columns that do not fit are read from global memory / L2 by their consumer jobs)
#ifdef TFGX_FUSED_DEBUG
extern "C" int tfgx_debug_fused_waves(uint64_t* out, int n_waves)
{
    TFGX_HIP_CHECK(hipDeviceSynchronize());
    TFGX_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(tfgx::g_fused_wave_end), sizeof(uint64_t) * 2 * size_t(n_waves)));
    return TFGX_OK;
}
#endif

extern "C" int tfgx_aggregate_gemm_fits(int64_t F, int64_t N)
{
    if (F < 4 || F > 128 || F % 4 != 0 || N < 1 || N > 256) return 0;
    const int kp = int((F + 1) / 2 * 2), np = int((N + 127) / 128) * 128;
    return fused_resident_cols(kp, np) > 0 ? 1 : 0;
}

extern "C" int tfgx_aggregate_gemm_f32(const tfgx_reduce_args* p, const float* B, int64_t ldb, const float* bias, int32_t act,
                                       float* C, int64_t ldc, int64_t N, tfgx_stream_t stream_)
{
    TFGX_RANGE();
    TFGX_REQUIRE(p != nullptr, "args is null");
    TFGX_REQUIRE(p->op == TFGX_SUM || p->op == TFGX_MEAN, "sum / mean only");
    TFGX_REQUIRE(act == TFGX_ACT_NONE || act == TFGX_ACT_RELU, "bad act");
    TFGX_REQUIRE(tfgx_aggregate_gemm_fits(p->F, N) == 1, "shape not supported (tfgx_aggregate_gemm_fits)");
    TFGX_REQUIRE(p->n_dst >= 0 && p->n_dst < (int64_t(1) << 31) * kTileRows / 64, "bad n_dst");
    if (p->n_dst == 0) return TFGX_OK;
    TFGX_REQUIRE(p->row_begin && p->row_end == p->row_begin + 1 && p->rp_stride == 1, "needs a plain CSR (row_ptr, row_ptr + 1)");
    TFGX_REQUIRE(p->x && B && C, "null pointer");          // (col may be NULL for a graph without edges)
    TFGX_REQUIRE(p->ldx >= (p->x_tail ? p->f_main : p->F) && p->ldx % 4 == 0 && aligned_to(p->x, 16) && ldb >= N && ldc >= N,
                 "bad leading dimension / alignment");
    TFGX_REQUIRE(!p->accumulate && !p->add_x && !p->track, "plain aggregation only (no accumulate / add_x / track)");
    TFGX_REQUIRE(p->ldx < (int64_t(1) << 31) && p->ld_tail < (int64_t(1) << 31) && p->ld_edge_tail < (int64_t(1) << 31),
                 "leading dimensions must be below 2^31 elements");
    if (p->x_tail) {      // split source rows (static feature layout): whole-line main rows + 16-byte aligned tails
        TFGX_REQUIRE(p->f_main >= 4 && p->f_main % 4 == 0 && p->f_main < p->F && p->ldx >= p->f_main && p->ld_tail >= p->F - p->f_main &&
                         p->ld_tail % 4 == 0 && aligned_to(p->x_tail, 16),
                     "split rows: bad f_main / ld_tail / alignment");
        TFGX_REQUIRE(p->edge_tail == nullptr || (p->ld_edge_tail >= p->F - p->f_main && p->ld_edge_tail % 4 == 0 &&
                                                 aligned_to(p->edge_tail, 16)),
                     "split rows: bad edge_tail stride / alignment");
    }
    TFGX_REQUIRE(p->out == nullptr || (p->ldo >= p->F && p->ldo % 4 == 0 && aligned_to(p->out, 16)),
                 "side output of the aggregate: rows of >= F floats, 16-byte aligned");
    const bool use_hub = p->hub_threshold > 0 && p->n_hub_rows > 0;
    if (use_hub) {
        TFGX_REQUIRE(p->hub_rows && p->hub_chunk_ptr && p->hub_chunk_begin && p->hub_chunk_end && p->hub_scratch &&
                         p->n_hub_chunks > 0 && p->n_hub_rows < (int64_t(1) << 31),
                     "hub rows given without chunk lists / scratch");
        // chunk partials first: every chunk is reduced like an ordinary row into hub_scratch (the unfused path's step 2)
        tfgx_reduce_args c = *p;
        c.row_begin = p->hub_chunk_begin; c.row_end = p->hub_chunk_end; c.rp_stride = 1;
        c.n_dst = p->n_hub_chunks; c.out = p->hub_scratch; c.ldo = p->F;
        c.op = TFGX_SUM; c.act = TFGX_ACT_NONE; c.accumulate = 0;
        c.self_coef = nullptr; c.bias = nullptr; c.add_x = nullptr; c.mean_count = nullptr;
        c.hub_threshold = 0; c.hub_rows = nullptr; c.hub_chunk_ptr = nullptr; c.hub_chunk_begin = nullptr;
        c.hub_chunk_end = nullptr; c.n_hub_rows = 0; c.n_hub_chunks = 0; c.hub_scratch = nullptr;
        c.row_order = nullptr;          // (a walk order names DESTINATION rows; the chunk launch walks chunks)
        c.hub_order_slot = nullptr;
        const int rc = tfgx_segment_reduce_f32(&c, stream_);
        if (rc != TFGX_OK) return rc;
    }
    FArgs a;
    a.hub_threshold = use_hub ? p->hub_threshold : 0;
    a.n_hub = use_hub ? int32_t(p->n_hub_rows) : 0;
    a.hub_rows = p->hub_rows; a.hub_chunk_ptr = p->hub_chunk_ptr; a.hub_scratch = p->hub_scratch;
    a.row_order = p->row_order;
    a.row_ptr = p->row_begin; a.col = p->col; a.w = p->w; a.n_dst = p->n_dst; a.x = p->x; a.ldx = p->ldx; a.F = int32_t(p->F);
    a.op = p->op; a.self_coef = p->self_coef; a.mean_count = p->mean_count;
    a.B = B; a.ldb = ldb; a.bias = bias; a.act = act; a.N = int32_t(N); a.C = C; a.ldc = ldc;
    a.hub_order_slot = use_hub ? p->hub_order_slot : nullptr;
    a.x_tail = p->x_tail; a.ld_tail = p->ld_tail; a.f_main = int32_t(p->x_tail ? p->f_main : p->F);
    a.edge_tail = p->x_tail ? p->edge_tail : nullptr; a.ld_edge_tail = p->ld_edge_tail;
    a.agg = p->out; a.ld_agg = p->ldo;
    a.KP = int32_t((p->F + 1) / 2 * 2);
    a.n_blocks = int32_t((N + 127) / 128) * 4;          // 32-column blocks, in groups of four (columns >= N are zero in LDS)
    a.NL = fused_resident_cols(a.KP, 32 * a.n_blocks);
    a.LDW = a.NL + 8;
    a.n_tiles = (p->n_dst + kTileRows - 1) / kTileRows;
#ifdef TFGX_FUSED_DEBUG
    a.dbg = getenv("TFGX_FUSED_DBG") ? atoi(getenv("TFGX_FUSED_DBG")) : 0;
    if (a.dbg) fprintf(stderr, "tfgx_aggregate_gemm_f32: DEBUG MODE %d — results are not valid\n", a.dbg);
#else
    a.dbg = 0;
#endif
    // per DEVICE (one process may drive several): compute-unit count, and the dynamic-LDS attribute of each instantiation
    constexpr int kMaxDev = 64;
    static int cus_of[kMaxDev] = {0};
    int dev = 0;
    TFGX_HIP_CHECK(hipGetDevice(&dev));
    TFGX_REQUIRE(dev >= 0 && dev < kMaxDev, "device ordinal out of range");
    if (cus_of[dev] == 0) {
        hipDeviceProp_t prop;
        TFGX_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        cus_of[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int cus = cus_of[dev];
    const size_t lds_bytes = fused_lds_bytes(a.KP, a.LDW);
    const int64_t wgs = a.n_tiles < cus ? a.n_tiles : cus;
    hipStream_t stream = as_stream(stream_);
    const bool weighted = p->w != nullptr;
#define TFGX_FUSED_GO(GG, WW)                                                                                        \
    do {                                                                                                             \
        static bool attr_set[kMaxDev] = {false};                                                                     \
        if (!attr_set[dev]) {                                                                                        \
            TFGX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(agg_gemm_kernel<GG, WW>),               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, int(kFusedLdsLimit)));    \
            attr_set[dev] = true;                                                                                    \
        }                                                                                                            \
        agg_gemm_kernel<GG, WW><<<dim3(unsigned(wgs)), dim3(kFusedThreads), lds_bytes, stream>>>(a);                 \
    } while (0)
    if (p->F <= 64) {
        if (weighted) TFGX_FUSED_GO(16, true);
        else TFGX_FUSED_GO(16, false);
    } else {
        if (weighted) TFGX_FUSED_GO(32, true);
        else TFGX_FUSED_GO(32, false);
    }
#undef TFGX_FUSED_GO
    TFGX_LAUNCH_CHECK("agg_gemm_kernel");
    return TFGX_OK;
}
