// C = act(A[M,K] @ B[K,N] + bias), fp32 in / fp32 accumulate on the matrix cores.
// Replaces `x @ kernel` (+ bias, activation) next to the aggregation:
// tf_geometric/nn/conv/gcn.py:272, nn/conv/gat.py:52-70, nn/conv/graph_sage.py:43-44,199,208-209.
//
// v_mfma_f32_32x32x2_f32 (f32-input MFMA): exact fp32, bitwise a k-ordered FMA chain per output element
// (no bf16/tf32 down-cast: parity is 1e-5 against an fp32 CPU reference).  M is the node count (10^5..10^8),
// K and N are feature widths (16..1433), so the kernel is a tall-skinny GEMM close to the HBM roofline:
// the A panel is streamed once, B (<= 1.5 MB) lives in L2/LDS.
//
// Two kernels:
//  * gemm_rows_kernel (N <= 256, K % 4 == 0, B within 160 KB of LDS, M >= 32768 — the shapes the layers produce on
//    large graphs): persistent, B resident in LDS, A loaded straight into the MFMA operand layout, no barriers after
//    the B load.  See its own header comment below.
//  * gemm_kernel (everything else): BM x BN tile per 256-thread workgroup (4 waves), BK = 16.  A is staged transposed
//    in LDS (As[k][m]) so the MFMA A operand (lane l: A[m = l&31][k = l>>5]) is a conflict-free ds_read_b32 over
//    consecutive m; B is staged as Bs[k][n].  Global loads for tile t+1 are issued before the MFMAs of tile t
//    (register staging), LDS is single-buffered.  On long K (>= 256) the 64 x 64 wave tile reads the LDS operands of k step
//    kk + 2 before the MFMAs of step kk (template flag AHEAD).  Optional split-K over blockIdx.y for small M with a long K.
#include "tfgx_common.h"
#include "tfgx_mfma.h"
#include <cstdlib>
#include <type_traits>

namespace tfgx {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

constexpr int BK = 16;   // 16 keeps load-staging registers low enough for 3 workgroups per CU (BK = 32: 2)
#ifndef TFGX_ROWS_EXPERIMENT
#define TFGX_ROWS_EXPERIMENT 0    // 3 = instrumented build: per-wave clocks of the tile loop and its phases (tfgx_debug_rows_stats / _waves)
#endif
#ifndef TFGX_GEMM_PREFETCH_TILES
#define TFGX_GEMM_PREFETCH_TILES 1   // developer A/B: 2 = gemm_kernel's long-K instantiation keeps two tiles of global loads in flight
#endif
#ifndef TFGX_ROWS_VEC_STORE
#define TFGX_ROWS_VEC_STORE 0     // 16-byte epilogue stores through a quad transpose of the accumulators: +4..9 % while every store carried
                                  // its own 64-bit address arithmetic; once the addresses moved off the vector ALU the 65 VALU ops per
                                  // 32 x 32 block of the transpose cost more than 96 fewer store instructions save (2.4 M x 100 -> 256:
                                  // 1.20 -> 1.155 ms, 170 k x 128 -> 256: 0.121 -> 0.113 without it) — kept as a switch
#endif

template <int BM, int BN, int WM, int WN, bool AV4, bool BV4, bool AHEAD>
__global__ __launch_bounds__(kBlock) void gemm_kernel(const float* __restrict__ A, int64_t lda,
                                                      const float* __restrict__ B, int64_t ldb,
                                                      const float* __restrict__ bias, int act,
                                                      float* __restrict__ C, int64_t ldc, int64_t M, int K, int N,
                                                      int n_tiles_n, int act_cols, int k_chunk, int64_t c_split_stride,
                                                      int two_level_on)
{
    // split-K (small M, long K — Cora's 2708 x 1433): blockIdx.y owns k in [y * k_chunk, (y + 1) * k_chunk) and writes
    // its partial product to its own slab of C (a workspace; bias / activation are applied by splitk_reduce_kernel)
    if (gridDim.y > 1) {
        const int k_lo = blockIdx.y * k_chunk;
        A += k_lo;
        B += int64_t(k_lo) * ldb;
        C += int64_t(blockIdx.y) * c_split_stride;
        K = min(k_chunk, K - k_lo);
    }
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per workgroup");
    constexpr int A_LOADS = BM * BK / 4 / kBlock;   // float4 per thread
    constexpr int B_LOADS = (BK * BN / 4 + kBlock - 1) / kBlock;
    constexpr int B_THREADS_PER_ROW = BN / 4;
    constexpr int LDA_S = BM + 1, LDB_S = BN + 4;   // +1: spreads the transposed A stores over banks

#ifndef TFGX_GEMM_DOUBLE_LDS
#define TFGX_GEMM_DOUBLE_LDS 0       // developer A/B: 1 = two LDS stages, ONE barrier per k tile.  Measured and dropped (round 6, same
#endif                               // box, alternating with hipBLASLt): 170 k x 1433 -> 256 1.103 -> 1.140 ms, 233 k x 602 -> 128
                                     // 0.367 -> 0.382, 2.4 M x 101 -> 256 1.578 -> 1.673 — 3-6 % slower on every shape of the sweep
    // two LDS stages: tile t + 1 is written into the other stage at the START of step t (its global loads were issued a whole
    // step earlier), so a step is store -> loads of t + 2 -> MFMAs of t -> one barrier instead of two
    constexpr int STAGES = TFGX_GEMM_DOUBLE_LDS ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float As_all[STAGES * BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs_all[STAGES * BK * LDB_S];
    float* As = As_all;
    float* Bs = Bs_all;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int64_t tile = blockIdx.x;
    const int64_t m0 = (tile / n_tiles_n) * BM;
    const int n0 = int(tile % n_tiles_n) * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.0f;

    // Long K on the narrow tiles (N <= 64: Q / K / V projections of a 602- or 1433-wide input): TWO-LEVEL accumulation.
    // A k-ordered fp32 chain of n terms carries rounding error ~ eps * n / sqrt(2) (in units of one term); flushing the
    // running MFMA accumulator into a second register set every 64 k (4 tiles) makes it ~ eps * sqrt(64 n) — 3x less at
    // K = 602 — for one v_add per accumulator register per 4 tiles.  Measured at Reddit shape (tools/
    // reddit_gat_attribution.py): the demo-literal GAT layer (d_head = 1: exp() turns the projections' absolute error
    // into relative error of the attention weights) was 2.2e-5 off the float64 value with the plain chain — the same as
    // hipBLASLt's fp32 GEMM — against 7e-6 for a blocked CPU BLAS.  Wide tiles (TM * TN = 4) keep the single chain: a
    // second set of 64 accumulator registers would cost a workgroup per CU.
    constexpr bool kTwoLevel = TM * TN <= 2;
    constexpr int kFlushTiles = 4;
    f32x16 tot[kTwoLevel ? TM : 1][kTwoLevel ? TN : 1];
    const bool two_level = kTwoLevel && two_level_on && K > 8 * BK;
    if (kTwoLevel) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int t = 0; t < 16; ++t) tot[i][j][t] = 0.0f;
    }
    int tiles_in_chain = 0;

    typedef float RegsA[A_LOADS * 4];
    typedef float4 RegsB[B_LOADS];

    auto load_tiles = [&](int k0, RegsA& ra, RegsB& rb) {
        if (AV4) {
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const int f = tid + i * kBlock;          // float4 index inside the BM x BK tile
                const int row = f / (BK / 4), kq = (f % (BK / 4)) * 4;
                const int64_t gm = m0 + row;
                const int gk = k0 + kq;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gm < M) {
                    const float* p = A + gm * lda + gk;
                    if (gk + 3 < K) {
                        // 4-byte alignment is enough for global_load_dwordx4: rows that are not 16-byte aligned (K = 1433,
                        // 602: lda % 4 != 0) may take this path too — one 16-byte load per lane instead of four dword loads
                        // (launch_gemm decides; a branch-free clamped-address form of this load was measured and LOST 10-50 %
                        // to its address arithmetic and selects: profiles/r03_gemm_sweep.jsonl)
                        const f32x4_a4 t = *reinterpret_cast<const f32x4_a4*>(p);
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    } else {
                        if (gk < K) v.x = p[0];
                        if (gk + 1 < K) v.y = p[1];
                        if (gk + 2 < K) v.z = p[2];
                    }
                }
                ra[4 * i + 0] = v.x;
                ra[4 * i + 1] = v.y;
                ra[4 * i + 2] = v.z;
                ra[4 * i + 3] = v.w;
            }
        } else {
            // rows that are not 16-byte aligned (K = 1433, 3703 ...): dword loads, 16 consecutive lanes on 16 consecutive
            // k of one row, so one instruction touches 4 rows x 64 contiguous bytes
#pragma unroll
            for (int j = 0; j < A_LOADS * 4; ++j) {
                const int e = tid + j * kBlock;
                const int64_t gm = m0 + e / BK;
                const int gk = k0 + e % BK;
                ra[j] = (gm < M && gk < K) ? A[gm * lda + gk] : 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int f = tid + i * kBlock;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < BK * BN / 4) {
                const int kr = f / B_THREADS_PER_ROW, nq = (f % B_THREADS_PER_ROW) * 4;
                const int gk = k0 + kr, gn = n0 + nq;
                if (gk < K) {
                    const float* p = B + int64_t(gk) * ldb + gn;
                    if (BV4 && gn + 3 < N) {
                        v = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (gn < N) v.x = p[0];
                        if (gn + 1 < N) v.y = p[1];
                        if (gn + 2 < N) v.z = p[2];
                        if (gn + 3 < N) v.w = p[3];
                    }
                }
            }
            rb[i] = v;
        }
    };

    auto store_tiles = [&](const RegsA& ra, const RegsB& rb) {
        if (AV4) {
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const int f = tid + i * kBlock;
                const int row = f / (BK / 4), kq = (f % (BK / 4)) * 4;
#pragma unroll
                for (int c = 0; c < 4; ++c) As[(kq + c) * LDA_S + row] = ra[4 * i + c];
            }
        } else {
#pragma unroll
            for (int j = 0; j < A_LOADS * 4; ++j) {
                const int e = tid + j * kBlock;
                As[(e % BK) * LDA_S + e / BK] = ra[j];
            }
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int f = tid + i * kBlock;
            if (f < BK * BN / 4) {
                const int kr = f / B_THREADS_PER_ROW, nq = (f % B_THREADS_PER_ROW) * 4;
                *reinterpret_cast<float4*>(&Bs[kr * LDB_S + nq]) = rb[i];
            }
        }
    };

    // one BK-deep tile: registers -> LDS, barrier, the global loads of tile k_next into the SAME registers, multiply, barrier
    // (DOUBLE: the tile is already in the current stage; only the multiply and the closing barrier)
    auto one_tile = [&](int k0, int k_next, RegsA& ra, RegsB& rb, auto dbl) {
        if constexpr (!decltype(dbl)::value) {
            store_tiles(ra, rb);
            __syncthreads();
            if (k_next < K) load_tiles(k_next, ra, rb);
        }
        const int kh = lane >> 5, l31 = lane & 31;
        const int kmax = min(BK, K - k0);   // the zero-padded tail of the last tile is skipped, not multiplied
        auto kstep = [&](int kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(kk + kh) * LDA_S + wm * WM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + kh) * LDB_S + wn * WN + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        };
        if constexpr (AHEAD) {
            // Long K on the wide wave tiles (the host picks this instantiation from K >= 256 and a 64 x 64 wave tile): one
            // straight-line block per tile, the operands of step kk + 2 read from LDS BEFORE the MFMAs of step kk are issued.  In the guarded form below every step is a
            // basic block of its own — ds_read, s_waitcnt lgkmcnt(0), 4 MFMAs — and the LDS round trip of a step starts only
            // after the previous step's last MFMA has been issued (+3.5 % at 170 k x 1433 -> 256 and 233 k x 602 -> 256,
            // profiles/r05_gemm_generic_ab.jsonl).  The padded tail of the last tile is multiplied, not skipped: A and B are
            // both zero-filled past K, so it adds +0 to every accumulator (same bits; at K = 101 the 11 wasted k of 112 cost
            // 4 %, which is why short K keeps the guards).  The sched_barriers pin the order: left alone the scheduler sinks
            // each read next to its MFMA again.  A guarded second body for the last tile inside this loop made the compiler
            // keep two accumulator sets (128 AGPRs).
            (void)kmax;
            auto fetch = [&](int kk, float (&a)[TM], float (&b)[TN]) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = As[(kk + kh) * LDA_S + wm * WM + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + kh) * LDB_S + wn * WN + j * 32 + l31];
            };
            auto mfmas = [&](const float (&a)[TM], const float (&b)[TN]) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            };
            float a0[TM], b0[TN], a1[TM], b1[TN];
            fetch(0, a0, b0);
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                fetch(kk + 2, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 4 < BK) fetch(kk + 4, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // the zero-padded tail of the last tile is skipped by a wave-uniform guard per step (one accumulator copy, no
            // second loop body)
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                if (kk < kmax) kstep(kk);
            }
        }
        if (kTwoLevel) {
            if (two_level && ++tiles_in_chain == kFlushTiles) {      // wave-uniform
                tiles_in_chain = 0;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            tot[i][j][t] += acc[i][j][t];
                            acc[i][j][t] = 0.0f;
                        }
            }
        }
        __syncthreads();
    };
    if constexpr (STAGES == 2) {
        RegsA ra;
        RegsB rb;
        load_tiles(0, ra, rb);
        store_tiles(ra, rb);                      // stage 0
        __syncthreads();
        if (BK < K) load_tiles(BK, ra, rb);
        int stage = 0;
        for (int k0 = 0; k0 < K; k0 += BK) {
            if (k0 + BK < K) {                    // tile t + 1 into the other stage (last read two steps ago, a barrier since)
                As = As_all + (stage ^ 1) * BK * LDA_S;
                Bs = Bs_all + (stage ^ 1) * BK * LDB_S;
                store_tiles(ra, rb);
            }
            if (k0 + 2 * BK < K) load_tiles(k0 + 2 * BK, ra, rb);
            As = As_all + stage * BK * LDA_S;
            Bs = Bs_all + stage * BK * LDB_S;
            one_tile(k0, k0 + BK, ra, rb, std::true_type{});
            stage ^= 1;
        }
    } else if constexpr (AHEAD && TFGX_GEMM_PREFETCH_TILES == 2) {
        // two register stages: the loads of tile t + 2 are issued when tile t starts multiplying, so they have two tiles of
        // MFMA time (not one) to come back from HBM before store_tiles waits for them
        RegsA ra0, ra1;
        RegsB rb0, rb1;
        load_tiles(0, ra0, rb0);
        if (BK < K) load_tiles(BK, ra1, rb1);
        for (int k0 = 0; k0 < K; k0 += 2 * BK) {
            one_tile(k0, k0 + 2 * BK, ra0, rb0, std::false_type{});
            if (k0 + BK < K) one_tile(k0 + BK, k0 + 3 * BK, ra1, rb1, std::false_type{});
        }
    } else {
        RegsA ra;
        RegsB rb;
        load_tiles(0, ra, rb);
        for (int k0 = 0; k0 < K; k0 += BK) one_tile(k0, k0 + BK, ra, rb, std::false_type{});
    }
    if (kTwoLevel) {
        if (two_level) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc[i][j][t] += tot[i][j][t];
        }
    }

    // epilogue: D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // Two phases on purpose: bias + activation are applied IN PLACE first, then every store reads its own accumulator
    // register.  Computing each result into a temporary right before its store makes all 64 stores share one data
    // register, and the compiler then drains vmcnt(0) between consecutive stores (write-after-read on the register of
    // an in-flight store): 64 fully serialised round trips per tile.
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gn = n0 + wn * WN + j * 32 + l31;
        const float bv = (bias && gn < N) ? bias[gn] : 0.0f;
        const int a_j = gn < act_cols ? act : TFGX_ACT_NONE;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = apply_act(acc[i][j][t] + bv, a_j);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gn = n0 + wn * WN + j * 32 + l31;
        if (gn >= N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float* cp = C + (m0 + wm * WM + i * 32 + 4 * lh) * ldc + gn;
            const int64_t rows_left = M - (m0 + wm * WM + i * 32 + 4 * lh);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int dr = (t & 3) + 8 * (t >> 2);
                if (dr < rows_left) cp[int64_t(dr) * ldc] = acc[i][j][t];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Row-streaming variant for the shape this library actually meets: M = nodes (10^5..10^8), K and N = feature widths
// with K*N small enough that the WHOLE weight matrix lives in LDS (160 KB per CU).
//
// One persistent 512-thread workgroup per CU loads B into LDS once; after that single barrier the 8 waves never
// synchronise again.  Each wave owns 32-row x N output tiles (TN = ceil(N/32) accumulators) and reads its A rows
// straight from global memory INTO THE MFMA OPERAND LAYOUT: v_mfma_f32_32x32x2_f32 pairs any two k's as long as A and
// B agree, so lane (m = lane & 31, kh = lane >> 5) loads the 64 contiguous bytes A[m][k0 + 16 kh .. + 15]
// and the i-th MFMA of the step multiplies k = k0 + 16 kh + i against B row k0 + 16 kh + i from LDS.  A never
// touches LDS, there is no transposition and no per-k-step barrier; the next step's A registers are loaded before
// the current step's 16 x TN MFMAs, B operands are double-buffered in registers one MFMA group ahead, and because
// waves drift apart the store epilogue of one wave overlaps the MFMAs of the other wave on its SIMD.
// Arithmetic: fp32 FMA chain per output element in the k order above (a permutation of 0..K-1).
// NG consecutive MFMA groups of gemm_rows_kernel (one group = one k pair x TN accumulators; a full step is 16 groups),
// B operands read from LDS one group ahead.  The sched_barriers pin that order: left alone, the scheduler sinks every
// ds_read next to its MFMA (fewest live registers), which puts a full LDS round trip in front of each group.
#ifndef TFGX_ROWS_LDS_AHEAD
#define TFGX_ROWS_LDS_AHEAD 1      // developer A/B: how many MFMA groups ahead the B operands are read from LDS (1 or 2)
#endif
constexpr int kRowsCounterStride = 32;          // uints between the per-slot tile counters (one 128-byte line each)
constexpr int kRowsCounterSlots = 16;           // >= waves per workgroup of any row kernel
#if TFGX_ROWS_EXPERIMENT == 3
__device__ uint64_t g_rows_dbg[16];     // [12]: cycles inside the MFMA groups of full steps, [13]: cycles in epilogues
__device__ uint64_t g_rows_wave[4096 * 4];   // per wave of the LAST launch: loop start tick, loop end tick, tiles, hw id     // loop cycles, loop 100 MHz ticks, tiles, waves, max / min loop ticks of a wave, prologue ticks, max end tick - min start tick
#endif
// One LDS dword as ONE ds_read_b32 with a 16-bit immediate offset.  A plain load lets the backend pair neighbours into
// ds_read2_b32, whose two 8-bit offsets reach only 1020 bytes: the 16 B rows of a k-step lie 9 KB apart, so it paid a
// v_add_u32 per pair — 13 vector-ALU instructions per step that the MFMA stream of the SIMD's other wave starves.  A relaxed
// wavefront-scope atomic load is never merged and costs nothing else (no fence, same instruction).
#ifndef TFGX_ROWS_LDS_SINGLE
#define TFGX_ROWS_LDS_SINGLE 1     // developer A/B: 0 = plain loads (the backend pairs them into ds_read2_b32 + v_add_u32)
#endif
#ifndef TFGX_ROWS_ZERO_FIRST
#define TFGX_ROWS_ZERO_FIRST 1     // developer A/B: 0 = accumulators zeroed with v_mov before every tile
#endif
__device__ __forceinline__ float lds_read_f32(const float* p)
{
#if TFGX_ROWS_LDS_SINGLE
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
    return *p;
#endif
}

// ZERO_FIRST: the first group multiplies into a literal zero (the MFMA's C operand as an inline constant) — a tile's first
// step then needs no zeroed accumulators (16 TN v_mov per tile otherwise).
template <int TN, int NG, bool ZERO_FIRST = false, class AOF>
__device__ __forceinline__ void rows_mfma_groups(f32x16 (&acc)[TN], AOF a_of, const float* b_s)
{
    constexpr int LDB_S = TN * 32 + 8;
    constexpr int AH = TFGX_ROWS_LDS_AHEAD, NB = AH + 1;
    float bb[NB][TN];
#pragma unroll
    for (int p = 0; p < AH; ++p)
        if (p < NG) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bb[p][j] = lds_read_f32(b_s + p * LDB_S + j * 32);
        }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        if (i + AH < NG) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bb[(i + AH) % NB][j] = lds_read_f32(b_s + (i + AH) * LDB_S + j * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (ZERO_FIRST && TFGX_ROWS_ZERO_FIRST && i == 0) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_of(i), bb[i % NB][j], zero, 0, 0, 0);
            } else {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_of(i), bb[i % NB][j], acc[j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// threads per workgroup: 8 waves, 2 per SIMD, <= 256 registers each (TN = 8 needs 218).  12 waves for the TN <= 4
// kernels (< 170 registers) measured 2 % slower than 8.
#ifndef TFGX_ROWS_THREADS_NARROW
#define TFGX_ROWS_THREADS_NARROW 512   // developer A/B: workgroup size of the TN <= 4 kernels
#endif
#ifndef TFGX_ROWS_THREADS_WIDE
#define TFGX_ROWS_THREADS_WIDE 512
#endif
template <int TN>
constexpr int rows_threads() { return TN <= 4 ? TFGX_ROWS_THREADS_NARROW : TFGX_ROWS_THREADS_WIDE; }

template <int TN>
__global__ __launch_bounds__(rows_threads<TN>()) void gemm_rows_kernel(const float* __restrict__ A, int64_t lda,
                                                        const float* __restrict__ B, int64_t ldb,
                                                        const float* __restrict__ bias, int act, int act_cols,
                                                        float* __restrict__ C, int64_t ldc, int64_t M, int K, int N,
                                                        int64_t n_tiles, int b_vec4, int two_level_on,
                                                        unsigned int* __restrict__ tile_counter)
{
    constexpr int LDB_S = TN * 32 + 8;   // (4 * LDB_S) % 64 == 32: the two half-waves hit disjoint banks
    constexpr int NQ = LDB_S / 4;
    extern __shared__ __attribute__((aligned(16))) float Bs[];   // [K][LDB_S], columns >= N zero

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, kh = lane >> 5;
#if TFGX_ROWS_EXPERIMENT == 3
    const uint64_t dbg_wk = wall_clock64();
#endif

    constexpr int NT = rows_threads<TN>();
#ifndef TFGX_ROWS_B_BATCH
#define TFGX_ROWS_B_BATCH 8
#endif
    // B -> LDS, TFGX_ROWS_B_BATCH loads in flight per thread before the first LDS store: with one load per iteration the
    // prologue was a chain of ~17 L2 round trips (128 x 256 floats over 512 threads) — a quarter of the whole launch at
    // ogbn-arxiv size (170 k rows)
    constexpr int BB = TFGX_ROWS_B_BATCH;
    for (int f0 = tid; f0 < K * NQ; f0 += NT * BB) {
        float4 v[BB];
#pragma unroll
        for (int u = 0; u < BB; ++u) {
            const int f = f0 + u * NT;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < K * NQ) {
                const int k = f / NQ, n4 = (f - k * NQ) * 4;
                const float* p = B + int64_t(k) * ldb + n4;
                if (b_vec4 && n4 + 3 < N) {
                    v[u] = *reinterpret_cast<const float4*>(p);
                } else {
                    if (n4 < N) v[u].x = p[0];
                    if (n4 + 1 < N) v[u].y = p[1];
                    if (n4 + 2 < N) v[u].z = p[2];
                    if (n4 + 3 < N) v[u].w = p[3];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < BB; ++u) {
            const int f = f0 + u * NT;
            if (f < K * NQ) {
                const int k = f / NQ, n4 = (f - k * NQ) * 4;
                *reinterpret_cast<float4*>(&Bs[k * LDB_S + n4]) = v[u];
            }
        }
    }
    __syncthreads();

    const int nfull = K / 32;                 // steps of 32 k's with every operand valid
    const int nsteps = (K + 31) / 32;
    const int64_t stride = int64_t(gridDim.x) * (NT / 64);

    // Per step a lane holds 16 consecutive k's of its row (4 dwordx4 loads).  Full step ks: k = 32 ks + 16 kh + i.
    // Tail step (T = K % 32 left, a multiple of 4): the two half-waves split it evenly, k = 32 nfull + kh T/2 + i for
    // i < T/2, so a tail costs T/2 fully used MFMA groups; T/2 may be only 8-byte aligned, hence the aligned(8) vector
    // type (global_load_dwordx4 itself has no 16-byte requirement).
    // No load is PREDICATED (an exec-masked load makes the waitcnt pass lose count and drain vmcnt(0), which serialises the
    // prefetch); the only branches around loads are wave-uniform.  Rows >= M of the last tile are clamped to row M - 1 through
    // the per-lane offset (their values are never stored), vectors of the tail step to k <= K - 4 (accounted for there).
    typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
    const int half_t = (K - nfull * 32) >> 1;
    // the A registers of a step, kept as the four 16-byte vectors they are loaded as: the per-step hand-over (cur = nxt) then
    // costs 8 64-bit moves instead of 16 v_mov_b32 (every vector-ALU instruction here waits ~38 cycles for an issue slot)
    f32x4 cur[4], nxt[4];
    // ADDRESSES.  In-kernel clocks (TFGX_ROWS_EXPERIMENT=3: cycles inside the MFMA groups / in the epilogue / elsewhere)
    // showed a wave spending a third of every tile OUTSIDE its MFMA groups, at ~38 cycles per VALU instruction: while the other
    // wave of the SIMD streams MFMAs, a vector ALU instruction gets an issue slot only now and then, and when both waves are
    // in such a stretch the MFMA port idles.  So the per-step and per-store address arithmetic is kept off the vector ALU:
    // every global address is  (wave-uniform 64-bit base, SALU)  +  (per-lane 32-bit byte offset, computed ONCE)  +  immediate
    // The tile index is made provably uniform with readfirstlane.  LOADS of the A rows: global loads on a uniform base pointer;
    // STORES of full tiles: raw buffer stores (a 128-bit resource descriptor in SGPRs rebuilt per tile with scalar instructions,
    // the per-lane offset as voffset, the row as a scalar offset, the column block as an immediate).  Raw buffer LOADS — whose
    // out-of-extent reads return 0, so the last tile and the prefetch past it would need no clamps — measured slower at
    // K = 256 and are a compile-time switch.
    constexpr int kRsrcFlags = 0x00020000;                       // raw buffer, dword data format (gfx9 family): the stores' descriptor
#ifndef TFGX_ROWS_BUFFER_LOADS
#define TFGX_ROWS_BUFFER_LOADS 0   // developer A/B: 1 = A rows through raw buffer loads too.  Same-box A/B: K = 100 / 128 gain 1-2 %,
                                   // K = 256 loses 4-10 % (2.4 M x 256 -> 40: 0.717 -> 0.790 ms) — global loads in saddr form stay
#endif
    const uint32_t a_voff = uint32_t((int64_t(l31) * lda + 16 * kh) * 4);
    const int64_t last_tile = n_tiles - 1;
    const uint32_t a_voff_last = uint32_t((min(int64_t(l31), M - 1 - last_tile * 32) * lda + 16 * kh) * 4);   // rows clamped to M - 1
#if TFGX_ROWS_BUFFER_LOADS
    // the tail step (K % 32 != 0): the half-waves split the K % 32 remaining k, vectors clamped to stay inside the row
    uint32_t a_voff_tail[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a_voff_tail[u] = uint32_t((int64_t(l31) * lda + min(nfull * 32 + half_t * kh + 4 * u, K - 4)) * 4);
#endif
    auto load_a = [&](f32x4 (&r)[4], int64_t t, int ks) {
#if TFGX_ROWS_BUFFER_LOADS
        // reads past the descriptor's extent return 0 and touch no memory: rows >= M of the last tile and the prefetch past the
        // last tile need no clamps
        const int64_t rows_left = M - t * 32;                                    // <= 0 past the last tile
        const int64_t bytes = rows_left > 0 ? rows_left * lda * 4 : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(A + (rows_left > 0 ? t : 0) * 32 * lda), 0, int(bytes < 0x7fffffff ? bytes : 0x7fffffff), kRsrcFlags);
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        if (ks < nfull) {                                                        // uniform
#pragma unroll
            for (int u = 0; u < 4; ++u)
                r[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, a_voff + 16 * u, ks * 128, 0));
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                r[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, a_voff_tail[u], 0, 0));
        }
#else
        // global loads in saddr form: (uniform 64-bit base) + (per-lane 32-bit byte offset) + immediate.  Past the end the last
        // tile is re-read (harmless); its rows >= M are clamped to M - 1 through the per-lane offset (values never stored)
        const int64_t tc = t < last_tile ? t : last_tile;                        // uniform
        if (ks < nfull) {                                                        // uniform
            const char* base = reinterpret_cast<const char*>(A + tc * 32 * lda + ks * 32);
            const uint32_t off = tc == last_tile ? a_voff_last : a_voff;
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const f32x4_a8*>(base + off + 16 * u);
        } else {
            // the tail step, once per tile: per-lane 64-bit addresses with their clamps (written differently from the branch
            // above ON PURPOSE: two branches of the same shape get merged into one per-lane address computation, and the
            // full steps lose the saddr form)
            const int64_t gm = min(tc * 32 + l31, M - 1);
            const int kb = ks * 32 + half_t * kh;
            const float* p = A + gm * lda;
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const f32x4_a8*>(p + min(kb + 4 * u, K - 4));
        }
#endif
    };

    f32x16 acc[TN];       // never zeroed: a tile's first MFMA group takes a literal zero as its C operand (ZERO_FIRST)
#if !TFGX_ROWS_ZERO_FIRST
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[j][t] = 0.0f;
            asm volatile("" : "+v"(acc[j]));
        }
    };
    zero_acc();
#endif
    // DYNAMIC tile order (tile_counter != nullptr): every wave claims its tiles from one device counter, one tile ahead — the
    // claim is ISSUED at the top of a tile and its value first READ at that tile's last k-step, where the next tile's A rows
    // are prefetched.  In-kernel clocks (TFGX_ROWS_EXPERIMENT=3, tools/rows_clock_probe.py) show the waves of one launch
    // spending 1087 .. 1512 us (mean 1300) over the same 36-37 tiles of 2.4 M x 128 -> 256: with a fixed tile -> wave map
    // the launch lasts as long as its slowest wave.
    unsigned int next_raw = 0;                 // lane 0: the claimed index of the next tile (valid once the atomic has returned)
    // kRowsCounterSlots pools (counters kRowsCounterStride uints = 128 bytes apart), pool p owning the tiles = p (mod pools):
    // a returning atomic on ONE address costs ~12 ns (measured: 4096 initial claims = 50 us), so a single counter would
    // serialise 75 000 claims into 0.9 ms.  A pool is shared by BOTH waves of a SIMD (wave & 3) of a quarter of the
    // workgroups (spread over all XCDs): the SIMD's older wave wins the MFMA port whenever both are ready and runs ~25 % faster than the younger one
    // for the whole launch (per-slot pools left the spread as it was) — in a shared pool it simply takes more tiles.
    const int pool = ((int(blockIdx.x) >> 3) & 3) * 4 + (wave & 3);      // blockIdx >> 3: workgroups i .. i + 7 sit on the 8 XCDs, so every pool spans all of them
    auto claim = [&]() {
        if (lane == 0) next_raw = atomicAdd(tile_counter + kRowsCounterStride * pool, 1u);
    };
    auto next_of = [&](int64_t t) -> int64_t {
        return tile_counter ? int64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(next_raw)))) * kRowsCounterSlots +
                                  __builtin_amdgcn_readfirstlane(pool)
                            : t + stride;
    };
    auto adv = [&](int64_t& t, int& ks) {
        const bool same = ks + 1 < nsteps;
        t = same ? t : next_of(t);
        ks = same ? ks + 1 : 0;
    };
    auto prefetch = [&](int64_t t, int ks) {
        adv(t, ks);
        load_a(nxt, t, ks);
    };
    auto rotate = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
    };

#if TFGX_ROWS_VEC_STORE
    // wave-uniform.  Wide outputs only: same-box A/B (profiles/r04_gemm_sweep.jsonl) 2.4 M x 100 -> 256: 1.349 -> 1.290 ms,
    // 170 k x 128 -> 256: 0.140 -> 0.128; at TN <= 4 (100 -> 128 / 64) the transposes cost more than the 4x fewer store
    // instructions give back (+3 .. 4 %), so narrow outputs keep the per-column stores
    const bool vec_store = TN >= 5 && (N % 4 == 0) && (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#endif
    const uint32_t c_off = uint32_t((int64_t(4 * kh) * ldc + l31) * 4);                               // per-column stores: row 4 kh, column l31
    const uint32_t c_off_vec = uint32_t((int64_t((lane & 3) + 4 * kh) * ldc + (l31 >> 2) * 4) * 4);   // 16-byte stores: row (lane & 3) + 4 kh
    float bv[TN];   // this lane's bias values, loaded once (a per-tile load would put a memory round trip in every epilogue)
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = (bias && j * 32 + l31 < N) ? bias[j * 32 + l31] : 0.0f;

    int64_t tile = int64_t(blockIdx.x) * (NT / 64) + __builtin_amdgcn_readfirstlane(wave);
    if (tile_counter) {
        claim();
        tile = next_of(0);
        claim();
    }
    if (tile < n_tiles) load_a(cur, tile, 0);
    // Consume the first A registers here so their wait sits in front of the loop.  Otherwise every step carries a
    // "first iteration" vmcnt wait; harmless in steady state (only the 4 prefetch loads are in flight), but right after an
    // epilogue the 16 * TN stores are in flight too and that wait stalls the wave until they have drained.
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" ::"v"(cur[u]));
    // narrow outputs with a long K (TN <= 2, K > 128): two-level accumulation like gemm_kernel's narrow tiles — the chain
    // is flushed into `tot` every 64 k (2 steps), rounding error ~ eps * sqrt(64 K) instead of ~ eps * K
    constexpr bool kTwoLevel = TN <= 2;
    const bool two_level = kTwoLevel && two_level_on && K > 128;
    f32x16 tot[kTwoLevel ? TN : 1];       // (assigned by the first fold of every tile)
#if TFGX_ROWS_EXPERIMENT == 3
    // instrumented build (results valid, timing perturbed by four scalar reads per wave): shader-clock cycles and constant
    // 100 MHz ticks spent inside the tile loop, summed over waves -> tfgx_debug_rows_stats
    const uint64_t dbg_c0 = __builtin_readcyclecounter(), dbg_w0 = wall_clock64();
    uint64_t dbg_tiles = 0, dbg_mf = 0, dbg_ep = 0, dbg_h0 = 0, dbg_h1 = 0;
#endif
    while (tile < n_tiles) {
#if TFGX_ROWS_EXPERIMENT == 3
        ++dbg_tiles;
#endif
        // one k-step: prefetch the next step's A rows, 16 MFMA groups on the current ones, hand the registers over.
        // zf (std::true_type / false_type): the step STARTS an accumulation chain — its first group multiplies into a literal 0
        auto step = [&](auto zf, int ks) {
            prefetch(tile, ks);
#if TFGX_ROWS_EXPERIMENT == 3
            const uint64_t dbg_a = __builtin_readcyclecounter();
#endif
            rows_mfma_groups<TN, 16, decltype(zf)::value>(acc, [&](int i) { return cur[i >> 2][i & 3]; },
                                                         Bs + (ks * 32 + 16 * kh) * LDB_S + l31);
#if TFGX_ROWS_EXPERIMENT == 3
            const uint64_t dbg_b = __builtin_readcyclecounter();
            dbg_mf += dbg_b - dbg_a;
#endif
            rotate();
#if TFGX_ROWS_EXPERIMENT == 3
            asm volatile("" ::"v"(cur[0]), "v"(cur[1]), "v"(cur[2]), "v"(cur[3]));
            (ks == 0 ? dbg_h0 : dbg_h1) += __builtin_readcyclecounter() - dbg_b;     // hand-over after the first / the other steps
#endif
        };
        // tail step (K % 32 != 0), kept OUTSIDE the step loop (an if/else inside it makes the compiler carry a second
        // accumulator set): half_t groups in pairs, skipped wave-uniformly past the end
        auto tail_step = [&](auto zf) {
            prefetch(tile, nfull);
            const float* b_s = Bs + (nfull * 32 + half_t * kh) * LDB_S + l31;
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                if (2 * q < half_t) {
                    // When T/2 % 4 == 2 the kh = 1 half's last vector would start at K - 2; load_a clamped it to K - 4,
                    // so the two values it needs sit in elements 2, 3 instead of 0, 1.
                    auto c = [&](int i) { return cur[i >> 2][i & 3]; };
                    float a[2] = {c(2 * q), c(2 * q + 1)};
                    if (q % 2 == 0 && kh == 1 && 2 * q + 2 == half_t) {
                        a[0] = c(2 * q + 2);
                        a[1] = c(2 * q + 3);
                    }
                    if (q == 0) rows_mfma_groups<TN, 2, decltype(zf)::value>(acc, [&](int i) { return a[i]; }, b_s);
                    else rows_mfma_groups<TN, 2>(acc, [&](int i) { return a[i]; }, b_s + 2 * q * LDB_S);
                }
            }
            rotate();
        };
        const bool has_tail = nfull < nsteps;
        if (kTwoLevel && two_level) {
            // narrow outputs of a long K (K > 128, so nfull >= 4): chains of TWO k-steps (64 k), each started from a literal
            // zero and folded into `tot` with one v_add per register — no accumulator is ever zeroed with the vector ALU
            // (the first fold is a copy); an odd last step and / or the tail form the last chain
            step(std::true_type{}, 0);
            step(std::false_type{}, 1);
#pragma unroll
            for (int j = 0; j < TN; ++j) tot[j] = acc[j];
            int ks = 2;
            for (; ks + 1 < nfull; ks += 2) {
                step(std::true_type{}, ks);
                step(std::false_type{}, ks + 1);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int t = 0; t < 16; ++t) tot[j][t] += acc[j][t];
            }
            bool chain = false;                                     // uniform: does `acc` hold a chain not yet in `tot`?
            if (ks < nfull) {
                step(std::true_type{}, ks);
                chain = true;
                if (has_tail) tail_step(std::false_type{});
            } else if (has_tail) {
                tail_step(std::true_type{});
                chain = true;
            }
            if (chain) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc[j][t] += tot[j][t];
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = tot[j];
            }
        } else {
            step(std::true_type{}, 0);                              // K >= 32: there always is a first step
            for (int ks = 1; ks < nfull; ++ks) step(std::false_type{}, ks);
            if (has_tail) tail_step(std::false_type{});
        }
#if TFGX_ROWS_EXPERIMENT == 3
        const uint64_t dbg_e0 = __builtin_readcyclecounter();
#endif
        // epilogue (D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)); bias + activation in
        // place first, then stores straight from the accumulator registers (see gemm_kernel's epilogue note)
        if (bias != nullptr || act != TFGX_ACT_NONE) {           // uniform: a plain x @ W (the h = x W of a GCN layer) skips 32 TN VALU ops
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int a_j = j * 32 + l31 < act_cols ? act : TFGX_ACT_NONE;
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[j][t] = apply_act(acc[j][t] + bv[j], a_j);
            }
        }
        const int64_t r0 = tile * 32 + 4 * kh;
#if TFGX_ROWS_VEC_STORE
        if (vec_store && tile * 32 + 32 <= M) {
            // 16-byte stores: quad-transposed accumulators (see quad_transpose4).  Register group g = t >> 2 holds rows
            // 8 g + (t & 3) + 4 kh; after the transpose lane i of a quad owns row 8 g + i + 4 kh, columns 4 q .. 4 q + 3
            const int qc = (l31 >> 2) * 4;
            char* const cb = reinterpret_cast<char*>(C + tile * 32 * ldc);     // uniform; row 8 g is a scalar add, column block j an immediate
            const int64_t ldc8b = 32 * ldc;                                     // 8 rows in bytes
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float r4[4][4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {                  // every lane takes part in the quad permutes (no branch around them)
#pragma unroll
                    for (int k = 0; k < 4; ++k) r4[g][k] = acc[j][4 * g + k];
                    quad_transpose4(r4[g], lane);
                }
                if (j * 32 + qc < N) {                         // N % 4 == 0 on this path: the four columns are valid together
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(cb + g * ldc8b + c_off_vec + j * 128) = f32x4{r4[g][0], r4[g][1], r4[g][2], r4[g][3]};
                }
            }
        } else
#endif
        if (tile * 32 + 32 <= M) {
            // a FULL tile: 36 rows x ldc x 4 bytes from the tile's first row (the descriptor's extent; no store relies on it)
            const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(C + tile * 32 * ldc, 0, int(36 * ldc * 4), kRsrcFlags);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int gn = j * 32 + l31;
                if (gn < N) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const float val = acc[j][t];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val), c_rs, c_off + j * 128,
                                                              int(((t & 3) + 8 * (t >> 2)) * ldc * 4), 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int gn = j * 32 + l31;
                if (gn < N) {
                    float* cp = C + r0 * ldc + gn;
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int dr = (t & 3) + 8 * (t >> 2);
                        if (r0 + dr < M) cp[int64_t(dr) * ldc] = acc[j][t];
                    }
                }
            }
        }
#if !TFGX_ROWS_ZERO_FIRST
        zero_acc();
#endif
#if TFGX_ROWS_EXPERIMENT == 3
        dbg_ep += __builtin_readcyclecounter() - dbg_e0;
#endif
        tile = next_of(tile);                  // dynamic: the claim issued a tile ago
        if (tile_counter) claim();             // ... and the one for the tile after the next
    }
#if TFGX_ROWS_EXPERIMENT == 3
    if (lane == 0 && dbg_tiles) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&g_rows_dbg[0]), (unsigned long long)(__builtin_readcyclecounter() - dbg_c0));
        atomicAdd(reinterpret_cast<unsigned long long*>(&g_rows_dbg[1]), (unsigned long long)(wall_clock64() - dbg_w0));
        atomicAdd(reinterpret_cast<unsigned long long*>(&g_rows_dbg[2]), (unsigned long long)dbg_tiles);
        atomicAdd(reinterpret_cast<unsigned long long*>(&g_rows_dbg[12]), (unsigned long long)dbg_mf);
        atomicAdd(reinterpret_cast<unsigned long long*>(&g_rows_dbg[13]), (unsigned long long)dbg_ep);
        atomicAdd(reinterpret_cast<unsigned long long*>(&g_rows_dbg[14]), (unsigned long long)dbg_h0);
        atomicAdd(reinterpret_cast<unsigned long long*>(&g_rows_dbg[15]), (unsigned long long)dbg_h1);
        atomicAdd(reinterpret_cast<unsigned long long*>(&g_rows_dbg[3]), 1ull);
        const unsigned long long w1 = wall_clock64();
        atomicMax(reinterpret_cast<unsigned long long*>(&g_rows_dbg[4]), (unsigned long long)(w1 - dbg_w0));
        atomicMin(reinterpret_cast<unsigned long long*>(&g_rows_dbg[5]), (unsigned long long)(w1 - dbg_w0));
        atomicAdd(reinterpret_cast<unsigned long long*>(&g_rows_dbg[6]), (unsigned long long)(dbg_w0 - dbg_wk));
        atomicMax(reinterpret_cast<unsigned long long*>(&g_rows_dbg[7]), (unsigned long long)(w1 - dbg_wk));
        atomicMin(reinterpret_cast<unsigned long long*>(&g_rows_dbg[8]), (unsigned long long)dbg_wk);      // earliest / latest kernel entry
        atomicMax(reinterpret_cast<unsigned long long*>(&g_rows_dbg[9]), (unsigned long long)dbg_wk);
        {
            const int gw = int(blockIdx.x) * (NT / 64) + wave;
            if (gw < 4096) {
                g_rows_wave[4 * gw + 0] = dbg_w0;
                g_rows_wave[4 * gw + 1] = w1;
                g_rows_wave[4 * gw + 2] = dbg_tiles;
                g_rows_wave[4 * gw + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
            }
        }
        atomicMin(reinterpret_cast<unsigned long long*>(&g_rows_dbg[10]), w1);                              // earliest / latest loop end
        atomicMax(reinterpret_cast<unsigned long long*>(&g_rows_dbg[11]), w1);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// dW[Ka, N] = X[M, Ka]^T @ G[M, N]  (+ db[N] = column sums of G): the weight gradient of `x @ kernel + bias`
// (what tf.GradientTape produces for the dense layers next to the aggregation, demo/demo_gcn.py:68-77).
// A reduction over M = nodes (10^5..10^8) into a small [Ka, N] output: every workgroup owns a strided set of R-row
// slabs, stages a slab of X and G in LDS with coalesced float4 loads, and its 8 waves accumulate TPW 32x32 output
// tiles each with v_mfma_f32_32x32x2_f32 — the MFMA's k index IS the row m, so lane (l31, kh) feeds
// A[i = l31][k = kh] = Xs[2s + kh][i0 + l31] and B[k = kh][n = l31] = Gs[2s + kh][n0 + l31]: consecutive floats of an
// LDS row, conflict-free, no transposition anywhere.  Per-workgroup partial outputs go to the caller's workspace and
// are summed in workgroup order by tn_reduce_kernel (deterministic; no atomics).  A virtual all-ones column Ka of X
// (a padding column of the LDS slab) makes row Ka of the product the bias gradient for free.
constexpr int kTnThreads = 512;

template <int TPW>
__global__ __launch_bounds__(kTnThreads) void gemm_tn_kernel(const float* __restrict__ X, int64_t ldx,
                                                             const float* __restrict__ G, int64_t ldg, int64_t M, int Ka,
                                                             int N, int n_first, int ng_cols, int R, int ka_pad,
                                                             int ng_pad, int want_bias, float* __restrict__ parts,
                                                             int64_t part_stride, int x_vec4, int g_vec4)
{
    extern __shared__ float lds[];
    float* Xs = lds;                              // [R][ka_pad]
    float* Gs = lds + size_t(R) * ka_pad;         // [R][ng_pad]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int Ti = ka_pad / 32, Tn = ng_pad / 32, T = Ti * Tn;
    int ti[TPW], tn[TPW];
    bool tv[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = wave * TPW + j;
        tv[j] = t < T;
        ti[j] = tv[j] ? t / Tn : 0;
        tn[j] = tv[j] ? t % Tn : 0;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;

    // padding columns are written once: zeros, except the virtual ones column at Ka (bias gradient)
    for (int idx = tid; idx < R * (ka_pad - Ka); idx += kTnThreads) {
        const int r = idx / (ka_pad - Ka), c = Ka + idx % (ka_pad - Ka);
        Xs[r * ka_pad + c] = (want_bias && c == Ka) ? 1.0f : 0.0f;
    }
    for (int idx = tid; idx < R * (ng_pad - ng_cols); idx += kTnThreads) {
        const int r = idx / (ng_pad - ng_cols), c = ng_cols + idx % (ng_pad - ng_cols);
        Gs[r * ng_pad + c] = 0.0f;
    }
    const int64_t n_slabs = (M + R - 1) / R;
    for (int64_t slab = blockIdx.x; slab < n_slabs; slab += gridDim.x) {
        const int64_t m0 = slab * R;
        __syncthreads();                           // previous slab fully consumed
        if (x_vec4) {
            const int per = Ka / 4;
            for (int idx = tid; idx < R * per; idx += kTnThreads) {
                const int r = idx / per, c = (idx % per) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m0 + r < M) v = *reinterpret_cast<const float4*>(X + (m0 + r) * ldx + c);
                *reinterpret_cast<float4*>(Xs + r * ka_pad + c) = v;
            }
        } else {
            for (int idx = tid; idx < R * Ka; idx += kTnThreads) {
                const int r = idx / Ka, c = idx % Ka;
                Xs[r * ka_pad + c] = (m0 + r < M) ? X[(m0 + r) * ldx + c] : 0.0f;
            }
        }
        if (want_bias && m0 + R > M) {             // rows past M must not count in the ones column
            for (int r = tid; r < R; r += kTnThreads) Xs[r * ka_pad + Ka] = (m0 + r < M) ? 1.0f : 0.0f;
        }
        if (g_vec4) {
            const int per = ng_cols / 4;
            for (int idx = tid; idx < R * per; idx += kTnThreads) {
                const int r = idx / per, c = (idx % per) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m0 + r < M) v = *reinterpret_cast<const float4*>(G + (m0 + r) * ldg + n_first + c);
                *reinterpret_cast<float4*>(Gs + r * ng_pad + c) = v;
            }
        } else {
            for (int idx = tid; idx < R * ng_cols; idx += kTnThreads) {
                const int r = idx / ng_cols, c = idx % ng_cols;
                Gs[r * ng_pad + c] = (m0 + r < M) ? G[(m0 + r) * ldg + n_first + c] : 0.0f;
            }
        }
        __syncthreads();
        // TFGX_TN_BATCH steps at a time
        // is not waiting on one ds_read round trip per instruction
#ifndef TFGX_TN_BATCH
#define TFGX_TN_BATCH 1
#endif
        for (int st0 = 0; st0 < R / 2; st0 += TFGX_TN_BATCH) {
            float av[TFGX_TN_BATCH][TPW], bv[TFGX_TN_BATCH][TPW];
#pragma unroll
            for (int u = 0; u < TFGX_TN_BATCH; ++u) {
                const int st = st0 + u < R / 2 ? st0 + u : R / 2 - 1;
                const float* xr = Xs + (2 * st + kh) * ka_pad + l31;
                const float* gr = Gs + (2 * st + kh) * ng_pad + l31;
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    av[u][j] = xr[ti[j] * 32];
                    bv[u][j] = gr[tn[j] * 32];
                }
            }
#pragma unroll
            for (int u = 0; u < TFGX_TN_BATCH; ++u) {
                if (st0 + u < R / 2) {
#pragma unroll
                    for (int j = 0; j < TPW; ++j)
                        if (tv[j]) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][j], bv[u][j], acc[j], 0, 0, 0);
                }
            }
        }
    }
    // partial [ka_pad rows used: Ka (+1 bias row)][ng_cols] of this workgroup; D layout: col = lane & 31,
    // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* out = parts + int64_t(blockIdx.x) * part_stride;
    const int rows_out = Ka + (want_bias ? 1 : 0);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        if (!tv[j]) continue;
        const int n = tn[j] * 32 + l31;
        if (n >= ng_cols) continue;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = ti[j] * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
            if (i < rows_out) out[int64_t(i) * ng_cols + n] = acc[j][q];
        }
    }
}

// ---- the same product without LDS: with k = the node row, the MFMA operand layouts ARE row-major reads — for
// v_mfma_f32_32x32x2_f32 lane (l % 32, l / 32) of the A operand wants X[r + l/32][i0 + l%32] (32 consecutive floats of a
// row) and of the B operand G[r + l/32][n0 + l%32]; for v_mfma_f32_16x16x4_f32 it is X[r + l/16][i0 + l%16] — so every
// wave streams its operands straight from global memory, keeps TM x TN output tiles in registers and reuses each A
// value TN times and each B value TM times.  No barriers: the four waves of a workgroup (WM x WN arrangement) walk the
// same rows and share them through L1/L2.  Loads of the next U k-steps are in flight while the current U are
// multiplied.  S = 16 pads Ka (+1) and N to multiples of 16 instead of 32: 101 x 256 costs 112 x 256 multiplies
// instead of 128 x 256.
template <int S>
struct TnTile;
template <>
struct TnTile<32> {
    typedef f32x16 acc_t;
    static constexpr int NQ = 16, KS = 2;             // accumulator registers per tile; node rows per MFMA
    __device__ static __forceinline__ acc_t mfma(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ int out_row(int q, int kk) { return (q & 3) + 8 * (q >> 2) + 4 * kk; }
};
template <>
struct TnTile<16> {
    typedef f32x4 acc_t;
    static constexpr int NQ = 4, KS = 4;
    __device__ static __forceinline__ acc_t mfma(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ int out_row(int q, int kk) { return 4 * kk + q; }
};

template <int S, int TM, int TN, int U, bool GATED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_tn_direct_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ G, int64_t ldg, const float* __restrict__ gate,
    int64_t ldgate, int64_t M, int Ka, int N, int want_bias, int WM, int m_groups, int tm_r, int tn_r,
    int64_t rows_per_wg, float* __restrict__ parts, int64_t part_stride)
{
    // GATED: G[m, n] counts only where gate[m, n] > 0 — the backward of a ReLU fused into the producing layer's epilogue
    // (gate = that layer's output), so the masked gradient is never written out
    // tm_r <= TM, tn_r <= TN: the tiles per wave the configuration asked for (the instantiation may be larger)
    typedef TnTile<S> T;
    typedef typename T::acc_t acc_t;
    constexpr int KS = T::KS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lc = lane % S, kk = lane / S;
    const int WN = 4 / WM;
    const int wm = wave % WM, wn = wave / WM;
    const int gm = blockIdx.y % m_groups, gn = blockIdx.y / m_groups;
    const int Ti = (Ka + (want_bias ? 1 : 0) + S - 1) / S, Tn = (N + S - 1) / S;
    const int mt0 = (gm * WM + wm) * tm_r, nt0 = (gn * WN + wn) * tn_r;
    int mcol[TM], ncol[TN];
    bool mval[TM], nval[TN], mtile[TM], ntile[TN];
    float mfill[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        mcol[i] = (mt0 + i) * S + lc;
        mval[i] = mcol[i] < Ka;
        mfill[i] = (want_bias && mcol[i] == Ka) ? 1.0f : 0.0f;       // the virtual all-ones column: row Ka = bias gradient
        mtile[i] = i < tm_r && mt0 + i < Ti;
        if (!mval[i]) mcol[i] = 0;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        ncol[j] = (nt0 + j) * S + lc;
        nval[j] = ncol[j] < N;
        ntile[j] = j < tn_r && nt0 + j < Tn;
        if (!nval[j]) ncol[j] = 0;
    }
    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < T::NQ; ++q) acc[i][j][q] = 0.0f;

    const int64_t r_begin = int64_t(blockIdx.x) * rows_per_wg;
    const int64_t r_end = r_begin + rows_per_wg < M ? r_begin + rows_per_wg : M;
    struct Ops {
        float a[U][TM], b[U][TN];
    };
    constexpr int BR = KS * U;                    // node rows per batch
    // OPERAND LOADS (end of round 4).  Every operand used to be a predicated per-lane load (`valid ? p[col] : fill`) behind
    // its own 64-bit address arithmetic: a branch and ~3 vector-ALU instructions per load — and a vector-ALU instruction
    // waits ~38 cycles for an issue slot beside an MFMA stream (see gemm_rows_kernel).  Now: RAW BUFFER loads.  One
    // descriptor per operand matrix and batch (scalar: base = first row of the batch, extent = the rows of the batch that
    // exist), one per-lane byte offset per (u, tile) computed ONCE per launch — 0xffffffff for a column past the matrix, so
    // that lane reads 0 without touching memory, exactly like a row past r_end.  No predication, no per-load arithmetic, and
    // the ragged last batch runs the same code as the full ones.
    constexpr int kFlags = 0x00020000;
    uint32_t off_a[U][TM], off_b[U][TN];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int i = 0; i < TM; ++i) off_a[u][i] = (mval[i] && mtile[i]) ? uint32_t((int64_t(KS * u + kk) * ldx + mcol[i]) * 4) : 0xffffffffu;
#pragma unroll
        for (int j = 0; j < TN; ++j) off_b[u][j] = (nval[j] && ntile[j]) ? uint32_t((int64_t(KS * u + kk) * ldg + ncol[j]) * 4) : 0xffffffffu;
    }
    uint32_t off_g[GATED ? U : 1][GATED ? TN : 1];
    if constexpr (GATED) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                off_g[u][j] = (nval[j] && ntile[j]) ? uint32_t((int64_t(KS * u + kk) * ldgate + ncol[j]) * 4) : 0xffffffffu;
    }
    // the virtual all-ones column Ka of X (row Ka of the product = the bias gradient) lives in ONE of this wave's tiles, if any
    int ib = -1;
    if (want_bias) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
            if (mtile[i] && (mt0 + i) == Ka / S) ib = i;                 // uniform
    }
    const bool fill_lane = want_bias && lc == Ka % S;                    // the lane that owns column Ka inside tile ib
    auto load = [&](Ops& o, int64_t r) {          // rows r .. min(r + BR, r_end) - 1; the rest of the batch reads zeros
        const int64_t rows = r_end - r < BR ? r_end - r : BR;           // uniform, >= 1
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + r * ldx), 0, int(rows * ldx * 4), kFlags);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G + r * ldg), 0, int(rows * ldg * 4), kFlags);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < TM; ++i) o.a[u][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off_a[u][i], 0, 0));
#pragma unroll
            for (int j = 0; j < TN; ++j) o.b[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, off_b[u][j], 0, 0));
        }
        if constexpr (GATED) {
            const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gate + r * ldgate), 0, int(rows * ldgate * 4), kFlags);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, off_g[u][j], 0, 0));
                    o.b[u][j] = t > 0.0f ? o.b[u][j] : 0.0f;
                }
        }
        if (ib >= 0) {                                                   // uniform; one tile of one wave per row group
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    if (i == ib) o.a[u][i] = fill_lane ? ((KS * u + kk < rows) ? 1.0f : 0.0f) : o.a[u][i];   // (that lane's real column is past Ka: it read 0)
        }
    };
    auto mul = [&](const Ops& o) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (mtile[i] && ntile[j]) acc[i][j] = T::mfma(o.a[u][i], o.b[u][j], acc[i][j]);
    };
    if (r_begin < r_end) {
        Ops cur;
        load(cur, r_begin);
        for (int64_t r = r_begin; r + BR < r_end; r += BR) {
            Ops nxt;
            load(nxt, r + BR);
            mul(cur);
            cur = nxt;
        }
        mul(cur);
    }
    // partial of this workgroup: [rows_out][N]; D layout: col = lane % S, row = TnTile::out_row(register, lane / S)
    float* out = parts + int64_t(blockIdx.x) * part_stride;
    const int rows_out = Ka + (want_bias ? 1 : 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (!mtile[i]) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!ntile[j]) continue;
            const int n = (nt0 + j) * S + lc;
            if (n >= N) continue;
#pragma unroll
            for (int q = 0; q < T::NQ; ++q) {
                const int row = (mt0 + i) * S + T::out_row(q, kk);
                if (row < rows_out) out[int64_t(row) * N + n] = acc[i][j][q];
            }
        }
    }
}

struct TnDirectCfg {
    int S, WM, TM, TN, m_groups, n_groups, wgs;
    int64_t rows_per_wg;
    double cost;          // multiplies on the critical path (the configuration search's figure of merit)
};

inline int tn_env(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// tile size, wave arrangement and tiles per wave.  Cost = multiplies on the critical path (the busiest wave's valid
// tiles x groups x tile area), then operand loads per MFMA.  Instantiated wave tiles: S = 32: TM <= 4, TN <= 2;
// S = 16: TM <= 8, TN <= 4 (128 accumulator registers either way).
inline TnDirectCfg tn_direct_config(int64_t M, int64_t Ka, int64_t N, bool want_bias)
{
    static const int s_env = tn_env("TFGX_TN_S", 0), wm_env = tn_env("TFGX_TN_WM", 0), w_env = tn_env("TFGX_TN_WGS_ENV", 0);
    TnDirectCfg best{};
    double best_cost = 1e30, best_loads = 1e30;
    for (int S = 16; S <= 32; S *= 2) {
        if (s_env > 0 && S != s_env) continue;
        const int Ti = int((Ka + (want_bias ? 1 : 0) + S - 1) / S), Tn = int((N + S - 1) / S);
        const int tm_max = S == 32 ? 4 : 8, tn_max = S == 32 ? 2 : 4;
        for (int WM = 1; WM <= 4; WM *= 2) {
            if (wm_env > 0 && WM != wm_env) continue;
            const int WN = 4 / WM;
            int TM = (Ti + WM - 1) / WM, TN = (Tn + WN - 1) / WN;
            int mg = 1, ng = 1;
            if (TM > tm_max) { mg = (TM + tm_max - 1) / tm_max; TM = (TM + mg - 1) / mg; }
            if (TN > tn_max) { ng = (TN + tn_max - 1) / tn_max; TN = (TN + ng - 1) / ng; }
            const double cost = double(mg) * ng * TM * TN * S * S, loads = double(TM + TN) / (TM * TN) / (S / 16);
            if (cost < best_cost * 0.999 || (cost < best_cost * 1.001 && loads < best_loads)) {
                best_cost = cost; best_loads = loads;
                best.S = S; best.WM = WM; best.TM = TM; best.TN = TN; best.m_groups = mg; best.n_groups = ng;
            }
        }
    }
    best.cost = best_cost;
    const int groups = best.m_groups * best.n_groups;
    int wgs = (w_env > 0 ? w_env : 512) / groups;
    if (wgs < 32) wgs = 32;
    int64_t rows = (M + wgs - 1) / wgs;
    rows = (rows + 7) / 8 * 8;                    // whole batches (8 node rows for both tile sizes)
    if (rows < 8) rows = 8;
    best.rows_per_wg = rows;
    best.wgs = int((M + rows - 1) / rows);
    if (best.wgs < 1) best.wgs = 1;
    return best;
}

// The bias gradient rides in the reduction as a virtual all-ones column Ka of X (row Ka of the product).  When Ka is a multiple of
// the tile size that one row opens a whole new row of tiles — Ka = 512, N = 128: 33 tile rows instead of 32, a second group of
// waves on grid.y that re-reads G and multiplies 15/16 empty tiles: 4.16 ms against 2.68 without the bias (products shape,
// tools/r06/tn_strides.py).  Then the column sums are taken by the two-phase column-sum kernel instead (one more read of G: 0.25 ms).
inline bool tn_bias_by_column_sum(int64_t M, int64_t Ka, int64_t N, bool want_bias, bool gated)
{
    static const int on = tn_env("TFGX_TN_SPLIT_BIAS", 1);      // developer A/B
    if (!want_bias || gated || !on) return false;
    if (M * N < (int64_t(1) << 26)) return false;               // small gradients: two more launches cost what the tile row does
    // the ones column opens a new row of 16-row tiles exactly when Ka is a multiple of 16 (the search's cost figure — multiplies
    // on the critical path — rates that row at +9 %; measured it is +55 %: the extra group of waves also re-reads G)
    return Ka % 16 == 0;
}

inline bool tn_use_direct()
{
    static const int v = tn_env("TFGX_TN_DIRECT", 1);      // 0: the LDS-staged kernel above (kept for the A/B, Ka <= 2016)
    return v != 0;
}

// dW[i, n_first + n] = sum over workgroups of parts[b][i][n]; row Ka -> db.  64 consecutive output elements per workgroup
// (one coalesced 256-byte read per part), the parts cut into kTnSlices contiguous ranges summed by different waves with 8
// loads in flight each, the slice sums folded in slice order through LDS: a fixed order (bit-reproducible), and enough
// parallelism that the 512 x (Ka + 1) x N floats stream at HBM/MALL rate (one thread per output element walking all
// 512 parts — the first version — took 118 us for 129 x 256 outputs, as long as the arxiv-shape reduction itself).
constexpr int kTnSlices = 8;
__global__ __launch_bounds__(64 * kTnSlices) void tn_reduce_kernel(const float* __restrict__ parts, int n_parts,
                                 int64_t part_stride, int Ka, int ng_cols,
                                 int n_first, int want_bias, float* __restrict__ dW, int64_t ldw, float* __restrict__ db)
{
    __shared__ float part_sum[kTnSlices][64];
    const int rows = Ka + (want_bias ? 1 : 0);
    const int64_t total = int64_t(rows) * ng_cols;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int per = (n_parts + kTnSlices - 1) / kTnSlices;
    const int b0 = min(slice * per, n_parts), b1 = min(b0 + per, n_parts);
    for (int64_t base = int64_t(blockIdx.x) * 64; base < total; base += int64_t(gridDim.x) * 64) {
        const int64_t t = base + lane;
        const int64_t tc = t < total ? t : total - 1;          // clamped: loads stay unconditional
        const float* p = parts + tc;
        float acc = 0.0f;
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[int64_t(b + u) * part_stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; b < b1; ++b) acc += p[int64_t(b) * part_stride];
        part_sum[slice][lane] = acc;
        __syncthreads();
        if (slice == 0 && t < total) {
            float sum = part_sum[0][lane];
#pragma unroll
            for (int q = 1; q < kTnSlices; ++q) sum += part_sum[q][lane];
            const int i = int(t / ng_cols), n = int(t % ng_cols);
            if (i < Ka) dW[int64_t(i) * ldw + n_first + n] = sum;
            else db[n_first + n] = sum;
        }
        __syncthreads();
    }
}

inline unsigned tn_reduce_grid(int64_t total)
{
    const int64_t g = (total + 63) / 64;
    return unsigned(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

#ifndef TFGX_TN_WGS
#define TFGX_TN_WGS 512
#endif
struct TnCfg {
    int ka_pad, tn_group, groups, R, wgs;
    size_t part_floats, lds_bytes;
};

inline TnCfg tn_config(int64_t M, int64_t Ka, int64_t N, bool want_bias)
{
    TnCfg c;
    c.ka_pad = int((Ka + (want_bias ? 1 : 0) + 31) / 32) * 32;
    const int Ti = c.ka_pad / 32, Tn = int((N + 31) / 32);
    c.tn_group = 64 / Ti < 1 ? 1 : (64 / Ti > Tn ? Tn : 64 / Ti);      // <= 64 output tiles per pass (8 waves x 8)
    c.groups = (Tn + c.tn_group - 1) / c.tn_group;
    c.R = 32;
    static int r_env = -1, w_env = -1;      // developer A/B: TFGX_TN_R (rows per slab), TFGX_TN_WGS_ENV (workgroups)
    if (r_env < 0) {
        const char* e = getenv("TFGX_TN_R");
        r_env = e ? atoi(e) : 0;
        const char* w = getenv("TFGX_TN_WGS_ENV");
        w_env = w ? atoi(w) : 0;
    }
    if (r_env >= 4) c.R = r_env;
    while (c.R > 4 && sizeof(float) * size_t(c.R) * size_t(c.ka_pad + c.tn_group * 32) > 72 * 1024) c.R /= 2;
    const int64_t slabs = (M + c.R - 1) / c.R;
    const int64_t wcap = w_env > 0 ? w_env : TFGX_TN_WGS;
    c.wgs = int(slabs < wcap ? (slabs < 1 ? 1 : slabs) : wcap);
    c.part_floats = size_t(Ka + (want_bias ? 1 : 0)) * size_t(c.tn_group) * 32;
    c.lds_bytes = sizeof(float) * size_t(c.R) * size_t(c.ka_pad + c.tn_group * 32);
    return c;
}

// developer A/B switch: TFGX_GEMM_TWO_LEVEL=0 restores the single k-ordered chain on the narrow tiles
static int two_level_default()
{
    static const int on = [] {
        const char* e = std::getenv("TFGX_GEMM_TWO_LEVEL");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return on;
}

// Dynamic tile order of the row kernel (see gemm_rows_kernel): its pool counters live in the CALLER's workspace
// (tfgx_gemm_bias_act_cols_ws_f32) — a buffer that belongs to one call in flight by construction — and are zeroed in stream
// order before the launch.  Without a workspace (or for launches too short to gain: a memset node costs ~2 us) the kernel
// walks its fixed tile -> wave map.  TFGX_ROWS_DYNAMIC=0 turns it off (developer A/B), =2 removes the size threshold (tests).
constexpr size_t kRowsCounterBytes = sizeof(unsigned int) * kRowsCounterStride * kRowsCounterSlots;
inline int rows_dynamic_mode()
{
    static const int mode = [] { const char* e = std::getenv("TFGX_ROWS_DYNAMIC"); return e ? atoi(e) : 1; }();
    return mode;
}
inline bool rows_dynamic_wanted(int64_t M)
{
    const int mode = rows_dynamic_mode();
    return mode == 2 || (mode == 1 && M >= (int64_t(1) << 18));
}
inline unsigned int* rows_tile_counter(int64_t M, void* workspace, size_t workspace_bytes)
{
    if (!rows_dynamic_wanted(M) || workspace == nullptr) return nullptr;
    const uintptr_t p = (reinterpret_cast<uintptr_t>(workspace) + 127) & ~uintptr_t(127);
    if (p + kRowsCounterBytes > reinterpret_cast<uintptr_t>(workspace) + workspace_bytes) return nullptr;
    return reinterpret_cast<unsigned int*>(p);
}

inline size_t rows_lds_bytes(int64_t K, int tn) { return sizeof(float) * size_t(K) * size_t(tn * 32 + 8); }

template <int TN>
int launch_gemm_rows(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, int act, float* C,
                     int64_t ldc, int64_t M, int K, int N, int act_cols, hipStream_t stream, unsigned int* tile_counter)
{
    // per DEVICE (one process may drive several GPUs): compute-unit count + the kernel's dynamic-LDS attribute
    constexpr int kMaxDev = 64;
    static int cus_of[kMaxDev] = {0};
    int dev = 0;
    TFGX_HIP_CHECK(hipGetDevice(&dev));
    TFGX_REQUIRE(dev >= 0 && dev < kMaxDev, "device ordinal out of range");
    if (cus_of[dev] == 0) {
        hipDeviceProp_t prop;
        TFGX_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        TFGX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows_kernel<TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        cus_of[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int cus = cus_of[dev];
    const int64_t n_tiles = (M + 31) / 32;
    constexpr int kWaves = rows_threads<TN>() / 64;
    const int64_t wgs = (n_tiles + kWaves - 1) / kWaves;
    const int b_vec4 = (ldb % 4 == 0) && aligned_to(B, 16);
    // one persistent workgroup per CU.  (Rounds 2-3 ran the narrow outputs, TN <= 2, with up to three per CU; the round-4
    // same-box sweep — TFGX_ROWS_WGS_MULT, profiles/r04_gemm_sweep.jsonl — has one per CU equal or faster on every narrow
    // shape: 2.4 M x 100 -> 16: 0.298 -> 0.265 ms, 170 k x 256 -> 40: 0.071 -> 0.063 (every workgroup stages B in LDS for
    // only a handful of tiles there), 2.4 M x 100 -> 64 and 2.4 M x 256 -> 40 unchanged)
    static const int mult_env = [] {          // developer A/B: TFGX_ROWS_WGS_MULT = workgroups per CU of the row kernel
        const char* e = std::getenv("TFGX_ROWS_WGS_MULT");
        return (e && atoi(e) > 0) ? atoi(e) : 0;
    }();
    const int64_t max_wgs = int64_t(cus) * (mult_env > 0 ? mult_env : 1);
    dim3 grid(static_cast<unsigned>(wgs < max_wgs ? wgs : max_wgs), 1, 1), block(rows_threads<TN>(), 1, 1);
    // claimed tiles: pool p (16 of them: ((blockIdx >> 3) & 3) * 4 + (wave & 3)) owns the tiles = p (mod 16) and is drained only
    // by waves whose own pool id is p, so every pool needs a claimant: at least 4 waves per workgroup and workgroups
    // 0, 8, 16, 24 present.  Smaller launches (few CUs, few tiles) take the fixed tile map instead — a pool without a
    // claimant would leave its tiles of C unwritten.
    static_assert(rows_threads<TN>() / 64 >= 4, "the claimed tile order needs all four (wave & 3) pool slots per workgroup");
    if (grid.x < 32) tile_counter = nullptr;
    if (tile_counter)
        TFGX_HIP_CHECK(hipMemsetAsync(tile_counter, 0, kRowsCounterBytes, stream));
    gemm_rows_kernel<TN><<<grid, block, rows_lds_bytes(K, TN), stream>>>(A, lda, B, ldb, bias, act, act_cols, C, ldc, M, K,
                                                                         N, n_tiles, b_vec4, two_level_default(), tile_counter);
    TFGX_LAUNCH_CHECK("gemm_rows_kernel");
    return TFGX_OK;
}

// the row-streaming kernel needs 16-byte aligned A rows with K % 4 == 0, 64 < N <= 256, all of B within 160 KB of LDS,
// and enough 32-row tiles to keep 8 waves on every CU busy for many tiles
inline bool rows_ok(const float* A, int64_t lda, int64_t M, int64_t K, int64_t N)
{
    if (N < 1 || N > 256 || K < 32 || K % 4 != 0 || lda % 4 != 0 || !aligned_to(A, 16) || M < 128 * 256) return false;
    if (lda >= (int64_t(1) << 25)) return false;          // per-lane byte offsets of the kernel are 32-bit (32 rows x lda x 4)
    return rows_lds_bytes(K, int((N + 31) / 32)) <= 160 * 1024;
}

// ---------------------------------------------------------------------------------------------------------
// Narrow outputs (N <= 48) of a K that need not be aligned (round 5; Cora-width first layers 1433 -> 16, GAT's Q / K
// projections 602 -> 8 / 16, hidden -> classes 256 -> 40): the product is a stream over A — 993 MB at 173 k x 1433 against
// 11 MB of output — that the LDS-staged kernel ran at 3.6 TB/s (two barriers per 16 k, half of every 32-column MFMA tile
// multiplying padding).  Here a persistent 1024-thread workgroup per CU keeps ALL of B in LDS (K x 16 NT floats, row pitch
// 16 NT + 4: the four k-groups of a wave land on disjoint banks) and every wave streams its own 16 rows of A straight into the
// operand layout of v_mfma_f32_16x16x4_f32 — lane (m = lane & 15, kq = lane >> 4) loads the 16 bytes A[m][k0 + 4 kq .. + 3]
// (4-byte alignment is enough for global_load_dwordx4, so K = 1433 needs no special path) and the i-th MFMA group of the step
// multiplies k = k0 + 4 kq + i against LDS row k of B: any pairing of k's is a valid product as long as A and B agree.  Eight
// steps (8 x 64 contiguous bytes per row) are in flight per lane before the first MFMA; no barrier after the B load.
// Arithmetic: one fp32 chain per output element in that k order, flushed into a second accumulator every 64 k (the two-level
// sum of the narrow tiles above: error ~ eps * sqrt(64 n) instead of eps * n).
constexpr int kSkinnyThreads = 1024;
inline int skinny_tiles(int64_t N) { return int((N + 15) / 16); }
inline size_t skinny_lds_bytes(int64_t K, int nt) { return sizeof(float) * size_t((K + 15) / 16 * 16) * size_t(16 * nt + 4); }
inline int skinny_mode()      // TFGX_GEMM_SKINNY: 0 = off, 1 = where the row-streaming kernel cannot go, 2 = every N <= 48 (developer A/B)
{
    static const int v = [] { const char* e = std::getenv("TFGX_GEMM_SKINNY"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; }();
    return v;
}
inline bool skinny_ok(const float* A, int64_t lda, int64_t M, int64_t K, int64_t N)
{
    return skinny_mode() != 0 && N <= 48 && K >= 64 && M >= 32768 && aligned_to(A, 4) && lda < (int64_t(1) << 28) &&
           skinny_lds_bytes(K, skinny_tiles(N)) <= 150 * 1024;
}

template <int NT>
__global__ __launch_bounds__(kSkinnyThreads) void gemm_skinny_kernel(const float* __restrict__ A, int64_t lda,
                                                                     const float* __restrict__ B, int64_t ldb,
                                                                     const float* __restrict__ bias, int act, int act_cols,
                                                                     float* __restrict__ C, int64_t ldc, int64_t M, int K, int N,
                                                                     int64_t n_tiles, int two_level)
{
    extern __shared__ __attribute__((aligned(16))) float Bsk[];      // [Kp][P], rows >= K and columns >= N are zero
    constexpr int P = 16 * NT + 4;
    const int Kp = (K + 15) / 16 * 16;
    for (int idx = threadIdx.x; idx < Kp * 16 * NT; idx += kSkinnyThreads) {
        const int k = idx / (16 * NT), n = idx % (16 * NT);
        Bsk[k * P + n] = (k < K && n < N) ? B[int64_t(k) * ldb + n] : 0.0f;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 15, kq = lane >> 4;
    constexpr int WAVES = kSkinnyThreads / 64;
    const int full = K / 16;                  // steps whose 16 k are all inside the row
    const float* bp0 = Bsk + 4 * kq * P + m;  // B[k0 + 4 kq + i][16 t + (lane & 15)] = bp0[(k0 + i) * P + 16 t]
    constexpr int U = 8;
    for (int64_t tile = int64_t(blockIdx.x) * WAVES + wave; tile < n_tiles; tile += int64_t(gridDim.x) * WAVES) {
        const int64_t row = tile * 16 + m;
        const float* ap = A + (row < M ? row : M - 1) * lda + 4 * kq;      // rows past M re-read the last row (never stored)
        f32x4 acc[NT], tot[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; tot[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        int s = 0;
        for (; s + U <= full; s += U) {
            f32x4_a4 a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = *reinterpret_cast<const f32x4_a4*>(ap + 16 * (s + u));
            __builtin_amdgcn_sched_barrier(0);     // all U loads first: left alone, the scheduler sinks each next to its MFMAs (two in flight)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float* bp = bp0 + 16 * (s + u) * P;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], bp[i * P + 16 * t], acc[t], 0, 0, 0);
                if ((u & 3) == 3 && two_level) {       // every 64 k (wave-uniform)
#pragma unroll
                    for (int t = 0; t < NT; ++t) { tot[t] += acc[t]; acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                }
            }
        }
        for (; s < full; ++s) {
            const f32x4_a4 a1 = *reinterpret_cast<const f32x4_a4*>(ap + 16 * s);
            const float* bp = bp0 + 16 * s * P;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], bp[i * P + 16 * t], acc[t], 0, 0, 0);
        }
        if (full * 16 < K) {                   // the last, partial step: elements past the row's end are read as zero
            const int k0 = full * 16 + 4 * kq;
            const float* bp = bp0 + 16 * full * P;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float av = (k0 + i < K) ? ap[16 * full + i] : 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[i * P + 16 * t], acc[t], 0, 0, 0);
            }
        }
        // D layout: column 16 t + (lane & 15), rows 4 * (lane >> 4) + i
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (two_level) acc[t] += tot[t];
            const int cn = 16 * t + m;
            if (cn < N) {
                const float bv = bias ? bias[cn] : 0.0f;
                const int a_j = cn < act_cols ? act : TFGX_ACT_NONE;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t r = tile * 16 + 4 * kq + i;
                    if (r < M) C[r * ldc + cn] = apply_act(acc[t][i] + bv, a_j);
                }
            }
        }
    }
}

template <int NT>
int launch_gemm_skinny(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, int act, float* C,
                       int64_t ldc, int64_t M, int K, int N, int act_cols, hipStream_t stream)
{
    constexpr int kMaxDev = 64;
    static int cus_of[kMaxDev] = {0};
    int dev = 0;
    TFGX_HIP_CHECK(hipGetDevice(&dev));
    TFGX_REQUIRE(dev >= 0 && dev < kMaxDev, "device ordinal out of range");
    if (cus_of[dev] == 0) {
        hipDeviceProp_t prop;
        TFGX_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        TFGX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_kernel<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        cus_of[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int64_t n_tiles = (M + 15) / 16;
    const int64_t wgs = (n_tiles + (kSkinnyThreads / 64) - 1) / (kSkinnyThreads / 64);
    dim3 grid(static_cast<unsigned>(wgs < cus_of[dev] ? wgs : cus_of[dev]), 1, 1), block(kSkinnyThreads, 1, 1);
    gemm_skinny_kernel<NT><<<grid, block, skinny_lds_bytes(K, NT), stream>>>(A, lda, B, ldb, bias, act, act_cols, C, ldc, M, K, N,
                                                                            n_tiles, two_level_default());
    TFGX_LAUNCH_CHECK("gemm_skinny_kernel");
    return TFGX_OK;
}

// sum of the split-K partials + bias + activation, in split order (deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ parts, int splits, int64_t M, int N, const float* __restrict__ bias,
                                     int act, int act_cols, float* __restrict__ C, int64_t ldc)
{
    int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const int64_t total = M * N;
    for (; t < total; t += stride) {
        const int64_t m = t / N;
        const int n = int(t - m * N);
        float acc = 0.0f;
        for (int s2 = 0; s2 < splits; ++s2) acc += parts[int64_t(s2) * total + t];
        if (bias) acc += bias[n];
        C[m * ldc + n] = apply_act(acc, n < act_cols ? act : TFGX_ACT_NONE);
    }
}

// how many K slices a generic-kernel launch with `tiles` output tiles should be cut into (1 = no split)
inline int splitk_factor(int64_t tiles, int64_t K)
{
    if (tiles >= 128 || K < 512) return 1;
    int64_t s = 256 / (tiles > 0 ? tiles : 1);
    if (s > K / 128) s = K / 128;
    if (s > 16) s = 16;
    return s < 2 ? 1 : int(s);
}

template <int BM, int BN, int WM, int WN>
int launch_gemm(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, int act, float* C,
                int64_t ldc, int64_t M, int K, int N, int act_cols, hipStream_t stream, float* split_ws = nullptr,
                int splits = 1)
{
    const int ntn = (N + BN - 1) / BN;
    const int64_t ntm = (M + BM - 1) / BM;
    const int64_t blocks = ntm * ntn;
    if (blocks >= (int64_t(1) << 31)) {
        set_error("tfgx_gemm_bias_act_f32: too many tiles");
        return TFGX_ERR_INVALID_ARG;
    }
    // developer A/B: TFGX_GEMM_UNALIGNED_V4=0 restores the dword loads for rows that are not 16-byte aligned
    static const bool unaligned_v4 = [] { const char* e = std::getenv("TFGX_GEMM_UNALIGNED_V4"); return !(e && e[0] == '0'); }();
    // measured (same box, TFGX_GEMM_UNALIGNED_V4 = 0 / 1): 173312 x 1433 -> 16: 0.379 -> 0.281 ms; 170000 x 1433 -> 256: 1.235 ->
    // 1.155; 233000 x 602 -> 64: 0.238 -> 0.214; 100000 x 301 -> 40: 0.073 -> 0.059; but 233000 x 602 -> 16 / 8: 0.188 -> 0.206 /
    // 0.176 -> 0.205 — narrow outputs of a medium K stay on the dword loads
    const bool av4 = ((lda % 4 == 0) && aligned_to(A, 16)) || (unaligned_v4 && aligned_to(A, 4) && (N > 32 || K >= 1024));
    const bool bv4 = (ldb % 4 == 0) && aligned_to(B, 16);
    const bool split = split_ws != nullptr && splits > 1;
    const int k_chunk = split ? int((((K + splits - 1) / splits) + 15) / 16 * 16) : K;
    const int ny = split ? (K + k_chunk - 1) / k_chunk : 1;
    float* out = split ? split_ws : C;
    const int64_t ldo = split ? N : ldc;
    const float* kb = split ? nullptr : bias;
    const int ka = split ? TFGX_ACT_NONE : act;
    dim3 grid(static_cast<unsigned>(blocks), static_cast<unsigned>(ny), 1), block(kBlock, 1, 1);
    // developer A/B: TFGX_GEMM_LDS_AHEAD=0 keeps the per-step operand reads at every K
    static const bool lds_ahead = [] { const char* e = std::getenv("TFGX_GEMM_LDS_AHEAD"); return !(e && e[0] == '0'); }();
    // wide wave tiles only (4 MFMAs per step): on the 32 x 64 / 64 x 32 wave tiles (2 MFMAs per 3 operand reads) the pinned
    // order LOST 8 % (233 k x 602 -> 64: 0.225 -> 0.244 ms, 170 k x 1433 -> 64: 0.385 -> 0.402)
    constexpr bool kWideWaveTile = (WM / 32) * (WN / 32) >= 4;
    const bool ahead = kWideWaveTile && lds_ahead && k_chunk >= 256;
#define TFGX_GEMM_GO(AV, BV)                                                                                              \
    do {                                                                                                                  \
        if (ahead)                                                                                                        \
            gemm_kernel<BM, BN, WM, WN, AV, BV, kWideWaveTile><<<grid, block, 0, stream>>>(                               \
                A, lda, B, ldb, kb, ka, out, ldo, M, K, N, ntn, act_cols, k_chunk, M * int64_t(N), two_level_default());  \
        else                                                                                                              \
            gemm_kernel<BM, BN, WM, WN, AV, BV, false><<<grid, block, 0, stream>>>(                                       \
                A, lda, B, ldb, kb, ka, out, ldo, M, K, N, ntn, act_cols, k_chunk, M * int64_t(N), two_level_default());  \
    } while (0)
    if (av4 && bv4) TFGX_GEMM_GO(true, true);
    else if (av4) TFGX_GEMM_GO(true, false);
    else if (bv4) TFGX_GEMM_GO(false, true);
    else TFGX_GEMM_GO(false, false);
#undef TFGX_GEMM_GO
    TFGX_LAUNCH_CHECK("gemm_kernel");
    if (split) {
        splitk_reduce_kernel<<<grid_for(M * N, kBlock), kBlock, 0, stream>>>(split_ws, ny, M, N, bias, act, act_cols, C, ldc);
        TFGX_LAUNCH_CHECK("splitk_reduce_kernel");
    }
    return TFGX_OK;
}

}  // namespace
}  // namespace tfgx

using namespace tfgx;

static inline int64_t generic_tiles(int64_t M, int64_t N)
{
    const int64_t bm = N <= 32 ? 256 : 128, bn = N <= 32 ? 32 : (N <= 64 ? 64 : 128);
    return ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
}

#if TFGX_ROWS_EXPERIMENT == 3
// developer build only: read and clear the row kernel's in-kernel clocks (see gemm_rows_kernel)
extern "C" int tfgx_debug_rows_stats(uint64_t* out4)
{
    uint64_t z[16] = {0, 0, 0, 0, 0, ~0ull, 0, 0, ~0ull, 0, ~0ull, 0, 0, 0, 0, 0};
    TFGX_HIP_CHECK(hipDeviceSynchronize());
    TFGX_HIP_CHECK(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_rows_dbg), sizeof(z)));
    TFGX_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_rows_dbg), z, sizeof(z)));
    return TFGX_OK;
}
#endif

#if TFGX_ROWS_EXPERIMENT == 3
extern "C" int tfgx_debug_rows_waves(uint64_t* out, int n_waves)
{
    TFGX_HIP_CHECK(hipDeviceSynchronize());
    TFGX_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rows_wave), sizeof(uint64_t) * 4 * size_t(n_waves)));
    return TFGX_OK;
}
#endif

extern "C" size_t tfgx_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N)
{
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    const int s = splitk_factor(generic_tiles(M, N), K);
    if (s > 1) return sizeof(float) * size_t(s) * size_t(M) * size_t(N);
    // the row kernel's tile counters (alignment slack included); whether the row kernel runs also depends on the pointers,
    // so this is an upper bound for shapes it can take
    const bool rows_shape = N <= 512 && K >= 32 && K % 4 == 0 && M >= 128 * 256;
    return (rows_shape && rows_dynamic_wanted(M)) ? kRowsCounterBytes + 128 : 0;
}

extern "C" int tfgx_gemm_bias_act_cols_ws_f32(const float* A, int64_t lda, const float* B, int64_t ldb,
                                              const float* bias, int32_t act, int64_t act_cols, float* C, int64_t ldc,
                                              int64_t M, int64_t K, int64_t N, void* workspace, size_t workspace_bytes,
                                              tfgx_stream_t stream_);

extern "C" int tfgx_gemm_bias_act_cols_f32(const float* A, int64_t lda, const float* B, int64_t ldb,
                                           const float* bias, int32_t act, int64_t act_cols, float* C, int64_t ldc,
                                           int64_t M, int64_t K, int64_t N, tfgx_stream_t stream_)
{
    TFGX_RANGE();
    return tfgx_gemm_bias_act_cols_ws_f32(A, lda, B, ldb, bias, act, act_cols, C, ldc, M, K, N, nullptr, 0, stream_);
}

extern "C" int tfgx_gemm_bias_act_cols_ws_f32(const float* A, int64_t lda, const float* B, int64_t ldb,
                                              const float* bias, int32_t act, int64_t act_cols, float* C, int64_t ldc,
                                              int64_t M, int64_t K, int64_t N, void* workspace, size_t workspace_bytes,
                                              tfgx_stream_t stream_)
{
    TFGX_RANGE();
    TFGX_REQUIRE(M >= 0 && K >= 1 && N >= 1, "bad M / K / N");
    TFGX_REQUIRE(K < (int64_t(1) << 30) && N < (int64_t(1) << 30), "K / N too large");
    TFGX_REQUIRE(act == TFGX_ACT_NONE || act == TFGX_ACT_RELU, "bad act");
    TFGX_REQUIRE(act_cols >= 0 && act_cols <= N, "act_cols outside [0, N]");
    if (M == 0) return TFGX_OK;
    TFGX_REQUIRE(A && B && C, "null pointer");
    TFGX_REQUIRE(lda >= K && ldb >= N && ldc >= N, "leading dimension too small");
    hipStream_t stream = as_stream(stream_);
    const int ac = int(act_cols);
    // K * N too large for LDS but a column slice fits (hidden -> hidden, 256 -> 256): run the row-streaming kernel once
    // per slice of 128 columns.  A is streamed once per slice; at these widths the MFMA time still dominates.
    const bool ldc_ok = ldc < (int64_t(1) << 25);          // same for the stores (36 rows x ldc x 4)
    if (ldc_ok && N > 128 && N <= 512 && N % 128 == 0 && !rows_ok(A, lda, M, K, N) && rows_ok(A, lda, M, K, 128)) {
        for (int64_t n0 = 0; n0 < N; n0 += 128) {
            const int64_t ac_slice = act_cols > n0 ? (act_cols - n0 < 128 ? act_cols - n0 : 128) : 0;
            const int rc = tfgx_gemm_bias_act_cols_ws_f32(A, lda, B + n0, ldb, bias ? bias + n0 : nullptr, act, ac_slice,
                                                          C + n0, ldc, M, K, 128, workspace, workspace_bytes, stream_);
            if (rc != TFGX_OK) return rc;
        }
        return TFGX_OK;
    }
    if (skinny_ok(A, lda, M, K, N) && (skinny_mode() == 2 || !rows_ok(A, lda, M, K, N))) {
        switch (skinny_tiles(N)) {
            case 1: return launch_gemm_skinny<1>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream);
            case 2: return launch_gemm_skinny<2>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream);
            default: return launch_gemm_skinny<3>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream);
        }
    }
    if (ldc_ok && rows_ok(A, lda, M, K, N)) {
        unsigned int* ctr = rows_tile_counter(M, workspace, workspace_bytes);
#define TFGX_ROWS_CASE(T) \
    case T: return launch_gemm_rows<T>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream, ctr)
        switch ((N + 31) / 32) {
            TFGX_ROWS_CASE(1);
            TFGX_ROWS_CASE(2);
            TFGX_ROWS_CASE(3);
            TFGX_ROWS_CASE(4);
            TFGX_ROWS_CASE(5);
            TFGX_ROWS_CASE(6);
            TFGX_ROWS_CASE(7);
            default: TFGX_ROWS_CASE(8);
        }
#undef TFGX_ROWS_CASE
    }
    // N = q * 128 + r with a SHORT remainder (r <= 64: 160, 144, 288 ...): the 128-column tiles would multiply a last tile column
    // that is 50 % or more padding (N = 160: 3/8 of all MFMA work).  The first q * 128 columns and the remainder are two products
    // — A is streamed once more, the remainder on a 32- / 64-column tile: 233 k x 602 -> 160 0.66 -> 0.51 ms (hipBLASLt 0.48).
    // Long K only: on a short K the second pass over A costs what the padding did.
    if (N > 128 && N % 128 != 0 && N % 128 <= 64 && K >= 256 && M >= 4096) {
        const int64_t n_main = N - N % 128;
        int rc = tfgx_gemm_bias_act_cols_ws_f32(A, lda, B, ldb, bias, act, act_cols < n_main ? act_cols : n_main, C, ldc, M, K,
                                                n_main, workspace, workspace_bytes, stream_);
        if (rc != TFGX_OK) return rc;
        return tfgx_gemm_bias_act_cols_ws_f32(A, lda, B + n_main, ldb, bias ? bias + n_main : nullptr, act,
                                              act_cols > n_main ? act_cols - n_main : 0, C + n_main, ldc, M, K, N - n_main,
                                              workspace, workspace_bytes, stream_);
    }
    // small M with a long K leaves most CUs idle: split K over blockIdx.y when the caller lent a workspace
    int splits = splitk_factor(generic_tiles(M, N), K);
    float* ws = static_cast<float*>(workspace);
    if (ws == nullptr || workspace_bytes < sizeof(float) * size_t(splits) * size_t(M) * size_t(N)) splits = 1;
    if (N <= 32)
        return launch_gemm<256, 32, 64, 32>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream, ws, splits);
    if (N <= 64)
        return launch_gemm<128, 64, 32, 64>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream, ws, splits);
    return launch_gemm<128, 128, 64, 64>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream, ws, splits);
}

extern "C" int tfgx_gemm_bias_act_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                                      int32_t act, float* C, int64_t ldc, int64_t M, int64_t K, int64_t N,
                                      tfgx_stream_t stream)
{
    TFGX_RANGE();
    return tfgx_gemm_bias_act_cols_f32(A, lda, B, ldb, bias, act, N, C, ldc, M, K, N, stream);
}

extern "C" size_t tfgx_gemm_tn_workspace_bytes(int64_t M, int64_t Ka, int64_t N, int32_t want_bias)
{
    if (M <= 0 || Ka <= 0 || N <= 0) return 0;
    if (tn_use_direct()) {
        // (gated calls never split: they get the larger of the two figures)
        const TnDirectCfg d = tn_direct_config(M, Ka, N, want_bias != 0);
        size_t bytes = sizeof(float) * size_t(Ka + (want_bias ? 1 : 0)) * size_t(N) * size_t(d.wgs);
        if (tn_bias_by_column_sum(M, Ka, N, want_bias != 0, false)) {
            const TnDirectCfg d0 = tn_direct_config(M, Ka, N, false);
            const size_t split = (sizeof(float) * size_t(Ka) * size_t(N) * size_t(d0.wgs) + 255) / 256 * 256 +
                                 tfgx_column_sum_workspace_bytes(M, N);
            bytes = split > bytes ? split : bytes;
        }
        return bytes;
    }
    if (Ka > 2016) return 0;
    const TnCfg c = tn_config(M, Ka, N, want_bias != 0);
    return sizeof(float) * c.part_floats * size_t(c.wgs);
}

extern "C" int tfgx_gemm_tn_f32(const float* X, int64_t ldx, const float* G, int64_t ldg, int64_t M, int64_t Ka,
                                int64_t N, float* dW, int64_t ldw, float* db, void* workspace, size_t workspace_bytes,
                                tfgx_stream_t stream_)
{
    return tfgx_gemm_tn_gated_f32(X, ldx, G, ldg, nullptr, 0, M, Ka, N, dW, ldw, db, workspace, workspace_bytes, stream_);
}

extern "C" int tfgx_gemm_tn_gated_f32(const float* X, int64_t ldx, const float* G, int64_t ldg, const float* gate,
                                      int64_t ld_gate, int64_t M, int64_t Ka, int64_t N, float* dW, int64_t ldw, float* db,
                                      void* workspace, size_t workspace_bytes, tfgx_stream_t stream_)
{
    TFGX_RANGE();
    TFGX_REQUIRE(gate == nullptr || (ld_gate >= N && tn_use_direct()), "gate: leading dimension too small / needs the direct kernel");
    TFGX_REQUIRE(M >= 0 && Ka >= 1 && N >= 1, "bad M / Ka / N");
    TFGX_REQUIRE(Ka < (int64_t(1) << 30) && N < (int64_t(1) << 30), "Ka / N too large");
    TFGX_REQUIRE(tn_use_direct() || Ka <= 2016, "Ka > 2016 is not supported by the LDS-staged kernel (TFGX_TN_DIRECT=0)");
    TFGX_REQUIRE(dW != nullptr && ldw >= N, "bad dW");
    hipStream_t stream = as_stream(stream_);
    if (M == 0) {
        for (int64_t i = 0; i < Ka; ++i) TFGX_HIP_CHECK(hipMemsetAsync(dW + i * ldw, 0, sizeof(float) * N, stream));
        if (db) TFGX_HIP_CHECK(hipMemsetAsync(db, 0, sizeof(float) * N, stream));
        return TFGX_OK;
    }
    TFGX_REQUIRE(X && G && ldx >= Ka && ldg >= N, "null pointer / leading dimension too small");
    bool want_bias = db != nullptr;
    if (tn_use_direct()) {
        if (tn_bias_by_column_sum(M, Ka, N, want_bias, gate != nullptr)) {
            // db by the column-sum kernel out of the tail of the workspace, dW by the reduction without the ones column
            const TnDirectCfg d0 = tn_direct_config(M, Ka, N, false);
            const size_t parts_bytes = (sizeof(float) * size_t(Ka) * size_t(N) * size_t(d0.wgs) + 255) / 256 * 256;
            const size_t cs_bytes = tfgx_column_sum_workspace_bytes(M, N);
            TFGX_REQUIRE(workspace != nullptr && workspace_bytes >= parts_bytes + cs_bytes, "workspace too small (tfgx_gemm_tn_workspace_bytes)");
            const int rc = tfgx_column_sum_f32(G, ldg, M, N, db, static_cast<char*>(workspace) + parts_bytes, cs_bytes, stream_);
            if (rc != TFGX_OK) return rc;
            want_bias = false;
            db = nullptr;
            workspace_bytes = parts_bytes;
        }
        const TnDirectCfg d = tn_direct_config(M, Ka, N, want_bias);
        const int64_t part_stride = int64_t(Ka + (want_bias ? 1 : 0)) * N;
        TFGX_REQUIRE(workspace != nullptr && workspace_bytes >= sizeof(float) * size_t(part_stride) * size_t(d.wgs),
                     "workspace too small (tfgx_gemm_tn_workspace_bytes)");
        float* parts = static_cast<float*>(workspace);
        dim3 grid(d.wgs, d.m_groups * d.n_groups, 1), block(256, 1, 1);
#define TFGX_TND(S_, TM_, TN_, U_)                                                                                      \
    if (gate != nullptr)                                                                                                \
        gemm_tn_direct_kernel<S_, TM_, TN_, U_, true><<<grid, block, 0, stream>>>(                                      \
            X, ldx, G, ldg, gate, ld_gate, M, int(Ka), int(N), want_bias ? 1 : 0, d.WM, d.m_groups, d.TM, d.TN,         \
            d.rows_per_wg, parts, part_stride);                                                                         \
    else                                                                                                                \
        gemm_tn_direct_kernel<S_, TM_, TN_, U_, false><<<grid, block, 0, stream>>>(                                     \
            X, ldx, G, ldg, nullptr, 0, M, int(Ka), int(N), want_bias ? 1 : 0, d.WM, d.m_groups, d.TM, d.TN,            \
            d.rows_per_wg, parts, part_stride)
        // wave tiles are instantiated at a few sizes; a smaller request runs on the next larger one (tiles past the
        // request are skipped by wave-uniform branches)
        if (d.S == 32) {
            if (d.TN <= 1) {
                if (d.TM <= 1) { TFGX_TND(32, 1, 1, 4); }
                else if (d.TM <= 2) { TFGX_TND(32, 2, 1, 4); }
                else { TFGX_TND(32, 4, 1, 4); }
            } else {
                if (d.TM <= 1) { TFGX_TND(32, 1, 2, 4); }
                else if (d.TM <= 2) { TFGX_TND(32, 2, 2, 4); }
                else if (d.TM <= 3) { TFGX_TND(32, 3, 2, 4); }
                else { TFGX_TND(32, 4, 2, 4); }
            }
        } else {
            if (d.TN <= 1) {
                if (d.TM <= 2) { TFGX_TND(16, 2, 1, 2); }
                else if (d.TM <= 4) { TFGX_TND(16, 4, 1, 2); }
                else { TFGX_TND(16, 8, 1, 2); }
            } else if (d.TN <= 2) {
                if (d.TM <= 2) { TFGX_TND(16, 2, 2, 2); }
                else if (d.TM <= 4) { TFGX_TND(16, 4, 2, 2); }
                else { TFGX_TND(16, 8, 2, 2); }
            } else {
                if (d.TM <= 2) { TFGX_TND(16, 2, 4, 2); }
                else if (d.TM <= 4) { TFGX_TND(16, 4, 4, 2); }
                else if (d.TM <= 6) { TFGX_TND(16, 6, 4, 2); }
                else if (d.TM <= 7) { TFGX_TND(16, 7, 4, 2); }
                else { TFGX_TND(16, 8, 4, 2); }
            }
        }
#undef TFGX_TND
        TFGX_LAUNCH_CHECK("gemm_tn_direct_kernel");
        tn_reduce_kernel<<<tn_reduce_grid(part_stride), 64 * kTnSlices, 0, stream>>>(parts, d.wgs, part_stride, int(Ka), int(N), 0,
                                                                        want_bias ? 1 : 0, dW, ldw, db);
        TFGX_LAUNCH_CHECK("tn_reduce_kernel");
        return TFGX_OK;
    }
    const TnCfg c = tn_config(M, Ka, N, want_bias);
    TFGX_REQUIRE(workspace != nullptr && workspace_bytes >= sizeof(float) * c.part_floats * size_t(c.wgs),
                 "workspace too small (tfgx_gemm_tn_workspace_bytes)");
    float* parts = static_cast<float*>(workspace);
    const int x_vec4 = (Ka % 4 == 0) && (ldx % 4 == 0) && aligned_to(X, 16);
    const int Ti = c.ka_pad / 32;
    for (int gidx = 0; gidx < c.groups; ++gidx) {
        const int n_first = gidx * c.tn_group * 32;
        const int ng_cols = int(N - n_first < c.tn_group * 32 ? N - n_first : c.tn_group * 32);
        const int ng_pad = (ng_cols + 31) / 32 * 32;
        const int g_vec4 = (ng_cols % 4 == 0) && (ldg % 4 == 0) && aligned_to(G + n_first, 16);
        const int tiles = Ti * (ng_pad / 32);
        const int tpw = (tiles + 7) / 8;
        const size_t lds = sizeof(float) * size_t(c.R) * size_t(c.ka_pad + ng_pad);
        const int64_t part_stride = int64_t(Ka + (want_bias ? 1 : 0)) * ng_cols;
#define TFGX_TN(TPW)                                                                                                    \
    {                                                                                                                   \
        static bool attr_set[64] = {false};      /* per device: one process may drive several GPUs */                  \
        int dev_ = 0;                                                                                                   \
        TFGX_HIP_CHECK(hipGetDevice(&dev_));                                                                            \
        TFGX_REQUIRE(dev_ >= 0 && dev_ < 64, "device ordinal out of range");                                            \
        if (!attr_set[dev_]) {                                                                                          \
            TFGX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel<TPW>),                      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                \
            attr_set[dev_] = true;                                                                                      \
        }                                                                                                               \
        gemm_tn_kernel<TPW><<<c.wgs, kTnThreads, lds, stream>>>(X, ldx, G, ldg, M, int(Ka), int(N), n_first, ng_cols,    \
                                                                c.R, c.ka_pad, ng_pad, want_bias ? 1 : 0, parts,        \
                                                                part_stride, x_vec4, g_vec4);                          \
    }
        if (tpw <= 1) TFGX_TN(1)
        else if (tpw <= 2) TFGX_TN(2)
        else if (tpw <= 4) TFGX_TN(4)
        else TFGX_TN(8)
#undef TFGX_TN
        TFGX_LAUNCH_CHECK("gemm_tn_kernel");
        tn_reduce_kernel<<<tn_reduce_grid(part_stride), 64 * kTnSlices, 0, stream>>>(parts, c.wgs, part_stride, int(Ka), ng_cols,
                                                                        n_first, want_bias ? 1 : 0, dW, ldw, db);
        TFGX_LAUNCH_CHECK("tn_reduce_kernel");
    }
    return TFGX_OK;
}

// out[c, r] = in[r, c]  (the [K, N] kernel of a dense layer, transposed for d/dx = g @ kernel^T on the forward GEMM)
namespace tfgx {
namespace {
__global__ void transpose_kernel(const float* __restrict__ in, int64_t ldi, int64_t rows, int64_t cols,
                                 float* __restrict__ out, int64_t ldo)
{
    __shared__ float tile[32][33];
    const int64_t r0 = int64_t(blockIdx.y) * 32, c0 = int64_t(blockIdx.x) * 32;
    for (int dy = threadIdx.y; dy < 32; dy += blockDim.y) {
        const int64_t r = r0 + dy, c = c0 + threadIdx.x;
        tile[dy][threadIdx.x] = (r < rows && c < cols) ? in[r * ldi + c] : 0.0f;
    }
    __syncthreads();
    for (int dy = threadIdx.y; dy < 32; dy += blockDim.y) {
        const int64_t c = c0 + dy, r = r0 + threadIdx.x;
        if (c < cols && r < rows) out[c * ldo + r] = tile[threadIdx.x][dy];
    }
}
}  // namespace
}  // namespace tfgx

extern "C" int tfgx_transpose_f32(const float* in, int64_t ldi, int64_t rows, int64_t cols, float* out, int64_t ldo,
                                  tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(rows >= 0 && cols >= 0 && ldi >= cols && ldo >= rows, "bad shape");
    if (rows == 0 || cols == 0) return TFGX_OK;
    TFGX_REQUIRE(in && out, "null pointer");
    dim3 grid(unsigned((cols + 31) / 32), unsigned((rows + 31) / 32), 1), block(32, 8, 1);
    tfgx::transpose_kernel<<<grid, block, 0, as_stream(stream)>>>(in, ldi, rows, cols, out, ldo);
    TFGX_LAUNCH_CHECK("transpose_kernel");
    return TFGX_OK;
}
