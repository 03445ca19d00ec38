// C = act(A[M,K] @ B[K,N] + bias), fp32 in / fp32 accumulate on the matrix cores.
// Replaces `x @ kernel` (+ bias, activation) next to the aggregation:
// tf_geometric/nn/conv/gcn.py:272, nn/conv/gat.py:52-70, nn/conv/graph_sage.py:43-44,199,208-209.
//
// v_mfma_f32_32x32x2_f32 (f32-input MFMA): exact fp32, bitwise a k-ordered FMA chain per output element
// (no bf16/tf32 down-cast: parity is 1e-5 against an fp32 CPU reference).  M is the node count (10^5..10^8),
// K and N are feature widths (16..1433), so the kernel is a tall-skinny GEMM close to the HBM roofline:
// the A panel is streamed once, B (<= 1.5 MB) lives in L2/LDS.
//
// Tile: BM x BN per 256-thread workgroup (4 waves), BK = 16.  A is staged transposed in LDS (As[k][m]) so
// the MFMA A operand (lane l: A[m = l&31][k = l>>5]) is a conflict-free ds_read_b32 over consecutive m;
// B is staged as Bs[k][n].  Global loads for tile t+1 are issued before the MFMAs of tile t (register
// staging), LDS is single-buffered.
#include "tfgx_common.h"

namespace tfgx {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;   // 16 keeps load-staging registers low enough for 3 workgroups per CU (BK = 32: 2)

template <int BM, int BN, int WM, int WN, bool AV4, bool BV4>
__global__ __launch_bounds__(kBlock) void gemm_kernel(const float* __restrict__ A, int64_t lda,
                                                      const float* __restrict__ B, int64_t ldb,
                                                      const float* __restrict__ bias, int act,
                                                      float* __restrict__ C, int64_t ldc, int64_t M, int K, int N,
                                                      int n_tiles_n, int act_cols)
{
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per workgroup");
    constexpr int A_LOADS = BM * BK / 4 / kBlock;   // float4 per thread
    constexpr int B_LOADS = (BK * BN / 4 + kBlock - 1) / kBlock;
    constexpr int B_THREADS_PER_ROW = BN / 4;
    constexpr int LDA_S = BM + 1, LDB_S = BN + 4;   // +1: spreads the transposed A stores over banks

    __shared__ __attribute__((aligned(16))) float As[BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB_S];

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

    float4 ra[A_LOADS], rb[B_LOADS];

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int f = tid + i * kBlock;          // float4 index inside the BM x BK tile
            const int row = f / (BK / 4), kq = (f % (BK / 4)) * 4;
            const int64_t gm = m0 + row;
            const int gk = k0 + kq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gm < M) {
                const float* p = A + gm * lda + gk;
                if (AV4 && gk + 3 < K) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (gk < K) v.x = p[0];
                    if (gk + 1 < K) v.y = p[1];
                    if (gk + 2 < K) v.z = p[2];
                    if (gk + 3 < K) v.w = p[3];
                }
            }
            ra[i] = v;
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

    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int f = tid + i * kBlock;
            const int row = f / (BK / 4), kq = (f % (BK / 4)) * 4;
            As[(kq + 0) * LDA_S + row] = ra[i].x;
            As[(kq + 1) * LDA_S + row] = ra[i].y;
            As[(kq + 2) * LDA_S + row] = ra[i].z;
            As[(kq + 3) * LDA_S + row] = ra[i].w;
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

    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        store_tiles();
        __syncthreads();
        if (k0 + BK < K) load_tiles(k0 + BK);
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
        // unrolled, so the LDS reads of step kk+1 issue under the MFMAs of step kk; the zero-padded tail of the last
        // tile is skipped by a wave-uniform guard per step (one accumulator copy, no second loop body)
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            if (kk < kmax) kstep(kk);
        }
        __syncthreads();
    }

    // epilogue: D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gn = n0 + wn * WN + j * 32 + l31;
        if (gn >= N) continue;
        const float bv = bias ? bias[gn] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int64_t gm = m0 + wm * WM + i * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh;
                if (gm < M) C[gm * ldc + gn] = apply_act(acc[i][j][t] + bv, gn < act_cols ? act : TFGX_ACT_NONE);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_gemm(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, int act, float* C,
                int64_t ldc, int64_t M, int K, int N, int act_cols, hipStream_t stream)
{
    const int ntn = (N + BN - 1) / BN;
    const int64_t ntm = (M + BM - 1) / BM;
    const int64_t blocks = ntm * ntn;
    if (blocks >= (int64_t(1) << 31)) {
        set_error("tfgx_gemm_bias_act_f32: too many tiles");
        return TFGX_ERR_INVALID_ARG;
    }
    const bool av4 = (lda % 4 == 0) && aligned_to(A, 16);
    const bool bv4 = (ldb % 4 == 0) && aligned_to(B, 16);
    dim3 grid(static_cast<unsigned>(blocks), 1, 1), block(kBlock, 1, 1);
#define TFGX_GEMM_GO(AV, BV) \
    gemm_kernel<BM, BN, WM, WN, AV, BV><<<grid, block, 0, stream>>>(A, lda, B, ldb, bias, act, C, ldc, M, K, N, ntn, act_cols)
    if (av4 && bv4) TFGX_GEMM_GO(true, true);
    else if (av4) TFGX_GEMM_GO(true, false);
    else if (bv4) TFGX_GEMM_GO(false, true);
    else TFGX_GEMM_GO(false, false);
#undef TFGX_GEMM_GO
    TFGX_LAUNCH_CHECK("gemm_kernel");
    return TFGX_OK;
}

}  // namespace
}  // namespace tfgx

using namespace tfgx;

extern "C" int tfgx_gemm_bias_act_cols_f32(const float* A, int64_t lda, const float* B, int64_t ldb,
                                           const float* bias, int32_t act, int64_t act_cols, float* C, int64_t ldc,
                                           int64_t M, int64_t K, int64_t N, tfgx_stream_t stream_)
{
    TFGX_REQUIRE(M >= 0 && K >= 1 && N >= 1, "bad M / K / N");
    TFGX_REQUIRE(K < (int64_t(1) << 30) && N < (int64_t(1) << 30), "K / N too large");
    TFGX_REQUIRE(act == TFGX_ACT_NONE || act == TFGX_ACT_RELU, "bad act");
    TFGX_REQUIRE(act_cols >= 0 && act_cols <= N, "act_cols outside [0, N]");
    if (M == 0) return TFGX_OK;
    TFGX_REQUIRE(A && B && C, "null pointer");
    TFGX_REQUIRE(lda >= K && ldb >= N && ldc >= N, "leading dimension too small");
    hipStream_t stream = as_stream(stream_);
    const int ac = int(act_cols);
    if (N <= 32) return launch_gemm<256, 32, 64, 32>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream);
    if (N <= 64) return launch_gemm<128, 64, 32, 64>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream);
    return launch_gemm<128, 128, 64, 64>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream);
}

extern "C" int tfgx_gemm_bias_act_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                                      int32_t act, float* C, int64_t ldc, int64_t M, int64_t K, int64_t N,
                                      tfgx_stream_t stream)
{
    return tfgx_gemm_bias_act_cols_f32(A, lda, B, ldb, bias, act, N, C, ldc, M, K, N, stream);
}
