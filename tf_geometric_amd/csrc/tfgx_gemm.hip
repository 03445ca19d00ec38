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
// Streaming variant for the shape this library actually meets: M = nodes (10^5..10^8), K and N = feature widths
// with K*N small enough that the WHOLE weight matrix lives in LDS (160 KB per CU).  One 512-thread workgroup per CU
// (8 waves: 4 along M x 2 along N, two waves per SIMD) is persistent: B is loaded into LDS once, then the workgroup
// walks M-tiles of 128 rows; A streams through a double-buffered LDS tile (BK = 16) with ONE barrier per k-step and
// the global loads of the next step (or of the next M-tile's first step) issued before the MFMAs of the current
// one, so there is no per-tile prologue bubble.  Same arithmetic as gemm_kernel: k-ordered fp32 FMA chain.
template <int TN>
__global__ __launch_bounds__(512) void gemm_stream_kernel(const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int64_t ldb,
                                                          const float* __restrict__ bias, int act, int act_cols,
                                                          float* __restrict__ C, int64_t ldc, int64_t M, int K, int N,
                                                          int64_t n_mtiles)
{
    constexpr int BM = 128, SBK = 16, LDA_S = BM + 1;
    constexpr int LDB_S = 2 * TN * 32 + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nk = (K + SBK - 1) / SBK;
    float* Bs = smem;                         // [nk*SBK][LDB_S], zero padded
    float* As = smem + nk * SBK * LDB_S;      // [2][SBK][LDA_S]

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;

    for (int idx = tid; idx < nk * SBK * LDB_S; idx += 512) {
        const int k = idx / LDB_S, n = idx - k * LDB_S;
        Bs[idx] = (k < K && n < N) ? B[int64_t(k) * ldb + n] : 0.0f;
    }

    const int arow = tid >> 2, akq = (tid & 3) * 4;   // one float4 of the 128 x 16 A tile per thread
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_a = [&](int64_t mt, int k0) {
        const int64_t gm = mt * BM + arow;
        const int gk = k0 + akq;
        ra = (gm < M && gk < K) ? *reinterpret_cast<const float4*>(A + gm * lda + gk) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_a = [&](int buf) {
        float* a = As + buf * SBK * LDA_S;
        a[(akq + 0) * LDA_S + arow] = ra.x;
        a[(akq + 1) * LDA_S + arow] = ra.y;
        a[(akq + 2) * LDA_S + arow] = ra.z;
        a[(akq + 3) * LDA_S + arow] = ra.w;
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[j][t] = 0.0f;

    int64_t mt = blockIdx.x;
    if (mt < n_mtiles) load_a(mt, 0);
    int buf = 0;
    for (; mt < n_mtiles; mt += gridDim.x) {
        for (int kt = 0; kt < nk; ++kt) {
            store_a(buf);
            __syncthreads();
            if (kt + 1 < nk) load_a(mt, (kt + 1) * SBK);
            else if (mt + gridDim.x < n_mtiles) load_a(mt + gridDim.x, 0);
            const float* a_s = As + buf * SBK * LDA_S;
            const float* b_s = Bs + kt * SBK * LDB_S;
            const int kmax = min(SBK, K - kt * SBK);
#pragma unroll
            for (int kk = 0; kk < SBK; kk += 2) {
                if (kk < kmax) {
                    const float a = a_s[(kk + kh) * LDA_S + wm * 32 + l31];
                    float b[TN];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = b_s[(kk + kh) * LDB_S + (wn * TN + j) * 32 + l31];
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], acc[j], 0, 0, 0);
                }
            }
            buf ^= 1;
        }
        // epilogue of this M-tile; the next tile's first A slice is already in flight.  Bias + activation in place
        // first, then stores straight from the accumulator registers (see gemm_kernel's epilogue note).
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = (wn * TN + j) * 32 + l31;
            const float bv = (bias && gn < N) ? bias[gn] : 0.0f;
            const int a_j = gn < act_cols ? act : TFGX_ACT_NONE;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[j][t] = apply_act(acc[j][t] + bv, a_j);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = (wn * TN + j) * 32 + l31;
            if (gn < N) {
                float* cp = C + (mt * BM + wm * 32 + 4 * kh) * ldc + gn;
                const int64_t rows_left = M - (mt * BM + wm * 32 + 4 * kh);
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int dr = (t & 3) + 8 * (t >> 2);
                    if (dr < rows_left) cp[int64_t(dr) * ldc] = acc[j][t];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[j][t] = 0.0f;
    }
}

template <int TN>
int launch_gemm_stream(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, int act, float* C,
                       int64_t ldc, int64_t M, int K, int N, int act_cols, hipStream_t stream)
{
    constexpr int LDB_S = 2 * TN * 32 + 4;
    const int nk = (K + 15) / 16;
    const size_t lds = sizeof(float) * (size_t(nk) * 16 * LDB_S + 2 * 16 * 129);
    static int cus = 0;
    static bool attr_set = false;
    if (!attr_set) {
        int dev = 0;
        hipDeviceProp_t prop;
        TFGX_HIP_CHECK(hipGetDevice(&dev));
        TFGX_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        TFGX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_stream_kernel<TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int64_t n_mtiles = (M + 127) / 128;
    dim3 grid(static_cast<unsigned>(n_mtiles < cus ? n_mtiles : cus), 1, 1), block(512, 1, 1);
    gemm_stream_kernel<TN><<<grid, block, lds, stream>>>(A, lda, B, ldb, bias, act, act_cols, C, ldc, M, K, N, n_mtiles);
    TFGX_LAUNCH_CHECK("gemm_stream_kernel");
    return TFGX_OK;
}

// the streaming kernel needs: 16-byte aligned A rows with K % 4 == 0, 64 < N <= 256, B + A buffers within 160 KB of LDS,
// and enough M-tiles to keep a persistent grid busy
inline bool stream_ok(const float* A, int64_t lda, int64_t M, int64_t K, int64_t N)
{
    if (N <= 64 || N > 256 || K % 4 != 0 || lda % 4 != 0 || !aligned_to(A, 16) || M < 128 * 256) return false;
    const int tn = int((N + 63) / 64);
    const size_t lds = sizeof(float) * (size_t((K + 15) / 16) * 16 * (2 * tn * 32 + 4) + 2 * 16 * 129);
    return lds <= 160 * 1024;
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
    if (stream_ok(A, lda, M, K, N)) {
        switch ((N + 63) / 64) {
            case 2: return launch_gemm_stream<2>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream);
            case 3: return launch_gemm_stream<3>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream);
            default: return launch_gemm_stream<4>(A, lda, B, ldb, bias, act, C, ldc, M, int(K), int(N), ac, stream);
        }
    }
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
