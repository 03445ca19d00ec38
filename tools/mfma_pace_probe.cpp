// What slows an MFMA stream down?  The row GEMM's inner structure rebuilt step by step (TN = 4 accumulators, 32 x 32 x 2 fp32):
//   0: MFMAs only                                    1: + B operands read from LDS one group ahead (2 ds_read2_b32 per 4 MFMAs)
//   2: + A rows streamed from HBM (4 dwordx4 per 32-k step, prefetched one step ahead)
//   3: + a 32 x 128 output tile stored every 8 steps (64 global_store_dword, the D layout)
// Prints shader-clock cycles per MFMA per SIMD (in-kernel clock) for 1 and 2 waves per SIMD, 256 workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_pace_probe.cpp -o tools/mfma_pace_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TN = 4, LDB_S = TN * 32 + 8, K = 256;

template <int MODE>
__global__ __launch_bounds__(512) void pace(const float* __restrict__ A, float* __restrict__ C, int64_t rows, int tiles_per_wave,
                                            unsigned long long* __restrict__ stats)
{
    extern __shared__ float Bs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    for (int i = tid; i < K * LDB_S; i += blockDim.x) {
        uint32_t h = uint32_t(i) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        Bs[i] = (float(h & 0xffffff) / 8388608.0f - 1.0f) * 0.1f;
    }
    __syncthreads();
    const int nw = blockDim.x / 64;
    const int64_t gw = int64_t(blockIdx.x) * nw + wave, total_w = int64_t(gridDim.x) * nw;
    f32x16 acc[TN];
    for (int j = 0; j < TN; ++j)
        for (int t = 0; t < 16; ++t) acc[j][t] = 0.f;
    float cur[16], nxt[16];
    for (int i = 0; i < 16; ++i) cur[i] = 1.0f + 1e-6f * float(lane + i), nxt[i] = cur[i];
    auto load_a = [&](float (&r)[16], int64_t tile, int ks) {
        const int64_t gm = (tile * 32 + l31) % rows;
        const float* p = A + gm * K + ks * 32 + 16 * kh;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * u);
            r[4 * u] = v[0]; r[4 * u + 1] = v[1]; r[4 * u + 2] = v[2]; r[4 * u + 3] = v[3];
        }
    };
    const uint64_t c0 = __builtin_readcyclecounter();
    uint64_t mfmas = 0;
    if (MODE >= 2) load_a(cur, gw, 0);
    for (int it = 0; it < tiles_per_wave; ++it) {
        const int64_t tile = gw + int64_t(it) * total_w;
        for (int ks = 0; ks < K / 32; ++ks) {
            if (MODE >= 2) load_a(nxt, ks + 1 < K / 32 ? tile : tile + total_w, ks + 1 < K / 32 ? ks + 1 : 0);
            const float* b_s = Bs + (ks * 32 + 16 * kh) * LDB_S + l31;
            float bb[2][TN];
            if (MODE >= 1) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bb[0][j] = b_s[j * 32];
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) bb[0][j] = bb[1][j] = 0.5f + float(j);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE >= 1 && i + 1 < 16) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) bb[(i + 1) & 1][j] = b_s[(i + 1) * LDB_S + j * 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i], bb[i & 1][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            mfmas += 16 * TN;
            if (MODE >= 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) cur[i] = nxt[i];
            }
        }
        if (MODE >= 3) {
            const int64_t r0 = (tile * 32 + 4 * kh) % (rows - 32);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float* cp = C + r0 * 128 + j * 32 + l31;
#pragma unroll
                for (int t = 0; t < 16; ++t) cp[int64_t((t & 3) + 8 * (t >> 2)) * 128] = acc[j][t];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[j][t] = 0.0f;
                asm volatile("" : "+v"(acc[j]));
            }
        }
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < TN; ++j)
        for (int t = 0; t < 16; ++t) s += acc[j][t];
    if (s == 1.2345e30f) C[0] = s;
    if (lane == 0) {
        atomicAdd(&stats[0], (unsigned long long)(c1 - c0));
        atomicAdd(&stats[1], (unsigned long long)mfmas);
        atomicAdd(&stats[2], 1ull);
    }
}

__global__ void fill_random(float* p, int64_t n, uint32_t seed)
{
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        uint32_t h = uint32_t(i) * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = (float(h & 0xffffff) / 8388608.0f - 1.0f) * 1.7f;       // uniform in (-1.7, 1.7): every mantissa bit toggles
    }
}

template <int MODE>
void run(int threads, const float* A, float* C, int64_t rows, unsigned long long* stats, int cus)
{
    const int tiles = 24;
    hipFuncSetAttribute(reinterpret_cast<const void*>(pace<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = sizeof(float) * K * LDB_S;
    pace<MODE><<<cus, threads, lds>>>(A, C, rows, 2, stats);
    hipDeviceSynchronize();
    hipMemset(stats, 0, 32);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    pace<MODE><<<cus, threads, lds>>>(A, C, rows, tiles, stats);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4];
    hipMemcpy(h, stats, 32, hipMemcpyDeviceToHost);
    const double waves_per_simd = threads / 256.0;
    const double cyc_per_mfma_wave = double(h[0]) / double(h[1]);
    const double flop = double(h[1]) * 4096.0;
    printf("{\"mode\": %d, \"waves_per_simd\": %.0f, \"ms\": %.3f, \"TFLOPs\": %.1f, \"cycles_per_mfma_per_wave\": %.1f, \"cycles_per_mfma_per_simd\": %.1f}\n",
           MODE, waves_per_simd, ms, flop / (ms * 1e-3) / 1e12, cyc_per_mfma_wave, cyc_per_mfma_wave / waves_per_simd);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int64_t rows = 2400000;
    float *A, *C;
    unsigned long long* stats;
    hipMalloc(&A, sizeof(float) * rows * K);
    hipMalloc(&C, sizeof(float) * rows * 128);
    hipMalloc(&stats, 32);
    if (getenv("PROBE_ZERO_A")) hipMemset(A, 0, sizeof(float) * rows * K);
    else fill_random<<<4096, 256>>>(A, rows * K, 12345u);
    hipDeviceSynchronize();
    for (int threads : {256, 512}) {
        run<0>(threads, A, C, rows, stats, cus);
        run<1>(threads, A, C, rows, stats, cus);
        run<2>(threads, A, C, rows, stats, cus);
        run<3>(threads, A, C, rows, stats, cus);
    }
    return 0;
}
