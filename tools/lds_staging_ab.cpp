// A/B for the north star's "LDS staging of the hot feature tile" on the gather - scale - segment-sum kernel.
//
//   A  register path (what tfgx_segment_reduce_f32 ships): every lane loads its 16 bytes of x[col] straight into the
//      registers it FMAs from; 8 gathered rows in flight per lane group.
//   B  LDS-staged path, in its strongest form on gfx950: global_load_lds_dwordx4 (the memory system writes the gathered
//      row into LDS without passing through VGPRs), DEPTH batches of 8 rows in flight per lane group, then ds_read_b128
//      + FMA.  No VGPR is held by a row in flight, so the staged variant can keep 2-4x more rows in flight.
//   B' plain staging (global_load -> VGPR -> ds_write -> ds_read), the textbook version.
// Same CSR plan, same weights, same FMA order per output element: the three variants must agree bit for bit.
// A gathered row is consumed exactly once by the lane group that fetched it — there is no reuse for LDS to serve — so
// the question this answers is only whether deeper prefetch through LDS beats the register path.
//
//   hipcc --offload-arch=gfx950 -O3 tools/lds_staging_ab.cpp -o tf_geometric_amd/lib/lds_staging_ab
//   ./lds_staging_ab [N=2400000] [E=123000000] [F=100]          -> one JSON line per variant
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define HIP_OK(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));   \
            std::exit(2);                                                     \
        }                                                                     \
    } while (0)

constexpr int G = 32;        // lanes per destination row (F <= 128, 4 columns per lane)
constexpr int U = 8;         // rows per batch
constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------------- A: registers
__global__ __launch_bounds__(kBlock) void reduce_regs(const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                      const float* __restrict__ w, int64_t n, const float* __restrict__ x,
                                                      int F, float* __restrict__ out)
{
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int c = lane * 4 < F ? lane * 4 : F - 4;
    const bool valid = lane * 4 < F;
    for (int64_t r = int64_t(blockIdx.x) * (kBlock / G) + grp; r < n; r += int64_t(gridDim.x) * (kBlock / G)) {
        const int s = row_ptr[r], e = row_ptr[r + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = s; base < e; base += G) {
            const int cj = base + lane < e ? col[base + lane] : 0;
            const float wj = base + lane < e ? w[base + lane] : 0.0f;
            const int cnt = min(G, e - base);
            int j = 0;
            for (; j + U <= cnt; j += U) {
                float4 v[U];
                float ww[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int cc = __shfl(cj, j + u, G);
                    ww[u] = __shfl(wj, j + u, G);
                    v[u] = *reinterpret_cast<const float4*>(x + int64_t(cc) * F + c);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    acc.x = fmaf(ww[u], v[u].x, acc.x); acc.y = fmaf(ww[u], v[u].y, acc.y);
                    acc.z = fmaf(ww[u], v[u].z, acc.z); acc.w = fmaf(ww[u], v[u].w, acc.w);
                }
            }
            for (; j < cnt; ++j) {
                const int cc = __shfl(cj, j, G);
                const float wv = __shfl(wj, j, G);
                const float4 v = *reinterpret_cast<const float4*>(x + int64_t(cc) * F + c);
                acc.x = fmaf(wv, v.x, acc.x); acc.y = fmaf(wv, v.y, acc.y);
                acc.z = fmaf(wv, v.z, acc.z); acc.w = fmaf(wv, v.w, acc.w);
            }
        }
        if (valid) *reinterpret_cast<float4*>(out + r * F + c) = acc;
    }
}

// ------------------------------------------------------------------------------------- B / B': staged through LDS
// Per wave a ring of DEPTH slots; a slot holds U gathered row pieces for the wave's 64 lanes (U * 64 * 16 bytes).
// DIRECT: global_load_lds_dwordx4 (LDS address = wave-uniform slot base + lane * 16); else load -> ds_write.
template <int DEPTH, bool DIRECT>
__global__ __launch_bounds__(kBlock) void reduce_lds(const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                     const float* __restrict__ w, int64_t n, const float* __restrict__ x,
                                                     int F, float* __restrict__ out)
{
    __shared__ float4 ring[kBlock / 64][DEPTH][U][64];
    const int lane = threadIdx.x % G, grp = threadIdx.x / G, wave = threadIdx.x / 64, wl = threadIdx.x % 64;
    const int c = lane * 4 < F ? lane * 4 : F - 4;
    const bool valid = lane * 4 < F;
    for (int64_t r0 = int64_t(blockIdx.x) * (kBlock / G); r0 < n; r0 += int64_t(gridDim.x) * (kBlock / G)) {
        const int64_t r = r0 + grp;                                  // the whole wave iterates together (uniform trip count)
        const int s = r < n ? row_ptr[r] : 0, e = r < n ? row_ptr[r + 1] : 0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int len = e - s;
        const int len_max = max(len, __shfl_xor(len, 32, 64));      // the wave's two rows walk in lock step
        for (int off = 0; off < len_max; off += G) {
            // (col, w) of up to 32 edges, one per lane, coalesced — exactly as the register path does
            const int base = s + off;
            const int cj = base + lane < e ? col[base + lane] : 0;
            const float wj = base + lane < e ? w[base + lane] : 0.0f;
            const int cnt = max(0, min(G, e - base));
            const int cnt_max = max(cnt, __shfl_xor(cnt, 32, 64));
            const int nsb = (cnt_max + U - 1) / U;                   // sub-batches of U rows (wave-uniform)
            auto issue = [&](int b) {
                const int slot = b % DEPTH;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int cc = __shfl(cj, (b * U + u) % G, G);   // lanes past cnt carry col 0: a valid row, weight 0
                    const float* src = x + int64_t(cc) * F + c;
                    if constexpr (DIRECT) {
#if defined(__HIP_DEVICE_COMPILE__)      // device-only builtin: the host pass must not see it
                        __builtin_amdgcn_global_load_lds(src, &ring[wave][slot][u][0], 16, 0, 0);
#endif
                    } else {
                        ring[wave][slot][u][wl] = *reinterpret_cast<const float4*>(src);
                    }
                }
            };
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (DIRECT) __builtin_amdgcn_s_waitcnt(0x0f70);    // cj / wj have landed; only row loads counted below
#endif
            const int pre = min(nsb, DEPTH - 1);
            for (int b = 0; b < pre; ++b) issue(b);
            for (int b = 0; b < nsb; ++b) {
                if (b + DEPTH - 1 < nsb) issue(b + DEPTH - 1);
                if constexpr (DIRECT) {
#if defined(__HIP_DEVICE_COMPILE__)
                    // s_waitcnt simm16 on gfx9: vmcnt = bits [3:0] | [15:14], expcnt [6:4] = 7, lgkmcnt [11:8] = 15
                    const int later = min(nsb - 1, b + DEPTH - 1) - b;   // sub-batches issued after b, still allowed in flight
                    if (later >= 3) __builtin_amdgcn_s_waitcnt(0x0f70 | ((3 * U) & 15) | (((3 * U) >> 4) << 14));
                    else if (later == 2) __builtin_amdgcn_s_waitcnt(0x0f70 | ((2 * U) & 15) | (((2 * U) >> 4) << 14));
                    else if (later == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | (U & 15));
                    else __builtin_amdgcn_s_waitcnt(0x0f70);
                    __builtin_amdgcn_wave_barrier();
#endif
                }
                const int slot = b % DEPTH;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = b * U + u;
                    const float wv = __shfl(wj, j % G, G);
                    const float4 v = ring[wave][slot][u][wl];
                    if (j < cnt) {
                        acc.x = fmaf(wv, v.x, acc.x); acc.y = fmaf(wv, v.y, acc.y);
                        acc.z = fmaf(wv, v.z, acc.z); acc.w = fmaf(wv, v.w, acc.w);
                    }
                }
            }
        }
        if (valid && r < n) *reinterpret_cast<float4*>(out + r * F + c) = acc;
    }
}

// explicit instantiations (the launches sit inside lambdas, which does not instantiate the host stubs)
template __global__ void reduce_lds<2, true>(const int*, const int*, const float*, int64_t, const float*, int, float*);
template __global__ void reduce_lds<4, true>(const int*, const int*, const float*, int64_t, const float*, int, float*);
template __global__ void reduce_lds<2, false>(const int*, const int*, const float*, int64_t, const float*, int, float*);

template <typename Fn>
static float time_ms(Fn launch, int reps)
{
    for (int i = 0; i < 3; ++i) launch();
    hipEvent_t a, b;
    HIP_OK(hipEventCreate(&a));
    HIP_OK(hipEventCreate(&b));
    HIP_OK(hipEventRecord(a, nullptr));
    for (int i = 0; i < reps; ++i) launch();
    HIP_OK(hipEventRecord(b, nullptr));
    HIP_OK(hipEventSynchronize(b));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 2400000;
    const int64_t E = argc > 2 ? atoll(argv[2]) : 123000000;
    const int F = argc > 3 ? atoi(argv[3]) : 100;
    if (F % 4 != 0 || F > 128) { std::fprintf(stderr, "F must be a multiple of 4, <= 128\n"); return 2; }
    // uniform random graph, CSR by destination built on the host (the plan is not what is being measured)
    std::mt19937_64 rng(7);
    std::vector<int> deg(N, 0), dst(E);
    for (int64_t i = 0; i < E; ++i) { dst[i] = int(rng() % uint64_t(N)); ++deg[dst[i]]; }
    std::vector<int> rp(N + 1, 0);
    for (int64_t r = 0; r < N; ++r) rp[r + 1] = rp[r] + deg[r];
    std::vector<int> col(E);
    std::vector<float> w(E);
    for (int64_t i = 0; i < E; ++i) { col[i] = int(rng() % uint64_t(N)); w[i] = 0.5f + float(rng() % 1000) / 1000.0f; }
    std::vector<float> x(size_t(N) * F);
    for (size_t i = 0; i < x.size(); ++i) x[i] = float(int(rng() % 2001) - 1000) / 500.0f;
    int *d_rp, *d_col;
    float *d_w, *d_x, *d_o[4];
    HIP_OK(hipMalloc(&d_rp, (N + 1) * 4));
    HIP_OK(hipMalloc(&d_col, E * 4));
    HIP_OK(hipMalloc(&d_w, E * 4));
    HIP_OK(hipMalloc(&d_x, x.size() * 4));
    for (auto& p : d_o) HIP_OK(hipMalloc(&p, x.size() * 4));
    HIP_OK(hipMemcpy(d_rp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_col, col.data(), E * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_w, w.data(), E * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    const int grid = int(std::min<int64_t>((N + kBlock / G - 1) / (kBlock / G), 1 << 20));
    const double bytes = double(E) * (4.0 * F + 8) + double(N) * 4 * F + 4.0 * (N + 1);
    struct V { const char* name; float ms; } res[4];
    res[0] = {"A registers (shipped path), 8 rows in flight",
              time_ms([&] { reduce_regs<<<grid, kBlock>>>(d_rp, d_col, d_w, N, d_x, F, d_o[0]); }, 10)};
    res[1] = {"B global_load_lds_dwordx4, ring depth 2 (8-16 rows in flight, no VGPR per row)",
              time_ms([&] { reduce_lds<2, true><<<grid, kBlock>>>(d_rp, d_col, d_w, N, d_x, F, d_o[1]); }, 10)};
    res[2] = {"B global_load_lds_dwordx4, ring depth 4 (24-32 rows in flight)",
              time_ms([&] { reduce_lds<4, true><<<grid, kBlock>>>(d_rp, d_col, d_w, N, d_x, F, d_o[2]); }, 10)};
    res[3] = {"B' load -> ds_write -> ds_read, ring depth 2",
              time_ms([&] { reduce_lds<2, false><<<grid, kBlock>>>(d_rp, d_col, d_w, N, d_x, F, d_o[3]); }, 10)};
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> ref(x.size()), got(x.size());
    HIP_OK(hipMemcpy(ref.data(), d_o[0], x.size() * 4, hipMemcpyDeviceToHost));
    for (int v = 0; v < 4; ++v) {
        bool same = true;
        if (v > 0) {
            HIP_OK(hipMemcpy(got.data(), d_o[v], x.size() * 4, hipMemcpyDeviceToHost));
            same = std::memcmp(ref.data(), got.data(), x.size() * 4) == 0;
        }
        std::printf("{\"probe\": \"lds_staging_ab\", \"variant\": \"%s\", \"N\": %lld, \"E\": %lld, \"F\": %d, \"ms\": %.4f, "
                    "\"G_edges_per_s\": %.3f, \"algorithmic_TBps\": %.3f, \"bit_identical_to_A\": %s}\n",
                    res[v].name, (long long)N, (long long)E, F, res[v].ms, double(E) / res[v].ms / 1e6,
                    bytes / res[v].ms / 1e9, same ? "true" : "false");
    }
    return 0;
}
