// Same-box A/B of the gather - scale - segment-sum walk on WIDE rows (F = 128 ... 512), VERDICT r4 item 1.
//
// Baseline = the shipped kernel through the C ABI (tfgx_segment_reduce_f32, libtfgx.so); every variant below reduces the same
// CSR plan with the same per-element FMA chain (edges in CSR order, then the implicit self-loop edge with ONE fma), so its output
// must equal the baseline's bit for bit (checked on the device).  Variants (template arguments <G, CH, U, MODE>):
//   G     lanes per destination row (64 / G rows per wave),   CH  16-byte column chunks per lane,   U  edges per batch
//   MODE  bit 0  MASKED TAIL  the last partial batch of a (col, w) block runs as ONE batch with clamped indices and
//                             predicated FMAs instead of one dependent load per edge
//         bit 1  EARLY SELF   self_coef[r] and x[r] are loaded before the edge walk, not after it (the epilogue then holds no
//                             dependent memory round trip)
//         bit 2  NT           gathered rows are loaded with the non-temporal hint
//         bit 4  ROT          the software-pipelined row headers are rotated after the row's walk instead of before it
//         bit 3  YSPLIT       rows wider than G*4*CH columns are cut into column blocks on blockIdx.y (col / w re-read per block)
// "probe" = the gather alone: random row ids from a hash, U rows in flight per lane, no index stream, one store per 64 rows:
// the ceiling of the access pattern at that table size.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/wide_row_ab.cpp -L tf_geometric_amd/lib -ltfgx \
//         -Wl,-rpath,'$ORIGIN' -o tf_geometric_amd/lib/wide_row_ab
//   ./wide_row_ab [N=2400000] [E=123000000] [F list, e.g. 128,256,512] [only=<substring of a variant name>] [reps=6]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tfgx.h"

#define HIP_OK(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));   \
            std::exit(2);                                                     \
        }                                                                     \
    } while (0)

constexpr int kBlock = 256;

__host__ __device__ inline uint32_t mix(uint64_t v)
{
    v ^= v >> 33; v *= 0xff51afd7ed558ccdULL; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ULL; v ^= v >> 33;
    return uint32_t(v);
}

__global__ void k_fill_x(float* x, size_t n)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
        x[i] = float(int(mix(i * 3 + 1) % 4001u) - 2000) / 1000.0f;
}
__global__ void k_deg(int* deg, int64_t E, int64_t N)
{
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < E; i += int64_t(gridDim.x) * blockDim.x)
        atomicAdd(&deg[mix(uint64_t(i) * 7 + 3) % uint64_t(N)], 1);
}
__global__ void k_edges(int* col, float* w, int64_t E, int64_t N)
{
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < E; i += int64_t(gridDim.x) * blockDim.x) {
        col[i] = int(mix(uint64_t(i) * 11 + 5) % uint64_t(N));
        w[i] = 0.5f + float(mix(uint64_t(i) * 13 + 9) % 1000u) / 1000.0f;
    }
}
__global__ void k_self(float* sc, int64_t N)
{
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < N; i += int64_t(gridDim.x) * blockDim.x)
        sc[i] = 0.01f + float(mix(uint64_t(i) * 17 + 1) % 100u) / 1000.0f;
}
__global__ void k_diff(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* cnt)
{
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(cnt, c);
}

struct Args {
    const int* row_ptr;
    const int* col;
    const float* w;
    const float* x;
    const float* sc;
    float* out;
    int64_t n;
    int F;
};

enum { M_MASK = 1, M_EARLY = 2, M_NT = 4, M_YSPLIT = 8, M_ROT = 16 };

template <int G>
__device__ __forceinline__ int bc_i(int v, int j)
{
    if constexpr (G == 64) return __builtin_amdgcn_readlane(v, j);
    else return __shfl(v, j, G);
}
template <int G>
__device__ __forceinline__ float bc_f(float v, int j)
{
    if constexpr (G == 64) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
    else return __shfl(v, j, G);
}
template <bool NT>
__device__ __forceinline__ float4 ld4(const float* p)
{
    if constexpr (NT) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    } else {
        return *reinterpret_cast<const float4*>(p);
    }
}

template <int G, int CH, int U, int MODE>
__global__ __launch_bounds__(kBlock) void reduce_var(const Args a)
{
    constexpr bool MASK = MODE & M_MASK, EARLY = MODE & M_EARLY, NT = MODE & M_NT, ROT = MODE & M_ROT;
    constexpr int RPB = kBlock / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int colbase = (MODE & M_YSPLIT) ? int(blockIdx.y) * G * 4 * CH : 0;
    int coff[CH];
    bool cvalid[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int c = colbase + (k * G + lane) * 4;
        cvalid[k] = c < a.F;
        coff[k] = cvalid[k] ? c : a.F - 4;
    }
    const int64_t rstride = int64_t(gridDim.x) * RPB;
    int64_t r = int64_t(blockIdx.x) * RPB + grp;
    int s = 0, e = 0, s1 = 0, e1 = 0;
    if (r < a.n) { s = a.row_ptr[r]; e = a.row_ptr[r + 1]; }
    if (r + rstride < a.n) { s1 = a.row_ptr[r + rstride]; e1 = a.row_ptr[r + rstride + 1]; }
    int cj_first = 0;
    float wj_first = 0.f;
    if (s + lane < e) { cj_first = a.col[s + lane]; wj_first = a.w[s + lane]; }
    for (; r < a.n; r += rstride) {
        int s2 = 0, e2 = 0;
        if (r + 2 * rstride < a.n) { s2 = a.row_ptr[r + 2 * rstride]; e2 = a.row_ptr[r + 2 * rstride + 1]; }
        int cj_first1 = 0;
        float wj_first1 = 0.f;
        if (s1 + lane < e1) { cj_first1 = a.col[s1 + lane]; wj_first1 = a.w[s1 + lane]; }
        const int sc_ = G == 64 ? __builtin_amdgcn_readfirstlane(s) : s;
        const int ec_ = G == 64 ? __builtin_amdgcn_readfirstlane(e) : e;
        int cj_next = cj_first;
        float wj_next = wj_first;
        if constexpr (!ROT) { s = s1; e = e1; s1 = s2; e1 = e2; cj_first = cj_first1; wj_first = wj_first1; }

        float4 acc[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        float self_c = 0.f;
        float4 xself[CH];
        if constexpr (EARLY) {
            self_c = a.sc[r];
#pragma unroll
            for (int k = 0; k < CH; ++k) xself[k] = ld4<false>(a.x + r * int64_t(a.F) + coff[k]);
        }
        for (int base = sc_; base < ec_; base += G) {
            const int cj = cj_next;
            const float wj = wj_next;
            const int nxt = base + G + lane;
            if (nxt < ec_) { cj_next = a.col[nxt]; wj_next = a.w[nxt]; }
            const int cnt = min(G, ec_ - base);
            int j = 0;
            for (; j + U <= cnt; j += U) {
                float4 xv[U][CH];
                float ww[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int c = bc_i<G>(cj, j + u);
                    ww[u] = bc_f<G>(wj, j + u);
#pragma unroll
                    for (int k = 0; k < CH; ++k) xv[u][k] = ld4<NT>(a.x + int64_t(c) * a.F + coff[k]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        acc[k].x = fmaf(ww[u], xv[u][k].x, acc[k].x);
                        acc[k].y = fmaf(ww[u], xv[u][k].y, acc[k].y);
                        acc[k].z = fmaf(ww[u], xv[u][k].z, acc[k].z);
                        acc[k].w = fmaf(ww[u], xv[u][k].w, acc[k].w);
                    }
            }
            if constexpr (MASK) {
                if (j < cnt) {      // one partial batch: loads from clamped edge slots (repeats hit the same lines), FMAs predicated
                    float4 xv[U][CH];
                    float ww[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int idx = min(j + u, cnt - 1);
                        const int c = bc_i<G>(cj, idx);
                        ww[u] = bc_f<G>(wj, idx);
#pragma unroll
                        for (int k = 0; k < CH; ++k) xv[u][k] = ld4<NT>(a.x + int64_t(c) * a.F + coff[k]);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool live = j + u < cnt;      // a select, not a branch: a branch lets the compiler sink the load into it
#pragma unroll
                        for (int k = 0; k < CH; ++k) {
                            const float tx = fmaf(ww[u], xv[u][k].x, acc[k].x), ty = fmaf(ww[u], xv[u][k].y, acc[k].y);
                            const float tz = fmaf(ww[u], xv[u][k].z, acc[k].z), tw = fmaf(ww[u], xv[u][k].w, acc[k].w);
                            acc[k].x = live ? tx : acc[k].x;
                            acc[k].y = live ? ty : acc[k].y;
                            acc[k].z = live ? tz : acc[k].z;
                            acc[k].w = live ? tw : acc[k].w;
                        }
                    }
                }
            } else {
                for (; j < cnt; ++j) {
                    const int c = bc_i<G>(cj, j);
                    const float wv = bc_f<G>(wj, j);
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const float4 v = ld4<NT>(a.x + int64_t(c) * a.F + coff[k]);
                        acc[k].x = fmaf(wv, v.x, acc[k].x);
                        acc[k].y = fmaf(wv, v.y, acc[k].y);
                        acc[k].z = fmaf(wv, v.z, acc[k].z);
                        acc[k].w = fmaf(wv, v.w, acc[k].w);
                    }
                }
            }
        }
        if constexpr (!EARLY) self_c = a.sc[r];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (!cvalid[k]) continue;
            if constexpr (!EARLY) xself[k] = ld4<false>(a.x + r * int64_t(a.F) + coff[k]);
            float4 res;
            res.x = fmaf(self_c, xself[k].x, acc[k].x);
            res.y = fmaf(self_c, xself[k].y, acc[k].y);
            res.z = fmaf(self_c, xself[k].z, acc[k].z);
            res.w = fmaf(self_c, xself[k].w, acc[k].w);
            *reinterpret_cast<float4*>(a.out + r * int64_t(a.F) + coff[k]) = res;
        }
        // ROT: the prefetched header / first (col, w) block of the next rows are taken over AFTER this row's walk — rotating them
        // before it makes the walk wait for those loads to land (a round trip in front of every row)
        if constexpr (ROT) { s = s1; e = e1; s1 = s2; e1 = e2; cj_first = cj_first1; wj_first = wj_first1; }
    }
}

// Column blocks of G*4 columns, one row x block per lane group, three ways to place the blocks of a row:
//   MAP 0  blockIdx.y = block (the blocks of a row run far apart in time)
//   MAP 1  inside the workgroup: wave w takes block w % nb (the blocks of a row run on the same CU at the same time)
//   MAP 2  1-D grid, XCD-aware: workgroups 8 apart in launch order (same XCD, same L2 for the col / w re-reads) take the blocks
//          of the same rows
template <int G, int U, int MAP>
__global__ __launch_bounds__(kBlock) void reduce_blk(const Args a, int nb)
{
    const int lane = threadIdx.x % G;
    int yb, grp, rpb;
    int64_t rb;
    if constexpr (MAP == 0) { yb = blockIdx.y; rb = blockIdx.x; rpb = kBlock / G; grp = threadIdx.x / G; }
    else if constexpr (MAP == 1) {
        const int wave = threadIdx.x / 64;
        yb = wave % nb; rpb = (4 / nb) * (64 / G); grp = (wave / nb) * (64 / G) + (threadIdx.x % 64) / G; rb = blockIdx.x;
    } else {
        const int64_t wg = blockIdx.x;
        const int64_t t = wg >> 3;
        yb = int(t % nb); rb = (t / nb) * 8 + (wg & 7); rpb = kBlock / G; grp = threadIdx.x / G;
    }
    const int64_t r = rb * rpb + grp;
    if (r >= a.n) return;
    const int c0 = yb * G * 4 + lane * 4;
    const bool cvalid = c0 < a.F;
    const int coff = cvalid ? c0 : a.F - 4;
    int s = a.row_ptr[r], e = a.row_ptr[r + 1];
    if constexpr (G == 64) { s = __builtin_amdgcn_readfirstlane(s); e = __builtin_amdgcn_readfirstlane(e); }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cj_next = 0;
    float wj_next = 0.f;
    if (s + lane < e) { cj_next = a.col[s + lane]; wj_next = a.w[s + lane]; }
    for (int base = s; base < e; base += G) {
        const int cj = cj_next;
        const float wj = wj_next;
        const int nxt = base + G + lane;
        if (nxt < e) { cj_next = a.col[nxt]; wj_next = a.w[nxt]; }
        const int cnt = min(G, e - base);
        int j = 0;
        for (; j + U <= cnt; j += U) {
            float4 xv[U];
            float ww[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = bc_i<G>(cj, j + u);
                ww[u] = bc_f<G>(wj, j + u);
                xv[u] = ld4<false>(a.x + int64_t(c) * a.F + coff);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc.x = fmaf(ww[u], xv[u].x, acc.x); acc.y = fmaf(ww[u], xv[u].y, acc.y);
                acc.z = fmaf(ww[u], xv[u].z, acc.z); acc.w = fmaf(ww[u], xv[u].w, acc.w);
            }
        }
        for (; j < cnt; ++j) {
            const int c = bc_i<G>(cj, j);
            const float wv = bc_f<G>(wj, j);
            const float4 v = ld4<false>(a.x + int64_t(c) * a.F + coff);
            acc.x = fmaf(wv, v.x, acc.x); acc.y = fmaf(wv, v.y, acc.y); acc.z = fmaf(wv, v.z, acc.z); acc.w = fmaf(wv, v.w, acc.w);
        }
    }
    if (!cvalid) return;
    const float self_c = a.sc[r];
    const float4 xs = ld4<false>(a.x + r * int64_t(a.F) + coff);
    float4 res;
    res.x = fmaf(self_c, xs.x, acc.x); res.y = fmaf(self_c, xs.y, acc.y); res.z = fmaf(self_c, xs.z, acc.z); res.w = fmaf(self_c, xs.w, acc.w);
    *reinterpret_cast<float4*>(a.out + r * int64_t(a.F) + coff) = res;
}

// the gather alone, in PIECES: LPR lanes fetch 16*LPR contiguous bytes at a random (row, column block) of the [n, F] table
template <int LPR, int U>
__global__ __launch_bounds__(kBlock) void probe_piece(const float* x, int64_t n, int F, int64_t per_group, float* sink)
{
    const int lane = threadIdx.x % LPR;
    const int64_t g = (blockIdx.x * int64_t(kBlock) + threadIdx.x) / LPR;
    const int nb = F / (LPR * 4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = 0; i < per_group; i += U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t h = mix(uint64_t(g) * 0x9E3779B97F4A7C15ULL + uint64_t(i + u));
            const int64_t row = h % uint64_t(n);
            const int blk = int((h >> 20) % uint32_t(nb));
            v[u] = *reinterpret_cast<const float4*>(x + row * F + (blk * LPR + lane) * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x == 12345.678f) sink[0] = acc.x + acc.y + acc.z + acc.w;
}

// the gather alone: LPR lanes fetch one row of 16*LPR bytes each, U rows in flight per lane
template <int LPR, int CH, int U>
__global__ __launch_bounds__(kBlock) void probe(const float* x, int64_t n, int F, int64_t rows_per_group, float* sink)
{
    const int lane = threadIdx.x % LPR;
    const int64_t g = (blockIdx.x * int64_t(kBlock) + threadIdx.x) / LPR;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = 0; i < rows_per_group; i += U) {
        float4 v[U][CH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = mix(uint64_t(g) * 0x9E3779B97F4A7C15ULL + uint64_t(i + u)) % uint64_t(n);
#pragma unroll
            for (int k = 0; k < CH; ++k) v[u][k] = *reinterpret_cast<const float4*>(x + row * F + (k * LPR + lane) * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < CH; ++k) { acc.x += v[u][k].x; acc.y += v[u][k].y; acc.z += v[u][k].z; acc.w += v[u][k].w; }
    }
    if (acc.x == 12345.678f) sink[0] = acc.x + acc.y + acc.z + acc.w;
}

// `only` = alternatives separated by '|': a variant runs if its name contains one of them
static bool wanted(const char* only, const char* name)
{
    if (!only) return true;
    std::string o(only);
    size_t p = 0;
    while (p <= o.size()) {
        size_t q = o.find('|', p);
        if (q == std::string::npos) q = o.size();
        if (q > p && std::strstr(name, o.substr(p, q - p).c_str())) return true;
        p = q + 1;
    }
    return false;
}

template <typename Fn>
static float time_ms(Fn fn, int reps)
{
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    fn();
    fn();
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) fn();
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    HIP_OK(hipGetLastError());
    return ms / reps;
}

struct Ctx {
    int64_t N, E;
    int F;
    Args a;
    float* ref;
    unsigned long long* d_cnt;
    double bytes;
    const char* only;
    int reps;
};

static void report(const Ctx& c, const char* name, float ms, long long mismatches)
{
    const int lines = (c.F * 4 + 127) / 128;
    std::printf("{\"probe\": \"wide_row_ab\", \"F\": %d, \"variant\": \"%s\", \"ms\": %.4f, \"frac_of_8TBps_alg\": %.4f, "
                "\"G_row_lines_per_s\": %.2f, \"mismatching_words_vs_shipped\": %lld}\n",
                c.F, name, ms, c.bytes / (ms * 1e-3) / 8e12, double(c.E + c.N) * lines / (ms * 1e-3) / 1e9, mismatches);
    std::fflush(stdout);
}

static long long diff(const Ctx& c)
{
    HIP_OK(hipMemset(c.d_cnt, 0, 8));
    k_diff<<<4096, 256>>>(reinterpret_cast<const uint32_t*>(c.ref), reinterpret_cast<const uint32_t*>(c.a.out), size_t(c.N) * c.F, c.d_cnt);
    unsigned long long h = 0;
    HIP_OK(hipMemcpy(&h, c.d_cnt, 8, hipMemcpyDeviceToHost));
    return (long long)h;
}

template <int G, int CH, int U, int MODE>
static void run_var(const Ctx& c, const char* name, int grid_cap = 1 << 20)
{
    if (!wanted(c.only, name)) return;
    constexpr int RPB = kBlock / G;
    const int cols = G * 4 * CH;
    const int ny = (MODE & M_YSPLIT) ? (c.F + cols - 1) / cols : 1;
    if (!(MODE & M_YSPLIT) && cols < c.F) return;
    if (cols >= 2 * c.F && G > 4) return;      // a narrower group covers this width
    const int grid = int(std::min<int64_t>((c.N + RPB - 1) / RPB, grid_cap));
    HIP_OK(hipMemset(c.a.out, 0xff, size_t(c.N) * c.F * 4));
    const float ms = time_ms([&] { reduce_var<G, CH, U, MODE><<<dim3(grid, ny), kBlock>>>(c.a); }, c.reps);
    report(c, name, ms, diff(c));
}

template <int LPR, int CH, int U>
static void run_probe(const Ctx& c, const char* name, float* sink)
{
    if (!wanted(c.only, name)) return;
    if (LPR * 4 * CH != c.F) return;
    const int64_t total_rows = c.E + c.N;
    const int grid = 256 * 32;
    const int64_t groups = int64_t(grid) * kBlock / LPR;
    const int64_t rpg = ((total_rows / groups + U - 1) / U) * U;
    const float ms = time_ms([&] { probe<LPR, CH, U><<<grid, kBlock>>>(c.a.x, c.N, c.F, rpg, sink); }, c.reps);
    const int lines = c.F * 4 / 128;
    std::printf("{\"probe\": \"wide_row_ab\", \"F\": %d, \"variant\": \"%s\", \"ms\": %.4f, \"rows\": %lld, \"G_row_lines_per_s\": %.2f, "
                "\"fetched_TBps\": %.3f}\n", c.F, name, ms, (long long)(rpg * groups), double(rpg * groups) * lines / (ms * 1e-3) / 1e9,
                double(rpg * groups) * c.F * 4 / (ms * 1e-3) / 1e12);
    std::fflush(stdout);
}

template <int G, int U, int MAP>
static void run_blk(const Ctx& c, const char* name)
{
    if (!wanted(c.only, name)) return;
    const int cols = G * 4;
    const int nb = (c.F + cols - 1) / cols;
    if (nb < 2) return;
    int rpb = kBlock / G;
    dim3 grid;
    if (MAP == 0) grid = dim3(unsigned((c.N + rpb - 1) / rpb), unsigned(nb));
    else if (MAP == 1) {
        if (nb != 2 && nb != 4) return;
        rpb = (4 / nb) * (64 / G);
        grid = dim3(unsigned((c.N + rpb - 1) / rpb));
    } else {
        int64_t rbs = (c.N + rpb - 1) / rpb;
        rbs = (rbs + 7) / 8 * 8;
        grid = dim3(unsigned(rbs * nb));
    }
    HIP_OK(hipMemset(c.a.out, 0xff, size_t(c.N) * c.F * 4));
    const float ms = time_ms([&] { reduce_blk<G, U, MAP><<<grid, kBlock>>>(c.a, nb); }, c.reps);
    report(c, name, ms, diff(c));
}

template <int LPR, int U>
static void run_piece(const Ctx& c, const char* name, float* sink)
{
    if (!wanted(c.only, name)) return;
    if (c.F % (LPR * 4) != 0) return;
    const int nb = c.F / (LPR * 4);
    const int64_t pieces = (c.E + c.N) * nb;
    const int grid = 256 * 32;
    const int64_t groups = int64_t(grid) * kBlock / LPR;
    const int64_t per = ((pieces / groups + U - 1) / U) * U;
    const float ms = time_ms([&] { probe_piece<LPR, U><<<grid, kBlock>>>(c.a.x, c.N, c.F, per, sink); }, c.reps);
    const double lines = double(per) * groups * (LPR * 16 / 128.0);
    std::printf("{\"probe\": \"wide_row_ab\", \"F\": %d, \"variant\": \"%s\", \"ms\": %.4f, \"pieces\": %lld, \"G_row_lines_per_s\": %.2f, "
                "\"fetched_TBps\": %.3f, \"ms_scaled_to_the_reduce\": %.3f}\n", c.F, name, ms, (long long)(per * groups),
                lines / (ms * 1e-3) / 1e9, lines * 128 / (ms * 1e-3) / 1e12, ms);
    std::fflush(stdout);
}

int main(int argc, char** argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 2400000;
    const int64_t E = argc > 2 ? atoll(argv[2]) : 123000000;
    std::vector<int> widths;
    {
        std::string s = argc > 3 ? argv[3] : "128,192,256,384,512";
        size_t p = 0;
        while (p < s.size()) { widths.push_back(atoi(s.c_str() + p)); p = s.find(',', p); if (p == std::string::npos) break; ++p; }
    }
    const char* only = (argc > 4 && std::strlen(argv[4]) > 0 && std::strcmp(argv[4], "all") != 0) ? argv[4] : nullptr;
    const int reps = argc > 5 ? atoi(argv[5]) : 6;
    const int fmax = *std::max_element(widths.begin(), widths.end());

    int *d_rp, *d_col;
    float *d_w, *d_x, *d_sc, *d_out, *d_ref, *d_sink;
    unsigned long long* d_cnt;
    HIP_OK(hipMalloc(&d_rp, (N + 1) * 4));
    HIP_OK(hipMalloc(&d_col, E * 4));
    HIP_OK(hipMalloc(&d_w, E * 4));
    HIP_OK(hipMalloc(&d_sc, N * 4));
    HIP_OK(hipMalloc(&d_x, size_t(N) * fmax * 4));
    HIP_OK(hipMalloc(&d_out, size_t(N) * fmax * 4));
    HIP_OK(hipMalloc(&d_ref, size_t(N) * fmax * 4));
    HIP_OK(hipMalloc(&d_sink, 64));
    HIP_OK(hipMalloc(&d_cnt, 8));
    HIP_OK(hipMemset(d_rp, 0, (N + 1) * 4));
    k_deg<<<8192, 256>>>(d_rp + 1, E, N);
    k_edges<<<8192, 256>>>(d_col, d_w, E, N);
    k_self<<<4096, 256>>>(d_sc, N);
    std::vector<int> rp(N + 1);
    HIP_OK(hipMemcpy(rp.data(), d_rp, (N + 1) * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < N; ++i) rp[i + 1] += rp[i];
    HIP_OK(hipMemcpy(d_rp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice));

    for (int F : widths) {
        k_fill_x<<<8192, 256>>>(d_x, size_t(N) * F);
        HIP_OK(hipDeviceSynchronize());
        Ctx c;
        c.N = N; c.E = E; c.F = F; c.ref = d_ref; c.d_cnt = d_cnt; c.only = only; c.reps = reps;
        c.a = Args{d_rp, d_col, d_w, d_x, d_sc, d_out, N, F};
        c.bytes = double(E + N) * (4.0 * F + 8) + double(N) * 4 * F + 4.0 * (N + 1);
        // baseline: the shipped kernel through the C ABI
        tfgx_reduce_args p;
        std::memset(&p, 0, sizeof(p));
        p.row_begin = d_rp; p.row_end = d_rp + 1; p.rp_stride = 1; p.col = d_col; p.w = d_w; p.n_dst = N;
        p.x = d_x; p.ldx = F; p.F = F; p.out = d_ref; p.ldo = F; p.op = TFGX_SUM; p.act = TFGX_ACT_NONE; p.self_coef = d_sc;
        char kname[256];
        tfgx_segment_reduce_describe(&p, kname, sizeof(kname));
        if (wanted(only, "shipped")) {
            const float ms = time_ms([&] {
                if (tfgx_segment_reduce_f32(&p, nullptr) != 0) { std::fprintf(stderr, "%s\n", tfgx_last_error()); std::exit(3); }
            }, reps);
            std::string nm = std::string("shipped ") + kname;
            Ctx c0 = c;
            report(c0, nm.c_str(), ms, 0);
        } else {
            tfgx_segment_reduce_f32(&p, nullptr);
            HIP_OK(hipDeviceSynchronize());
        }
        // replicas of the shipped walk (sanity: should time like "shipped")
        run_var<32, 1, 8, 0>(c, "replica G32 CH1 U8");
        run_var<64, 1, 8, 0>(c, "replica G64 CH1 U8");
        run_var<64, 2, 4, 0>(c, "replica G64 CH2 U4 cap4096", 4096);
        // masked tail / early self
        run_var<32, 1, 8, M_MASK>(c, "G32 CH1 U8 mask");
        run_var<32, 1, 8, M_MASK | M_EARLY>(c, "G32 CH1 U8 mask early");
        run_var<64, 1, 8, M_MASK>(c, "G64 CH1 U8 mask");
        run_var<64, 1, 8, M_EARLY>(c, "G64 CH1 U8 early");
        run_var<64, 1, 8, M_MASK | M_EARLY>(c, "G64 CH1 U8 mask early");
        run_var<64, 1, 8, M_MASK | M_EARLY | M_ROT>(c, "G64 CH1 U8 mask early rot");
        run_var<32, 1, 8, M_MASK | M_EARLY | M_ROT>(c, "G32 CH1 U8 mask early rot");
        run_var<64, 1, 8, M_ROT>(c, "G64 CH1 U8 rot");
        run_var<64, 1, 8, M_MASK | M_EARLY | M_ROT | M_NT>(c, "G64 CH1 U8 mask early rot nt");
        run_var<64, 1, 16, M_MASK | M_EARLY | M_ROT>(c, "G64 CH1 U16 mask early rot");
        run_var<32, 2, 4, M_MASK | M_EARLY | M_ROT>(c, "G32 CH2 U4 mask early rot");
        run_var<64, 2, 4, M_MASK | M_EARLY | M_ROT>(c, "G64 CH2 U4 mask early rot nocap");
        run_var<64, 2, 4, M_MASK | M_EARLY | M_ROT>(c, "G64 CH2 U4 mask early rot cap4096", 4096);
        run_var<64, 2, 8, M_MASK | M_EARLY | M_ROT>(c, "G64 CH2 U8 mask early rot nocap");
        run_var<64, 1, 8, M_MASK | M_EARLY | M_ROT | M_YSPLIT>(c, "G64 CH1 U8 mask early rot ysplit");
        run_var<64, 1, 8, M_MASK | M_EARLY | M_NT>(c, "G64 CH1 U8 mask early nt");
        run_var<64, 1, 4, M_MASK | M_EARLY>(c, "G64 CH1 U4 mask early");
        run_var<64, 1, 16, M_MASK | M_EARLY>(c, "G64 CH1 U16 mask early");
        // two rows per wave at F = 256
        run_var<32, 2, 4, M_MASK | M_EARLY>(c, "G32 CH2 U4 mask early");
        run_var<32, 2, 8, M_MASK | M_EARLY>(c, "G32 CH2 U8 mask early");
        run_var<16, 4, 2, M_MASK | M_EARLY>(c, "G16 CH4 U2 mask early");
        // F = 384 / 512: two chunks per lane, or column blocks on grid.y
        run_var<64, 2, 4, M_MASK | M_EARLY>(c, "G64 CH2 U4 mask early cap4096", 4096);
        run_var<64, 2, 4, M_MASK | M_EARLY>(c, "G64 CH2 U4 mask early nocap");
        run_var<64, 2, 8, M_MASK | M_EARLY>(c, "G64 CH2 U8 mask early nocap");
        run_var<64, 2, 2, M_MASK | M_EARLY>(c, "G64 CH2 U2 mask early nocap");
        run_var<64, 1, 8, M_MASK | M_EARLY | M_YSPLIT>(c, "G64 CH1 U8 mask early ysplit");
        run_var<32, 1, 8, M_MASK | M_EARLY | M_YSPLIT>(c, "G32 CH1 U8 mask early ysplit");
        run_var<32, 4, 2, M_MASK | M_EARLY>(c, "G32 CH4 U2 mask early");
        run_var<32, 4, 4, M_MASK | M_EARLY>(c, "G32 CH4 U4 mask early");
        // column blocks (the finding of call 1: 512-byte pieces beat 1 KB / 2 KB pieces): piece size, batch depth, placement
        run_blk<32, 8, 0>(c, "blk G32 U8 grid.y");
        run_blk<32, 4, 0>(c, "blk G32 U4 grid.y");
        run_blk<32, 16, 0>(c, "blk G32 U16 grid.y");
        run_blk<16, 8, 0>(c, "blk G16 U8 grid.y");
        run_blk<16, 4, 0>(c, "blk G16 U4 grid.y");
        run_blk<16, 16, 0>(c, "blk G16 U16 grid.y");
        run_blk<8, 8, 0>(c, "blk G8 U8 grid.y");
        run_blk<32, 8, 1>(c, "blk G32 U8 in-workgroup");
        run_blk<16, 8, 1>(c, "blk G16 U8 in-workgroup");
        run_blk<32, 8, 2>(c, "blk G32 U8 xcd-siblings");
        run_blk<16, 8, 2>(c, "blk G16 U8 xcd-siblings");
        run_blk<16, 4, 2>(c, "blk G16 U4 xcd-siblings");
        run_piece<8, 8>(c, "piece probe 128B U8", d_sink);
        run_piece<16, 8>(c, "piece probe 256B U8", d_sink);
        run_piece<16, 4>(c, "piece probe 256B U4", d_sink);
        run_piece<32, 8>(c, "piece probe 512B U8", d_sink);
        run_piece<32, 4>(c, "piece probe 512B U4", d_sink);
        run_piece<64, 8>(c, "piece probe 1KB U8", d_sink);
        run_piece<64, 4>(c, "piece probe 1KB U4", d_sink);
        run_piece<64, 2>(c, "piece probe 1KB U2", d_sink);
        // the gather alone
        run_probe<32, 1, 8>(c, "probe 512B rows U8", d_sink);
        run_probe<64, 1, 8>(c, "probe 1KB rows U8", d_sink);
        run_probe<64, 1, 16>(c, "probe 1KB rows U16", d_sink);
        run_probe<32, 2, 4>(c, "probe 1KB rows as 2x512B per half wave U4", d_sink);
        run_probe<64, 2, 4>(c, "probe 2KB rows U4", d_sink);
        run_probe<64, 2, 8>(c, "probe 2KB rows U8", d_sink);
    }
    return 0;
}
