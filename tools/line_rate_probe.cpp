// How many 128-byte lines per second can one MI355X fetch from HBM when the lines are addressed at random?
// The ceiling the gather kernels are measured against: a gathered row costs whole line requests.
//
// Kernel: a group of G = 8*L lanes fetches L consecutive lines (L*128 bytes, one dwordx4 per lane) starting at a random
// line-aligned (or 16-byte-aligned, --unaligned) offset of a table much larger than the caches, 8 independent fetches
// in flight per lane, and folds them into one float per group so nothing is optimised away.  Also times a plain
// sequential read of the same table.  Prints one JSON object per configuration.
//
//   hipcc --offload-arch=gfx950 -O3 tools/line_rate_probe.cpp -o tf_geometric_amd/lib/line_rate_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define HIP_OK(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));   \
            std::exit(2);                                                     \
        }                                                                     \
    } while (0)

template <int G>   // lanes per fetched span; each lane loads 16 bytes
__global__ __launch_bounds__(256) void gather_spans(const float* __restrict__ table, const uint32_t* __restrict__ start16,
                                                    int64_t n_spans, int span_floats, float* __restrict__ out)
{
    constexpr int U = 8;
    const int lane = threadIdx.x % G;
    const int loff = (lane * 4) % span_floats;   // lanes past the span re-read its first bytes (same lines)
    const int64_t group = (int64_t(blockIdx.x) * 256 + threadIdx.x) / G;
    const int64_t n_groups = int64_t(gridDim.x) * 256 / G;
    float acc = 0.0f;
    for (int64_t i = group * U; i < n_spans; i += n_groups * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + u < n_spans ? i + u : n_spans - 1;
            v[u] = *reinterpret_cast<const float4*>(table + int64_t(start16[j]) * 4 + loff);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123456.789f) out[group] = acc;   // never true for the probe's data; keeps the loads alive
}

__global__ __launch_bounds__(256) void stream_read(const float4* __restrict__ table, int64_t n4, float* __restrict__ out)
{
    constexpr int U = 8;   // 8 independent 16-byte loads in flight per lane, like the gather probe
    float acc = 0.0f;
    const int64_t stride = int64_t(gridDim.x) * 256;
    int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = table[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) {
        const float4 v = table[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123456.789f) out[threadIdx.x] = acc;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

template <int G>
static void run(const float* table, int64_t table_bytes, int lines, bool aligned, int64_t n_spans, float* out)
{
    const int64_t span_bytes = int64_t(lines) * 128;
    std::vector<uint32_t> h(n_spans);
    const int64_t slots = (table_bytes - span_bytes - 128) / (aligned ? 128 : 16);
    for (auto& s : h) s = uint32_t((rnd() % slots) * (aligned ? 8 : 1));   // in units of 16 bytes
    uint32_t* d;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&d), sizeof(uint32_t) * n_spans));
    HIP_OK(hipMemcpy(d, h.data(), sizeof(uint32_t) * n_spans, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    const int grid = 256 * 32;
    for (int it = 0; it < 2; ++it) gather_spans<G><<<grid, 256>>>(table, d, n_spans, lines * 32, out);
    HIP_OK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) gather_spans<G><<<grid, 256>>>(table, d, n_spans, lines * 32, out);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    // an unaligned span of L*128 bytes touches L+1 lines unless it happens to start on a line boundary (1 in 8)
    const double lines_touched = aligned ? lines : lines + 7.0 / 8.0;
    std::printf("{\"probe\": \"random_spans\", \"span_bytes\": %lld, \"aligned\": %s, \"spans\": %lld, \"ms\": %.4f, "
                "\"G_spans_per_s\": %.2f, \"G_lines_per_s\": %.2f, \"useful_TBps\": %.3f, \"fetched_TBps\": %.3f}\n",
                (long long)span_bytes, aligned ? "true" : "false", (long long)n_spans, ms, n_spans / ms / 1e6,
                n_spans * lines_touched / ms / 1e6, n_spans * double(span_bytes) / ms / 1e9,
                n_spans * lines_touched * 128.0 / ms / 1e9);
    std::fflush(stdout);
    HIP_OK(hipFree(d));
}

int main(int argc, char** argv)
{
    const int64_t table_bytes = int64_t(argc > 1 ? std::atoll(argv[1]) : 4096) << 20;   // MiB, default 4 GiB
    const bool quick = argc > 2 && std::strcmp(argv[2], "quick") == 0;    // bench.py: aligned spans only, a quarter of the spans
    const int64_t n_spans = quick ? (16 << 20) : (64 << 20);
    float *table, *out;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&table), table_bytes));
    HIP_OK(hipMemset(table, 0, table_bytes));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&out), sizeof(float) * 256 * 32 * 256));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) stream_read<<<256 * 32, 256>>>(reinterpret_cast<const float4*>(table), table_bytes / 16, out);
    HIP_OK(hipEventRecord(e0));
    for (int it = 0; it < 5; ++it) stream_read<<<256 * 32, 256>>>(reinterpret_cast<const float4*>(table), table_bytes / 16, out);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("{\"probe\": \"sequential_read\", \"bytes\": %lld, \"ms\": %.4f, \"TBps\": %.3f, \"G_lines_per_s\": %.2f}\n",
                (long long)table_bytes, ms / 5, table_bytes / (ms / 5) / 1e9, table_bytes / 128.0 / (ms / 5) / 1e6);
    for (int aligned = 1; aligned >= (quick ? 1 : 0); --aligned) {
        run<8>(table, table_bytes, 1, aligned, n_spans, out);
        run<16>(table, table_bytes, 2, aligned, n_spans, out);
        run<32>(table, table_bytes, 3, aligned, n_spans / 2, out);   // G = 32 lanes, lanes 24..31 re-read inside the span
        run<32>(table, table_bytes, 4, aligned, n_spans / 2, out);
    }
    return 0;
}
