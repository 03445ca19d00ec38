// Segmented top-k selection: tfgx_segment_topk.
//
// Replaces topk_pool (tf_geometric/nn/pool/topk_pool.py:6-87), which scatters the scores into a dense
// [num_sources, max_targets_per_source] matrix padded with min_score - 1, argsorts every row and masks the first
// node_k columns.  Here: ONE stable radix sort of 64-bit keys (source id << 32 | descending-order score bits) with the
// caller's position as payload, segment starts from the sorted keys' boundaries, node_k per source, an exclusive scan
// for the output offsets and a masked copy.  O(n) memory instead of O(num_sources * max_targets), and the output
// order is the reference's: sources ascending, scores descending, equal scores in the caller's order (tf.argsort
// DESCENDING is top_k underneath: the lower index wins a tie; the sort by source id is the same top_k on negated ids).
#include "tfgx_common.h"
#include <hipcub/hipcub.hpp>

namespace tfgx {
namespace {

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

inline int id_bits(int64_t n)
{
    int b = 1;
    while (b < 31 && (int64_t(1) << b) < n) ++b;
    return b;
}

// ascending unsigned order of the result == DESCENDING float order; -0.0 and +0.0 compare equal (as in top_k)
__device__ __forceinline__ uint32_t descending_bits(float f)
{
    uint32_t u = __float_as_uint(f == 0.0f ? 0.0f : f);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;   // order-preserving map of IEEE-754 onto unsigned
    return ~u;
}

__global__ void topk_make_keys(const int32_t* __restrict__ seg, const float* __restrict__ score, int64_t n,
                               int32_t num_segments, uint64_t* __restrict__ keys, int32_t* __restrict__ vals,
                               int32_t* __restrict__ bad)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    int any_bad = 0;
    for (; i < n; i += stride) {
        const int32_t s = seg[i];
        any_bad |= (s < 0) | (s >= num_segments);
        keys[i] = (uint64_t(uint32_t(s)) << 32) | descending_bits(score[i]);
        vals[i] = static_cast<int32_t>(i);
    }
    if (__any(any_bad) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

// start[s] = first sorted position whose segment id >= s, for s in [0, num_segments]
__global__ void topk_segment_starts(const uint64_t* __restrict__ keys_s, int64_t n, int32_t num_segments,
                                    int32_t* __restrict__ start)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; i <= n; i += stride) {
        const int32_t lo = (i == 0) ? 0 : int32_t(keys_s[i - 1] >> 32) + 1;
        const int32_t hi = (i == n) ? num_segments : int32_t(keys_s[i] >> 32);
        for (int32_t s = lo; s <= hi; ++s) start[s] = static_cast<int32_t>(i);
    }
}

// node_k (topk_pool.py:62-71): min(k, count), or ceil(float32(count) * float32(ratio)) — clamped to count, where the
// reference would start selecting its padding columns for ratio > 1
__global__ void topk_counts(const int32_t* __restrict__ start, int32_t num_segments, int32_t k, float ratio,
                            int32_t* __restrict__ node_k)
{
    int64_t s = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; s <= num_segments; s += stride) {
        int32_t v = 0;
        if (s < num_segments) {
            const int32_t cnt = start[s + 1] - start[s];
            v = (k >= 0) ? min(k, cnt) : min(cnt, int32_t(ceilf(float(cnt) * ratio)));
        }
        node_k[s] = v;   // node_k[num_segments] = 0: the exclusive scan leaves the total there
    }
}

__global__ void topk_emit(const uint64_t* __restrict__ keys_s, const int32_t* __restrict__ vals_s, int64_t n,
                          const int32_t* __restrict__ start, const int32_t* __restrict__ node_k,
                          const int32_t* __restrict__ offset, int32_t num_segments, int32_t* __restrict__ out_index,
                          int32_t* __restrict__ out_count)
{
    int64_t p = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    if (p == 0) *out_count = offset[num_segments];
    for (; p < n; p += stride) {
        const int32_t s = int32_t(keys_s[p] >> 32);
        const int32_t r = int32_t(p) - start[s];
        if (r < node_k[s]) out_index[offset[s] + r] = vals_s[p];
    }
}

struct TopkLayout {
    size_t keys, keys_s, vals, vals_s, start, node_k, offset, bad, temp, total;
};

TopkLayout topk_layout(int64_t n, int64_t num_segments)
{
    const size_t nn = size_t(n > 0 ? n : 1), ns = size_t(num_segments + 1);
    size_t t1 = 0, t2 = 0;
    const uint64_t* k = nullptr;
    uint64_t* ko = nullptr;
    const int32_t* v = nullptr;
    int32_t* vo = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, k, ko, v, vo, static_cast<int>(nn), 0,
                                             32 + id_bits(num_segments));
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t2, v, vo, static_cast<int>(ns));
    TopkLayout L;
    size_t o = 0;
    L.keys = o, o += align_up(8 * nn);
    L.keys_s = o, o += align_up(8 * nn);
    L.vals = o, o += align_up(4 * nn);
    L.vals_s = o, o += align_up(4 * nn);
    L.start = o, o += align_up(4 * ns);
    L.node_k = o, o += align_up(4 * ns);
    L.offset = o, o += align_up(4 * ns);
    L.bad = o, o += 256;
    L.temp = o, o += align_up(t1 > t2 ? t1 : t2) + 256;
    L.total = o;
    return L;
}

}  // namespace
}  // namespace tfgx

using namespace tfgx;

extern "C" size_t tfgx_segment_topk_workspace_bytes(int64_t n, int64_t num_segments)
{
    if (n < 0 || num_segments < 0) return 0;
    return topk_layout(n, num_segments).total;
}

extern "C" int tfgx_segment_topk(const int32_t* segment, const float* score, int64_t n, int64_t num_segments,
                                 int32_t k, float ratio, int32_t* out_index, int32_t* out_count, void* workspace,
                                 size_t workspace_bytes, tfgx_stream_t stream_)
{
    TFGX_RANGE();
    hipStream_t stream = as_stream(stream_);
    TFGX_REQUIRE(n >= 0 && num_segments >= 0, "negative size");
    TFGX_REQUIRE(n < (int64_t(1) << 31) - 1 && num_segments < (int64_t(1) << 31) - 1, "sizes must fit int32");
    TFGX_REQUIRE(k >= 0 || ratio >= 0.0f, "give k >= 0, or k < 0 and ratio >= 0");
    TFGX_REQUIRE(out_count != nullptr, "out_count is null");
    if (n == 0) {
        TFGX_HIP_CHECK(hipMemsetAsync(out_count, 0, sizeof(int32_t), stream));
        return TFGX_OK;
    }
    TFGX_REQUIRE(segment && score && out_index && workspace, "null pointer");
    const TopkLayout lay = topk_layout(n, num_segments);
    if (workspace_bytes < lay.total) {
        set_error("tfgx_segment_topk: workspace too small (%zu < %zu)", workspace_bytes, lay.total);
        return TFGX_ERR_WORKSPACE;
    }
    char* ws = static_cast<char*>(workspace);
    uint64_t* keys = reinterpret_cast<uint64_t*>(ws + lay.keys);
    uint64_t* keys_s = reinterpret_cast<uint64_t*>(ws + lay.keys_s);
    int32_t* vals = reinterpret_cast<int32_t*>(ws + lay.vals);
    int32_t* vals_s = reinterpret_cast<int32_t*>(ws + lay.vals_s);
    int32_t* start = reinterpret_cast<int32_t*>(ws + lay.start);
    int32_t* node_k = reinterpret_cast<int32_t*>(ws + lay.node_k);
    int32_t* offset = reinterpret_cast<int32_t*>(ws + lay.offset);
    int32_t* bad = reinterpret_cast<int32_t*>(ws + lay.bad);
    void* temp = ws + lay.temp;
    const size_t temp_bytes = workspace_bytes - lay.temp;
    const int32_t ns = int32_t(num_segments);

    TFGX_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int32_t), stream));
    topk_make_keys<<<grid_for(n, kBlock), kBlock, 0, stream>>>(segment, score, n, ns, keys, vals, bad);
    TFGX_LAUNCH_CHECK("topk_make_keys");
    int32_t bad_host = 0;
    TFGX_HIP_CHECK(hipMemcpyAsync(&bad_host, bad, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    TFGX_HIP_CHECK(hipStreamSynchronize(stream));
    if (bad_host) {
        set_error("tfgx_segment_topk: segment id outside [0, %lld)", (long long)num_segments);
        return TFGX_ERR_INDEX;
    }
    size_t tb = temp_bytes;
    TFGX_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(temp, tb, keys, keys_s, vals, vals_s, static_cast<int>(n), 0,
                                                      32 + id_bits(num_segments), stream));
    topk_segment_starts<<<grid_for(n + 1, kBlock), kBlock, 0, stream>>>(keys_s, n, ns, start);
    TFGX_LAUNCH_CHECK("topk_segment_starts");
    topk_counts<<<grid_for(num_segments + 1, kBlock), kBlock, 0, stream>>>(start, ns, k, ratio, node_k);
    TFGX_LAUNCH_CHECK("topk_counts");
    tb = temp_bytes;
    TFGX_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, tb, node_k, offset, static_cast<int>(num_segments + 1),
                                                    stream));
    topk_emit<<<grid_for(n, kBlock), kBlock, 0, stream>>>(keys_s, vals_s, n, start, node_k, offset, ns, out_index,
                                                          out_count);
    TFGX_LAUNCH_CHECK("topk_emit");
    return TFGX_OK;
}
