// CSR-by-destination plan build + small edge-attribute helpers.
//
// Replaces the per-call scatter order of tf.math.unsorted_segment_* and the TF1 path's
// tf.argsort + gathers (reference: tf_geometric/nn/kernel/segment.py:7-11) with a plan built once
// per graph: a stable LSD radix sort of (row, edge id) (hipcub::DeviceRadixSort, plumbing only),
// then row_ptr from the sorted keys' boundaries (no atomics, no scan).
#include "tfgx_common.h"
#include <hipcub/hipcub.hpp>

namespace tfgx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

inline int key_bits(int64_t n)
{
    int b = 1;
    while (b < 31 && (int64_t(1) << b) < n) ++b;
    return b;
}

__global__ void validate_and_iota(const int32_t* __restrict__ row, const int32_t* __restrict__ col, int64_t E,
                                  int32_t n_dst, int32_t n_src, int32_t* __restrict__ iota,
                                  int32_t* __restrict__ bad)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    int any_bad = 0;
    for (; i < E; i += stride) {
        const int32_t r = row[i], c = col[i];
        any_bad |= (r < 0) | (r >= n_dst) | (c < 0) | (c >= n_src);
        iota[i] = static_cast<int32_t>(i);
    }
    if (__any(any_bad) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

// keys sorted ascending; row_ptr[k] = first position whose key >= k
__global__ void row_ptr_from_sorted(const int32_t* __restrict__ keys, int64_t E, int32_t n_dst,
                                    int32_t* __restrict__ row_ptr)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; i <= E; i += stride) {
        const int32_t lo = (i == 0) ? 0 : keys[i - 1] + 1;
        const int32_t hi = (i == E) ? n_dst : keys[i];
        for (int32_t k = lo; k <= hi; ++k) row_ptr[k] = static_cast<int32_t>(i);
    }
}

__global__ void gather_i32(const int32_t* __restrict__ src, const int32_t* __restrict__ idx, int64_t n,
                           int32_t* __restrict__ dst)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[idx[i]];
}

__global__ void permute_rows(const float* __restrict__ src, const int32_t* __restrict__ perm, int64_t E,
                             int64_t width, float* __restrict__ dst)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const int64_t total = E * width;
    for (; i < total; i += stride) {
        const int64_t e = i / width, j = i - e * width;
        dst[i] = src[int64_t(perm[e]) * width + j];
    }
}

}  // namespace
}  // namespace tfgx

using namespace tfgx;

extern "C" int tfgx_version(void) { return 100; }

extern "C" const char* tfgx_last_error(void) { return g_err; }

extern "C" size_t tfgx_csr_plan_workspace_bytes(int64_t n_dst, int64_t E)
{
    if (E < 0 || n_dst < 0) return 0;
    size_t temp = 0;
    const int32_t* kin = nullptr;
    int32_t* kout = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, kin, kout, kin, kout, static_cast<int>(E > 0 ? E : 1), 0,
                                       key_bits(n_dst));
    // [bad flag | keys_out | iota | hipcub temp]
    return 256 + 2 * align_up(sizeof(int32_t) * size_t(E > 0 ? E : 1)) + align_up(temp) + 256;
}

extern "C" int tfgx_build_csr_by_dst(const int32_t* row, const int32_t* col, int64_t E, int64_t n_dst,
                                     int64_t n_src, int32_t* row_ptr, int32_t* col_sorted, int32_t* perm,
                                     void* workspace, size_t workspace_bytes, tfgx_stream_t stream_)
{
    hipStream_t stream = as_stream(stream_);
    TFGX_REQUIRE(E >= 0 && n_dst >= 0 && n_src >= 0, "negative size");
    TFGX_REQUIRE(E < (int64_t(1) << 31) - 1 && n_dst < (int64_t(1) << 31) - 1 && n_src < (int64_t(1) << 31) - 1,
                 "sizes must fit int32");
    TFGX_REQUIRE(row_ptr != nullptr, "row_ptr is null");
    if (E == 0) {
        TFGX_HIP_CHECK(hipMemsetAsync(row_ptr, 0, sizeof(int32_t) * size_t(n_dst + 1), stream));
        return TFGX_OK;
    }
    TFGX_REQUIRE(row && col && col_sorted && perm && workspace, "null pointer");
    if (workspace_bytes < tfgx_csr_plan_workspace_bytes(n_dst, E)) {
        set_error("tfgx_build_csr_by_dst: workspace too small (%zu < %zu)", workspace_bytes,
                  tfgx_csr_plan_workspace_bytes(n_dst, E));
        return TFGX_ERR_WORKSPACE;
    }
    char* ws = static_cast<char*>(workspace);
    int32_t* bad = reinterpret_cast<int32_t*>(ws);
    int32_t* keys_out = reinterpret_cast<int32_t*>(ws + 256);
    int32_t* iota = reinterpret_cast<int32_t*>(ws + 256 + align_up(sizeof(int32_t) * size_t(E)));
    void* temp = ws + 256 + 2 * align_up(sizeof(int32_t) * size_t(E));
    size_t temp_bytes = workspace_bytes - (256 + 2 * align_up(sizeof(int32_t) * size_t(E)));

    TFGX_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int32_t), stream));
    validate_and_iota<<<grid_for(E, kBlock), kBlock, 0, stream>>>(row, col, E, int32_t(n_dst), int32_t(n_src),
                                                                   iota, bad);
    TFGX_LAUNCH_CHECK("validate_and_iota");
    int32_t bad_host = 0;
    TFGX_HIP_CHECK(hipMemcpyAsync(&bad_host, bad, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    TFGX_HIP_CHECK(hipStreamSynchronize(stream));
    if (bad_host) {
        set_error("tfgx_build_csr_by_dst: edge endpoint outside [0, %lld) x [0, %lld)", (long long)n_dst,
                  (long long)n_src);
        return TFGX_ERR_INDEX;
    }
    TFGX_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, row, keys_out, iota, perm,
                                                      static_cast<int>(E), 0, key_bits(n_dst), stream));
    gather_i32<<<grid_for(E, kBlock), kBlock, 0, stream>>>(col, perm, E, col_sorted);
    TFGX_LAUNCH_CHECK("gather_i32");
    row_ptr_from_sorted<<<grid_for(E + 1, kBlock), kBlock, 0, stream>>>(keys_out, E, int32_t(n_dst), row_ptr);
    TFGX_LAUNCH_CHECK("row_ptr_from_sorted");
    return TFGX_OK;
}

extern "C" int tfgx_permute_rows_f32(const float* src, const int32_t* perm, int64_t E, int64_t width, float* dst,
                                     tfgx_stream_t stream)
{
    TFGX_REQUIRE(E >= 0 && width >= 1, "bad size");
    if (E == 0) return TFGX_OK;
    TFGX_REQUIRE(src && perm && dst, "null pointer");
    permute_rows<<<grid_for(E * width, kBlock), kBlock, 0, as_stream(stream)>>>(src, perm, E, width, dst);
    TFGX_LAUNCH_CHECK("permute_rows");
    return TFGX_OK;
}
