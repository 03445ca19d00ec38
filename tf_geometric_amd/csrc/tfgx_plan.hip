// CSR-by-destination plan build + small edge-attribute helpers.
//
// Replaces the per-call scatter order of tf.math.unsorted_segment_* and the TF1 path's
// tf.argsort + gathers (reference: tf_geometric/nn/kernel/segment.py:7-11) with a plan built once
// per graph: a stable LSD radix sort of (row, edge id) (hipcub::DeviceRadixSort, plumbing only),
// then row_ptr from the sorted keys' boundaries (no atomics, no scan).
#include "tfgx_common.h"
#include <hipcub/hipcub.hpp>
#include <dlfcn.h>
#include <cstdlib>

namespace tfgx {

namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)();
struct RoctxApi {
    roctx_push_fn push = nullptr;
    roctx_pop_fn pop = nullptr;
    RoctxApi()
    {
        const char* on = getenv("TFGX_ROCTX");
        if (on == nullptr || on[0] == '\0' || on[0] == '0') return;
        void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (h == nullptr) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (h == nullptr) return;
        push = reinterpret_cast<roctx_push_fn>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<roctx_pop_fn>(dlsym(h, "roctxRangePop"));
        if (push == nullptr || pop == nullptr) push = nullptr, pop = nullptr;
    }
};
const RoctxApi& roctx_api()
{
    static const RoctxApi api;
    return api;
}
}  // namespace

void roctx_push(const char* name)
{
    const RoctxApi& a = roctx_api();
    if (a.push) a.push(name);
}
void roctx_pop()
{
    const RoctxApi& a = roctx_api();
    if (a.pop) a.pop();
}

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

// ---- duplicate-edge merge (tf.unique on n*row+col, first-occurrence order) -------------------------------
__global__ void edge_hash_iota(const int32_t* __restrict__ row, const int32_t* __restrict__ col, int64_t E, int64_t n,
                               int64_t* __restrict__ keys, int32_t* __restrict__ vals)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; i < E; i += stride) {
        keys[i] = n * int64_t(row[i]) + int64_t(col[i]);
        vals[i] = static_cast<int32_t>(i);
    }
}

__global__ void mark_heads(const int64_t* __restrict__ keys_s, int64_t E, int32_t* __restrict__ head)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; i < E; i += stride) head[i] = (i == 0 || keys_s[i] != keys_s[i - 1]) ? 1 : 0;
}

// gid_incl = inclusive scan of head (group id + 1 per sorted position)
__global__ void record_first_pos(const int32_t* __restrict__ head, const int32_t* __restrict__ gid_incl,
                                 const int32_t* __restrict__ vals_s, int64_t E, int32_t* __restrict__ first_pos,
                                 int32_t* __restrict__ mark)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; i < E; i += stride) {
        if (head[i]) {   // stable sort: the head of a group is its FIRST occurrence in the caller's order
            first_pos[gid_incl[i] - 1] = vals_s[i];
            mark[vals_s[i]] = 1;
        }
    }
}

__global__ void emit_unique(const int64_t* __restrict__ keys_s, const int32_t* __restrict__ vals_s,
                            const int32_t* __restrict__ head, const int32_t* __restrict__ gid_incl,
                            const int32_t* __restrict__ first_pos, const int32_t* __restrict__ posrank, int64_t E,
                            int64_t n, int32_t* __restrict__ out_row, int32_t* __restrict__ out_col,
                            int32_t* __restrict__ unique_index, int32_t* __restrict__ n_unique)
{
    int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (; i < E; i += stride) {
        const int32_t g = gid_incl[i] - 1;
        const int32_t r = posrank[first_pos[g]];      // rank of the group by first occurrence
        unique_index[vals_s[i]] = r;
        if (head[i]) {
            out_row[r] = static_cast<int32_t>(keys_s[i] / n);
            out_col[r] = static_cast<int32_t>(keys_s[i] % n);
        }
        if (i == E - 1) *n_unique = gid_incl[i];
    }
}

}  // namespace
}  // namespace tfgx

using namespace tfgx;

extern "C" int tfgx_version(void) { return TFGX_ABI_VERSION; }

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
    TFGX_RANGE();
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
    TFGX_RANGE();
    TFGX_REQUIRE(E >= 0 && width >= 1, "bad size");
    if (E == 0) return TFGX_OK;
    TFGX_REQUIRE(src && perm && dst, "null pointer");
    permute_rows<<<grid_for(E * width, kBlock), kBlock, 0, as_stream(stream)>>>(src, perm, E, width, dst);
    TFGX_LAUNCH_CHECK("permute_rows");
    return TFGX_OK;
}

static inline int hash_bits(int64_t n)
{
    int b = 1;
    while (b < 62 && (int64_t(1) << b) < n * n) ++b;
    return b;
}

extern "C" size_t tfgx_merge_edges_workspace_bytes(int64_t E, int64_t n)
{
    if (E <= 0) return 256;
    size_t t1 = 0, t2 = 0;
    const int64_t* k = nullptr;
    int64_t* ko = nullptr;
    const int32_t* v = nullptr;
    int32_t* vo = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, k, ko, v, vo, static_cast<int>(E), 0, hash_bits(n));
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, t2, v, vo, static_cast<int>(E));
    const size_t a8 = align_up(sizeof(int64_t) * size_t(E)), a4 = align_up(sizeof(int32_t) * size_t(E));
    return 2 * a8 + 7 * a4 + align_up(t1 > t2 ? t1 : t2) + 256;
}

extern "C" int tfgx_merge_duplicated_edges(const int32_t* row, const int32_t* col, int64_t E, int64_t n,
                                           int32_t* out_row, int32_t* out_col, int32_t* unique_index,
                                           int32_t* n_unique, void* workspace, size_t workspace_bytes,
                                           tfgx_stream_t stream_)
{
    TFGX_RANGE();
    hipStream_t stream = as_stream(stream_);
    TFGX_REQUIRE(E >= 0 && n >= 1 && n < (int64_t(1) << 31) && n_unique, "bad argument");
    if (E == 0) {
        TFGX_HIP_CHECK(hipMemsetAsync(n_unique, 0, sizeof(int32_t), stream));
        return TFGX_OK;
    }
    TFGX_REQUIRE(row && col && out_row && out_col && unique_index && workspace, "null pointer");
    if (workspace_bytes < tfgx_merge_edges_workspace_bytes(E, n)) {
        set_error("tfgx_merge_duplicated_edges: workspace too small");
        return TFGX_ERR_WORKSPACE;
    }
    const size_t a8 = align_up(sizeof(int64_t) * size_t(E)), a4 = align_up(sizeof(int32_t) * size_t(E));
    char* ws = static_cast<char*>(workspace);
    int64_t* keys = reinterpret_cast<int64_t*>(ws);
    int64_t* keys_s = reinterpret_cast<int64_t*>(ws + a8);
    int32_t* vals = reinterpret_cast<int32_t*>(ws + 2 * a8);
    int32_t* vals_s = reinterpret_cast<int32_t*>(ws + 2 * a8 + a4);
    int32_t* head = reinterpret_cast<int32_t*>(ws + 2 * a8 + 2 * a4);
    int32_t* gid = reinterpret_cast<int32_t*>(ws + 2 * a8 + 3 * a4);
    int32_t* first_pos = reinterpret_cast<int32_t*>(ws + 2 * a8 + 4 * a4);
    int32_t* mark = reinterpret_cast<int32_t*>(ws + 2 * a8 + 5 * a4);
    int32_t* posrank = reinterpret_cast<int32_t*>(ws + 2 * a8 + 6 * a4);
    void* temp = ws + 2 * a8 + 7 * a4;
    size_t temp_bytes = workspace_bytes - (2 * a8 + 7 * a4);
    const int g = grid_for(E, kBlock);
    edge_hash_iota<<<g, kBlock, 0, stream>>>(row, col, E, n, keys, vals);
    TFGX_LAUNCH_CHECK("edge_hash_iota");
    size_t tb = temp_bytes;
    TFGX_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(temp, tb, keys, keys_s, vals, vals_s, static_cast<int>(E), 0,
                                                      hash_bits(n), stream));
    mark_heads<<<g, kBlock, 0, stream>>>(keys_s, E, head);
    TFGX_LAUNCH_CHECK("mark_heads");
    tb = temp_bytes;
    TFGX_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(temp, tb, head, gid, static_cast<int>(E), stream));
    TFGX_HIP_CHECK(hipMemsetAsync(mark, 0, sizeof(int32_t) * size_t(E), stream));
    record_first_pos<<<g, kBlock, 0, stream>>>(head, gid, vals_s, E, first_pos, mark);
    TFGX_LAUNCH_CHECK("record_first_pos");
    tb = temp_bytes;
    TFGX_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, tb, mark, posrank, static_cast<int>(E), stream));
    emit_unique<<<g, kBlock, 0, stream>>>(keys_s, vals_s, head, gid, first_pos, posrank, E, n, out_row, out_col,
                                          unique_index, n_unique);
    TFGX_LAUNCH_CHECK("emit_unique");
    return TFGX_OK;
}
