// Weight gradient of the max-pool GraphSAGE aggregator's MLP WITHOUT the gradient of its hidden rows (round 6).
//
//   reference (nn/conv/graph_sage.py:228-287):  h = relu(x W_mlp + b_mlp)  [N, Fp];  red[r, j] = max_{e: row_e = r} h[col_e, j]
//   layer 0 of a model (x is data: no d/dx wanted) needs, of this part,
//       dW_mlp[k, j] = sum_c x[c, k] dh[c, j] [h[c, j] > 0],   db_mlp[j] = sum_c dh[c, j] [h[c, j] > 0],
//       dh[c, j]     = sum_{r: c wins (r, j)} g[r, j] / count[r, j]            (tf.math.unsorted_segment_max's registered gradient)
//   The composition materialises dh [N, Fp]: per-edge winner masks (E x Fp bits), a source-major gather of ~N Fp scattered
//   4-byte elements (one 128-byte line request each: 1.23 G requests at products shape, 24.8 ms), a ReLU-mask pass and the
//   reduction x^T dh — 39.8 of the layer's 94 ms.  Exchanging the sums,
//       dW_mlp[:, j] = sum_r  gn[r, j] * x[w(r, j), :],   gn = [red > 0] g / count,  w(r, j) = the source that attains the maximum,
//   the sum runs DESTINATION-major: a workgroup stages the x rows of a destination's in-edges in LDS (the forward's own gather:
//   one 400-byte burst per edge) and every (column, feature) accumulator lives in a register of ONE thread for the whole
//   launch — no atomics, a fixed summation order, one partial [F_in + 1, Fp] per workgroup folded by a second kernel.
//
//   pool_wgrad_kernel     one workgroup per CU, ONE THREAD PER COLUMN j, its F_in + 1 accumulators dW[0..F_in, j] (feature F_in is
//                         the constant 1: db) in registers.  The thread fetches its own (position, gn) pair straight from the
//                         tracked forward's packed (count << 16 | position) array — no broadcast of anything — and reads the
//                         winner's staged row X[p_j][0..F_in] from LDS with immediate offsets: 2 instructions per multiply-add.
//                         (ds_read_b128: 256 bytes per clock).  Rows are staged XS = KMAX + 4 floats apart with XS / 4 ODD: the 16
//                         lanes one LDS cycle serves read different rows at the same features, and p -> p XS / 4 mod 16 is a
//                         bijection, so up to 16 distinct winners sit in distinct bank groups (equal winners are a broadcast).  Per (row, chunk of <= 96 edges) the x rows of the next TWO chunks
//                         are in flight (ids one chunk further ahead) while this one is consumed — which holds only because no load
//                         of the loop is predicated (see load_ids): the exec-masked form drained vmcnt(0) in every step.  A column with tied maxima walks the chunk
//                         exactly (every tied edge receives g / count)
//   pool_wgrad_reduce     dW / db = the workgroups' partials added in workgroup order
#include "tfgx_common.h"

namespace tfgx {
namespace {

#ifndef TFGX_POOL_DEPTH
#define TFGX_POOL_DEPTH 2             // developer A/B: chunks of gathered x rows in flight (2 or 3)
#endif
#ifndef TFGX_POOL_LOAD_ORDER
#define TFGX_POOL_LOAD_ORDER 1        // developer A/B: 0 = the gathers of a step are issued before its id / scalar loads
#endif
#ifndef TFGX_POOL_EXPERIMENT
#define TFGX_POOL_EXPERIMENT 0      // developer A/B, TIMING ONLY (wrong results): 1 = no LDS reads / FMAs, 2 = no gather of x rows
#endif
constexpr int kPoolChunk = 96;      // edges staged at a time, at most (products shape: in-degree 51 +- 7)
#ifndef TFGX_POOL_SLOTS
#define TFGX_POOL_SLOTS 4             // developer A/B
#endif
constexpr int kPoolSlots = TFGX_POOL_SLOTS;       // 16-byte loads per thread and chunk: a chunk holds min(96, 4 * threads / (F_in / 4)) edges

struct PoolArgs {
    const int32_t* row_ptr;
    const int32_t* col;
    int64_t n_dst;
    const float* x; int64_t ldx; int32_t F_in;
    const float* h; int64_t ldh;
    const float* red; int64_t ldr;
    const int32_t* packed; int64_t ldp;
    const float* g; int64_t ldg;
    int32_t Fp;
    float* partial;          // [gridDim.x, F_in + 1, Fp]
    int64_t rows_per_block;
    int32_t chunk;           // edges per work item, at most
    const int4* items;       // work items (row, first CSR position, edges, offset of the chunk inside its row), block by block
    const int32_t* item_ptr; // [gridDim.x + 1] item range of every workgroup
};

// Work items of the main kernel: the rows of a workgroup's range with at least one in-edge, cut into chunks of at most `chunk`
// edges — written once by these two small kernels, so that the main loop reads its next items as plain 16-byte loads issued
// steps ahead instead of walking row_ptr (round 6: the scalar walk — LDS read, s_waitcnt, v_readfirstlane, a data-dependent
// loop over empty rows — sat in front of every step's loads).
__global__ void pool_item_count_kernel(const int32_t* __restrict__ row_ptr, int64_t n_dst, int64_t rows_per_block, int chunk,
                                       int32_t* __restrict__ counts)
{
    const int64_t b = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t r0 = b * rows_per_block;
    if (r0 >= n_dst) return;
    const int64_t r1 = min(n_dst, r0 + rows_per_block);
    int c = 0;
    for (int64_t r = r0; r < r1; ++r) c += (row_ptr[r + 1] - row_ptr[r] + chunk - 1) / chunk;
    counts[b] = c;
}

__global__ void pool_item_scan_kernel(const int32_t* __restrict__ counts, int n_blocks, int32_t* __restrict__ item_ptr)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < n_blocks; ++b) { item_ptr[b] = acc; acc += counts[b]; }
        item_ptr[n_blocks] = acc;
    }
}

__global__ void pool_item_fill_kernel(const int32_t* __restrict__ row_ptr, int64_t n_dst, int64_t rows_per_block, int chunk,
                                      const int32_t* __restrict__ item_ptr, int4* __restrict__ items)
{
    const int64_t b = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t r0 = b * rows_per_block;
    if (r0 >= n_dst) return;
    const int64_t r1 = min(n_dst, r0 + rows_per_block);
    int4* out = items + item_ptr[b];
    for (int64_t r = r0; r < r1; ++r) {
        const int s = row_ptr[r], e = row_ptr[r + 1];
        for (int p = s; p < e; p += chunk) *out++ = make_int4(int(r), p, min(chunk, e - p), p - s);
    }
}

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));   // (arrays of HIP's float4 struct were left in scratch once their loads lost their branches)

template <int KMAX>       // accumulators per thread: F_in + 1 <= KMAX
__global__ __launch_bounds__(512, 1) void pool_wgrad_kernel(const PoolArgs a)
{
    constexpr int XS = KMAX + 4;                 // floats between staged rows: 16-byte aligned, XS / 4 ODD (KMAX % 8 == 0) — see the header
    static_assert(KMAX % 8 == 0, "KMAX must be a multiple of 8");
    extern __shared__ float smem[];
    float* const Xs = smem;                                                   // [2][kPoolChunk][XS]
    int* const Cs = reinterpret_cast<int*>(smem + 2 * kPoolChunk * XS);       // [2][kPoolChunk] source ids of the staged rows
    const int tid = threadIdx.x, nthr = blockDim.x;                           // nthr == Fp: thread tid owns column tid
    const int j = tid;
    const int Q4 = a.F_in / 4;                      // 16-byte pieces per x row

    float2v acc[KMAX / 2];                       // pairs of accumulators: v_pk_fma_f32 does two multiply-adds per issue slot
#pragma unroll
    for (int k = 0; k < KMAX / 2; ++k) acc[k] = float2v{0.0f, 0.0f};

    // slot u of this thread: piece q (16 bytes) of edge e of the chunk
    int slot_e[kPoolSlots];
#pragma unroll
    for (int u = 0; u < kPoolSlots; ++u) slot_e[u] = (tid + u * nthr) / Q4;
    auto slot_q = [&](int u) { return tid + u * nthr - slot_e[u] * Q4; };

    {
        const int64_t r_end = a.n_dst;          // (item.row == r_end: no item)
        // a work item: edges [s, s + len) of row `row`, the chunk starting `off` edges into the row; row == r_end: none left
        struct Item { int64_t row; int s; int len; int off; };
        const int item_begin = a.item_ptr[blockIdx.x], item_end = a.item_ptr[blockIdx.x + 1];
        auto fetch = [&](int i) __attribute__((always_inline)) {                // the raw 16 bytes of item i (a load every thread issues: one request per wave)
            return (i < item_end) ? a.items[i] : make_int4(-1, 0, 0, 0);
        };
        auto decode = [&](const int4& v) __attribute__((always_inline)) {
            Item it;
            const int row = __builtin_amdgcn_readfirstlane(v.x);
            it.row = row < 0 ? r_end : int64_t(row);
            it.s = __builtin_amdgcn_readfirstlane(v.y);
            it.len = __builtin_amdgcn_readfirstlane(v.z);
            it.off = __builtin_amdgcn_readfirstlane(v.w);
            return it;
        };
        // NO LOAD OF THE LOOP IS PREDICATED.  A load under a per-lane branch (`if (slot < len) v = p[..]`) is an exec-masked block
        // of its own, and the compiler's wait-count pass then drains vmcnt(0) at the next use of ANY loaded value — the ISA of the
        // predicated form waited for the gathers of chunk i + 2 in the middle of step i, right after issuing them: one chunk in
        // flight, every step a full gather latency.  Slots past the chunk's end load the chunk's LAST edge again (same addresses
        // as a live slot: no new lines), rows past the end load row n_dst - 1; the values are discarded by selects.
        auto load_ids = [&](const Item& it, int (&ids)[kPoolSlots]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < kPoolSlots; ++u) ids[u] = a.col[it.s + max(min(slot_e[u], it.len - 1), 0)];
        };
        auto load_x = [&](const Item& it, const int (&ids)[kPoolSlots], float4v (&xv)[kPoolSlots]) __attribute__((always_inline)) {
            (void)it;
#pragma unroll
            for (int u = 0; u < kPoolSlots; ++u) {
                if (TFGX_POOL_EXPERIMENT != 2)
                    xv[u] = *reinterpret_cast<const float4v*>(a.x + uint64_t(uint32_t(ids[u])) * uint64_t(a.ldx) + 4 * slot_q(u));
                else
                    xv[u] = float4v{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto commit = [&](const Item& it, int buf, const int (&ids)[kPoolSlots], const float4v (&xv)[kPoolSlots]) __attribute__((always_inline)) {
            float* Xb = Xs + buf * kPoolChunk * XS;
            int* Cb = Cs + buf * kPoolChunk;
#pragma unroll
            for (int u = 0; u < kPoolSlots; ++u) {
                if (slot_e[u] < it.len) {
                    *reinterpret_cast<float4v*>(Xb + slot_e[u] * XS + 4 * slot_q(u)) = xv[u];
                    if (slot_q(u) == 0) {
                        Cb[slot_e[u]] = ids[u];
                        Xb[slot_e[u] * XS + a.F_in] = 1.0f;               // the constant feature: its accumulator is the bias gradient
                    }
                }
            }
        };
        // (position of the winner inside its row, gn) of this thread's column for row r, straight from the tracked forward's
        // packed array: gn = g where the maximum is unique and positive (ReLU under the max: a maximum of 0 passes nothing); a
        // tied maximum (count > 1) is walked exactly below (pos = -1, gn = g / count)
        struct Pair { int pos; float gn; float rv; };
        struct RawPair { uint32_t pk; float rv; float gv; bool row_ok; };
        // the row's three scalars are LOADED a step before they are used and DECODED at the use (decoding at the load made the
        // step wait for them at once, and with them for every load issued before)
        auto load_pair = [&](const Item& it) __attribute__((always_inline)) {
            RawPair o;
            const int64_t row = min(it.row, r_end - 1);                    // three unconditional loads (see above)
            o.pk = uint32_t(a.packed[row * a.ldp + j]);
            o.rv = a.red[row * a.ldr + j];
            o.gv = a.g[row * a.ldg + j];
            o.row_ok = it.row < r_end;
            return o;
        };
        auto decode_pair = [&](const RawPair& w) __attribute__((always_inline)) {
            Pair o;
            const uint32_t cnt = w.pk >> 16;
            const bool live = w.row_ok && cnt > 0u && w.rv > 0.0f;
            const bool tied = live && cnt > 1u;
            o.pos = live ? (tied ? -1 : int(w.pk & 0xFFFFu)) : 0;
            o.gn = live ? (tied ? w.gv / float(cnt) : w.gv) : 0.0f;
            o.rv = tied ? w.rv : 0.0f;
            return o;
        };

        // software pipeline, two chunks deep: while chunk i is consumed from LDS, the x rows of chunk i + 1 are in registers
        // (loaded during chunk i - 1, written to the other buffer after the compute) and those of chunk i + 2 are in flight;
        // the source ids run one chunk further ahead still.  (Three register bundles trading roles in a loop unrolled by three —
        // no register rotation at the end of a step — measured SLOWER: 28.5 vs 25.8 ms at products shape; two rows per step
        // — 128-edge spans, 7 load slots, one barrier for two work items — 24.0 vs 24.4 ms at 255 VGPRs: not kept.)
#if TFGX_POOL_DEPTH == 3
        // three chunks deep: the gathers of chunk i + 3 are issued in step i and written to LDS at the end of step i + 1
        int inext = item_begin + 4;
        Item it0 = decode(fetch(item_begin));
        Item it1 = decode(fetch(item_begin + 1));
        Item it2 = decode(fetch(item_begin + 2));
        Item it3 = decode(fetch(item_begin + 3));
        int4 raw4 = fetch(inext);
        int ids1[kPoolSlots], ids2[kPoolSlots], ids3[kPoolSlots], ids4[kPoolSlots];
        float4v x1[kPoolSlots], x2[kPoolSlots], x3[kPoolSlots];
        {
            int ids0[kPoolSlots];
            float4v x0[kPoolSlots];
            load_ids(it0, ids0);
            load_ids(it1, ids1);
            load_ids(it2, ids2);
            load_ids(it3, ids3);
            load_x(it0, ids0, x0);
            load_x(it1, ids1, x1);
            load_x(it2, ids2, x2);
            commit(it0, 0, ids0, x0);
        }
        RawPair prw = load_pair(it0);
        __syncthreads();
        int buf = 0;
        while (it0.row < r_end) {
            const Item it4 = decode(raw4);
            raw4 = fetch(++inext);
            load_ids(it4, ids4);
            const RawPair prw_n = load_pair(it1);
            __builtin_amdgcn_sched_barrier(0);
            load_x(it3, ids3, x3);              // three chunks ahead
            __builtin_amdgcn_sched_barrier(0);
#else
        int inext = item_begin + 3;
        Item it0 = decode(fetch(item_begin));
        Item it1 = decode(fetch(item_begin + 1));
        Item it2 = decode(fetch(item_begin + 2));
        int4 raw3 = fetch(inext);                // decoded one step later: the load has a whole step to arrive
        int ids1[kPoolSlots], ids2[kPoolSlots], ids3[kPoolSlots];
        float4v x1[kPoolSlots], x2[kPoolSlots];
        {
            int ids0[kPoolSlots];
            float4v x0[kPoolSlots];
            load_ids(it0, ids0);
            load_ids(it1, ids1);
            load_ids(it2, ids2);
            load_x(it0, ids0, x0);
            load_x(it1, ids1, x1);
            commit(it0, 0, ids0, x0);
        }
        RawPair prw = load_pair(it0);
        __syncthreads();
        int buf = 0;
        while (it0.row < r_end) {
            const Item it3 = decode(raw3);
            raw3 = fetch(++inext);
#endif
#if TFGX_POOL_DEPTH == 3
#elif TFGX_POOL_LOAD_ORDER
            // PROGRAM ORDER of the step's loads: vector-memory loads return in order (vmcnt), so whatever a later step waits for also
            // waits for every load issued before it.  The x gathers of chunk i + 2 go LAST: the ids of chunk i + 3 and the (packed,
            // red, g) scalars of row i + 1 — which the next step needs at its start — are then never queued behind a random gather
            // (they were: every step began by waiting for the previous step's gathers, one chunk in flight at a time)
            load_ids(it3, ids3);
            const RawPair prw_n = load_pair(it1);
            __builtin_amdgcn_sched_barrier(0);
            load_x(it2, ids2, x2);              // two chunks ahead
            __builtin_amdgcn_sched_barrier(0);
#else
            load_x(it2, ids2, x2);              // two chunks ahead
            load_ids(it3, ids3);
            const RawPair prw_n = load_pair(it1);
#endif
            {
                const Pair pr = decode_pair(prw);
                const float* Xb = Xs + buf * kPoolChunk * XS;
                const int rel = pr.pos - it0.off;
                const bool in = pr.pos >= 0 && uint32_t(rel) < uint32_t(it0.len);       // the winner is staged in this chunk
                const float gv = in ? pr.gn : 0.0f;
                const float2v g2 = {gv, gv};
                const float* xr = Xb + (in ? rel : 0) * XS;
#pragma unroll
                for (int k = 0; k < (TFGX_POOL_EXPERIMENT == 1 ? 4 : KMAX); k += 4) {      // ds_read_b128 (immediate offsets), 2 x v_pk_fma_f32
                    const float4 v = *reinterpret_cast<const float4*>(xr + k);
                    const float2v lo = {v.x, v.y}, hi = {v.z, v.w};
                    acc[k / 2] = __builtin_elementwise_fma(g2, lo, acc[k / 2]);
                    acc[k / 2 + 1] = __builtin_elementwise_fma(g2, hi, acc[k / 2 + 1]);
                }
                if (pr.pos < 0) {
                    // tied maximum in this column: every edge of the chunk that attains it receives g / count (duplicate edges,
                    // exact ties of quantised features); the hidden rows are consulted, as the mask form does
                    const int* Cb = Cs + buf * kPoolChunk;
                    const float2v t2 = {pr.gn, pr.gn};
                    for (int i = 0; i < it0.len; ++i) {
                        if (a.h[int64_t(Cb[i]) * a.ldh + j] == pr.rv) {
                            const float* xt = Xb + i * XS;
#pragma unroll
                            for (int k = 0; k < KMAX; k += 2) {
                                const float2v v = {xt[k], xt[k + 1]};
                                acc[k / 2] = __builtin_elementwise_fma(t2, v, acc[k / 2]);
                            }
                        }
                    }
                }
            }
            commit(it1, buf ^ 1, ids1, x1);
            __syncthreads();
            buf ^= 1;
            prw = prw_n;
#if TFGX_POOL_DEPTH == 3
            it0 = it1; it1 = it2; it2 = it3; it3 = it4;
#pragma unroll
            for (int u = 0; u < kPoolSlots; ++u) { ids1[u] = ids2[u]; ids2[u] = ids3[u]; ids3[u] = ids4[u]; x1[u] = x2[u]; x2[u] = x3[u]; }
#else
            it0 = it1; it1 = it2; it2 = it3;
#pragma unroll
            for (int u = 0; u < kPoolSlots; ++u) { ids1[u] = ids2[u]; ids2[u] = ids3[u]; x1[u] = x2[u]; }
#endif
        }
    }
    float* out = a.partial + int64_t(blockIdx.x) * (a.F_in + 1) * a.Fp;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k <= a.F_in) out[int64_t(k) * a.Fp + j] = (k & 1) ? acc[k / 2].y : acc[k / 2].x;
}

__global__ __launch_bounds__(kBlock) void pool_wgrad_reduce_kernel(const float* __restrict__ partial, int n_blocks, int F_in,
                                                                   int Fp, float* __restrict__ dW, int64_t lddw,
                                                                   float* __restrict__ db)
{
    const int64_t total = int64_t(F_in + 1) * Fp;
    int64_t t = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (; t < total; t += stride) {
        float s = 0.0f;
        for (int b = 0; b < n_blocks; ++b) s += partial[int64_t(b) * total + t];
        const int k = int(t / Fp), j = int(t - int64_t(k) * Fp);
        if (k < F_in) dW[int64_t(k) * lddw + j] = s;
        else if (db != nullptr) db[j] = s;
    }
}

int pool_grid()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

inline size_t pool_lds_bytes(int kmax)
{
    return size_t(2) * kPoolChunk * size_t(kmax + 4) * sizeof(float) + size_t(2) * kPoolChunk * sizeof(int);
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace
}  // namespace tfgx

using namespace tfgx;

extern "C" int tfgx_pool_mlp_max_wgrad_applies(int64_t F_in, int64_t Fp)
{
    return (F_in % 4 == 0 && F_in >= 4 && F_in <= 124 && (Fp == 128 || Fp == 256 || Fp == 512)) ? 1 : 0;
}

static int pool_chunk(int64_t F_in, int64_t Fp)      // edges per work item: every 16-byte piece of a chunk has a load slot
{
    const int64_t c = (kPoolSlots * Fp) / (F_in / 4);
    return int(c < kPoolChunk ? c : kPoolChunk);
}

static size_t pool_partial_bytes(int64_t F_in, int64_t Fp) { return align256(size_t(1024) * size_t(F_in + 1) * size_t(Fp) * sizeof(float)); }
static size_t pool_item_capacity(int64_t n_dst, int64_t E, int64_t F_in, int64_t Fp) { return size_t(n_dst) + size_t(E) / size_t(pool_chunk(F_in, Fp)) + 1; }

// ---- the work-item table: a function of the plan and of (F_in, Fp) alone — built once per graph, kept by the caller
struct PoolPlanHeader {
    int32_t magic, grid, chunk, reserved;
    int64_t rows_per_block, n_dst, E, F_in, Fp;
};
constexpr int32_t kPoolMagic = 0x706f6f6c;
static size_t pool_plan_items_offset() { return align256(sizeof(PoolPlanHeader)) + align256(sizeof(int32_t) * 2 * 1032); }

extern "C" size_t tfgx_pool_mlp_max_wgrad_plan_bytes(int64_t n_dst, int64_t E, int64_t F_in, int64_t Fp)
{
    if (n_dst < 0 || E < 0 || !tfgx_pool_mlp_max_wgrad_applies(F_in, Fp)) return 0;
    return pool_plan_items_offset() + align256(sizeof(int4) * pool_item_capacity(n_dst, E, F_in, Fp));
}

extern "C" int tfgx_pool_mlp_max_wgrad_plan(const int32_t* row_ptr, int64_t n_dst, int64_t E, int64_t F_in, int64_t Fp,
                                            void* plan_buf, size_t plan_bytes, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n_dst >= 0 && E >= 0 && tfgx_pool_mlp_max_wgrad_applies(F_in, Fp), "bad size (tfgx_pool_mlp_max_wgrad_applies)");
    TFGX_REQUIRE(plan_buf != nullptr && plan_bytes >= tfgx_pool_mlp_max_wgrad_plan_bytes(n_dst, E, F_in, Fp), "plan buffer too small");
    TFGX_REQUIRE(n_dst == 0 || row_ptr != nullptr, "null pointer");
    hipStream_t st = as_stream(stream);
    PoolPlanHeader h;
    h.magic = kPoolMagic; h.reserved = 0; h.n_dst = n_dst; h.E = E; h.F_in = F_in; h.Fp = Fp;
    h.chunk = pool_chunk(F_in, Fp);
    int grid = pool_grid();
    if (grid > 1024) grid = 1024;
    if (int64_t(grid) > n_dst) grid = int(n_dst > 0 ? n_dst : 1);
    h.rows_per_block = n_dst > 0 ? (n_dst + grid - 1) / grid : 1;
    h.grid = n_dst > 0 ? int((n_dst + h.rows_per_block - 1) / h.rows_per_block) : 0;
    char* base = static_cast<char*>(plan_buf);
    // (the header's fields are functions of (device, n_dst, F_in, Fp) and are recomputed by the launch: nothing is copied)
    if (n_dst == 0) return TFGX_OK;
    int32_t* counts = reinterpret_cast<int32_t*>(base + align256(sizeof(PoolPlanHeader)));
    int32_t* item_ptr = counts + 1032;
    int4* items = reinterpret_cast<int4*>(base + pool_plan_items_offset());
    pool_item_count_kernel<<<(h.grid + 63) / 64, 64, 0, st>>>(row_ptr, n_dst, h.rows_per_block, h.chunk, counts);
    pool_item_scan_kernel<<<1, 64, 0, st>>>(counts, h.grid, item_ptr);
    pool_item_fill_kernel<<<(h.grid + 63) / 64, 64, 0, st>>>(row_ptr, n_dst, h.rows_per_block, h.chunk, item_ptr, items);
    TFGX_LAUNCH_CHECK("pool_item_*_kernel");
    return TFGX_OK;
}

extern "C" size_t tfgx_pool_mlp_max_wgrad_workspace_bytes(int64_t n_dst, int64_t F_in, int64_t Fp)
{
    if (n_dst < 0 || !tfgx_pool_mlp_max_wgrad_applies(F_in, Fp)) return 0;
    return pool_partial_bytes(F_in, Fp);                      // partials of up to 1024 workgroups
}

extern "C" int tfgx_pool_mlp_max_wgrad_f32(const int32_t* row_ptr, const int32_t* col, int64_t n_dst, int64_t E, const float* x,
                                           int64_t ldx, int64_t F_in, const float* h, int64_t ldh, const float* red,
                                           int64_t ldr, const int32_t* packed, int64_t ldp, const float* g, int64_t ldg,
                                           int64_t Fp, const void* plan_buf, float* dW, int64_t lddw, float* db,
                                           void* workspace, size_t workspace_bytes, tfgx_stream_t stream)
{
    TFGX_RANGE();
    TFGX_REQUIRE(n_dst >= 0 && E >= 0 && tfgx_pool_mlp_max_wgrad_applies(F_in, Fp),
                 "F_in a multiple of 4 in [4, 124] and Fp in {128, 256, 512} (tfgx_pool_mlp_max_wgrad_applies)");
    TFGX_REQUIRE(row_ptr && col && x && h && red && packed && g && dW && workspace && plan_buf, "null pointer");
    TFGX_REQUIRE(ldx >= F_in && ldx % 4 == 0 && aligned_to(x, 16) && ldx < (int64_t(1) << 31), "x: 16-byte aligned rows, ldx % 4 == 0");
    TFGX_REQUIRE(ldh >= Fp && ldr >= Fp && ldp >= Fp && ldg >= Fp && lddw >= Fp, "leading dimension too small");
    TFGX_REQUIRE(workspace_bytes >= tfgx_pool_mlp_max_wgrad_workspace_bytes(n_dst, F_in, Fp), "workspace too small");
    hipStream_t st = as_stream(stream);
    float* partial = static_cast<float*>(workspace);
    if (n_dst == 0 || E == 0) {      // (no edge: every maximum is float lowest, nothing passes the ReLU — and the row loop's
                                     //  unconditional loads need col[0] to exist)
        TFGX_HIP_CHECK(hipMemset2DAsync(dW, sizeof(float) * size_t(lddw), 0, sizeof(float) * size_t(Fp), size_t(F_in), st));
        if (db) TFGX_HIP_CHECK(hipMemsetAsync(db, 0, sizeof(float) * size_t(Fp), st));
        return TFGX_OK;
    }
    // grid / rows per workgroup / chunk are functions of (device, n_dst, F_in, Fp): recomputed here exactly as the plan call did
    PoolArgs a;
    a.row_ptr = row_ptr; a.col = col; a.n_dst = n_dst; a.x = x; a.ldx = ldx; a.F_in = int(F_in); a.h = h; a.ldh = ldh;
    a.red = red; a.ldr = ldr; a.packed = packed; a.ldp = ldp; a.g = g; a.ldg = ldg; a.Fp = int(Fp);
    a.partial = partial;
    int grid = pool_grid();
    if (grid > 1024) grid = 1024;
    if (int64_t(grid) > n_dst) grid = int(n_dst);
    a.rows_per_block = (n_dst + grid - 1) / grid;
    grid = int((n_dst + a.rows_per_block - 1) / a.rows_per_block);
    const char* base = static_cast<const char*>(plan_buf);
    a.chunk = pool_chunk(F_in, Fp);
    a.item_ptr = reinterpret_cast<const int32_t*>(base + align256(sizeof(PoolPlanHeader))) + 1032;
    a.items = reinterpret_cast<const int4*>(base + pool_plan_items_offset());
#define TFGX_POOL_LAUNCH(KMAX_)                                                                                       \
    {                                                                                                                 \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            TFGX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_wgrad_kernel<KMAX_>),             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, int(pool_lds_bytes(KMAX_)))); \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        pool_wgrad_kernel<KMAX_><<<grid, int(Fp), pool_lds_bytes(KMAX_), st>>>(a);                                    \
    }
    if (F_in + 1 <= 8) TFGX_POOL_LAUNCH(8)
    else if (F_in + 1 <= 40) TFGX_POOL_LAUNCH(40)
    else if (F_in + 1 <= 72) TFGX_POOL_LAUNCH(72)
    else if (F_in + 1 <= 104) TFGX_POOL_LAUNCH(104)
    else TFGX_POOL_LAUNCH(128)
#undef TFGX_POOL_LAUNCH
    TFGX_LAUNCH_CHECK("pool_wgrad_kernel");
    pool_wgrad_reduce_kernel<<<grid_for((F_in + 1) * Fp, kBlock), kBlock, 0, st>>>(partial, grid, int(F_in), int(Fp), dW, lddw, db);
    TFGX_LAUNCH_CHECK("pool_wgrad_reduce_kernel");
    return TFGX_OK;
}
