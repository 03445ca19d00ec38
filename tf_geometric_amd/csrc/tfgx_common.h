// Shared helpers for the tfgx HIP sources (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/tfgx.h"

namespace tfgx {

void set_error(const char* fmt, ...);

inline int hip_fail(hipError_t e, const char* what)
{
    set_error("%s: %s", what, hipGetErrorString(e));
    return TFGX_ERR_HIP;
}

#define TFGX_HIP_CHECK(expr)                                        \
    do {                                                            \
        hipError_t _e = (expr);                                     \
        if (_e != hipSuccess) return ::tfgx::hip_fail(_e, #expr);   \
    } while (0)

#define TFGX_LAUNCH_CHECK(name)                                     \
    do {                                                            \
        hipError_t _e = hipGetLastError();                          \
        if (_e != hipSuccess) return ::tfgx::hip_fail(_e, name);    \
    } while (0)

#define TFGX_REQUIRE(cond, msg)                                     \
    do {                                                            \
        if (!(cond)) {                                              \
            ::tfgx::set_error("%s: %s", __func__, msg);             \
            return TFGX_ERR_INVALID_ARG;                            \
        }                                                           \
    } while (0)

inline hipStream_t as_stream(tfgx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// roctx ranges around every C-ABI entry point (the tracing hook): with TFGX_ROCTX=1 in the environment
// each call shows up as a named range in rocprofv3 --marker-trace next to the kernels it launched.  libroctx64.so is
// opened lazily with dlopen, so the library has no link-time dependency on it and costs one predictable branch per
// call when tracing is off.
void roctx_push(const char* name);
void roctx_pop();
struct RoctxRange {
    explicit RoctxRange(const char* name) { roctx_push(name); }
    ~RoctxRange() { roctx_pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};
#define TFGX_RANGE() ::tfgx::RoctxRange tfgx_roctx_range_(__func__)

constexpr int kWave = 64;
constexpr int kBlock = 256;
// memory-bound grids: enough workgroups to fill 256 CUs x 8 XCDs several times over
constexpr int kMaxGrid = 256 * 32;

inline int grid_for(int64_t work_items, int per_block, int cap = kMaxGrid)
{
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return static_cast<int>(g);
}

__device__ __forceinline__ float apply_act(float v, int act) { return (act == TFGX_ACT_RELU) ? fmaxf(v, 0.0f) : v; }

// ---- vector load/store helpers shared by the streaming kernels (VEC floats per lane: 4 -> global_load_dwordx4)
template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC])
{
    using T = typename VecT<VEC>::type;
    const T t = *reinterpret_cast<const T*>(p);
    if constexpr (VEC == 1) { v[0] = t; }
    if constexpr (VEC == 2) { v[0] = t.x; v[1] = t.y; }
    if constexpr (VEC == 4) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC])
{
    using T = typename VecT<VEC>::type;
    T t;
    if constexpr (VEC == 1) { t = v[0]; }
    if constexpr (VEC == 2) { t.x = v[0]; t.y = v[1]; }
    if constexpr (VEC == 4) { t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; }
    *reinterpret_cast<T*>(p) = t;
}

// Streamed-once rows (the raw softmax state chained between source-block launches, gradients accumulated block by block): loaded /
// stored NON-TEMPORALLY, so that they do not displace the gathered rows the launch keeps in the L2s.
typedef float tfgx_f32x4 __attribute__((ext_vector_type(4)));
typedef float tfgx_f32x2 __attribute__((ext_vector_type(2)));
template <int VEC>
__device__ __forceinline__ void load_vec_nt(const float* p, float (&v)[VEC])
{
    if constexpr (VEC == 1) { v[0] = __builtin_nontemporal_load(p); }
    if constexpr (VEC == 2) { const tfgx_f32x2 t = __builtin_nontemporal_load(reinterpret_cast<const tfgx_f32x2*>(p)); v[0] = t[0]; v[1] = t[1]; }
    if constexpr (VEC == 4) {
        const tfgx_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const tfgx_f32x4*>(p));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
}

template <int VEC>
__device__ __forceinline__ void store_vec_nt(float* p, const float (&v)[VEC])
{
    if constexpr (VEC == 1) { __builtin_nontemporal_store(v[0], p); }
    if constexpr (VEC == 2) { tfgx_f32x2 t; t[0] = v[0]; t[1] = v[1]; __builtin_nontemporal_store(t, reinterpret_cast<tfgx_f32x2*>(p)); }
    if constexpr (VEC == 4) {
        tfgx_f32x4 t; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
        __builtin_nontemporal_store(t, reinterpret_cast<tfgx_f32x4*>(p));
    }
}

// ---- dropout decisions: a pure function of (seed, item), so the forward kernel, both backward kernels and the host
// (tfgx_dropout_keep, used by the tests) regenerate the same mask.  murmur3 finaliser over item ^ seed_lo, mixed with
// seed_hi between the two multiplies; keep <=> top 24 bits >= rate * 2^24.
__host__ __device__ inline uint32_t drop_hash(uint64_t seed, uint32_t item)
{
    uint32_t x = item ^ static_cast<uint32_t>(seed);
    x ^= x >> 16;
    x *= 0x85ebca6bu;
    x ^= static_cast<uint32_t>(seed >> 32);
    x ^= x >> 13;
    x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}

struct DropCfg {
    uint32_t thr;       // 0 = dropout off
    float keep_scale;   // 1 / (1 - rate)
    uint64_t seed;
    int64_t self_base;
    const uint64_t* seed_dev;   // non-NULL: the seed is READ FROM DEVICE MEMORY by the kernel (hipGraph replays: the step's
                                // own captured seed-advance kernel leaves a fresh value there before every replayed launch)
};

inline DropCfg make_drop(float rate, uint64_t seed, int64_t self_base, const uint64_t* seed_dev = nullptr)
{
    DropCfg c;
    c.thr = rate > 0.0f ? static_cast<uint32_t>(rate * 16777216.0f) : 0u;
    c.keep_scale = rate > 0.0f ? 1.0f / (1.0f - rate) : 1.0f;
    c.seed = seed;
    c.self_base = self_base;
    c.seed_dev = rate > 0.0f ? seed_dev : nullptr;
    return c;
}

__host__ __device__ inline float drop_scale(const DropCfg& c, uint32_t item)
{
    if (c.thr == 0u) return 1.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t seed = c.seed_dev != nullptr ? *c.seed_dev : c.seed;     // wave-uniform address: one scalar load, cached
#else
    const uint64_t seed = c.seed;
#endif
    return (drop_hash(seed, item) >> 8) >= c.thr ? c.keep_scale : 0.0f;
}

inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace tfgx
