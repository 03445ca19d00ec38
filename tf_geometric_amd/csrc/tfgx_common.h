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

}  // namespace tfgx
