// Small helpers shared by the MFMA kernels (tfgx_gemm.hip, tfgx_fused.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace tfgx {

// 4 x 4 transpose between the four lanes of a quad and four registers: lane i (= lane & 3) enters with r[k] = M[k][i] and
// leaves with r[0..3] = M[i][0..3] — two butterfly stages over DPP quad permutes (xor 1, then xor 2).  The MFMA D layout puts
// ONE output column in a lane (rows in registers); after this a lane holds four consecutive columns of one row and the
// epilogue stores 16 bytes per lane: 32 store instructions per 32 x 256 tile instead of 128.
__device__ __forceinline__ void quad_transpose4(float (&r)[4], int lane)
{
    const bool odd = lane & 1, hi = lane & 2;
#pragma unroll
    for (int k = 0; k < 4; k += 2) {
        const float send = odd ? r[k] : r[k + 1];
        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        r[k] = odd ? recv : r[k];
        r[k + 1] = odd ? r[k + 1] : recv;
    }
    const float sa = hi ? r[0] : r[2], sb = hi ? r[1] : r[3];
    const float ra = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sa), 0x4E, 0xF, 0xF, true));          // quad_perm [2,3,0,1]
    const float rb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sb), 0x4E, 0xF, 0xF, true));
    const float o0 = hi ? ra : r[0], o1 = hi ? rb : r[1], o2 = hi ? r[2] : ra, o3 = hi ? r[3] : rb;
    r[0] = o0; r[1] = o1; r[2] = o2; r[3] = o3;
}

}  // namespace tfgx
