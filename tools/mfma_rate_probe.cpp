// Pure-MFMA issue rate of gfx950 for the three fp32 shapes (no memory traffic): TFLOP/s with 1 / 2 / 4 waves per SIMD and
// 4 or 8 independent accumulator chains per wave.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.cpp -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int CHAINS>
__global__ __launch_bounds__(1024) void probe(float* out, int iters, float a0, float b0)
{
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    if constexpr (KIND == 0) {                       // v_mfma_f32_32x32x2_f32: 16 passes, 4096 flop
        f32x16 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c)
            for (int t = 0; t < 16; ++t) acc[c][t] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
        float s = 0.f;
        for (int c = 0; c < CHAINS; ++c)
            for (int t = 0; t < 16; ++t) s += acc[c][t];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else if constexpr (KIND == 1) {                // v_mfma_f32_16x16x4_f32: 8 passes, 2048 flop
        f32x4 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c)
            for (int t = 0; t < 4; ++t) acc[c][t] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        }
        float s = 0.f;
        for (int c = 0; c < CHAINS; ++c)
            for (int t = 0; t < 4; ++t) s += acc[c][t];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {                                         // v_mfma_f32_16x16x1_f32 (4 blocks): 8 passes, 2048 flop
        f32x16 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c)
            for (int t = 0; t < 16; ++t) acc[c][t] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[c], 0, 0, 0);
        }
        float s = 0.f;
        for (int c = 0; c < CHAINS; ++c)
            for (int t = 0; t < 16; ++t) s += acc[c][t];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int KIND, int CHAINS>
void run(const char* name, double flop_per, int threads, float* out, int cus)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<KIND, CHAINS><<<cus, threads>>>(out, 200, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND, CHAINS><<<cus, threads>>>(out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = double(cus) * (threads / 64) * double(iters) * CHAINS * flop_per;
    printf("{\"mfma\": \"%s\", \"chains\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f, \"cycles_per_mfma_at_2.4GHz\": %.1f}\n",
           name, CHAINS, threads / 256, ms, flop / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / (double(iters) * CHAINS * (threads / 256)));
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* out;
    hipMalloc(&out, sizeof(float) * size_t(cus) * 1024);
    for (int threads : {256, 512, 1024}) {
        run<0, 4>("32x32x2", 4096, threads, out, cus);
        run<0, 8>("32x32x2", 4096, threads, out, cus);
        run<1, 4>("16x16x4", 2048, threads, out, cus);
        run<1, 8>("16x16x4", 2048, threads, out, cus);
        run<2, 4>("16x16x1", 2048, threads, out, cus);
        run<2, 8>("16x16x1", 2048, threads, out, cus);
    }
    return 0;
}
