// expf on the device vs double: worst relative error over [-40, 0] (hipcc -O3, the library's flags)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, float* y2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { y[i] = expf(x[i]); y2[i] = __expf(x[i]); }
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> x(n), y(n), y2(n);
    for (int i = 0; i < n; ++i) x[i] = -40.0f * float(i) / float(n);
    float *dx, *dy, *dy2;
    hipMalloc(&dx, 4 * n); hipMalloc(&dy, 4 * n); hipMalloc(&dy2, 4 * n);
    hipMemcpy(dx, x.data(), 4 * n, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dy, dy2, n);
    hipMemcpy(y.data(), dy, 4 * n, hipMemcpyDeviceToHost);
    hipMemcpy(y2.data(), dy2, 4 * n, hipMemcpyDeviceToHost);
    double w = 0, w2 = 0; float at = 0, at2 = 0;
    for (int i = 0; i < n; ++i) {
        const double r = std::exp(double(x[i]));
        const double e = std::fabs(double(y[i]) - r) / r, e2 = std::fabs(double(y2[i]) - r) / r;
        if (e > w) { w = e; at = x[i]; }
        if (e2 > w2) { w2 = e2; at2 = x[i]; }
    }
    printf("expf: worst rel err %.3e at x = %.4f;  __expf: %.3e at x = %.4f\n", w, at, w2, at2);
    return 0;
}
