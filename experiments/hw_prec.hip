// Measures the relative error of the gfx950 hardware approximations used by device_math.hpp (max over a sweep of arguments).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    o[0 * n + i] = __builtin_amdgcn_rcp(v);
    o[1 * n + i] = __builtin_amdgcn_rsq(v);
    o[2 * n + i] = __builtin_amdgcn_sqrt(v);
    o[3 * n + i] = (double)__builtin_amdgcn_rcpf((float)v);
    o[4 * n + i] = (double)__builtin_amdgcn_rsqf((float)v);
    o[5 * n + i] = (double)__builtin_amdgcn_sqrtf((float)v);
    o[6 * n + i] = (double)__logf((float)v);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), o(7 * n);
    for (int i = 0; i < n; ++i) x[i] = std::exp(-20.0 + 40.0 * (i + 0.37) / n) * (1.0 + 1e-3 * std::sin(12345.678 * i));
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 7 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 7 * n * 8, hipMemcpyDeviceToHost);
    const char *nm[7] = {"v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "__logf (abs err)"};
    for (int f = 0; f < 7; ++f) {
        double worst = 0;
        for (int i = 0; i < n; ++i) {
            const double v = f >= 3 ? (double)(float)x[i] : x[i];
            const double ex = f % 3 == 0 && f < 6 ? 1.0 / v : f % 3 == 1 && f < 6 ? 1.0 / std::sqrt(v) : f < 6 ? std::sqrt(v) : std::log(v);
            const double e = f == 6 ? std::fabs(o[f * n + i] - ex) : std::fabs(o[f * n + i] / ex - 1.0);
            if (e > worst) worst = e;
        }
        printf("%-18s max rel err %.3e = 2^%.1f\n", nm[f], worst, std::log2(worst));
    }
    return 0;
}
