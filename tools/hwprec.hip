// Accuracy of the gfx950 double-precision reciprocal / rsqrt estimates and of the refinements used in
// mw_kernels.hip.h, against host long double.  hipcc --offload-arch=gfx950 -O3 tools/hwprec.hip -o hwprec
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    o[i] = y;
    double e = __builtin_fma(-v * y, y, 1.0);
    double y1 = __builtin_fma(y, e * __builtin_fma(e, 0.375, 0.5), y);
    o[n + i] = y1;
    e = __builtin_fma(-v * y1, y1, 1.0);
    o[2 * n + i] = __builtin_fma(y1 * 0.5, e, y1);
    double r = __builtin_amdgcn_rcp(-v);
    o[3 * n + i] = r;
    double f = __builtin_fma(v, r, 1.0);
    double r1 = __builtin_fma(r, f, r);
    o[4 * n + i] = r1;
    f = __builtin_fma(v, r1, 1.0);
    o[5 * n + i] = __builtin_fma(r1, f, r1);
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(6 * n);
    for (int i = 0; i < n; ++i) x[i] = 0.5 + 80.0 * (double)i / n + 1e-7 * (i % 977);
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
    double m[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        long double rs = 1.0L / sqrtl((long double)x[i]), rc = -1.0L / (long double)x[i];
        for (int c = 0; c < 3; ++c) m[c] = fmax(m[c], (double)fabsl((o[c * n + i] - rs) / rs));
        for (int c = 3; c < 6; ++c) m[c] = fmax(m[c], (double)fabsl((o[c * n + i] - rc) / rc));
    }
    printf("rsq raw %.3e  +1 step(2nd order) %.3e  +2 steps %.3e\nrcp raw %.3e  +1 step %.3e  +2 steps %.3e\n", m[0], m[1], m[2], m[3], m[4], m[5]);
    return 0;
}
