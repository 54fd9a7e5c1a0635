// Accuracy of the gfx950 FP64 reciprocal / rsqrt seeds and of 1-2 Newton steps on top (what ceg_math.h builds on).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/seed_accuracy scripts/probes/seed_accuracy.hip && /tmp/seed_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i];
    double y0 = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, y0, 1.0);
    double y1 = __builtin_fma(y0, e, y0);
    e = __builtin_fma(-a, y1, 1.0);
    double y2 = __builtin_fma(y1, e, y1);
    double r0 = __builtin_amdgcn_rsq(a);
    out[5 * i] = y0; out[5 * i + 1] = y1; out[5 * i + 2] = y2; out[5 * i + 3] = r0;
    double g = a * r0, h = 0.5 * r0, r = __builtin_fma(-h, g, 0.5);
    h = __builtin_fma(h, r, h);
    out[5 * i + 4] = h + h;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(5 * n);
    for (int i = 0; i < n; ++i) x[i] = 1.0 + 255.0 * (i + 0.37) / n;
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 5 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 5 * n * 8, hipMemcpyDeviceToHost);
    double w[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const long double a = x[i], inv = 1.0L / a, rs = 1.0L / sqrtl(a);
        for (int c = 0; c < 5; ++c) {
            const long double ref = c < 3 ? inv : rs;
            w[c] = fmax(w[c], (double)fabsl((o[5 * i + c] - ref) / ref));
        }
    }
    printf("v_rcp_f64 seed %.3e  +1 Newton %.3e  +2 Newton %.3e   v_rsq_f64 seed %.3e  +1 Goldschmidt %.3e\n", w[0], w[1], w[2], w[3], w[4]);
    return 0;
}
