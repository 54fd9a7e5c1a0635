// Does a SIMD skip the 16-lane passes of a wave64 VALU instruction whose lanes are all inactive in EXEC?
// (The boundary candidates of the grid-build hot loop run with ~60 % of the lanes active, and a tile's lanes are ordered so that
// 16 consecutive lanes are one x-plane of 4 x 4 points: whole planes are often out of range together.  If inactive passes were
// skipped, part of the lane-utilisation loss would cost no time and the lane -> point order would matter.)
// A stream of independent v_fma_f64 chains, all CUs, 4 waves per SIMD, under different EXEC masks.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_skip scripts/probes/exec_skip.hip && /tmp/exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>

#define NF 8
__global__ __launch_bounds__(256) void k_probe(double* out, int iters, double seed, unsigned long long mask)
{
    double x[NF];
    for (int c = 0; c < NF; ++c) x[c] = seed + c + threadIdx.x * 1e-3;
    const double y = seed * 0.999;
    const int lane = threadIdx.x & 63;
    if ((mask >> lane) & 1ull) {                  // divergent region: EXEC = mask for the whole loop
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int f = 0; f < NF; ++f) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x[f]) : "v"(y));
        }
    }
    double s = 0.0;
    for (int c = 0; c < NF; ++c) s += x[c];
    if (s == 12345.678) out[threadIdx.x] = s;
}

static double run(unsigned long long mask, int blocks, int iters)
{
    double* d;
    (void)hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, d, 4, 1.5, mask);
    hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5, mask);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)hipFree(d);
    return best;
}

int main()
{
    const int blocks = 256 * 4, iters = 2000;      // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    struct { const char* what; unsigned long long m; } cases[] = {
        {"all 64 lanes", ~0ull},
        {"lanes 0-31 (two 16-lane passes empty)", 0xffffffffull},
        {"lanes 0-15 (three passes empty)", 0xffffull},
        {"lanes 16-31 only", 0xffff0000ull},
        {"every other lane (no pass empty)", 0x5555555555555555ull},
        {"lanes 0-7 of every 16 (no pass empty)", 0x00ff00ff00ff00ffull},
        {"one lane", 1ull},
    };
    const double flops = (double)blocks * 256 * iters * 16.0 * NF * 2.0;
    for (auto& c : cases) {
        const double ms = run(c.m, blocks, iters);
        printf("%-42s %8.3f ms   (%.1f TFLOP/s if all 64 lanes counted)\n", c.what, ms, flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
