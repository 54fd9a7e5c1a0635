// Does the FP64 matrix pipe run beside the FP64 vector ALU on MI355X?  
// The grid-build hot loop leaves v_mfma_f64_* idle; its 16 accumulation FMAs per candidate could in principle be rewritten as a
// contraction over candidates.  That only pays if a SIMD issues v_fma_f64 at full rate WHILE v_mfma_f64 instructions
// are in flight.  Measured here, all CUs, 4 waves per SIMD, independent accumulator chains:
//   (a) v_fma_f64 alone                      (b) v_mfma_f64_16x16x4_f64 alone          (b') v_mfma_f64_4x4x4_4b_f64 alone
//   (c) both in EVERY wave, interleaved      (d) half the waves of a SIMD do (a), the other half (b)
// If the pipes overlap, time(c) and time(d) approach max(time(a), time(b)); if they share issue / datapath, the sum.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_coissue scripts/probes/mfma_coissue.hip && /tmp/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));
#define NF 8          // independent FMA chains per thread
#define NM 4          // independent MFMA accumulators per wave

template <int FMA_PER_IT, int MFMA_PER_IT, int KIND, bool SPLIT_WAVES>
__global__ __launch_bounds__(256) void k_probe(double* out, int iters, double seed, int per_round)
{
    double x[NF];
    for (int c = 0; c < NF; ++c) x[c] = seed + c + threadIdx.x * 1e-3;
    const double y = seed * 0.999;
    v4d acc[NM];
    for (int c = 0; c < NM; ++c) acc[c] = v4d{seed, seed + c, 0.5, 0.25};
    double d1 = 0.0;
    const double a = seed * 1e-3 + threadIdx.x * 1e-6, b = 1.0 + threadIdx.x * 1e-7;
    // split: the workgroups of one CU are (with the dispatcher's round-robin) blocks b, b + per_round, b + 2 per_round, ...: those of
    // even "rounds" run the vector stream, those of odd rounds the matrix stream, so that every SIMD holds two waves of each
    const int round = blockIdx.x / per_round;
    const bool do_fma = !SPLIT_WAVES || (round & 1) == 0, do_mfma = !SPLIT_WAVES || (round & 1) == 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (do_fma) {
#pragma unroll
                for (int f = 0; f < FMA_PER_IT; ++f) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x[f % NF]) : "v"(y));
            }
            if (do_mfma) {
#pragma unroll
                for (int m = 0; m < MFMA_PER_IT; ++m) {
                    if (KIND == 0) acc[m % NM] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m % NM], 0, 0, 0);
                    else d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1 + 0.0 * m, 0, 0, 0);
                }
            }
        }
    }
    double s = d1;
    for (int c = 0; c < NF; ++c) s += x[c];
    for (int c = 0; c < NM; ++c) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
    if (s == 12345.678) out[threadIdx.x] = s;
}

typedef void (*kern_t)(double*, int, double, int);
static double run(kern_t k, int blocks, int iters)
{
    double* d;
    (void)hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 4, 1.5, blocks / 4);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5, blocks / 4);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    hipFree(d);
    return best;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 4;       // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    const int iters = 1000;
    const double clk = p.clockRate * 1e3;
    auto cyc = [&](double ms, double instr_per_wave, double waves_doing_it) { return ms * 1e-3 * clk / (waves_doing_it * instr_per_wave); };
    const double tf = run(k_probe<16, 0, 0, false>, blocks, iters);
    const double tm = run(k_probe<0, 4, 0, false>, blocks, iters);
    const double tm4 = run(k_probe<0, 4, 1, false>, blocks, iters);
    printf("%d CUs, nominal clock %.0f MHz, 4 waves per SIMD\n", cus, p.clockRate / 1e3);
    printf("(a)  v_fma_f64 alone, 16 per step                 %8.3f ms  = %.2f cycles per instruction per SIMD  (%.1f TFLOP/s)\n", tf, cyc(tf, iters * 8.0 * 16, 4),
           2.0 * 64 * iters * 8.0 * 16 * blocks * 4 / (tf * 1e-3) / 1e12);
    printf("(b)  v_mfma_f64_16x16x4 alone, 4 per step          %8.3f ms  = %.2f cycles per instruction per SIMD  (%.1f TFLOP/s)\n", tm, cyc(tm, iters * 8.0 * 4, 4),
           2.0 * 1024 * iters * 8.0 * 4 * blocks * 4 / (tm * 1e-3) / 1e12);
    printf("(b') v_mfma_f64_4x4x4_4b alone, 4 per step (dependent) %8.3f ms  = %.2f cycles per instruction per SIMD  (%.1f TFLOP/s)\n", tm4, cyc(tm4, iters * 8.0 * 4, 4),
           2.0 * 256 * iters * 8.0 * 4 * blocks * 4 / (tm4 * 1e-3) / 1e12);
    const double tc = run(k_probe<16, 4, 0, false>, blocks, iters);
    printf("(c)  16 v_fma_f64 + 4 v_mfma_f64_16x16x4 per step in every wave   %8.3f ms   sum (a)+(b) %.3f   max %.3f   -> overlap %.0f %%\n", tc, tf + tm,
           tf > tm ? tf : tm, 100.0 * (tf + tm - tc) / (tf < tm ? tf : tm));
    const double tc2 = run(k_probe<16, 2, 0, false>, blocks, iters), tm2 = run(k_probe<0, 2, 0, false>, blocks, iters);
    printf("(c2) 16 v_fma_f64 + 2 v_mfma_f64_16x16x4 per step in every wave   %8.3f ms   sum %.3f   max %.3f   -> overlap %.0f %%\n", tc2, tf + tm2,
           tf > tm2 ? tf : tm2, 100.0 * (tf + tm2 - tc2) / (tf < tm2 ? tf : tm2));
    // split: waves 0, 2 of a workgroup (2 of the 4 waves of each SIMD... one workgroup = 4 waves = 1 per SIMD; wave parity alternates per SIMD)
    const double tfs = run(k_probe<16, 0, 0, true>, blocks, iters), tms = run(k_probe<0, 4, 0, true>, blocks, iters);
    const double td = run(k_probe<16, 4, 0, true>, blocks, iters);
    printf("(d)  half the waves of each SIMD: 16 v_fma_f64 per step (alone %.3f ms); the other half: 4 v_mfma per step (alone %.3f ms); together %8.3f ms   sum %.3f   max %.3f\n",
           tfs, tms, td, tfs + tms, tfs > tms ? tfs : tms);
    return 0;
}
