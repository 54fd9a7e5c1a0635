// Issue cost of the instructions the k_culled hot loops are made of, in units of one v_fma_f64 (4 cycles per wave64 on a
// 16-lane SIMD): every thread runs 8 independent chains of one instruction, 4 waves per SIMD, all CUs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates scripts/probes/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v2d __attribute__((ext_vector_type(2)));
#define CHAINS 8
#define UNROLL 16

#define KERNEL(NAME, DECL, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(double* out, int iters, double seed)        \
    {                                                                                        \
        DECL;                                                                                \
        for (int it = 0; it < iters; ++it) {                                                 \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                             \
                _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { ASM; }                  \
            }                                                                                \
        }                                                                                    \
        double s = 0;                                                                        \
        for (int c = 0; c < CHAINS; ++c) s += (double)x[c];                                  \
        if (s == 12345.678) out[threadIdx.x] = s;                                            \
    }

#define DECL_F64 double x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = seed + c + threadIdx.x * 1e-3; double y = seed * 0.999
#define DECL_U32 unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = (unsigned)(seed * 1000) + c + threadIdx.x; unsigned y = (unsigned)seed + 3

KERNEL(k_fma_f64, DECL_F64, asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x[c]) : "v"(y)))
KERNEL(k_mul_f64, DECL_F64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
KERNEL(k_add_f64, DECL_F64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
KERNEL(k_rcp_f64, DECL_F64, asm volatile("v_rcp_f64 %0, %0" : "+v"(x[c])))
KERNEL(k_rsq_f64, DECL_F64, asm volatile("v_rsq_f64 %0, %0" : "+v"(x[c])))
KERNEL(k_sqrt_f64, DECL_F64, asm volatile("v_sqrt_f64 %0, %0" : "+v"(x[c])))
KERNEL(k_mov_b64, DECL_F64, asm volatile("v_mov_b64 %0, %1" : "+v"(x[c]) : "v"(y)))
KERNEL(k_cmp_f64, DECL_F64, asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x[c]), "v"(y) : "vcc"))
KERNEL(k_cndmask, DECL_U32, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(y) : "vcc"))
KERNEL(k_mov_b32, DECL_U32, asm volatile("v_mov_b32 %0, %1" : "+v"(x[c]) : "v"(y)))
KERNEL(k_and_b32, DECL_U32, asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
KERNEL(k_lshr_b32, DECL_U32, asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x[c])))
KERNEL(k_sub_u32, DECL_U32, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
KERNEL(k_cmp_u32, DECL_U32, asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(x[c]), "v"(y) : "vcc"))
KERNEL(k_mad_u24, DECL_U32, asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x[c]) : "v"(y)))
KERNEL(k_rcp_f32, DECL_U32, asm volatile("v_rcp_f32 %0, %0" : "+v"(x[c])))
KERNEL(k_cvt_f32_f64, DECL_F64, unsigned t; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(t) : "v"(x[c])))
KERNEL(k_cvt_f64_f32, DECL_F64, asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(x[c]) : "v"((float)y)))
KERNEL(k_pk_fma_f32, DECL_F64, asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x[c]) : "v"(y)))
KERNEL(k_fma_f32, DECL_U32, asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[c]) : "v"(y)))
KERNEL(k_ldexp_f64, DECL_F64, asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x[c])))
KERNEL(k_frexp_mant, DECL_F64, asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(x[c])))
KERNEL(k_floor_f64, DECL_F64, asm volatile("v_floor_f64 %0, %0" : "+v"(x[c])))
KERNEL(k_fract_f64, DECL_F64, asm volatile("v_fract_f64 %0, %0" : "+v"(x[c])))

__global__ __launch_bounds__(256) void k_lds_b128(double* out, int iters, double seed)
{
    __shared__ double4 tab[1024];
    for (int t = threadIdx.x; t < 1024; t += 256) tab[t] = make_double4(seed, t, 1, 2);
    __syncthreads();
    unsigned a = ((unsigned)seed & 1023u) * 32u;          // wave-uniform address: broadcast read, as in the hot loop
    double4 acc = make_double4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL * CHAINS; ++u) {
            v2d v;
            asm volatile("ds_read_b128 %0, %1 offset:0\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a + 32u * (u & 7)));
            acc.x += v.x;
        }
    }
    if (acc.x == 12345.678) out[threadIdx.x] = acc.x;
}
__global__ __launch_bounds__(256) void k_lds_b128_lane(double* out, int iters, double seed)
{
    __shared__ double4 tab[2048];
    for (int t = threadIdx.x; t < 2048; t += 256) tab[t] = make_double4(seed, t, 1, 2);
    __syncthreads();
    unsigned a = ((threadIdx.x * 7u) & 255u) * 112u;       // per-lane table rows of 112 bytes, as the Ewald table reads
    double4 acc = make_double4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL * CHAINS; ++u) {
            v2d v;
            asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(v) : "v"(a + 16u * (u % 7)));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc.x += v.x;
        }
    }
    if (acc.x == 12345.678) out[threadIdx.x] = acc.x;
}

typedef void (*kern_t)(double*, int, double);
static double run(kern_t k, int blocks, int iters)
{
    double* d;
    (void)hipMalloc(&d, 4096);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 4, 1.5);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
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
    const int iters = 2000;
    struct { const char* name; kern_t k; } list[] = {
        {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_rsq_f64", k_rsq_f64},
        {"v_sqrt_f64", k_sqrt_f64}, {"v_mov_b64", k_mov_b64}, {"v_cmp_lt_f64", k_cmp_f64}, {"v_cndmask_b32", k_cndmask}, {"v_mov_b32", k_mov_b32},
        {"v_and_b32", k_and_b32}, {"v_lshrrev_b32", k_lshr_b32}, {"v_sub_u32", k_sub_u32}, {"v_cmp_gt_u32", k_cmp_u32}, {"v_mad_u32_u24", k_mad_u24},
        {"v_rcp_f32", k_rcp_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_fma_f32", k_fma_f32}, {"v_ldexp_f64", k_ldexp_f64}, {"v_frexp_mant_f64", k_frexp_mant}, {"v_floor_f64", k_floor_f64}, {"v_fract_f64", k_fract_f64},
        {"ds_read_b128 (uniform address)", k_lds_b128}, {"ds_read_b128 (per-lane rows of 112 B)", k_lds_b128_lane}};
    const double base = run(k_fma_f64, blocks, iters);
    const double per_wave_instr = (double)iters * UNROLL * CHAINS;
    // 4 waves per SIMD share it: time = 4 waves * instr * cycles / clock
    printf("%d CUs, clock %.0f MHz; v_fma_f64: %.3f ms for %.0f instructions per wave, 4 waves per SIMD -> %.2f cycles per instruction at the nominal clock\n",
           cus, p.clockRate / 1e3, base, per_wave_instr, base * 1e-3 * p.clockRate * 1e3 / (4.0 * per_wave_instr));
    for (auto& e : list) {
        const double ms = run(e.k, blocks, iters);
        printf("%-40s %8.3f ms  = %5.2f x v_fma_f64\n", e.name, ms, ms / base);
    }
    return 0;
}
