// ceg_math.h -- FP64 building blocks of the hot loops, written for the CDNA4 vector ALU:
// no IEEE division / sqrt / libm calls (each costs 10-40 VALU instructions), only
// v_rsq_f64 / v_rcp_f64 seeds (4.6e-8 / 5.2e-8 relative, scripts/probes/seed_accuracy.hip) refined by ONE
// Newton / coupled Goldschmidt step (2e-15 / 4e-15), a range-reduced table exp and an erfcx table for the
// variants that still evaluate exp / erfc in the loop (EWK = 1, per-candidate Buckingham classes), and the
// scalar-operand FMA forms the compiler does not emit by itself.  The r^2-indexed tables of EWK = 2 / VDWK = 3
// are described in ceg_internal.h (layout) and ceg_api.hip (host-side fit).  Accuracy of every path against the
// oracle's libm: tests/test_gpu_parity.py::test_fast_math_accuracy_single_pair.
#pragma once

#include <hip/hip_runtime.h>

namespace ceg {

// p*t + c with the constant c in a scalar register pair and a 3-address v_fma_f64.  Written as
// inline asm because hipcc otherwise keeps literal FP64 constants in VGPRs and emits
// v_mov_b64 + v_fmac_f64 for every Horner step (two VALU issues instead of one).
__device__ __forceinline__ double fma_sc(double p, double t, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(t), "s"(c));
    return r;
}

// a + c, a*2 + c, a*4 + c with c in a scalar register pair (2.0 / 4.0 are inline constants)
__device__ __forceinline__ double add_sc(double a, double c)
{
    double r;
    asm("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(c));
    return r;
}
__device__ __forceinline__ double fma2_sc(double a, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, 2.0, %2" : "=v"(r) : "v"(a), "s"(c));
    return r;
}
__device__ __forceinline__ double fma4_sc(double a, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, 4.0, %2" : "=v"(r) : "v"(a), "s"(c));
    return r;
}

// a * c with the constant c in a scalar register pair
__device__ __forceinline__ double mul_sc(double a, double c)
{
    double r;
    asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(c));
    return r;
}

// a*b + c with b in a scalar pair and c a loop-invariant VGPR constant (3-address, no copy)
__device__ __forceinline__ double fma_vsv(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

// a*b - c with b in a scalar pair (3-address, no copy)
__device__ __forceinline__ double fms_vsv(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, -%3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

// min of two non-NaN doubles in one instruction (fmin() adds two canonicalising v_max_f64)
__device__ __forceinline__ double min_nonan(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// sqrt(a) and 1/sqrt(a) for a normal positive a: v_rsq_f64 seed y (4.6e-8 relative, scripts/probes/seed_accuracy.hip) + ONE
// third-order step: with e = 1 - a y^2, 1/sqrt(a) = y (1 - e)^(-1/2) = y (1 + e/2 + 3 e^2/8 + O(e^3)), O(e^3) ~ 3e-22.
// Five instructions for 1/sqrt alone (the sqrt is one more product, dropped when unused) -- what the coupled Goldschmidt step of
// round 2 cost, whose second-order remainder (3/8) e^2 = 3e-15 reached the stored Float32 values where the repulsive and
// dispersive sums of a derivative channel cancel (profiles/r02_parity_report.txt: 77 ULP); now <= 3 roundings.
__device__ __forceinline__ void fast_sqrt_rsqrt(double a, double& s, double& rs)
{
    const double y = __builtin_amdgcn_rsq(a);
    const double t = a * y;
    const double e = __builtin_fma(-t, y, 1.0);
    double p;
    asm("v_fma_f64 %0, %1, %2, 0.5" : "=v"(p) : "v"(e), "s"(0.375));      // 1/2 + 3 e / 8 (0.375 in a scalar pair: no inline constant)
    const double q = y * e;
    rs = __builtin_fma(q, p, y);
    s = a * rs;
}

// 1/a for a normal a: hardware seed (4.6e-8 relative, measured: scripts/probes/seed_accuracy.hip) + CEG_RCP_ITERS Newton steps
// (1: 2.2e-15, 2: 1.1e-16).
#ifndef CEG_RCP_ITERS
#define CEG_RCP_ITERS 1
#endif
__device__ __forceinline__ double fast_rcp(double a)
{
    double y = __builtin_amdgcn_rcp(a);
#pragma unroll
    for (int it = 0; it < CEG_RCP_ITERS; ++it) {
        const double e = __builtin_fma(-a, y, 1.0);
        y = __builtin_fma(y, e, y);
    }
    return y;
}

// exp(y) for y in [-700, 0] (the hot loop passes -alpha^2 r^2 in [-30, 0]).
__device__ __forceinline__ double fast_exp_neg(double y)
{
    const double log2e = 1.4426950408889634074;
    const double ln2_hi = 6.93147180369123816490e-01;   // ln2 split: hi has 32 trailing zero bits
    const double ln2_lo = 1.90821492927058770002e-10;
    const double k = __builtin_rint(y * log2e);
    double r = __builtin_fma(-k, ln2_hi, y);
    r = __builtin_fma(-k, ln2_lo, r);                   // |r| <= ln2/2
    // exp(r) = sum r^n/n!, n <= 13 (truncation < 1e-17 for |r| <= 0.3466)
    double p = 1.6059043836821613e-10;                   // 1/13!
    p = fma_sc(p, r, 2.08767569878681e-09);       // 1/12!
    p = fma_sc(p, r, 2.505210838544172e-08);      // 1/11!
    p = fma_sc(p, r, 2.755731922398589e-07);      // 1/10!
    p = fma_sc(p, r, 2.7557319223985893e-06);     // 1/9!
    p = fma_sc(p, r, 2.48015873015873e-05);       // 1/8!
    p = fma_sc(p, r, 1.984126984126984e-04);      // 1/7!
    p = fma_sc(p, r, 1.388888888888889e-03);      // 1/6!
    p = fma_sc(p, r, 8.333333333333333e-03);      // 1/5!
    p = fma_sc(p, r, 4.1666666666666664e-02);     // 1/4!
    p = fma_sc(p, r, 1.6666666666666666e-01);     // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}

// ---- LDS-table variants used by the hot loop --------------------------------------------
// erfcx on [x0, x0 + N*h]: N = ERFCX_TAB_N intervals, degree-5 polynomial in the local
// coordinate s in [0, 1) per interval (6 doubles, 48 B, 16-B aligned), fitted per plan on
// the host in long double for the plan's [alpha*R_EXACT, alpha*cutoff] (max rel err 2e-15).
// Coefficients arrive from LDS as fresh VGPRs, so every Horner step is one v_fma/v_fmac.
constexpr int ERFCX_TAB_N = 128;

__device__ __forceinline__ double erfcx_tab(const double* __restrict__ tab, double x, double inv_h, double mx0_inv_h)
{
    const double u = __builtin_fma(x, inv_h, mx0_inv_h);      // (x - x0)/h in [0, N)
    unsigned idx = (unsigned)u;                                 // u >= 0: truncation = floor
    idx = idx < (unsigned)(ERFCX_TAB_N - 1) ? idx : (unsigned)(ERFCX_TAB_N - 1);
    const double sl = __builtin_amdgcn_fract(u);                // local coordinate in [0, 1)
    const double2* c = reinterpret_cast<const double2*>(tab + idx * 6);
    const double2 c01 = c[0], c23 = c[1], c45 = c[2];
    double p = __builtin_fma(c45.y, sl, c45.x);
    p = __builtin_fma(p, sl, c23.y);
    p = __builtin_fma(p, sl, c23.x);
    p = __builtin_fma(p, sl, c01.y);
    p = __builtin_fma(p, sl, c01.x);
    return p;
}

// exp(y), y in [-700, 0]: k = rint(64 y / ln2), exp(y) = 2^(k>>6) * 2^((k&63)/64) * exp(r),
// |r| <= ln2/128, exp(r) by a degree-5 Taylor polynomial (truncation 4e-17).
__device__ __forceinline__ double exp_neg_tab(const double* __restrict__ exp2_tab, double y)
{
    const double c64_log2e = 92.33248261689366;               // 64/ln2
    const double ln2_64_hi = 0.01083042469326756;             // ln2/64 with 21 trailing zero bits: k*hi exact
    const double ln2_64_lo = 2.9815858269852933e-12;
    const double kf = __builtin_rint(y * c64_log2e);
    const int k = (int)kf;
    double r = __builtin_fma(-kf, ln2_64_hi, y);
    r = __builtin_fma(-kf, ln2_64_lo, r);
    const double T = exp2_tab[k & 63];
    double q = fma_vsv(r, 8.3333333333333332e-03, 4.1666666666666664e-02);    // r/120 + 1/24
    q = fma_sc(q, r, 1.6666666666666666e-01);
    q = __builtin_fma(q, r, 0.5);
    q = __builtin_fma(q, r, 1.0);
    const double pr = q * r;                                  // exp(r) - 1
    return __builtin_ldexp(__builtin_fma(T, pr, T), k >> 6);
}

// erfcx(x) = exp(x^2) erfc(x) for x in [0, ERFCX_XMAX]: degree-18 polynomial in
// t = 2.8 u - 1.8, u = 2/(2+x) (Chebyshev interpolant on u in [2/7, 1] converted to the
// monomial basis with 60-digit arithmetic; sum |coef| = 1.006, so Horner is well conditioned;
// max relative error 2.2e-15 in double).
constexpr double ERFCX_XMAX = 5.0;

__device__ __forceinline__ double erfcx_poly(double x)
{
    const double u = 2.0 * fast_rcp(2.0 + x);
    const double t = __builtin_fma(u, 2.8, -1.8);
    double p = -8.831938387231295e-11;
    p = fma_sc(p, t, -4.135667802654713e-11);
    p = fma_sc(p, t, 1.3922900473941904e-09);
    p = fma_sc(p, t, -2.706262020862157e-09);
    p = fma_sc(p, t, -5.482959765234865e-09);
    p = fma_sc(p, t, 4.311416522498039e-08);
    p = fma_sc(p, t, -8.16041092885496e-08);
    p = fma_sc(p, t, -2.2284362873574516e-07);
    p = fma_sc(p, t, 1.6349752166110725e-06);
    p = fma_sc(p, t, -1.9074493734062743e-06);
    p = fma_sc(p, t, -1.641101524432363e-05);
    p = fma_sc(p, t, 6.508546403940596e-05);
    p = fma_sc(p, t, 0.0001235891242710475);
    p = fma_sc(p, t, -0.0012713100507524312);
    p = fma_sc(p, t, -0.0016962844625334375);
    p = fma_sc(p, t, 0.028193785197004564);
    p = fma_sc(p, t, 0.15791059062700571);
    p = fma_sc(p, t, 0.41766221044962965);
    p = fma_sc(p, t, 0.3990292854009171);
    return p;
}

// sin(2 pi u), cos(2 pi u) for any finite u of moderate size (|u| < 2^20, say): u is reduced to [-1/2, 1/2] and to the octant
// |2 pi v| <= pi/4 exactly (v = u - k/4), then the Cephes minimax polynomials of sin / cos on that interval; max abs error 1.9e-16
// (checked against extended precision on 2e6 random arguments).  ~35 instructions where ocml's sincospi, which also handles
// huge and non-finite arguments, takes ~230: the reciprocal-space tables of ceg_recip / ceg_mc are built from it.
__device__ __forceinline__ void sincos_2pi(double u, double& s, double& c)
{
    u -= rint(u);
    const double k = rint(4.0 * u);                       // -2 .. 2
    const double v = __builtin_fma(k, -0.25, u);          // exact
    const double x = v * 6.283185307179586476925;
    const double z = x * x;
    double ps = 1.58962301576546568060e-10;
    ps = __builtin_fma(ps, z, -2.50507477628578072866e-8);
    ps = __builtin_fma(ps, z, 2.75573136213857245213e-6);
    ps = __builtin_fma(ps, z, -1.98412698295895385996e-4);
    ps = __builtin_fma(ps, z, 8.33333333332211858878e-3);
    ps = __builtin_fma(ps, z, -1.66666666666666307295e-1);
    const double sinx = __builtin_fma(x * z, ps, x);
    double pc = -1.13585365213876817300e-11;
    pc = __builtin_fma(pc, z, 2.08757008419747316778e-9);
    pc = __builtin_fma(pc, z, -2.75573141792967388112e-7);
    pc = __builtin_fma(pc, z, 2.48015872888517045348e-5);
    pc = __builtin_fma(pc, z, -1.38888888888730564116e-3);
    pc = __builtin_fma(pc, z, 4.16666666666665929218e-2);
    const double cosx = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)k & 3;                             // angle = x + q pi/2
    const double a = (q & 1) ? cosx : sinx, b = (q & 1) ? sinx : cosx;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

}  // namespace ceg
