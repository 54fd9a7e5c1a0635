// ceg_math.h -- FP64 building blocks of the hot loop, written for the CDNA4 vector ALU:
// no IEEE division / sqrt / libm calls (each costs 10-40 VALU instructions), only
// v_rsq_f64 / v_rcp_f64 seeds refined by FMA Newton steps, a range-reduced exp and a
// polynomial for erfcx.  Every function is accurate to a few 1e-16 relative on its stated
// domain (tests/test_gpu_parity.py::test_radial_functions_accuracy), far inside the 1e-6
// parity tolerance, so the result differs from the reference's Base.exp /
// SpecialFunctions.erfc by rounding only.
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

// min of two non-NaN doubles in one instruction (fmin() adds two canonicalising v_max_f64)
__device__ __forceinline__ double min_nonan(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// sqrt(a) and 1/sqrt(a) for a normal positive a: hardware seed + 2 Goldschmidt steps.
__device__ __forceinline__ void fast_sqrt_rsqrt(double a, double& s, double& rs)
{
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y;          // ~ sqrt(a)
    double h = 0.5 * y;        // ~ 1/(2 sqrt(a))
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    s = g;
    rs = h + h;
}

// 1/a for a normal a: hardware seed + 2 Newton steps.
__device__ __forceinline__ double fast_rcp(double a)
{
    double y = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-a, y, 1.0);
    y = __builtin_fma(y, e, y);
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

}  // namespace ceg
