// ceg_minimage.h -- the reference's minimum-image routine, literally (shared by the grid kernels
// and the blocking-sphere scan).
#pragma once
#include <hip/hip_runtime.h>

namespace ceg {

// periodic_distance2_fromcartesian! (src/utils.jl:210-246).  d in/out: cartesian difference ->
// the image vector the reference leaves in `buffer` (stale on the fall-through path).
__device__ __forceinline__ double periodic_distance2_literal_m(const double* M, const double* I, int ortho, double safemin2,
                                                               double& dx, double& dy, double& dz)
{
    // same operation sequence as the Julia source (StaticArrays mat-vec = plain mul/add, no FMA):
    // components the reference's wrap arithmetic makes exactly zero stay exactly zero
#pragma clang fp contract(off)
    double f0 = I[0] * dx + I[3] * dy + I[6] * dz;
    double f1 = I[1] * dx + I[4] * dy + I[7] * dz;
    double f2 = I[2] * dx + I[5] * dy + I[8] * dz;
    double t;
    t = f0 + 0.5; f0 = t - floor(t) - 0.5;
    t = f1 + 0.5; f1 = t - floor(t) - 0.5;
    t = f2 + 0.5; f2 = t - floor(t) - 0.5;
    dx = M[0] * f0 + M[3] * f1 + M[6] * f2;
    dy = M[1] * f0 + M[4] * f1 + M[7] * f2;
    dz = M[2] * f0 + M[5] * f1 + M[8] * f2;
    const double ref2 = dx * dx + dy * dy + dz * dz;
    if (ortho || ref2 <= safemin2) return ref2;
    // first strictly closer image among +a, -a, +b, -b, +c, -c, formed like the reference does (src/utils.jl:234-244):
    // the fractional component is stepped IN PLACE (+1, -2, +1) and every trial image is the full product mat * f, so a
    // tie within an ulp of `newnorm2 < ref2` falls on the same side as in the Julia source (and the +1 -2 +1 round trip
    // leaves in f_i whatever rounding it leaves there for the later axes).
    double f[3] = {f0, f1, f2};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        f[ax] += 1.0;
        dx = M[0] * f[0] + M[3] * f[1] + M[6] * f[2];
        dy = M[1] * f[0] + M[4] * f[1] + M[7] * f[2];
        dz = M[2] * f[0] + M[5] * f[1] + M[8] * f[2];
        double n2 = dx * dx + dy * dy + dz * dz;
        if (n2 < ref2) return n2;
        f[ax] -= 2.0;
        dx = M[0] * f[0] + M[3] * f[1] + M[6] * f[2];
        dy = M[1] * f[0] + M[4] * f[1] + M[7] * f[2];
        dz = M[2] * f[0] + M[5] * f[1] + M[8] * f[2];
        n2 = dx * dx + dy * dy + dz * dz;
        if (n2 < ref2) return n2;
        f[ax] += 1.0;
    }
    // fall-through: the reference returns ref2 but leaves buffer at the last trial image, mat * (f with f_3 - 1)
    return ref2;
}

}  // namespace ceg
