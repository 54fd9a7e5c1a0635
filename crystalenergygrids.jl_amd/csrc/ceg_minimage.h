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
    // first strictly closer image among +a, -a, +b, -b, +c, -c (src/utils.jl:234-244)
    const double wx = dx, wy = dy, wz = dz;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const double cx = M[3 * ax], cy = M[3 * ax + 1], cz = M[3 * ax + 2];
        double ex = wx + cx, ey = wy + cy, ez = wz + cz;     // (f_ax + 1)
        double n2 = ex * ex + ey * ey + ez * ez;
        if (n2 < ref2) { dx = ex; dy = ey; dz = ez; return n2; }
        ex = wx - cx; ey = wy - cy; ez = wz - cz;            // (f_ax - 1)
        n2 = ex * ex + ey * ey + ez * ez;
        if (n2 < ref2) { dx = ex; dy = ey; dz = ez; return n2; }
    }
    // fall-through: the reference returns ref2 but leaves buffer at the last trial image
    // (f_3 - 1), i.e. wrapped - c
    dx = wx - M[6]; dy = wy - M[7]; dz = wz - M[8];
    return ref2;
}

}  // namespace ceg
