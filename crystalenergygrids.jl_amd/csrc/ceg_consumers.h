// ceg_consumers.h -- device code shared by the grid-consumer kernels (rows f1-f3 of SURVEY 8f: ceg_interp.hip,
// ceg_pairs.hip) and the fused Monte-Carlo trial kernel (ceg_mc.hip).  Not installed.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "../../include/ceg_hip.h"
#include "ceg_internal.h"
#include "ceg_math.h"

namespace ceg_consumers {

using ceg::DevRule;

// ------------------------------------------------------------------ f1: tricubic interpolation
struct InterpGeom {
    double mat[9], invmat[9];
    double size[3], shift[3];
    int32_t dims[3];
    int32_t is_vdw;
    int32_t trilinear;      // EnergyGrid.higherorder == false: the "no derivatives" branch of interpolate_grid (grids.jl:259-269)
    int32_t _pad;
};

// 1-D cubic Hermite basis on [0,1]: value at 0, value at 1, slope at 0, slope at 1
__device__ __forceinline__ void hermite(double t, double w[2][2])
{
    const double t2 = t * t, t3 = t2 * t;
    w[0][0] = 2.0 * t3 - 3.0 * t2 + 1.0;     // f(0)
    w[0][1] = -2.0 * t3 + 3.0 * t2;          // f(1)
    w[1][0] = t3 - 2.0 * t2 + t;             // f'(0)
    w[1][1] = t3 - t2;                       // f'(1)
}

// interpolate_grid (src/grids.jl:212-273) at one point of a node-major grid [x][y][z][8]
__device__ __forceinline__ double interp_point(const InterpGeom& g, const float* __restrict__ grid, double px, double py, double pz)
{
    double sh[3];
    {
#pragma clang fp contract(off)
        // wrap_atom: abc = invmat * p;  newpoint = mat * (abc - floor(abc))      coordinates.jl:58-61
        const double* I = g.invmat;
        const double* M = g.mat;
        double a0 = (I[0] * px + I[3] * py) + I[6] * pz;
        double a1 = (I[1] * px + I[4] * py) + I[7] * pz;
        double a2 = (I[2] * px + I[5] * py) + I[8] * pz;
        a0 -= floor(a0); a1 -= floor(a1); a2 -= floor(a2);
        const double q0 = (M[0] * a0 + M[3] * a1) + M[6] * a2;
        const double q1 = (M[1] * a0 + M[4] * a1) + M[7] * a2;
        const double q2 = (M[2] * a0 + M[5] * a1) + M[8] * a2;
        // offsetpoint: (newpoint - shift)*dims/size + 1                           coordinates.jl:63-66
        sh[0] = (q0 - g.shift[0]) * (double)g.dims[0] / g.size[0] + 1.0;
        sh[1] = (q1 - g.shift[1]) * (double)g.dims[1] / g.size[1] + 1.0;
        sh[2] = (q2 - g.shift[2]) * (double)g.dims[2] / g.size[2] + 1.0;
    }
    const int nx = g.dims[0] + 1, ny = g.dims[1] + 1, nz = g.dims[2] + 1;
    // p0 = floor.(Int, shifted);  p1 = p0 .+ (p0 != extent)   (1-based)           grids.jl:216-218
    int p0[3], p1[3];
    double r[3];
    const int ext[3] = {nx, ny, nz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double f = floor(sh[a]);
        int i0 = (int)f;
        r[a] = sh[a] - f;
        // memory safety only: a wrapped point always lands in [1, extent]
        i0 = i0 < 1 ? 1 : (i0 > ext[a] ? ext[a] : i0);
        p0[a] = i0;
        p1[a] = i0 + (i0 != ext[a] ? 1 : 0);
    }
    if (g.trilinear) {
        // "no derivatives" (grids.jl:259-269): channel 1 at the 8 corners, trilinear weights, no blocking rule.  The reference
        // indexes this branch as g.grid[x, y, z, 1] although the array is [z, y, x, channel] (:126-133, :227-244): the FIRST
        // array index -- the one that runs along z -- receives the x cell index and vice versa.  Reproduced as written: an index
        // beyond the axis it is applied to is a BoundsError in Julia and NaN here (cubic grids never get there).
        const double mrx = 1.0 - r[0], mry = 1.0 - r[1], mrz = 1.0 - r[2];
        auto at = [&](int a, int b, int c) -> double {        // g.grid[a, b, c, 1], 1-based; a indexes z, b y, c x
            if (a < 1 || a > nz || b < 1 || b > ny || c < 1 || c > nx) return __builtin_nan("");
            return (double)grid[8 * (((int64_t)(c - 1) * ny + (b - 1)) * nz + (a - 1))];
        };
        const int x0 = p0[0], y0 = p0[1], z0 = p0[2], x1 = p1[0], y1 = p1[1], z1 = p1[2];
        {
#pragma clang fp contract(off)
            double ret = at(x0, y0, z0) * mrx * mry * mrz + at(x1, y0, z0) * r[0] * mry * mrz;
            ret = ret + at(x0, y1, z0) * mrx * r[1] * mrz;
            ret = ret + at(x0, y0, z1) * mrx * mry * r[2];
            ret = ret + at(x1, y1, z0) * r[0] * r[1] * mrz;
            ret = ret + at(x1, y0, z1) * r[0] * mry * r[2];
            ret = ret + at(x0, y1, z1) * mrx * r[1] * r[2];
            ret = ret + at(x1, y1, z1) * r[0] * r[1] * r[2];
            return ret;
        }
    }
    // node-major layout [x][y][z][8 channels]: the 8 channels of a corner are 32 contiguous bytes and
    // the two z neighbours of an (x, y) row 64 contiguous bytes
    const int64_t sx = (int64_t)ny * nz, sy = nz;
    const int64_t bx[2] = {(int64_t)(p0[0] - 1) * sx, (int64_t)(p1[0] - 1) * sx};
    const int64_t by[2] = {(int64_t)(p0[1] - 1) * sy, (int64_t)(p1[1] - 1) * sy};
    const int z0 = p0[2] - 1, z1 = p1[2] - 1;

    double wx[2][2], wy[2][2], wz[2][2];
    hermite(r[0], wx);
    hermite(r[1], wy);
    hermite(r[2], wz);

    double ret = 0.0;
    bool blocked = false;
    const float4* g4 = reinterpret_cast<const float4*>(grid);
#pragma unroll
    for (int ax = 0; ax < 2; ++ax)
#pragma unroll
        for (int ay = 0; ay < 2; ++ay) {
            const int64_t node0 = bx[ax] + by[ay] + z0, node1 = bx[ax] + by[ay] + z1;
            const float4 a0 = g4[2 * node0], b0 = g4[2 * node0 + 1];      // channels 0-3, 4-7 at z0
            const float4 a1 = g4[2 * node1], b1 = g4[2 * node1 + 1];      // ... at z1
            blocked = blocked || (a0.x > 5e6f) || (a1.x > 5e6f);
            // channels: value, dx, dy, dz, dxy, dxz, dyz, dxyz (derivatives pre-scaled by the grid step)
            const double v0[8] = {a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, b0.z, b0.w};
            const double v1[8] = {a1.x, a1.y, a1.z, a1.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ox = (c == 1 || c == 4 || c == 5 || c == 7) ? 1 : 0;
                const int oy = (c == 2 || c == 4 || c == 6 || c == 7) ? 1 : 0;
                const int oz = (c == 3 || c == 5 || c == 6 || c == 7) ? 1 : 0;
                const double wxy = wx[ox][ax] * wy[oy][ay];
                ret += wxy * (v0[c] * wz[oz][0] + v1[c] * wz[oz][1]);
            }
        }
    // VdW grid with any corner value > 5e6 -> 1e100 K                             grids.jl:245-248
    return (g.is_vdw && blocked) ? 1e100 : ret;
}

// The same interpolant with ONE corner per lane (the wave-per-placement Monte-Carlo kernel: 8 lanes per (atom, grid), their partial
// sums added by the caller with three lane shuffles).  corner = ax << 2 | ay << 1 | az.  Returns the corner's share of the 64-term sum;
// `blocked` is set when THIS corner's value exceeds 5e6 (the caller ORs the eight).  Trilinear grids ("no derivatives", a hand-made
// EnergyGrid only) are evaluated whole by the lane of corner 0.
__device__ __forceinline__ double interp_corner(const InterpGeom& g, const float* __restrict__ grid, double px, double py, double pz, int corner,
                                                bool& blocked)
{
    blocked = false;
    if (g.trilinear) return corner == 0 ? interp_point(g, grid, px, py, pz) : 0.0;
    double sh[3];
    {
#pragma clang fp contract(off)
        const double* I = g.invmat;
        const double* M = g.mat;
        double a0 = (I[0] * px + I[3] * py) + I[6] * pz;
        double a1 = (I[1] * px + I[4] * py) + I[7] * pz;
        double a2 = (I[2] * px + I[5] * py) + I[8] * pz;
        a0 -= floor(a0); a1 -= floor(a1); a2 -= floor(a2);
        const double q0 = (M[0] * a0 + M[3] * a1) + M[6] * a2;
        const double q1 = (M[1] * a0 + M[4] * a1) + M[7] * a2;
        const double q2 = (M[2] * a0 + M[5] * a1) + M[8] * a2;
        sh[0] = (q0 - g.shift[0]) * (double)g.dims[0] / g.size[0] + 1.0;
        sh[1] = (q1 - g.shift[1]) * (double)g.dims[1] / g.size[1] + 1.0;
        sh[2] = (q2 - g.shift[2]) * (double)g.dims[2] / g.size[2] + 1.0;
    }
    const int ext[3] = {g.dims[0] + 1, g.dims[1] + 1, g.dims[2] + 1};
    const int sel[3] = {(corner >> 2) & 1, (corner >> 1) & 1, corner & 1};
    int node[3];
    double w0[3], w1[3];                       // Hermite weights of this corner along each axis: value, slope
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double f = floor(sh[a]);
        int i0 = (int)f;
        const double t = sh[a] - f;
        i0 = i0 < 1 ? 1 : (i0 > ext[a] ? ext[a] : i0);
        const int i1 = i0 + (i0 != ext[a] ? 1 : 0);
        node[a] = (sel[a] ? i1 : i0) - 1;
        const double t2 = t * t, t3 = t2 * t;
        w0[a] = sel[a] ? (-2.0 * t3 + 3.0 * t2) : (2.0 * t3 - 3.0 * t2 + 1.0);
        w1[a] = sel[a] ? (t3 - t2) : (t3 - 2.0 * t2 + t);
    }
    const int64_t n = ((int64_t)node[0] * ext[1] + node[1]) * ext[2] + node[2];
    const float4* g4 = reinterpret_cast<const float4*>(grid);
    const float4 a = g4[2 * n], b = g4[2 * n + 1];
    blocked = a.x > 5e6f;
    // channels: value, dx, dy, dz, dxy, dxz, dyz, dxyz
    const double x0 = w0[0], x1 = w1[0], y0 = w0[1], y1 = w1[1], z0 = w0[2], z1 = w1[2];
    double ret = (x0 * y0) * (z0 * (double)a.x + z1 * (double)a.w);        // value, dz
    ret += (x1 * y0) * (z0 * (double)a.y + z1 * (double)b.y);              // dx, dxz
    ret += (x0 * y1) * (z0 * (double)a.z + z1 * (double)b.z);              // dy, dyz
    ret += (x1 * y1) * (z0 * (double)b.x + z1 * (double)b.w);              // dxy, dxyz
    return ret;
}

// ------------------------------------------------------------------ f3: pair rule energies
// (rule::InteractionRule)(r2) -- src/interactions.jl:392-406 (r2 forms) and :367-390 (r forms)
__device__ __forceinline__ double rule_energy(const DevRule& R, double r2, double coulombic)
{
#pragma clang fp contract(off)
    double v;
    switch (R.kind) {
    case CEG_LENNARDJONES: {
        const double s2 = R.p1 * R.p1;
        const double q = s2 / r2;
        const double x6 = q * q * q;
        v = 4.0 * R.p0 * x6 * (x6 - 1.0);
        break;
    }
    case CEG_HARDSPHERE: {
        const double s = R.p0 + R.p1;
        v = (r2 < s * s) ? __builtin_huge_val() : 0.0;
        break;
    }
    case CEG_NOINTERACTION: v = 0.0; break;
    case CEG_MONOMIAL: v = R.p0 / pow(r2, R.p1 / 2.0); break;
    case CEG_COULOMB_EWALD_DIRECT: {
        const double r = sqrt(r2);
        v = coulombic * R.p1 * R.p2 * erfc(R.p0 * r) / r;
        break;
    }
    case CEG_COULOMB: v = coulombic * R.p0 * R.p1 / sqrt(r2); break;
    case CEG_BUCKINGHAM: {
        const double r = sqrt(r2);
        const double r3 = r * r * r;
        v = R.p0 * exp(-R.p1 * r) - R.p2 / (r3 * r3);
        break;
    }
    case CEG_EXPONENTIAL: v = R.p0 * exp(-R.p1 * sqrt(r2)); break;
    default: v = __builtin_nan(""); break;          // UndefinedInteraction is refused at create time
    }
    return v - R.shift;
}

// Same energies with the shared sqrt / 1/r of the pair and the ceg_math.h functions (each <= 1.3e-13 relative):
// valid for 0.25 A^2 <= r2 and alpha*r <= ERFCX_XMAX for every CoulombEwaldDirect rule (checked by the host);
// closer pairs take rule_energy so that r -> 0 gives the reference's Inf / NaN.
__device__ __forceinline__ double rule_energy_fast(const DevRule& R, double r2, double r, double rinv, double coulombic)
{
    double v;
    switch (R.kind) {
    case CEG_LENNARDJONES: {
        const double q = (R.p1 * R.p1) * (rinv * rinv);
        const double x6 = q * q * q;
        v = 4.0 * R.p0 * x6 * (x6 - 1.0);
        break;
    }
    case CEG_HARDSPHERE: {
        const double s = R.p0 + R.p1;
        v = (r2 < s * s) ? __builtin_huge_val() : 0.0;
        break;
    }
    case CEG_NOINTERACTION: v = 0.0; break;
    case CEG_COULOMB_EWALD_DIRECT: {
        const double x = R.p0 * r;
        v = (coulombic * R.p1 * R.p2) * (ceg::fast_exp_neg(-(x * x)) * ceg::erfcx_poly(x)) * rinv;
        break;
    }
    case CEG_COULOMB: v = coulombic * R.p0 * R.p1 * rinv; break;
    case CEG_BUCKINGHAM: {
        const double i2 = rinv * rinv;
        v = R.p0 * ceg::fast_exp_neg(-(R.p1 * r)) - R.p2 * (i2 * i2 * i2);
        break;
    }
    case CEG_EXPONENTIAL: v = R.p0 * ceg::fast_exp_neg(-(R.p1 * r)); break;
    default: return rule_energy(R, r2, coulombic);      // Monomial (pow) and anything else
    }
    return v - R.shift;
}


// ---- the pair distance of unsafe_periodic_distance2! (src/utils.jl:294-302) for the consumer kernels.
// The reference measures the ONE image whose fractional difference lies in [-1/2, 1/2): invmat*d, wrap, mat*f, norm -- 47 flops as
// separate instructions when its operation order is kept (no contraction), of which the hot loop of k_pairs spent 91 % on non-FMA
// instructions (profiles/r03_consumers_pairs.txt).  With every perpendicular cell width above 2 cutoff (`fastwrap`, decided on the
// host: what the reference demands of an MC cell, montecarlo.jl "perpendicular length lower than 24.0") a pair whose fractional
// difference is within rounding of +-1/2 lies beyond the cutoff under EITHER image, so the wrap may be done as f - rint(f) with fused
// multiply-adds; only a pair within 1e-9 of the cutoff itself -- where the truncated potentials jump -- is measured again with the
// reference's operation order.  TRI: mat and invmat are upper triangular (a along x, b in the xy plane: the convention of every cell the
// reference builds, utils.jl:140-144), six of the eighteen products vanish.
__device__ __forceinline__ double pair_distance2_literal(const double* M, const double* I, double dx, double dy, double dz)
{
#pragma clang fp contract(off)
    double f0 = I[0] * dx + I[3] * dy + I[6] * dz;
    double f1 = I[1] * dx + I[4] * dy + I[7] * dz;
    double f2 = I[2] * dx + I[5] * dy + I[8] * dz;
    f0 = ((f0 + 0.5) - floor(f0 + 0.5)) - 0.5;
    f1 = ((f1 + 0.5) - floor(f1 + 0.5)) - 0.5;
    f2 = ((f2 + 0.5) - floor(f2 + 0.5)) - 0.5;
    const double vx = M[0] * f0 + M[3] * f1 + M[6] * f2;
    const double vy = M[1] * f0 + M[4] * f1 + M[7] * f2;
    const double vz = M[2] * f0 + M[5] * f1 + M[8] * f2;
    return vx * vx + vy * vy + vz * vz;
}

// the same behind a call, matrices read from memory (`geom` = mat[9], invmat[9] in device memory): what the fast form falls back to
// for the rare pair in the cutoff band -- inlined, the literal form kept all eighteen matrix elements live in scalar registers across
// the hot loop, and the kernels ran out of them (v_readlane reloads of spilled SGPRs: 16 VALU instructions per pair test in k_pairs)
__device__ __attribute__((noinline)) double pair_distance2_literal_call(const double* __restrict__ geom, double dx, double dy, double dz)
{
    return pair_distance2_literal(geom, geom + 9, dx, dy, dz);
}

template <bool TRI>
__device__ __forceinline__ double pair_distance2_fast(const double* M, const double* I, const double* __restrict__ geom, double dx, double dy, double dz,
                                                      double cutoff2, double band)
{
    double f0, f1, f2;
    if (TRI) {
        f2 = I[8] * dz;
        f1 = __builtin_fma(I[7], dz, I[4] * dy);
        f0 = __builtin_fma(I[6], dz, __builtin_fma(I[3], dy, I[0] * dx));
    } else {
        f0 = __builtin_fma(I[6], dz, __builtin_fma(I[3], dy, I[0] * dx));
        f1 = __builtin_fma(I[7], dz, __builtin_fma(I[4], dy, I[1] * dx));
        f2 = __builtin_fma(I[8], dz, __builtin_fma(I[5], dy, I[2] * dx));
    }
    f0 -= __builtin_rint(f0);
    f1 -= __builtin_rint(f1);
    f2 -= __builtin_rint(f2);
    double vx, vy, vz;
    if (TRI) {
        vx = __builtin_fma(M[6], f2, __builtin_fma(M[3], f1, M[0] * f0));
        vy = __builtin_fma(M[7], f2, M[4] * f1);
        vz = M[8] * f2;
    } else {
        vx = __builtin_fma(M[6], f2, __builtin_fma(M[3], f1, M[0] * f0));
        vy = __builtin_fma(M[7], f2, __builtin_fma(M[4], f1, M[1] * f0));
        vz = __builtin_fma(M[8], f2, __builtin_fma(M[5], f1, M[2] * f0));
    }
    double r2 = __builtin_fma(vz, vz, __builtin_fma(vy, vy, vx * vx));
    if (fabs(r2 - cutoff2) <= band) r2 = pair_distance2_literal_call(geom, dx, dy, dz);       // the cutoff decision is the reference's
    return r2;
}

// host side: 0 literal, 1 fast wrap, 2 fast wrap with upper-triangular matrices
inline int wrap_mode(const double mat[9], const double invmat[9], const double hfrac[3])
{
    for (int i = 0; i < 3; ++i)
        if (!(hfrac[i] < 0.5 * (1.0 - 1e-6))) return 0;
    const bool tri = mat[1] == 0.0 && mat[2] == 0.0 && mat[5] == 0.0 && invmat[1] == 0.0 && invmat[2] == 0.0 && invmat[5] == 0.0;
    return tri ? 2 : 1;
}

// ---- neighbour cells of the guest atoms (what the reference gets from CellListMap above 1200 atom x threads,
// src/energy.jl:340-349,398-404): bins of the FRACTIONAL coordinates of the MC cell.  unsafe_periodic_distance2!
// (src/utils.jl:294-302) measures the one image whose fractional difference lies in [-1/2, 1/2), and |f_i| <= cutoff / width_i
// for a pair inside the cutoff (width_i = perpendicular width = 1 / |row i of invmat|): the partners of an atom at fractional
// coordinate f sit in the bins of [f - hfrac, f + hfrac] mod 1 whatever the cell shape.  The kernels run every atom of those
// bins through the same exact distance test as the exhaustive loop, so the sums are the same sets of pairs.
struct CellBins {
    int32_t on;
    int32_t nb[3];
    double hfrac[3];          // cutoff / width_i
};

// Cells pay when the bins a molecule can reach (cutoff sphere + its own extent + a bin either side) hold well under half the MC cell: never for the
// reference's fixtures (24-36 A wide, cutoff 12 A), by a factor 2-6 for the north-star framework (57 x 57 x 85 A).
// CEG_HIP_MC_CELLS=1 / 0 forces them on / off, CEG_HIP_MC_BIN sets the bin width (A, default 4).
inline CellBins choose_cell_bins(const double invmat[9], double cutoff)
{
    CellBins cb{};
    double bin = 4.0;
    if (const char* e = getenv("CEG_HIP_MC_BIN")) { const double b = atof(e); if (b >= 0.5 && b <= 100.0) bin = b; }
    double covered = 1.0;
    int64_t ncells = 1;
    for (int i = 0; i < 3; ++i) {
        const double norm = std::sqrt(invmat[i] * invmat[i] + invmat[3 + i] * invmat[3 + i] + invmat[6 + i] * invmat[6 + i]);
        const double width = 1.0 / norm;
        cb.hfrac[i] = cutoff * norm;
        cb.nb[i] = std::max(1, std::min(1024, (int)std::floor(width / bin)));
        ncells *= cb.nb[i];
        covered *= std::min(1.0, (2.0 * cutoff + 2.0 * bin + 3.0) / width);
    }
    cb.on = covered < 0.5 ? 1 : 0;
    if (const char* e = getenv("CEG_HIP_MC_CELLS")) cb.on = atoi(e) != 0 ? 1 : 0;
    if (ncells > (1 << 22)) cb.on = 0;
    return cb;
}

inline int cell_of_position(const CellBins& cb, const double invmat[9], const double* p)
{
    int b[3];
    for (int i = 0; i < 3; ++i) {
        double f = invmat[i] * p[0] + invmat[3 + i] * p[1] + invmat[6 + i] * p[2];
        f -= std::floor(f);
        const int q = (int)(f * cb.nb[i]);
        b[i] = q < 0 ? 0 : (q >= cb.nb[i] ? cb.nb[i] - 1 : q);
    }
    return (b[0] * cb.nb[1] + b[1]) * cb.nb[2] + b[2];
}

// bins along fractional axis `axis` that the cutoff spheres of the m atoms at pos[3m] can reach: `n` bins starting at `first`
// (periodic, first in [0, nb), n <= nb so that no bin comes twice); 1e-6 of slack covers the rounding of the bin assignment
__device__ __forceinline__ void cell_range(const double* I, const double* pos, int m, int axis, int nb, double hfrac, int& first, int& n)
{
    double lo = 1e300, hi = -1e300;
    for (int a = 0; a < m; ++a) {
        const double f = I[axis] * pos[3 * a] + I[3 + axis] * pos[3 * a + 1] + I[6 + axis] * pos[3 * a + 2];
        lo = fmin(lo, f);
        hi = fmax(hi, f);
    }
    const double slack = hfrac + 1e-6;
    const long long blo = (long long)floor((lo - slack) * nb), bhi = (long long)floor((hi + slack) * nb);
    const long long span = bhi - blo + 1;
    long long f0 = blo % nb;
    if (f0 < 0) f0 += nb;
    first = (int)f0;
    n = (int)(span < nb ? span : nb);
}


// Device buffers of a handle's host-array entry point (ceg_interp_points / ceg_recip_energy / ceg_pairs_energy), kept between calls and only
// ever grown: hipMalloc + hipFree of two arrays per call cost more than a batch of a few thousand rows takes (hipFree synchronises the device).
struct HostIo {
    double* d_in = nullptr;
    double* d_out = nullptr;
    size_t in_cap = 0, out_cap = 0;
    bool ensure(size_t in_bytes, size_t out_bytes)
    {
        if (in_bytes > in_cap) {
            if (d_in) (void)hipFree(d_in);
            d_in = nullptr; in_cap = 0;
            const size_t cap = in_bytes + in_bytes / 2;
            if (hipMalloc((void**)&d_in, cap) != hipSuccess) return false;
            in_cap = cap;
        }
        if (out_bytes > out_cap) {
            if (d_out) (void)hipFree(d_out);
            d_out = nullptr; out_cap = 0;
            const size_t cap = out_bytes + out_bytes / 2;
            if (hipMalloc((void**)&d_out, cap) != hipSuccess) return false;
            out_cap = cap;
        }
        return true;
    }
    void release()
    {
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        d_in = d_out = nullptr;
        in_cap = out_cap = 0;
    }
};

}  // namespace ceg_consumers

// the interpolation handle (ceg_interp.hip owns its life cycle; ceg_mc.hip reads geometry and grid pointer)
struct ceg_interp {
    int device = 0;
    ceg_consumers::InterpGeom g{};
    const float* d_grid = nullptr;
    float* owned = nullptr;
    ceg_consumers::HostIo io;          // ceg_interp_points
};

