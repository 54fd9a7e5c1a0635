// ceg_kernels.hip -- hand-written HIP kernels (gfx950 / MI355X, wave64) for the grid-build
// hot path of CrystalEnergyGrids.jl:
//
//   compute_derivatives_vdw    src/probes.jl:71-92
//   compute_derivatives_ewald  src/probes.jl:94-117
//   derivativesGrid            src/interactions.jl:432-472, 599-610
//   derivatives_ewald          src/ewald.jl:299-312
//   periodic_distance2_fromcartesian!  src/utils.jl:210-246
//   abc_to_xyz                 src/coordinates.jl:72-76
//   _set_gridpoint!            src/grids.jl:118-135
//
// Two algorithms, same results up to FP64 summation order:
//
//  * k_bruteforce: the reference's loop shape -- one thread per grid point, every
//    ProbeSystem atom streamed through LDS in tiles of 256 and run through the literal
//    min-image routine.  O(N_grid * N_atoms).
//
//  * k_culled: one wave (64 lanes) owns a 4x4x4 tile of grid points.  The atoms have
//    been expanded on the host into explicit lattice images binned on a cartesian
//    lattice; the wave gathers the images whose bin rows intersect the tile's cutoff
//    neighbourhood, prunes them against the tile box, compacts the survivors into LDS
//    and every lane then loops over the same (broadcast) candidate.  For each
//    (point, image) pair inside the cutoff the reference's *selection rule* (is this
//    image the one periodic_distance2! would return?) is evaluated exactly, including
//    the `ortho` shortcut and the stale-vector fall-through of src/utils.jl:234-245.
//    O(N_grid * n_cut).
//
// No MFMA: this is a pairwise FP64 reduction, bound by the FP64 vector ALU.
#include "ceg_internal.h"

namespace ceg {

// ------------------------------------------------------------------ small helpers
struct Accum {
    double v, d1x, d1y, d1z, d2xy, d2xz, d2yz, d3;
};

__device__ __forceinline__ void accum_zero(Accum& a)
{
    a.v = a.d1x = a.d1y = a.d1z = a.d2xy = a.d2xz = a.d2yz = a.d3 = 0.0;
}

// src/probes.jl:85-89
__device__ __forceinline__ void accum_add(Accum& a, double v, double p1, double p2, double p3,
                                          double dx, double dy, double dz)
{
    a.v += v;
    a.d1x += p1 * dx;
    a.d1y += p1 * dy;
    a.d1z += p1 * dz;
    const double d13 = dx * dz;
    a.d2xy += p2 * (dx * dy);
    a.d2xz += p2 * d13;
    a.d2yz += p2 * (dy * dz);
    a.d3 += p3 * d13 * dy;
}

// src/coordinates.jl:72-76, evaluated as (i*size)/dims + shift with no contraction so the
// grid coordinates are bit-identical to the reference's.
__device__ __forceinline__ double grid_coord(int i, double size, int dims, double shift)
{
#pragma clang fp contract(off)
    return __dadd_rn(__ddiv_rn(__dmul_rn((double)i, size), (double)dims), shift);
}

// ------------------------------------------------------------------ pair potentials
// derivativesGrid over the rule run of one atom kind (src/interactions.jl:432-472,599-610).
// `rb`,`re` are wave-uniform.
__device__ __forceinline__ void vdw_terms(const DevRule* __restrict__ rules, int rb, int re, double r2,
                                          double& v, double& p1, double& p2, double& p3)
{
    v = p1 = p2 = p3 = 0.0;
    for (int t = rb; t < re; ++t) {
        const int kind = rules[t].kind;
        const double q0 = rules[t].p0, q1 = rules[t].p1, q2 = rules[t].p2, sh = rules[t].shift;
        if (kind == CEG_LENNARDJONES) {               // :434-441
            const double inv = 1.0 / r2;
            const double s = q1 * inv;                // sigma^2 / r2
            const double x6 = s * s * s;
            const double inv2 = inv * inv;
            v += 4.0 * q0 * x6 * (x6 - 1.0) - sh;
            p1 += 24.0 * q0 * (x6 * (1.0 - 2.0 * x6)) * inv;
            p2 += 96.0 * q0 * (x6 * (7.0 * x6 - 2.0)) * inv2;
            p3 += 384.0 * q0 * (x6 * (5.0 - 28.0 * x6)) * (inv2 * inv2);
        } else if (kind == CEG_BUCKINGHAM) {          // :447-457
            const double A = q0, B = q1, C = q2;
            const double r4 = r2 * r2;
            const double r = sqrt(r2);
            const double r6 = r4 * r2;
            const double x6 = C / r6;
            const double xe = A * exp(-B * r);
            v += (xe - x6) - sh;
            p1 += -B * xe / r + 6.0 * x6 / r2;
            p2 += -48.0 * x6 / r4 + B * xe * (1.0 + B * r) / (r2 * r);
            p3 += -(3.0 * B * r + B * B * r2 + 3.0) * B * xe * r / r6 + 480.0 * C / (r6 * r6);
        } else {                                      // CEG_HARDSPHERE :444-446
            v += ((r2 < q0) ? __builtin_huge_val() : 0.0) - sh;
        }
    }
}

// derivatives_ewald (src/ewald.jl:299-312)
__device__ __forceinline__ void ewald_terms(double alpha, double charge, double r2,
                                            double& v, double& p1, double& p2, double& p3)
{
    const double inv_sqrtpi = 0.56418958354775628695;
    const double r = sqrt(r2);
    const double r3 = r2 * r;
    const double r5 = r3 * r2;
    const double r2a2 = r2 * (alpha * alpha);
    const double er2a2 = 2.0 * alpha * r * exp(-r2a2) * inv_sqrtpi;
    const double erfar = erfc(alpha * r);
    v = charge * erfar / r;
    p1 = -charge * (er2a2 + erfar) / r3;
    p2 = charge * (er2a2 * (3.0 + 2.0 * r2a2) + 3.0 * erfar) / r5;
    p3 = charge * (-er2a2 * (15.0 + 10.0 * r2a2 + 4.0 * (r2a2 * r2a2)) - 15.0 * erfar) / (r5 * r2);
}

// ------------------------------------------------------------------ result store
// Base.clamp semantics: NaN passes through.
__device__ __forceinline__ double clamp_julia(double x, double lo, double hi)
{
    return (x > hi) ? hi : ((x < lo) ? lo : x);
}

// _set_gridpoint! (src/grids.jl:118-135)
__device__ __forceinline__ void store_gridpoint(float* __restrict__ out, int64_t idx, int64_t cs,
                                                const double* delta, double lambda, double thr, Accum a)
{
#pragma clang fp contract(off)
    if (a.v > thr) {
        a.v = 2.0 * thr;
        a.d1x = clamp_julia(a.d1x, -thr, thr);
        a.d1y = clamp_julia(a.d1y, -thr, thr);
        a.d1z = clamp_julia(a.d1z, -thr, thr);
        a.d2xy = a.d2xz = a.d2yz = 0.0;
        a.d3 = 0.0;
    }
    const double D1 = delta[0], D2 = delta[1], D3 = delta[2];
    out[idx] = (float)__dmul_rn(a.v, lambda);
    out[idx + cs] = (float)__dmul_rn(__dmul_rn(a.d1x, D1), lambda);
    out[idx + 2 * cs] = (float)__dmul_rn(__dmul_rn(a.d1y, D2), lambda);
    out[idx + 3 * cs] = (float)__dmul_rn(__dmul_rn(a.d1z, D3), lambda);
    out[idx + 4 * cs] = (float)__dmul_rn(__dmul_rn(a.d2xy, __dmul_rn(D1, D2)), lambda);
    out[idx + 5 * cs] = (float)__dmul_rn(__dmul_rn(a.d2xz, __dmul_rn(D1, D3)), lambda);
    out[idx + 6 * cs] = (float)__dmul_rn(__dmul_rn(a.d2yz, __dmul_rn(D2, D3)), lambda);
    out[idx + 7 * cs] = (float)__dmul_rn(__dmul_rn(a.d3, __dmul_rn(__dmul_rn(D1, D2), D3)), lambda);
}

__device__ __forceinline__ void store_raw(double* __restrict__ out, int64_t p, const Accum& a)
{
    double* o = out + 8 * p;
    o[0] = a.v;  o[1] = a.d1x;  o[2] = a.d1y;  o[3] = a.d1z;
    o[4] = a.d2xy;  o[5] = a.d2xz;  o[6] = a.d2yz;  o[7] = a.d3;
}

template <int MODE>
__device__ __forceinline__ void write_results(const Geom& g, const Output& out, bool points_mode,
                                              int64_t pidx, int i, int j, int k,
                                              Accum av, Accum ac, double smallest_d2)
{
    if (MODE != MODE_VDW) {
        // src/probes.jl:116
        ac.v = (smallest_d2 < 1.0) ? __builtin_huge_val() : ac.v;
    }
    if (points_mode) {
        if (MODE != MODE_COULOMB && out.raw_vdw) store_raw(out.raw_vdw, pidx, av);
        if (MODE != MODE_VDW && out.raw_coulomb) store_raw(out.raw_coulomb, pidx, ac);
        return;
    }
    const int64_t nz = g.dims[2] + 1, ny = g.dims[1] + 1;
    const int64_t idx = (int64_t)k + nz * ((int64_t)j + ny * (int64_t)(i - out.i_origin));
    if (MODE != MODE_COULOMB && out.vdw)
        store_gridpoint(out.vdw, idx, out.channel_stride, g.delta, out.lambda_vdw, out.thr_vdw, av);
    if (MODE != MODE_VDW && out.coulomb)
        store_gridpoint(out.coulomb, idx, out.channel_stride, g.delta, out.lambda_coulomb, out.thr_coulomb, ac);
}

// ------------------------------------------------------------------ literal min-image
// periodic_distance2_fromcartesian! (src/utils.jl:210-246).  d in/out: cartesian difference ->
// the image vector the reference leaves in `buffer` (stale on the fall-through path).
__device__ __forceinline__ double periodic_distance2_literal(const Geom& g, double& dx, double& dy, double& dz)
{
    // same operation sequence as the Julia source (StaticArrays mat-vec = plain mul/add, no FMA):
    // components the reference's wrap arithmetic makes exactly zero stay exactly zero
#pragma clang fp contract(off)
    const double* M = g.mat;
    const double* I = g.invmat;
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
    if (g.ortho || ref2 <= g.safemin2) return ref2;
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

// ------------------------------------------------------------------ brute force kernel
template <int MODE, bool POINTS>
__global__ __launch_bounds__(256) void k_bruteforce(Geom g, AtomTable atoms, RuleTable rt, Output out,
                                                     Points pts, int64_t total)
{
    __shared__ double4 s_xyzq[256];
    __shared__ int32_t s_kind[256];

    const int tid = threadIdx.x;
    const int64_t t = (int64_t)blockIdx.x * 256 + tid;
    const bool valid = t < total;
    int i = 0, j = 0, k = 0;
    double px, py, pz;
    if (POINTS) {
        const int64_t tt = valid ? t : 0;
        px = pts.xyz[3 * tt]; py = pts.xyz[3 * tt + 1]; pz = pts.xyz[3 * tt + 2];
    } else {
        const int64_t nz = g.dims[2] + 1, ny = g.dims[1] + 1;
        const int64_t tt = valid ? t : 0;
        k = (int)(tt % nz);
        j = (int)((tt / nz) % ny);
        i = out.i_begin + (int)(tt / (nz * ny));
        px = grid_coord(i, g.size[0], g.dims[0], g.shift[0]);
        py = grid_coord(j, g.size[1], g.dims[1], g.shift[1]);
        pz = grid_coord(k, g.size[2], g.dims[2], g.shift[2]);
    }

    Accum av, ac;
    accum_zero(av);
    accum_zero(ac);
    double smallest_d2 = __builtin_huge_val();

    for (int64_t base = 0; base < atoms.n; base += 256) {
        __syncthreads();
        const int64_t a = base + tid;
        if (a < atoms.n) {
            s_xyzq[tid] = atoms.xyzq[a];
            s_kind[tid] = atoms.kind ? atoms.kind[a] : -1;
        }
        __syncthreads();
        const int m = (int)((atoms.n - base < 256) ? (atoms.n - base) : 256);
        for (int q = 0; q < m; ++q) {
            const double4 A = s_xyzq[q];
            double dx = px - A.x, dy = py - A.y, dz = pz - A.z;          // src/probes.jl:81
            const double d2 = periodic_distance2_literal(g, dx, dy, dz); // :82
            if (d2 >= g.cutoff2) continue;                               // :83
            if (MODE != MODE_COULOMB) {
                const int kd = __builtin_amdgcn_readfirstlane(s_kind[q]);
                if (kd >= 0) {
                    const int rb = rt.offset[kd], re = rt.offset[kd + 1];
                    if (re > rb) {
                        double v, p1, p2, p3;
                        vdw_terms(rt.rules, rb, re, d2, v, p1, p2, p3);
                        accum_add(av, v, p1, p2, p3, dx, dy, dz);
                    }
                }
            }
            if (MODE != MODE_VDW) {
                smallest_d2 = fmin(smallest_d2, d2);                     // :108
                double v, p1, p2, p3;
                ewald_terms(g.alpha, A.w, d2, v, p1, p2, p3);
                accum_add(ac, v, p1, p2, p3, dx, dy, dz);
            }
        }
    }
    if (valid) write_results<MODE>(g, out, POINTS, t, i, j, k, av, ac, smallest_d2);
}

// ------------------------------------------------------------------ culled kernel
// Is image P (d = pos - P, |d|^2 = r2 < cutoff2, hence the unique nearest image) the one
// periodic_distance2! (src/utils.jl:226-246) returns for this pair?  On true, d is what the
// reference leaves in `buffer`.
__device__ __forceinline__ bool select_image(const Geom& g, double& dx, double& dy, double& dz, double r2)
{
    const double* M = g.mat;
    const double* I = g.invmat;
    const double gx = I[0] * dx + I[3] * dy + I[6] * dz;
    const double gy = I[1] * dx + I[4] * dy + I[7] * dz;
    const double gz = I[2] * dx + I[5] * dy + I[8] * dz;
    const double mx = floor(gx + 0.5), my = floor(gy + 0.5), mz = floor(gz + 0.5);
    if (mx == 0.0 && my == 0.0 && mz == 0.0) {
        // P is the wrapped image.  Returned directly if ortho or within safemin; otherwise the
        // neighbour search finds nothing closer (all other images are >= cutoff away) and
        // falls through with buffer = wrapped - c.
        if (!g.ortho && r2 > g.safemin2) { dx -= M[6]; dy -= M[7]; dz -= M[8]; }
        return true;
    }
    if (g.ortho) return false;           // the wrapped image (out of cutoff) is returned
    const double vx = dx - (M[0] * mx + M[3] * my + M[6] * mz);
    const double vy = dy - (M[1] * mx + M[4] * my + M[7] * mz);
    const double vz = dz - (M[2] * mx + M[5] * my + M[8] * mz);
    const double ref2 = vx * vx + vy * vy + vz * vz;
    if (ref2 <= g.safemin2) return false;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const double cx = M[3 * ax], cy = M[3 * ax + 1], cz = M[3 * ax + 2];
        const double m_ax = (ax == 0) ? mx : ((ax == 1) ? my : mz);
        const double m_o1 = (ax == 0) ? my : mx;
        const double m_o2 = (ax == 2) ? my : mz;
        const bool others_zero = (m_o1 == 0.0) && (m_o2 == 0.0);
        double ex = vx + cx, ey = vy + cy, ez = vz + cz;
        double n2 = ex * ex + ey * ey + ez * ez;
        if (n2 < ref2) return others_zero && (m_ax == 1.0);
        ex = vx - cx; ey = vy - cy; ez = vz - cz;
        n2 = ex * ex + ey * ey + ez * ez;
        if (n2 < ref2) return others_zero && (m_ax == -1.0);
    }
    return false;                        // fall-through returns |wrapped|^2 >= cutoff2
}

__device__ __forceinline__ double wave_min(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o));
    return x;
}
__device__ __forceinline__ double wave_max(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o));
    return x;
}

// pairs closer than this (A^2) take the literal min-image arithmetic in the culled kernel
constexpr double R_EXACT2 = 4.0;
constexpr int META_SIMPLE = 1 << 24;
constexpr int META_KINDMASK = (1 << 24) - 1;

template <int MODE, bool POINTS>
__global__ __launch_bounds__(64) void k_culled(Geom g, ImageBins ib, RuleTable rt, Output out, Points pts,
                                                int tiles_j, int tiles_k)
{
    __shared__ double4 s_cand[64];
    __shared__ int32_t s_meta[64];
    __shared__ int32_t s_atom[64];
    __shared__ int32_t s_rowstart[64];
    __shared__ int32_t s_rowprefix[65];

    const int lane = threadIdx.x;

    // ---- this lane's point
    int i = 0, j = 0, k = 0;
    int64_t pidx = 0;
    bool valid;
    double px, py, pz;
    if (POINTS) {
        pidx = (int64_t)blockIdx.x * 64 + lane;
        valid = pidx < pts.n;
        const int64_t tt = valid ? pidx : (pts.n - 1);
        px = pts.xyz[3 * tt]; py = pts.xyz[3 * tt + 1]; pz = pts.xyz[3 * tt + 2];
    } else {
        const int tk = blockIdx.x % tiles_k;
        const int tj = (blockIdx.x / tiles_k) % tiles_j;
        const int ti = blockIdx.x / (tiles_k * tiles_j);
        i = out.i_begin + 4 * ti + (lane >> 4);
        j = 4 * tj + ((lane >> 2) & 3);
        k = 4 * tk + (lane & 3);
        valid = (i < out.i_end) && (j <= g.dims[1]) && (k <= g.dims[2]);
        // out-of-range lanes take the tile's first point (always valid) so they stay inside the box
        const int ci = valid ? i : (out.i_begin + 4 * ti);
        const int cj = valid ? j : 4 * tj;
        const int ck = valid ? k : 4 * tk;
        px = grid_coord(ci, g.size[0], g.dims[0], g.shift[0]);
        py = grid_coord(cj, g.size[1], g.dims[1], g.shift[1]);
        pz = grid_coord(ck, g.size[2], g.dims[2], g.shift[2]);
    }

    // ---- tile box (exact hull of the lanes' points)
    const double blx = wave_min(px), bhx = wave_max(px);
    const double bly = wave_min(py), bhy = wave_max(py);
    const double blz = wave_min(pz), bhz = wave_max(pz);
    const double cx = 0.5 * (blx + bhx), cy = 0.5 * (bly + bhy), cz = 0.5 * (blz + bhz);
    const double hx = 0.5 * (bhx - blx), hy = 0.5 * (bhy - bly), hz = 0.5 * (bhz - blz);
    // cutoff with a rounding margin: anything a lane can see with r2 < cutoff2 is kept
    const double rc2 = g.cutoff2 * (1.0 + 1e-9) + 1e-9;
    const double rc = sqrt(rc2);
    // fractional half-extent of the tile box (+ margin), for the "always the wrapped image" test
    const double* I = g.invmat;
    const double e0 = fabs(I[0]) * hx + fabs(I[3]) * hy + fabs(I[6]) * hz + 1e-9;
    const double e1 = fabs(I[1]) * hx + fabs(I[4]) * hy + fabs(I[7]) * hz + 1e-9;
    const double e2 = fabs(I[2]) * hx + fabs(I[5]) * hy + fabs(I[8]) * hz + 1e-9;

    // ---- bin rows (bx, by) intersecting the neighbourhood
    auto bin_of = [&](double x, int ax) -> int {
        int b = (int)floor((x - ib.lo[ax]) * ib.inv_bin[ax]);
        b = b < 0 ? 0 : b;
        return b >= ib.nb[ax] ? ib.nb[ax] - 1 : b;
    };
    const int bx0 = bin_of(blx - rc, 0), bx1 = bin_of(bhx + rc, 0);
    const int by0 = bin_of(bly - rc, 1), by1 = bin_of(bhy + rc, 1);
    const int nrx = bx1 - bx0 + 1, nry = by1 - by0 + 1;
    const int nrows = nrx * nry;

    Accum av, ac;
    accum_zero(av);
    accum_zero(ac);
    double smallest_d2 = __builtin_huge_val();
    // widths of the "decide with the reference's arithmetic" bands (negative = band unused)
    const double band_cut = 1e-9 * g.cutoff2;
    const double band_safe = (g.ortho || g.safemin2 > g.cutoff2 * (1.0 + 1e-8)) ? -1.0 : 1e-9 * g.safemin2;

    for (int rbase = 0; rbase < nrows; rbase += 64) {
        // -- one row per lane: image range [start, start+count)
        int count = 0, start = 0;
        const int r = rbase + lane;
        if (r < nrows) {
            const int bx = bx0 + r / nry, by = by0 + r % nry;
            const double colx0 = ib.lo[0] + bx * ib.bin[0], colx1 = colx0 + ib.bin[0];
            const double coly0 = ib.lo[1] + by * ib.bin[1], coly1 = coly0 + ib.bin[1];
            const double gx = fmax(0.0, fmax(blx - colx1, colx0 - bhx));
            const double gy = fmax(0.0, fmax(bly - coly1, coly0 - bhy));
            const double dxy2 = gx * gx + gy * gy;
            if (dxy2 < rc2) {
                const double zr = sqrt(rc2 - dxy2);
                const int bz0 = bin_of(blz - zr, 2), bz1 = bin_of(bhz + zr, 2);
                const int64_t rowbase = ((int64_t)bx * ib.nb[1] + by) * ib.nb[2];
                start = ib.bin_start[rowbase + bz0];
                count = ib.bin_start[rowbase + bz1 + 1] - start;
            }
        }
        // inclusive scan of counts over the wave
        int incl = count;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        __syncthreads();                       // previous batch's readers are done
        s_rowstart[lane] = start;
        s_rowprefix[lane + 1] = incl;
        if (lane == 0) s_rowprefix[0] = 0;
        __syncthreads();
        const int total = s_rowprefix[64];

        for (int cbase = 0; cbase < total; cbase += 64) {
            // -- stage: lane loads one image of the flattened row list and tests it against the tile
            const int t = cbase + lane;
            bool keep = false;
            double4 P = make_double4(0, 0, 0, 0);
            int meta = 0, aidx = 0;
            if (t < total) {
                int lo = 0, hi = 64;           // largest lo with prefix[lo] <= t
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    const int mid = (lo + hi) >> 1;
                    if (s_rowprefix[mid] <= t) lo = mid; else hi = mid;
                }
                const int img = s_rowstart[lo] + (t - s_rowprefix[lo]);
                P = ib.xyzq[img];
                aidx = ib.atom[img];
                const int kd = ib.kind ? ib.kind[img] : -1;
                const double qx = fmax(0.0, fabs(cx - P.x) - hx);
                const double qy = fmax(0.0, fabs(cy - P.y) - hy);
                const double qz = fmax(0.0, fabs(cz - P.z) - hz);
                keep = (qx * qx + qy * qy + qz * qz) < rc2;
                if (MODE == MODE_VDW)          // kinds without a rule contribute exact zeros
                    keep = keep && (kd >= 0) && (rt.offset[kd + 1] > rt.offset[kd]);
                bool simple = g.diag != 0;
                if (!simple) {
                    const double ux = cx - P.x, uy = cy - P.y, uz = cz - P.z;
                    const double f0 = I[0] * ux + I[3] * uy + I[6] * uz;
                    const double f1 = I[1] * ux + I[4] * uy + I[7] * uz;
                    const double f2 = I[2] * ux + I[5] * uy + I[8] * uz;
                    simple = (fabs(f0) + e0 < 0.5) && (fabs(f1) + e1 < 0.5) && (fabs(f2) + e2 < 0.5);
                }
                meta = (kd & META_KINDMASK) | (simple ? META_SIMPLE : 0);
            }
            const unsigned long long mask = __ballot(keep);
            const int nkeep = __popcll(mask);
            __syncthreads();                   // previous chunk's readers are done
            if (keep) {
                const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                s_cand[slot] = P;
                s_meta[slot] = meta;
                s_atom[slot] = aidx;
            }
            __syncthreads();

            // -- every lane against every kept image (LDS broadcast reads)
            for (int q = 0; q < nkeep; ++q) {
                const double4 A = s_cand[q];
                const int mt = __builtin_amdgcn_readfirstlane(s_meta[q]);
                double dx = px - A.x, dy = py - A.y, dz = pz - A.z;
                double r2 = dx * dx + dy * dy + dz * dz;
                bool in = r2 < g.cutoff2;
                if (r2 < R_EXACT2 || fabs(r2 - g.cutoff2) <= band_cut || fabs(r2 - g.safemin2) <= band_safe) {
                    // (1) Very close pair: the radial factors are astronomically large (LJ ~ r^-14), so
                    // a one-ulp difference in a component of d that the reference's wrap arithmetic
                    // makes exactly zero would be amplified into the result.  (2) Pair within 1e-9 of
                    // a decision threshold (cutoff, safemin): the decision must be taken on the
                    // reference's own rounding of d2 (truncated potentials jump at the cutoff).
                    // Redo such pairs exactly like the reference: original atom position, invmat*d,
                    // wrap, mat*f (src/utils.jl:210-246).  ~0.5 % of the in-cutoff pairs.
                    const double4 O = ib.atoms[s_atom[q]];
                    dx = px - O.x; dy = py - O.y; dz = pz - O.z;
                    r2 = periodic_distance2_literal(g, dx, dy, dz);
                    in = r2 < g.cutoff2;
                } else if (mt & META_SIMPLE) {
                    // wrapped image for the whole tile: returned as is, except for the
                    // stale-vector fall-through when safemin2 < r2 (src/utils.jl:233-245)
                    if (!g.ortho && r2 > g.safemin2) { dx -= g.mat[6]; dy -= g.mat[7]; dz -= g.mat[8]; }
                } else if (in) {
                    in = select_image(g, dx, dy, dz, r2);
                }
                if (!in) continue;
                if (MODE != MODE_COULOMB) {
                    int kd = mt & META_KINDMASK;
                    if (kd != META_KINDMASK) {
                        const int rb = rt.offset[kd], re = rt.offset[kd + 1];
                        if (re > rb) {
                            double v, p1, p2, p3;
                            vdw_terms(rt.rules, rb, re, r2, v, p1, p2, p3);
                            accum_add(av, v, p1, p2, p3, dx, dy, dz);
                        }
                    }
                }
                if (MODE != MODE_VDW) {
                    smallest_d2 = fmin(smallest_d2, r2);
                    double v, p1, p2, p3;
                    ewald_terms(g.alpha, A.w, r2, v, p1, p2, p3);
                    accum_add(ac, v, p1, p2, p3, dx, dy, dz);
                }
            }
        }
    }
    if (valid) write_results<MODE>(g, out, POINTS, pidx, i, j, k, av, ac, smallest_d2);
}

// ------------------------------------------------------------------ launchers
template <bool POINTS>
static hipError_t launch_bf_t(int mode, const Geom& g, const AtomTable& atoms, const RuleTable& rt,
                              const Output& out, const Points& pts, int64_t total, hipStream_t stream)
{
    if (total <= 0) return hipSuccess;
    const int64_t nblocks = (total + 255) / 256;
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)nblocks), block(256);
    switch (mode) {
    case MODE_VDW:
        hipLaunchKernelGGL((k_bruteforce<MODE_VDW, POINTS>), grid, block, 0, stream, g, atoms, rt, out, pts, total);
        break;
    case MODE_COULOMB:
        hipLaunchKernelGGL((k_bruteforce<MODE_COULOMB, POINTS>), grid, block, 0, stream, g, atoms, rt, out, pts, total);
        break;
    default:
        hipLaunchKernelGGL((k_bruteforce<MODE_FUSED, POINTS>), grid, block, 0, stream, g, atoms, rt, out, pts, total);
        break;
    }
    return hipGetLastError();
}

hipError_t launch_bruteforce(int mode, const Geom& g, const AtomTable& atoms, const RuleTable& rt,
                             const Output& out, const Points& pts, hipStream_t stream)
{
    if (pts.xyz) return launch_bf_t<true>(mode, g, atoms, rt, out, pts, pts.n, stream);
    const int64_t total = (int64_t)(out.i_end - out.i_begin) * (g.dims[1] + 1) * (g.dims[2] + 1);
    return launch_bf_t<false>(mode, g, atoms, rt, out, pts, total, stream);
}

template <bool POINTS>
static hipError_t launch_cull_t(int mode, const Geom& g, const ImageBins& ib, const RuleTable& rt,
                                const Output& out, const Points& pts, int64_t nblocks, int tj, int tk,
                                hipStream_t stream)
{
    if (nblocks <= 0) return hipSuccess;
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)nblocks), block(64);
    switch (mode) {
    case MODE_VDW:
        hipLaunchKernelGGL((k_culled<MODE_VDW, POINTS>), grid, block, 0, stream, g, ib, rt, out, pts, tj, tk);
        break;
    case MODE_COULOMB:
        hipLaunchKernelGGL((k_culled<MODE_COULOMB, POINTS>), grid, block, 0, stream, g, ib, rt, out, pts, tj, tk);
        break;
    default:
        hipLaunchKernelGGL((k_culled<MODE_FUSED, POINTS>), grid, block, 0, stream, g, ib, rt, out, pts, tj, tk);
        break;
    }
    return hipGetLastError();
}

hipError_t launch_culled(int mode, const Geom& g, const ImageBins& ib, const RuleTable& rt,
                         const Output& out, const Points& pts, hipStream_t stream)
{
    if (pts.xyz) return launch_cull_t<true>(mode, g, ib, rt, out, pts, (pts.n + 63) / 64, 1, 1, stream);
    const int ni = out.i_end - out.i_begin;
    const int ti = (ni + 3) / 4, tj = (g.dims[1] + 1 + 3) / 4, tk = (g.dims[2] + 1 + 3) / 4;
    return launch_cull_t<false>(mode, g, ib, rt, out, pts, (int64_t)ti * tj * tk, tj, tk, stream);
}

}  // namespace ceg
