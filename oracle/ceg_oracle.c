/*
 * ceg_oracle.c -- CPU restatement (plain C, FP64) of the grid-build hot path of
 * CrystalEnergyGrids.jl.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker the HIP kernels are compared against.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the
 * product path (libceg_hip.so) never links, loads or falls back to it.
 *
 * Pinning status: the Julia reference cannot run here (no Julia toolchain) and it ships no
 * .grid golden file (grids are generated at test time, .gitignore:4-5).  The restatement is
 * pinned through the literals of test/runtests.jl (tests/golden/pins.json,
 * tests/test_reference_pins.py): the reference's own tolerance is rtol 1e-3; what this file +
 * the host mirror actually reproduce is 4e-16 (Na/CHA VdW, runtests.jl:44), 1e-16 (Ar/CHA+Na
 * minimum, :38), 2e-8 (Ar/CIT-7 triclinic supercell, :169, a 12-digit literal) and 5e-10 for the
 * Coulomb value (:45, limited by third-party constants / erfc).  Analytic checks:
 * tests/test_oracle_analytic.py.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference).  Loop order, operation order and branch structure follow
 * the Julia source; build with -ffp-contract=off so no FMA is introduced that
 * the source does not spell.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/ceg_hip.h"   /* ceg_rule_t and the kind enum only */

#define ORACLE_API __attribute__((visibility("default")))

/* ---- src/utils.jl:226-246  periodic_distance2!(buffer, mat, ortho, safemin2, buffer2)
 * mat is column-major: mat[i + 3*k] = M[i,k].
 * On return buffer holds the cartesian image vector *as the reference leaves it*:
 * on the fall-through path (no closer single-axis image found) that is the last
 * trial image (f3 - 1), not the wrapped one, reproduced here. */
static inline void matvec3(double out[3], const double m[9], const double v[3])
{
    /* StaticArrays mul!: out[i] = sum_k M[i,k]*v[k], k ascending */
    for (int i = 0; i < 3; ++i)
        out[i] = (m[i] * v[0] + m[i + 3] * v[1]) + m[i + 6] * v[2];
}

static inline double norm2_3(const double u[3])
{
    /* src/utils.jl:190-196 norm2(u): r2 += x^2 in order */
    double r2 = 0.0;
    r2 += u[0] * u[0];
    r2 += u[1] * u[1];
    r2 += u[2] * u[2];
    return r2;
}

static double periodic_distance2(double buffer[3], const double mat[9], int ortho,
                                 double safemin2, double buffer2[3])
{
    for (int i = 0; i < 3; ++i) {               /* utils.jl:227-230 */
        double diff = buffer2[i] + 0.5;
        buffer2[i] = diff - floor(diff) - 0.5;
    }
    matvec3(buffer, mat, buffer2);              /* :231 */
    double ref2 = norm2_3(buffer);              /* :232 */
    if (ortho || ref2 <= safemin2) return ref2; /* :233 */
    for (int i = 0; i < 3; ++i) {               /* :234-244 */
        buffer2[i] += 1;
        matvec3(buffer, mat, buffer2);
        double newnorm2 = norm2_3(buffer);
        if (newnorm2 < ref2) return newnorm2;
        buffer2[i] -= 2;
        matvec3(buffer, mat, buffer2);
        newnorm2 = norm2_3(buffer);
        if (newnorm2 < ref2) return newnorm2;
        buffer2[i] += 1;
    }
    return ref2;                                /* :245 */
}

/* ---- src/utils.jl:210-213 periodic_distance2_fromcartesian! */
static double periodic_distance2_fromcartesian(double buffer[3], const double mat[9],
                                               const double invmat[9], int ortho,
                                               double safemin2, double buffer2[3])
{
    matvec3(buffer2, invmat, buffer);
    return periodic_distance2(buffer, mat, ortho, safemin2, buffer2);
}

ORACLE_API double oracle_periodic_distance2_fromcartesian(double buffer[3], const double mat[9],
                                                          const double invmat[9], int ortho,
                                                          double safemin2)
{
    double buffer2[3];
    return periodic_distance2_fromcartesian(buffer, mat, invmat, ortho, safemin2, buffer2);
}

/* ---- src/interactions.jl:432-472 derivativesGrid(rule::InteractionRule, d2)
 * returns 0 ok, <0 for the kinds on which the reference throws. */
static int derivatives_grid_rule(const ceg_rule_t* rule, double r2, double out[4])
{
    double value, d1, d2, d3;
    switch (rule->kind) {
    case CEG_LENNARDJONES: {                       /* :434-441 */
        double eps = rule->p[0], sigma = rule->p[1];
        double s = (sigma * sigma) / r2;
        double x6 = s * s * s;                      /* (σ^2/r2)^3 = Base.literal_pow: x*x*x */
        double r4 = r2 * r2;
        value = 4 * eps * x6 * (x6 - 1);
        d1 = 24 * eps * (x6 * (1 - 2 * x6)) / r2;
        d2 = 96 * eps * (x6 * (7 * x6 - 2)) / r4;
        d3 = 384 * eps * (x6 * (5 - 28 * x6)) / (r4 * r4);
        break;
    }
    case CEG_COULOMB:                               /* :442-443 error(...) */
        return CEG_ERR_RULE;
    case CEG_HARDSPHERE: {                          /* :444-446 */
        double rr = rule->p[0] + rule->p[1];
        value = (r2 < rr * rr) ? INFINITY : 0.0;
        d1 = d2 = d3 = 0.0;
        break;
    }
    case CEG_BUCKINGHAM: {                          /* :447-457 */
        double A = rule->p[0], B = rule->p[1], C = rule->p[2];
        double r4 = r2 * r2;
        double r = sqrt(r2);
        double r6 = r4 * r2;
        double x6 = C / r6;
        double xe = A * exp(-B * r);
        value = xe - x6;
        d1 = -B * xe / r + 6 * x6 / r2;
        d2 = -48 * x6 / r4 + B * xe * (1 + B * r) / (r2 * r);
        d3 = -(3 * B * r + B * B * r2 + 3) * B * xe * r / r6 + 480 * C / (r6 * r6);
        break;
    }
    case CEG_NOINTERACTION:
    case CEG_COULOMB_EWALD_DIRECT:                  /* :458-461 early return, shift not applied */
        out[0] = out[1] = out[2] = out[3] = 0.0;
        return 0;
    case CEG_MONOMIAL:                              /* :462-465 error(...) */
    case CEG_EXPONENTIAL:
    case CEG_UNDEFINED_INTERACTION:                 /* :466-467 throw */
    default:
        return CEG_ERR_RULE;
    }
    out[0] = value - rule->shift;                   /* :471 */
    out[1] = d1;
    out[2] = d2;
    out[3] = d3;
    return 0;
}

/* ---- src/interactions.jl:599-610 derivativesGrid(f::InteractionRuleSum, d2)
 *      src/forcefields.jl:302-304 derivatives_nocutoff
 * A single InteractionRule is a run of length 1 (0 + x == x exactly for the
 * finite, inf and nan values that occur, except -0.0 which cannot be observed
 * after accumulation). */
static int derivatives_nocutoff(const ceg_rule_t* rules, int nrules, double r2, double out[4])
{
    if (nrules == 1) return derivatives_grid_rule(rules, r2, out);
    double value = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
    for (int x = 0; x < nrules; ++x) {
        double t[4];
        int rc = derivatives_grid_rule(rules + x, r2, t);
        if (rc) return rc;
        value += t[0];
        d1 += t[1];
        d2 += t[2];
        d3 += t[3];
    }
    out[0] = value; out[1] = d1; out[2] = d2; out[3] = d3;
    return 0;
}

ORACLE_API int oracle_derivatives_grid(const ceg_rule_t* rules, int nrules, double r2, double out[4])
{
    return derivatives_nocutoff(rules, nrules, r2, out);
}

/* ---- src/ewald.jl:299-312 derivatives_ewald(ewald, charge, r2) */
static void derivatives_ewald(double alpha, double charge, double r2, double out[4])
{
    const double sqrtpi = 1.7724538509055160273; /* sqrt(π) */
    double r = sqrt(r2);
    double r3 = r2 * r;
    double r5 = r3 * r2;
    double a = alpha;
    double r2a2 = r2 * (a * a);
    double er2a2 = 2 * a * r * exp(-r2a2) / sqrtpi;
    double erfar = erfc(a * r);
    out[0] = charge * erfar / r;
    out[1] = -charge * (er2a2 + erfar) / r3;
    out[2] = charge * (er2a2 * (3 + 2 * r2a2) + 3 * erfar) / r5;
    out[3] = charge * (-er2a2 * (15 + 10 * r2a2 + 4 * (r2a2 * r2a2)) - 15 * erfar) / (r5 * r2);
}

ORACLE_API void oracle_derivatives_ewald(double alpha, double charge, double r2, double out[4])
{
    derivatives_ewald(alpha, charge, r2, out);
}

/* ---- src/probes.jl:71-92 compute_derivatives_vdw(s, pos)
 * out8 = value, d1[3], d2[3], d3 */
ORACLE_API int oracle_compute_derivatives_vdw(
    const double* positions, const int64_t* atomkinds, int64_t natoms,
    const double mat[9], const double invmat[9], int ortho, double safemin2, double cutoff2,
    const ceg_rule_t* rules, const int32_t* rule_offset,
    const double pos[3], double out8[8])
{
    double buffer[3], buffer2[3];
    double value = 0.0, d1[3] = {0, 0, 0}, d2[3] = {0, 0, 0}, d3 = 0.0;
    for (int64_t i = 0; i < natoms; ++i) {                       /* :80 */
        buffer[0] = pos[0] - positions[3 * i + 0];               /* :81 */
        buffer[1] = pos[1] - positions[3 * i + 1];
        buffer[2] = pos[2] - positions[3 * i + 2];
        double dd2 = periodic_distance2_fromcartesian(buffer, mat, invmat, ortho, safemin2, buffer2);
        if (dd2 >= cutoff2) continue;                            /* :83 */
        int64_t k = atomkinds[i] - 1;
        double t[4];
        int rc = derivatives_nocutoff(rules + rule_offset[k], rule_offset[k + 1] - rule_offset[k], dd2, t);
        if (rc) return rc;
        value += t[0];                                           /* :85 */
        d1[0] += t[1] * buffer[0];                               /* :86 */
        d1[1] += t[1] * buffer[1];
        d1[2] += t[1] * buffer[2];
        double d13 = buffer[0] * buffer[2];                      /* :87 */
        d2[0] += t[2] * (buffer[0] * buffer[1]);                 /* :88 */
        d2[1] += t[2] * d13;
        d2[2] += t[2] * (buffer[1] * buffer[2]);
        d3 += t[3] * d13 * buffer[1];                            /* :89 */
    }
    out8[0] = value;
    out8[1] = d1[0]; out8[2] = d1[1]; out8[3] = d1[2];
    out8[4] = d2[0]; out8[5] = d2[1]; out8[6] = d2[2];
    out8[7] = d3;
    return 0;
}

/* ---- src/probes.jl:94-117 compute_derivatives_ewald(s, ewald, pos) */
ORACLE_API int oracle_compute_derivatives_ewald(
    const double* positions, const double* charges, int64_t natoms,
    const double mat[9], const double invmat[9], int ortho, double safemin2, double cutoff2,
    double alpha, const double pos[3], double out8[8])
{
    double buffer[3], buffer2[3];
    double value = 0.0, d1[3] = {0, 0, 0}, d2[3] = {0, 0, 0}, d3 = 0.0;
    double smallest_d2 = INFINITY;                               /* :104 */
    for (int64_t i = 0; i < natoms; ++i) {
        buffer[0] = pos[0] - positions[3 * i + 0];
        buffer[1] = pos[1] - positions[3 * i + 1];
        buffer[2] = pos[2] - positions[3 * i + 2];
        double dd2 = periodic_distance2_fromcartesian(buffer, mat, invmat, ortho, safemin2, buffer2);
        if (dd2 >= cutoff2) continue;
        smallest_d2 = (dd2 < smallest_d2) ? dd2 : smallest_d2;   /* :108 min */
        double t[4];
        derivatives_ewald(alpha, charges[i], dd2, t);
        value += t[0];
        d1[0] += t[1] * buffer[0];
        d1[1] += t[1] * buffer[1];
        d1[2] += t[1] * buffer[2];
        double d13 = buffer[0] * buffer[2];
        d2[0] += t[2] * (buffer[0] * buffer[1]);
        d2[1] += t[2] * d13;
        d2[2] += t[2] * (buffer[1] * buffer[2]);
        d3 += t[3] * d13 * buffer[1];
    }
    out8[0] = (smallest_d2 < 1.0) ? INFINITY : value;            /* :116 */
    out8[1] = d1[0]; out8[2] = d1[1]; out8[3] = d1[2];
    out8[4] = d2[0]; out8[5] = d2[1]; out8[6] = d2[2];
    out8[7] = d3;
    return 0;
}

/* ---- src/grids.jl:118-135 _set_gridpoint!(grid,i,j,k,Δ,λ,λ⁻¹e7,derivatives)
 * npts = (dims[2]+1)*(dims[1]+1)*(dims[0]+1): channel stride of the column-major
 * [z,y,x,c] array; idx = k + nz*(j + ny*i). */
static inline double clamp_julia(double x, double lo, double hi)
{
    /* Base.clamp: ifelse(x > hi, hi, ifelse(x < lo, lo, x)) -- NaN passes through */
    return (x > hi) ? hi : ((x < lo) ? lo : x);
}

ORACLE_API void oracle_set_gridpoint(float* grid, int64_t idx, int64_t npts, const double delta[3],
                                     double lambda, double thr, const double d[8])
{
    double value = d[0];
    double d1[3] = {d[1], d[2], d[3]};
    double d2[3] = {d[4], d[5], d[6]};
    double d3 = d[7];
    if (value > thr) {                                   /* :120-125 */
        value = 2 * thr;
        d1[0] = clamp_julia(d1[0], -thr, thr);
        d1[1] = clamp_julia(d1[1], -thr, thr);
        d1[2] = clamp_julia(d1[2], -thr, thr);
        d2[0] = d2[1] = d2[2] = 0.0;
        d3 = 0.0;
    }
    grid[idx + 0 * npts] = (float)(value * lambda);                              /* :126 */
    grid[idx + 1 * npts] = (float)(d1[0] * delta[0] * lambda);                   /* :127 */
    grid[idx + 2 * npts] = (float)(d1[1] * delta[1] * lambda);
    grid[idx + 3 * npts] = (float)(d1[2] * delta[2] * lambda);
    grid[idx + 4 * npts] = (float)(d2[0] * (delta[0] * delta[1]) * lambda);      /* :130 */
    grid[idx + 5 * npts] = (float)(d2[1] * (delta[0] * delta[2]) * lambda);
    grid[idx + 6 * npts] = (float)(d2[2] * (delta[1] * delta[2]) * lambda);
    grid[idx + 7 * npts] = (float)(d3 * (delta[0] * delta[1] * delta[2]) * lambda); /* :133 */
}

/* ---- src/coordinates.jl:72-76 abc_to_xyz: (i*size)/dims + shift */
static inline void abc_to_xyz(const int32_t dims[3], const double size[3], const double shift[3],
                              int i, int j, int k, double pos[3])
{
    pos[0] = (double)i * size[0] / (double)dims[0] + shift[0];
    pos[1] = (double)j * size[1] / (double)dims[1] + shift[1];
    pos[2] = (double)k * size[2] / (double)dims[2] + shift[2];
}

ORACLE_API void oracle_abc_to_xyz(const int32_t dims[3], const double size[3], const double shift[3],
                                  int32_t i, int32_t j, int32_t k, double pos[3])
{
    abc_to_xyz(dims, size, shift, i, j, k, pos);
}

/* ---- src/grids.jl:144-150 loop nest of create_grid_vdw, restricted to x-planes
 * [i_begin, i_end) and y-rows [j_begin, j_end) so a bounded sample can be timed (the full
 * grid is i in [0, dims[0]+1), j in [0, dims[1]+1)); threaded over i like Threads.@threads
 * (grids.jl:144).  raw8 (optional, may be NULL) receives the
 * FP64 derivatives of every point before _set_gridpoint!, [8*idx + c]. */
ORACLE_API int oracle_grid_vdw(
    const double* positions, const int64_t* atomkinds, int64_t natoms,
    const double mat[9], const double invmat[9], int ortho, double safemin2, double cutoff2,
    const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds,
    const int32_t dims[3], const double size[3], const double shift[3], const double delta[3],
    double lambda, double thr, int32_t i_begin, int32_t i_end, int32_t j_begin, int32_t j_end,
    float* grid, double* raw8, int32_t nthreads)
{
    (void)nkinds;
    const int64_t nz = dims[2] + 1, ny = dims[1] + 1, nx = dims[0] + 1;
    const int64_t npts = nz * ny * nx;
    int err = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = i_begin; i < i_end; ++i) {
        for (int j = j_begin; j < j_end; ++j)
            for (int k = 0; k <= dims[2]; ++k) {
                double pos[3], d[8];
                abc_to_xyz(dims, size, shift, i, j, k, pos);
                int rc = oracle_compute_derivatives_vdw(positions, atomkinds, natoms, mat, invmat,
                                                        ortho, safemin2, cutoff2, rules, rule_offset,
                                                        pos, d);
                if (rc) { err = rc; continue; }
                int64_t idx = k + nz * (j + ny * (int64_t)i);
                if (raw8) memcpy(raw8 + 8 * idx, d, sizeof d);
                if (grid) oracle_set_gridpoint(grid, idx, npts, delta, lambda, thr, d);
            }
    }
    return err;
}

/* ---- src/grids.jl:171-177 loop nest of create_grid_coulomb */
ORACLE_API int oracle_grid_coulomb(
    const double* positions, const double* charges, int64_t natoms,
    const double mat[9], const double invmat[9], int ortho, double safemin2, double cutoff2,
    double alpha,
    const int32_t dims[3], const double size[3], const double shift[3], const double delta[3],
    double lambda, double thr, int32_t i_begin, int32_t i_end, int32_t j_begin, int32_t j_end,
    float* grid, double* raw8, int32_t nthreads)
{
    const int64_t nz = dims[2] + 1, ny = dims[1] + 1, nx = dims[0] + 1;
    const int64_t npts = nz * ny * nx;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = i_begin; i < i_end; ++i) {
        for (int j = j_begin; j < j_end; ++j)
            for (int k = 0; k <= dims[2]; ++k) {
                double pos[3], d[8];
                abc_to_xyz(dims, size, shift, i, j, k, pos);
                oracle_compute_derivatives_ewald(positions, charges, natoms, mat, invmat, ortho,
                                                 safemin2, cutoff2, alpha, pos, d);
                int64_t idx = k + nz * (j + ny * (int64_t)i);
                if (raw8) memcpy(raw8 + 8 * idx, d, sizeof d);
                if (grid) oracle_set_gridpoint(grid, idx, npts, delta, lambda, thr, d);
            }
    }
    return 0;
}

/* compute_derivatives_* at a list of arbitrary points (for sampled fixtures) */
ORACLE_API int oracle_points_vdw(
    const double* positions, const int64_t* atomkinds, int64_t natoms,
    const double mat[9], const double invmat[9], int ortho, double safemin2, double cutoff2,
    const ceg_rule_t* rules, const int32_t* rule_offset,
    const double* points, int64_t npoints, double* out8, int32_t nthreads)
{
    int err = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < npoints; ++p) {
        int rc = oracle_compute_derivatives_vdw(positions, atomkinds, natoms, mat, invmat, ortho,
                                                safemin2, cutoff2, rules, rule_offset,
                                                points + 3 * p, out8 + 8 * p);
        if (rc) err = rc;
    }
    return err;
}

ORACLE_API int oracle_points_coulomb(
    const double* positions, const double* charges, int64_t natoms,
    const double mat[9], const double invmat[9], int ortho, double safemin2, double cutoff2,
    double alpha, const double* points, int64_t npoints, double* out8, int32_t nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < npoints; ++p)
        oracle_compute_derivatives_ewald(positions, charges, natoms, mat, invmat, ortho, safemin2,
                                         cutoff2, alpha, points + 3 * p, out8 + 8 * p);
    return 0;
}

/* ---- src/grids.jl:212-273 interpolate_grid(g, point) with src/coordinates.jl:58-66
 * (wrap_atom, offsetpoint).  grid: float[8][nx][ny][nz] already in K; mat/invmat: unit-cell
 * matrix (column-major); coeff: the 64x64 matrix COEFF of src/constants.jl:24-89, row-major,
 * passed in by the caller.  Literal evaluation: a = COEFF*X, then the 64-term polynomial in the
 * reference's loop order. */
ORACLE_API double oracle_interpolate_grid(const float* grid, const int32_t dims[3], const double size[3],
                                          const double shift[3], const double mat[9], const double invmat[9],
                                          int is_vdw, const double* coeff, const double point[3])
{
    double abc[3], frac[3], np_[3], shifted[3];
    matvec3(abc, invmat, point);                               /* coordinates.jl:59 */
    for (int i = 0; i < 3; ++i) frac[i] = abc[i] - floor(abc[i]);
    matvec3(np_, mat, frac);                                   /* :60 */
    for (int i = 0; i < 3; ++i)                                /* :65 */
        shifted[i] = (np_[i] - shift[i]) * (double)dims[i] / size[i] + 1;
    const int64_t nx = dims[0] + 1, ny = dims[1] + 1, nz = dims[2] + 1;
    const int64_t ext[3] = {nx, ny, nz};
    int64_t p0[3], p1[3];
    double r[3];
    for (int i = 0; i < 3; ++i) {                              /* grids.jl:216-219 */
        p0[i] = (int64_t)floor(shifted[i]);
        p1[i] = p0[i] + (p0[i] != ext[i]);
        r[i] = shifted[i] - (double)p0[i];
    }
    const int64_t x0 = p0[0] - 1, y0 = p0[1] - 1, z0 = p0[2] - 1, x1 = p1[0] - 1, y1 = p1[1] - 1, z1 = p1[2] - 1;
    const int64_t cs = nx * ny * nz;
#define G_(c, x, y, z) grid[(c) * cs + ((x) * ny + (y)) * nz + (z)]
    float X[64];
    for (int c = 0; c < 8; ++c) {                              /* :227-244 */
        X[8 * c + 0] = G_(c, x0, y0, z0); X[8 * c + 1] = G_(c, x1, y0, z0);
        X[8 * c + 2] = G_(c, x0, y1, z0); X[8 * c + 3] = G_(c, x1, y1, z0);
        X[8 * c + 4] = G_(c, x0, y0, z1); X[8 * c + 5] = G_(c, x1, y0, z1);
        X[8 * c + 6] = G_(c, x0, y1, z1); X[8 * c + 7] = G_(c, x1, y1, z1);
    }
#undef G_
    if (is_vdw)                                                /* :245-248 */
        for (int t = 0; t < 8; ++t)
            if (X[t] > 5e6f) return 1e100;
    double a[64];
    for (int row = 0; row < 64; ++row) {                       /* :252 mul!(a, COEFF, X) */
        double acc = 0.0;
        for (int col = 0; col < 64; ++col) acc += coeff[row * 64 + col] * (double)X[col];
        a[row] = acc;
    }
    const double rx = r[0], ry = r[1], rz = r[2];
    const double rx2 = rx * rx, ry2 = ry * ry, rz2 = rz * rz;
    const double rxs[4] = {1.0, rx, rx2, rx2 * rx};
    const double rys[4] = {1.0, ry, ry2, ry2 * ry};
    const double rzs[4] = {1.0, rz, rz2, rz2 * rz};
    double ret = 0.0;
    for (int k = 0; k < 4; ++k)                                /* :256-258 */
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 4; ++i)
                ret += a[i + 4 * j + 16 * k] * rxs[i] * rys[j] * rzs[k];
    return ret;
}

/* interpolate_grid, branch `higherorder == false` (src/grids.jl:259-269): channel 1 only, trilinear weights, no blocking rule,
 * sum in the reference's order.  The reference writes g.grid[x0,y0,z0,1] ... although the array is [z, y, x, channel]
 * (:126-133, :227-244): the first array index, which runs along z, gets the x cell index and the third, which runs along x,
 * the z cell index.  Restated as written; an index outside the axis it lands on is a BoundsError in Julia -> NaN here. */
ORACLE_API double oracle_interpolate_grid_noderiv(const float* grid, const int32_t dims[3], const double size[3],
                                                  const double shift[3], const double mat[9], const double invmat[9],
                                                  const double point[3])
{
    double abc[3], frac[3], np_[3], shifted[3];
    matvec3(abc, invmat, point);
    for (int i = 0; i < 3; ++i) frac[i] = abc[i] - floor(abc[i]);
    matvec3(np_, mat, frac);
    for (int i = 0; i < 3; ++i) shifted[i] = (np_[i] - shift[i]) * (double)dims[i] / size[i] + 1;
    const int64_t nx = dims[0] + 1, ny = dims[1] + 1, nz = dims[2] + 1;
    const int64_t ext[3] = {nx, ny, nz};
    int64_t p0[3], p1[3];
    double r[3];
    for (int i = 0; i < 3; ++i) {
        p0[i] = (int64_t)floor(shifted[i]);
        p1[i] = p0[i] + (p0[i] != ext[i]);
        r[i] = shifted[i] - (double)p0[i];
    }
    const double rx = r[0], ry = r[1], rz = r[2], mrx = 1 - rx, mry = 1 - ry, mrz = 1 - rz;
    const int64_t x0 = p0[0], y0 = p0[1], z0 = p0[2], x1 = p1[0], y1 = p1[1], z1 = p1[2];
    /* g.grid[a, b, c, 1] of the [z, y, x, channel] array, 1-based: a runs along z, b along y, c along x */
#define GA_(a, b, c) (((a) < 1 || (a) > nz || (b) < 1 || (b) > ny || (c) < 1 || (c) > nx) ? (double)NAN \
                      : (double)grid[(((c) - 1) * ny + ((b) - 1)) * nz + ((a) - 1)])
    double ret = GA_(x0, y0, z0) * mrx * mry * mrz + GA_(x1, y0, z0) * rx * mry * mrz;
    ret = ret + GA_(x0, y1, z0) * mrx * ry * mrz;
    ret = ret + GA_(x0, y0, z1) * mrx * mry * rz;
    ret = ret + GA_(x1, y1, z0) * rx * ry * mrz;
    ret = ret + GA_(x1, y0, z1) * rx * mry * rz;
    ret = ret + GA_(x0, y1, z1) * mrx * ry * rz;
    ret = ret + GA_(x1, y1, z1) * rx * ry * rz;
#undef GA_
    return ret;
}

ORACLE_API void oracle_interpolate_points_noderiv(const float* grid, const int32_t dims[3], const double size[3],
                                                  const double shift[3], const double mat[9], const double invmat[9],
                                                  const double* points, int64_t n, double* out, int32_t nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p)
        out[p] = oracle_interpolate_grid_noderiv(grid, dims, size, shift, mat, invmat, points + 3 * p);
}

ORACLE_API void oracle_interpolate_points(const float* grid, const int32_t dims[3], const double size[3],
                                          const double shift[3], const double mat[9], const double invmat[9],
                                          int is_vdw, const double* coeff, const double* points, int64_t n,
                                          double* out, int32_t nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p)
        out[p] = oracle_interpolate_grid(grid, dims, size, shift, mat, invmat, is_vdw, coeff, points + 3 * p);
}

/* ---- reciprocal-space Ewald energy of ONE rigid molecule in the framework (row f2):
 * compute_ewald(ctx) (src/ewald.jl:555-577) after move_one_system!(ctx, 1, positions) (:352-366)
 * for a context holding a single molecule, i.e. the `coulomb_reciprocal` term of energy_point
 * (src/grids.jl:319-325).  Tables as make_line_pos!/make_line_neg! build them (:73-92, powers by
 * repeated multiplication), sums in the order of ewald_main_loop! (:148-185).
 *  kind[t] = (j, k, i_first, i_last, rangeidx) rows of kspace.kindices (:213-236)
 *  invmat  column-major inverse of the supercell matrix (eframework.invmat)
 *  energy_net_charges, static_contribution: the two context constants (:497-544) */
#include <complex.h>
ORACLE_API double oracle_reciprocal_energy(const int32_t* kind, int64_t nkind, const int32_t ks[3],
                                           const double* kfactors, const double* sf_re, const double* sf_im,
                                           int64_t num_kvecs, const double invmat[9],
                                           const double* positions, const double* charges, int32_t natoms,
                                           double energy_net_charges, double static_contribution)
{
    const int kx = ks[0], ky = ks[1], kz = ks[2];
    const int kxp = kx + 1, tkyp = 2 * ky + 1, tkzp = 2 * kz + 1;
    double complex* Eikx = malloc(sizeof(double complex) * kxp * natoms);
    double complex* Eiky = malloc(sizeof(double complex) * tkyp * natoms);
    double complex* Eikz = malloc(sizeof(double complex) * tkzp * natoms);
    double complex* sums = calloc(num_kvecs, sizeof(double complex));
    const double twopi = 6.283185307179586476925286766559;
    for (int a = 0; a < natoms; ++a) {
        double f[3];
        matvec3(f, invmat, positions + 3 * a);                     /* :359 */
        const double complex ex = cexp(I * (twopi * f[0])), ey = cexp(I * (twopi * f[1])), ez = cexp(I * (twopi * f[2]));
        double complex* X = Eikx + (size_t)kxp * a;                /* make_line_pos! :73-81 */
        X[0] = 1.0;
        if (kxp > 1) X[1] = ex;
        for (int i = 2; i < kxp; ++i) X[i] = X[i - 1] * ex;
        double complex* Y = Eiky + (size_t)tkyp * a;               /* make_line_neg! :83-91 */
        if (ky > 0) {
            Y[ky - 1] = conj(ey);
            for (int i = ky - 2; i >= 0; --i) Y[i] = Y[i + 1] * conj(ey);
        }
        Y[ky] = 1.0;
        if (ky > 0) Y[ky + 1] = ey;
        for (int i = ky + 2; i < tkyp; ++i) Y[i] = Y[i - 1] * ey;
        double complex* Z = Eikz + (size_t)tkzp * a;
        if (kz > 0) {
            Z[kz - 1] = conj(ez);
            for (int i = kz - 2; i >= 0; --i) Z[i] = Z[i + 1] * conj(ez);
        }
        Z[kz] = 1.0;
        if (kz > 0) Z[kz + 1] = ez;
        for (int i = kz + 2; i < tkzp; ++i) Z[i] = Z[i - 1] * ez;
    }
    for (int a = 0; a < natoms; ++a) {                             /* ewald_main_loop! :158-176 */
        const double c = charges[a];
        for (int64_t t = 0; t < nkind; ++t) {
            const int jy = kind[5 * t], jz = kind[5 * t + 1], i0 = kind[5 * t + 2], i1 = kind[5 * t + 3];
            const int64_t ridx = kind[5 * t + 4];
            const double complex eyz = c * Eiky[(size_t)tkyp * a + ky + jy] * Eikz[(size_t)tkzp * a + kz + jz];
            for (int i = i0; i <= i1; ++i) sums[ridx + (i - i0)] += Eikx[(size_t)kxp * a + i] * eyz;
        }
    }
    double framework_adsorbate = 0.0, adsorbate_adsorbate = 0.0;   /* compute_ewald :563-570 */
    for (int64_t q = 0; q < num_kvecs; ++q) {
        const double temp = kfactors[q];
        const double re_a = creal(sums[q]), im_a = cimag(sums[q]);
        framework_adsorbate += temp * (sf_re[q] * re_a + sf_im[q] * im_a);
        adsorbate_adsorbate += temp * (re_a * re_a + im_a * im_a);
    }
    free(Eikx); free(Eiky); free(Eikz); free(sums);
    return 2 * (framework_adsorbate + energy_net_charges) + (adsorbate_adsorbate + static_contribution);
}

ORACLE_API void oracle_reciprocal_energies(const int32_t* kind, int64_t nkind, const int32_t ks[3],
                                           const double* kfactors, const double* sf_re, const double* sf_im,
                                           int64_t num_kvecs, const double invmat[9],
                                           const double* positions, const double* charges, int32_t natoms, int64_t n,
                                           double energy_net_charges, double static_contribution, double* out,
                                           int32_t nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p)
        out[p] = oracle_reciprocal_energy(kind, nkind, ks, kfactors, sf_re, sf_im, num_kvecs, invmat,
                                          positions + 3 * (size_t)natoms * p, charges, natoms,
                                          energy_net_charges, static_contribution);
}

ORACLE_API int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ======================================================================================
 * Row f3: guest-guest pair energy of a rigid molecule at a trial placement.
 * src/energy.jl:407-427 single_contribution_vdw_noneighbour (rigid molecule: no intra term),
 * src/utils.jl:294-302 unsafe_periodic_distance2!, src/interactions.jl:367-406 rule energies
 * (r2 forms :392-406 for LJ / HardSphere / NoInteraction / Monomial, r forms otherwise),
 * :589-595 for rule sums.
 * ====================================================================================== */
static double rule_energy_r(const ceg_rule_t* R, double r, double coulombic)
{   /* interactions.jl:367-390 */
    double v;
    switch (R->kind) {
    case CEG_LENNARDJONES: { double x6 = pow(R->p[1] / r, 6); v = 4 * R->p[0] * x6 * (x6 - 1); break; }
    case CEG_COULOMB_EWALD_DIRECT: v = coulombic * R->p[1] * R->p[2] * erfc(R->p[0] * r) / r; break;
    case CEG_COULOMB: v = coulombic * R->p[0] * R->p[1] / r; break;
    case CEG_HARDSPHERE: v = (r < R->p[0] + R->p[1]) ? INFINITY : 0.0; break;
    case CEG_BUCKINGHAM: v = R->p[0] * exp(-R->p[1] * r) - R->p[2] / pow(r, 6); break;
    case CEG_NOINTERACTION: v = 0.0; break;
    case CEG_MONOMIAL: v = R->p[0] / pow(r, R->p[1]); break;
    case CEG_EXPONENTIAL: v = R->p[0] * exp(-R->p[1] * r); break;
    default: v = NAN; break;
    }
    return v - R->shift;
}

static double rule_energy_r2(const ceg_rule_t* R, double r2, double coulombic)
{   /* interactions.jl:392-406 */
    switch (R->kind) {
    case CEG_LENNARDJONES: { double s2 = R->p[1] * R->p[1]; double q = s2 / r2; double x6 = q * q * q;
                             return 4 * R->p[0] * x6 * (x6 - 1) - R->shift; }
    case CEG_HARDSPHERE: { double s = R->p[0] + R->p[1]; return ((r2 < s * s) ? INFINITY : 0.0) - R->shift; }
    case CEG_NOINTERACTION: return 0.0 - R->shift;
    case CEG_MONOMIAL: return R->p[0] / pow(r2, R->p[1] / 2) - R->shift;
    default: return rule_energy_r(R, sqrt(r2), coulombic);
    }
}

ORACLE_API void oracle_single_contribution_vdw(const double mat[9], const double invmat[9], double cutoff2,
                                               const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds,
                                               double coulombic, const double* positions, const int32_t* kinds,
                                               const int32_t* molecule, int64_t natoms, const double* trial,
                                               const int32_t* trial_kinds, int32_t m, int64_t n, int32_t exclude,
                                               double* out, int32_t nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t p = 0; p < n; ++p) {
        double energy = 0.0;
        for (int32_t k2 = 0; k2 < m; ++k2) {                     /* energy.jl:415 */
            const double* pos2 = trial + ((size_t)p * m + k2) * 3;
            for (int64_t l1 = 0; l1 < natoms; ++l1) {             /* :417 */
                if (molecule[l1] == exclude) continue;            /* :419 */
                double buffer[3] = {pos2[0] - positions[3 * l1], pos2[1] - positions[3 * l1 + 1], pos2[2] - positions[3 * l1 + 2]};
                double f[3];
                matvec3(f, invmat, buffer);                       /* utils.jl:295 */
                for (int i = 0; i < 3; ++i) { double diff = f[i] + 0.5; f[i] = diff - floor(diff) - 0.5; }
                matvec3(buffer, mat, f);
                const double d2 = norm2_3(buffer);
                if (d2 < cutoff2) {                               /* energy.jl:422 */
                    const int32_t t = kinds[l1] * nkinds + trial_kinds[k2];
                    double e = 0.0;                               /* rule sum: interactions.jl:589-595 */
                    for (int32_t q = rule_offset[t]; q < rule_offset[t + 1]; ++q) e += rule_energy_r2(&rules[q], d2, coulombic);
                    energy += e;
                }
            }
        }
        out[p] = energy;
    }
}

/* ======================================================================================
 * Row f4: blocking masks.
 * src/grids.jl:188-204 BlockFile(g::EnergyGrid); src/coordinates.jl:139-152 the scan of parse_blockfile.
 * Masks are uint8 [x][y][z], z fastest (Julia's block[i,j,k] with i = x).
 * ====================================================================================== */
ORACLE_API void oracle_block_from_grid(const float* value, const int32_t dims[3], double threshold, uint8_t* block)
{
    const int64_t a = dims[0] + 1, b = dims[1] + 1, c = dims[2] + 1;
    memset(block, 0, (size_t)(a * b * c));
#define BLK(i, j, k) block[(k) + c * ((j) + b * (i))]
    for (int64_t i = 0; i < a - 1; ++i)                            /* grids.jl:191: 1:a-1 etc. */
        for (int64_t j = 0; j < b - 1; ++j)
            for (int64_t k = 0; k < c - 1; ++k)
                if (value[k + c * (j + b * i)] > (float)threshold) {   /* g.grid[k,j,i,1] > 5e6 */
                    BLK(i, j, k) = 1; BLK(i + 1, j, k) = 1; BLK(i, j + 1, k) = 1; BLK(i + 1, j + 1, k) = 1;
                    BLK(i, j, k + 1) = 1; BLK(i + 1, j, k + 1) = 1; BLK(i, j + 1, k + 1) = 1; BLK(i + 1, j + 1, k + 1) = 1;
                }
#undef BLK
}

ORACLE_API void oracle_block_spheres(const int32_t dims[3], const double delta[3], const double shift[3], const double mat[9],
                                     const double invmat[9], int32_t ortho, double safemin2, const double* centers,
                                     const double* radius2, int32_t nspheres, uint8_t* block, int32_t nthreads)
{
    const int64_t a = dims[0] + 1, b = dims[1] + 1, c = dims[2] + 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < a; ++i)
        for (int64_t j = 0; j < b; ++j)
            for (int64_t k = 0; k < c; ++k) {
                /* inverse_offsetpoint(SVector(i,j,k)), coordinates.jl:68-70, 1-based there */
                const double p[3] = {(double)i * delta[0] + shift[0], (double)j * delta[1] + shift[1], (double)k * delta[2] + shift[2]};
                uint8_t blocked = 0;
                for (int32_t s = 0; s < nspheres; ++s) {              /* :145-151 */
                    double buffer[3] = {centers[3 * s] - p[0], centers[3 * s + 1] - p[1], centers[3 * s + 2] - p[2]};
                    double buffer2[3];
                    if (periodic_distance2_fromcartesian(buffer, mat, invmat, ortho, safemin2, buffer2) < radius2[s]) { blocked = 1; break; }
                }
                block[k + c * (j + b * i)] = blocked;
            }
}
