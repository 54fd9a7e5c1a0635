/*
 * ceg_oracle_mc.c -- CPU restatement (plain C, FP64) of the host-side pieces the consumers of the grids
 * need: cell analysis, the EwaldFramework tables, the per-molecule structure factors of an
 * IncrementalEwaldContext and single_contribution_ewald.  TEST INFRASTRUCTURE ONLY (see ceg_oracle.c):
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 *
 * Why it exists (round 4): until round 3 oracle/oracle.py borrowed these from the PRODUCT package's host
 * mirror (ceg_hip.ewald / ceg_hip.utils / ceg_hip.constants), so a bug there was common-mode to the GPU
 * path and to its checker.  Everything the checker needs is restated here (and in oracle/hostlogic.py)
 * from the Julia source; tests/test_oracle_hostlogic.py compares the two restatements with each other
 * and with the literals of test/runtests.jl.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Build with -ffp-contract=off.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

static inline void matvec3(double out[3], const double m[9], const double v[3])
{
    /* StaticArrays mat-vec, column-major m: out[i] = sum_k M[i,k]*v[k], k ascending */
    for (int i = 0; i < 3; ++i)
        out[i] = (m[i] * v[0] + m[i + 3] * v[1]) + m[i + 6] * v[2];
}

/* Float16(x) for a Float64 x in the normal / subnormal range of binary16: round to nearest, ties to even
 * (11 significant bits; quantum 2^-24 below 2^-14).  Overflow is not handled (callers pass angles). */
static double round_to_half(double x)
{
    if (x == 0.0 || !isfinite(x)) return x;
    int e;
    (void)frexp(x, &e);                       /* |x| = m * 2^e, m in [0.5, 1) */
    int q = e - 11;                           /* quantum exponent for 11 significant bits */
    if (q < -24) q = -24;
    return ldexp(nearbyint(ldexp(x, -q)), q); /* default rounding mode: to nearest, ties to even */
}

/* ---- src/utils.jl:129-138 cell_parameters: lengths and angles (degrees) of the columns of mat.
 * acosd(x) = rad2deg(acos(x)) = (acos(x)/pi)*180 (Base). */
static void cell_parameters(const double mat[9], double len[3], double ang[3])
{
    const double* a_ = mat, * b_ = mat + 3, * c_ = mat + 6;
    const double pi = 3.14159265358979323846;
    len[0] = sqrt(a_[0] * a_[0] + a_[1] * a_[1] + a_[2] * a_[2]);
    len[1] = sqrt(b_[0] * b_[0] + b_[1] * b_[1] + b_[2] * b_[2]);
    len[2] = sqrt(c_[0] * c_[0] + c_[1] * c_[1] + c_[2] * c_[2]);
    const double bc = b_[0] * c_[0] + b_[1] * c_[1] + b_[2] * c_[2];
    const double ca = c_[0] * a_[0] + c_[1] * a_[1] + c_[2] * a_[2];
    const double ab = a_[0] * b_[0] + a_[1] * b_[1] + a_[2] * b_[2];
    ang[0] = acos(bc / (len[1] * len[2])) / pi * 180;
    ang[1] = acos(ca / (len[2] * len[0])) / pi * 180;
    ang[2] = acos(ab / (len[0] * len[1])) / pi * 180;
}

static void cross3(double out[3], const double u[3], const double v[3])
{
    out[0] = u[1] * v[2] - u[2] * v[1];
    out[1] = u[2] * v[0] - u[0] * v[2];
    out[2] = u[0] * v[1] - u[1] * v[0];
}

static double dot3(const double u[3], const double v[3])
{
    return u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
}

/* ---- src/utils.jl:146-155 prepare_periodic_distance_computations(mat) -> (ortho, safemin)
 * ortho = all(x -> isapprox(Float16(x), 90; rtol=0.02), angles):
 *   isapprox(x::Float16, 90) = x == 90 || abs(x - 90) <= 0.02*max(abs(x), 90), the difference formed in Float16
 *   (Float16 - Int promotes to Float16), the bound in Float64 (Float64 * Float16). */
ORACLE_API void oracle_prepare_periodic_distance_computations(const double mat[9], int32_t* ortho, double* safemin)
{
    double len[3], ang[3];
    cell_parameters(mat, len, ang);
    int all90 = 1;
    for (int t = 0; t < 3; ++t) {
        const double x16 = round_to_half(ang[t]);
        const double diff = fabs(round_to_half(x16 - 90.0));
        const double bound = 0.02 * fmax(fabs(x16), 90.0);
        if (!(x16 == 90.0 || diff <= bound)) all90 = 0;
    }
    *ortho = all90;
    const double* a_ = mat, * b_ = mat + 3, * c_ = mat + 6;
    double bxc[3], cxa[3], axb[3];
    cross3(bxc, b_, c_);
    cross3(cxa, c_, a_);
    cross3(axb, a_, b_);
    const double w0 = dot3(bxc, a_) / (len[1] * len[2]);
    const double w1 = dot3(cxa, b_) / (len[0] * len[2]);
    const double w2 = dot3(axb, c_) / (len[0] * len[1]);
    *safemin = fmin(fmin(w0, w1), w2) / 2;    /* half-distance between opposite planes of the cell */
}

/* cispi(2x) = exp(2 pi i x) (Base.cispi: sine and cosine of pi*y with exact argument reduction).
 * Here: reduce 2x modulo 2 exactly (fmod is exact), then libm sin / cos of pi*r, |r| <= 1. */
static double complex cispi2(double x)
{
    const double pi = 3.14159265358979323846;
    double r = fmod(2 * x, 2.0);              /* exact */
    if (r > 1.0) r -= 2.0;
    else if (r < -1.0) r += 2.0;
    /* the exact quarter points, like sincospi returns them */
    if (r == 0.0) return 1.0;
    if (r == 1.0 || r == -1.0) return -1.0;
    if (r == 0.5) return I;
    if (r == -0.5) return -I;
    return cos(pi * r) + I * sin(pi * r);
}

/* ---- src/ewald.jl:69-91 make_line_pos! / make_line_neg!: powers by repeated multiplication */
static void make_line_pos(double complex* E, int n, double complex eikt)
{
    E[0] = 1.0;
    if (n > 1) E[1] = eikt;                   /* (the reference writes Eikt[2,j] unconditionally: k >= 1 there) */
    for (int i = 2; i < n; ++i) E[i] = E[i - 1] * eikt;
}

static void make_line_neg(double complex* E, int k, double complex eikt)
{   /* E has 2k+1 entries; E[k] is the zeroth power */
    const double complex ceikt = conj(eikt);
    if (k > 0) {
        E[k - 1] = ceikt;
        for (int i = k - 2; i >= 0; --i) E[i] = E[i + 1] * ceikt;
    }
    make_line_pos(E + k, k + 1, eikt);
}

/* ---- src/ewald.jl:248-261 kfactors of initialize_ewald.
 *  kind[5*t..] = (j, k, i_first, i_last, rangeidx) rows of kspace.kindices (0-based rangeidx)
 *  invmat column-major inverse of the SUPERCELL matrix:
 *  il_ax, il_ay, il_az, il_bx, ... = invmat destructures column by column: il_ax = invmat[1,1], il_ay = invmat[2,1],
 *  il_az = invmat[3,1], il_bx = invmat[1,2], ... */
ORACLE_API void oracle_ewald_kfactors(const int32_t* kind, int64_t nkind, const double invmat[9],
                                      double volume_factor, double alpha_factor, double* kfactors)
{
    const double twopi = 2 * 3.14159265358979323846;
    const double il_ax = invmat[0], il_ay = invmat[1], il_az = invmat[2];
    const double il_bx = invmat[3], il_by = invmat[4], il_bz = invmat[5];
    const double il_cx = invmat[6], il_cy = invmat[7], il_cz = invmat[8];
    for (int64_t t = 0; t < nkind; ++t) {
        const int j = kind[5 * t], k = kind[5 * t + 1], i0 = kind[5 * t + 2], i1 = kind[5 * t + 3];
        const int64_t rangeidx = kind[5 * t + 4];
        const double rk0x = j * il_ay + k * il_az;
        const double rk0y = j * il_by + k * il_bz;
        const double rk0z = j * il_cy + k * il_cz;
        for (int i = i0; i <= i1; ++i) {
            const double rkx = twopi * (rk0x + i * il_ax);
            const double rky = twopi * (rk0y + i * il_bx);
            const double rkz = twopi * (rk0z + i * il_cx);
            const double rksqr = rkx * rkx + rky * rky + rkz * rkz;
            kfactors[rangeidx + (i - i0)] = volume_factor * (1 + (i != 0)) * exp(alpha_factor * rksqr) / rksqr;
        }
    }
}

/* one site's contribution to the sums, ewald_main_loop! / update_sums! inner part (src/ewald.jl:164-176, 669-678) */
static void accumulate_site(double complex* sums, const int32_t* kind, int64_t nkind, int ky, int kz, double c,
                            const double complex* X, const double complex* Y, const double complex* Z)
{
    for (int64_t t = 0; t < nkind; ++t) {
        const int jy = kind[5 * t], jz = kind[5 * t + 1], i0 = kind[5 * t + 2], i1 = kind[5 * t + 3];
        const int64_t rangeidx = kind[5 * t + 4];
        const double complex Eik_yz = c * Y[ky + jy] * Z[kz + jz];
        for (int i = i0; i <= i1; ++i) sums[rangeidx + (i - i0)] += X[i] * Eik_yz;
    }
}

/* ---- src/ewald.jl:265-273: StoreRigidChargeFramework = ewald_main_loop! over the sites of setup_Eik (:109-146)
 * for the framework tiled by `supercell`: site order atom-major, then pi_a, pi_b, pi_c with pi_c fastest
 * (jofs = 1 + (j-1)*PiABC + pi_a*PiBC + pi_b*PiC + pi_c), fractional position invmat*position + pi/Pi on each axis,
 * charges = repeat(charges; inner = Pi).  positions[3*natoms] cartesian (unit cell), invmat: inverse SUPERCELL matrix. */
ORACLE_API void oracle_framework_structure_factor(const int32_t* kind, int64_t nkind, const int32_t ks[3], int64_t num_kvecs,
                                                  const double invmat[9], const int32_t supercell[3],
                                                  const double* positions, const double* charges, int64_t natoms,
                                                  double* out_re, double* out_im)
{
    const int kx = ks[0], ky = ks[1], kz = ks[2];
    const int PA = supercell[0], PB = supercell[1], PC = supercell[2];
    double complex* X = malloc(sizeof(double complex) * (size_t)(kx + 1));
    double complex* Y = malloc(sizeof(double complex) * (size_t)(2 * ky + 1));
    double complex* Z = malloc(sizeof(double complex) * (size_t)(2 * kz + 1));
    double complex* sums = calloc((size_t)(num_kvecs > 0 ? num_kvecs : 1), sizeof(double complex));
    for (int64_t a = 0; a < natoms; ++a) {
        double p[3];
        matvec3(p, invmat, positions + 3 * a);
        for (int pa = 0; pa < PA; ++pa)
            for (int pb = 0; pb < PB; ++pb)
                for (int pc = 0; pc < PC; ++pc) {
                    make_line_pos(X, kx + 1, cispi2(p[0] + (double)pa / PA));
                    make_line_neg(Y, ky, cispi2(p[1] + (double)pb / PB));
                    make_line_neg(Z, kz, cispi2(p[2] + (double)pc / PC));
                    accumulate_site(sums, kind, nkind, ky, kz, charges[a], X, Y, Z);
                }
    }
    for (int64_t q = 0; q < num_kvecs; ++q) { out_re[q] = creal(sums[q]); out_im[q] = cimag(sums[q]); }
    free(X); free(Y); free(Z); free(sums);
}

/* ---- one molecule's structure factor: move_one_system!(tmpEiks, ctx, nothing, positions) (src/ewald.jl:352-366)
 * followed by update_sums! (:660-684) -- also what ewald_main_loop! leaves in sums[:, ij+1] (:158-176).
 * invmat: inverse of the MC-cell (= supercell) matrix, column-major. */
ORACLE_API void oracle_molecule_sums(const int32_t* kind, int64_t nkind, const int32_t ks[3], int64_t num_kvecs,
                                     const double invmat[9], const double* positions, const double* charges, int32_t natoms,
                                     double* out_re, double* out_im)
{
    const int kx = ks[0], ky = ks[1], kz = ks[2];
    double complex* X = malloc(sizeof(double complex) * (size_t)(kx + 1));
    double complex* Y = malloc(sizeof(double complex) * (size_t)(2 * ky + 1));
    double complex* Z = malloc(sizeof(double complex) * (size_t)(2 * kz + 1));
    double complex* sums = calloc((size_t)(num_kvecs > 0 ? num_kvecs : 1), sizeof(double complex));
    for (int a = 0; a < natoms; ++a) {
        double p[3];
        matvec3(p, invmat, positions + 3 * a);
        make_line_pos(X, kx + 1, cispi2(p[0]));
        make_line_neg(Y, ky, cispi2(p[1]));
        make_line_neg(Z, kz, cispi2(p[2]));
        accumulate_site(sums, kind, nkind, ky, kz, charges[a], X, Y, Z);
    }
    for (int64_t q = 0; q < num_kvecs; ++q) { out_re[q] = creal(sums[q]); out_im[q] = cimag(sums[q]); }
    free(X); free(Y); free(Z); free(sums);
}

/* ---- src/ewald.jl:717-737 the energy loop of single_contribution_ewald:
 *   rest   = StoreRigidChargeFramework + sums[:,1]            (ij < 0: own_* == NULL)
 *          = StoreRigidChargeFramework + (sums[:,1] - sums[:,ij+1]) otherwise (in that association)
 *   single = tmpsums (or sums[:,ij+1] when positions === nothing)
 *   returns 2*rest_single + single_single */
ORACLE_API double oracle_single_contribution_ewald(const double* kfactors, int64_t num_kvecs,
                                                   const double* fw_re, const double* fw_im,
                                                   const double* total_re, const double* total_im,
                                                   const double* own_re, const double* own_im,
                                                   const double* single_re, const double* single_im)
{
    double rest_single = 0.0, single_single = 0.0;
    for (int64_t q = 0; q < num_kvecs; ++q) {
        double re_f, im_f;
        if (!own_re) { re_f = fw_re[q] + total_re[q]; im_f = fw_im[q] + total_im[q]; }
        else { re_f = fw_re[q] + (total_re[q] - own_re[q]); im_f = fw_im[q] + (total_im[q] - own_im[q]); }
        const double re_a = single_re[q], im_a = single_im[q];
        const double temp = kfactors[q];
        rest_single += temp * (re_f * re_a + im_f * im_a);
        single_single += temp * (re_a * re_a + im_a * im_a);
    }
    return 2 * rest_single + single_single;
}

/* ---- src/ewald.jl:630-652 compute_ewald(::IncrementalEwaldContext), the energy loop:
 *   2*(framework_adsorbate + energy_net_charges) + (adsorbate_adsorbate + static_contribution) */
ORACLE_API double oracle_compute_ewald_total(const double* kfactors, int64_t num_kvecs,
                                             const double* fw_re, const double* fw_im,
                                             const double* total_re, const double* total_im,
                                             double energy_net_charges, double static_contribution)
{
    double framework_adsorbate = 0.0, adsorbate_adsorbate = 0.0;
    for (int64_t q = 0; q < num_kvecs; ++q) {
        const double temp = kfactors[q];
        framework_adsorbate += temp * (fw_re[q] * total_re[q] + fw_im[q] * total_im[q]);
        adsorbate_adsorbate += temp * (total_re[q] * total_re[q] + total_im[q] * total_im[q]);
    }
    return 2 * (framework_adsorbate + energy_net_charges) + (adsorbate_adsorbate + static_contribution);
}
