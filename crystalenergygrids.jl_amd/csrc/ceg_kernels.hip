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
//    grouped by class (VdW-active / Coulomb only, each split into candidates the whole tile sees at a
//    regular distance -- no range test in their loop -- and the rest; plus images near a cell-wrap
//    boundary) and every lane
//    then loops over the same (broadcast) candidate, one branch-free loop per class.  For each
//    (point, image) pair inside the cutoff the reference's *selection rule* (is this
//    image the one periodic_distance2! would return?) is evaluated exactly, including
//    the `ortho` shortcut and the stale-vector fall-through of src/utils.jl:234-245.
//    O(N_grid * n_cut).  Radial arithmetic of the hot loops: 1/r^2 by v_rcp_f64 + one Newton
//    step; real-space Ewald factors from r^2-indexed polynomial tables + the B_n recurrence
//    (EWK = 2: no sqrt / exp / erfc); Lennard-Jones in closed form; a single Buckingham class
//    from an r^2-indexed table of A exp(-B r) (VDWK = 3); everything the tables do not cover
//    (r < 2 A, threshold bands, wrap-boundary images) is redone with the reference's literal
//    arithmetic after the loops (slow_pairs).
//
// No MFMA: this is a pairwise FP64 reduction, bound by the FP64 vector ALU.
#include <type_traits>

#include "ceg_internal.h"
#include "ceg_math.h"
#include "ceg_minimage.h"

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
// exp for the exact path of k_culled: out of line, so that its polynomial constants stay inside the call instead of being
// hoisted to the top of the kernel and parked in scratch once per tile (they were 64 of the 128 B per lane of the fused
// Buckingham variants)
__device__ __attribute__((noinline)) double exp_out_of_line(double x) { return exp(x); }

template <bool LJONLY = false, bool COLD = false>
__device__ __forceinline__ void vdw_terms(const DevRule* __restrict__ rules, int rb, int re, double r2,
                                          double& v_out, double& p1_out, double& p2_out, double& p3_out)
{
    // local accumulators (plain values, not the caller's references: keeps them in registers)
    double v = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
    for (int t = rb; t < re; ++t) {
        const int kind = rules[t].kind;
        const double q0 = rules[t].p0, q1 = rules[t].p1, q2 = rules[t].p2, sh = rules[t].shift;
        double tv, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        if (LJONLY || kind == CEG_LENNARDJONES) {     // :434-441 (LJONLY: the plan has no other kind)
            const double inv = 1.0 / r2;
            const double s = q1 * inv;                // sigma^2 / r2
            const double x6 = s * s * s;
            const double inv2 = inv * inv;
            tv = 4.0 * q0 * x6 * (x6 - 1.0);
            t1 = 24.0 * q0 * (x6 * (1.0 - 2.0 * x6)) * inv;
            t2 = 96.0 * q0 * (x6 * (7.0 * x6 - 2.0)) * inv2;
            t3 = 384.0 * q0 * (x6 * (5.0 - 28.0 * x6)) * (inv2 * inv2);
        } else if (kind == CEG_BUCKINGHAM) {          // :447-457
            const double A = q0, B = q1, C = q2;
            const double r4 = r2 * r2;
            const double r = sqrt(r2);
            const double r6 = r4 * r2;
            const double x6 = C / r6;
            const double xe = A * (COLD ? exp_out_of_line(-B * r) : exp(-B * r));
            tv = xe - x6;
            t1 = -B * xe / r + 6.0 * x6 / r2;
            t2 = -48.0 * x6 / r4 + B * xe * (1.0 + B * r) / (r2 * r);
            t3 = -(3.0 * B * r + B * B * r2 + 3.0) * B * xe * r / r6 + 480.0 * C / (r6 * r6);
        } else {                                      // CEG_HARDSPHERE :444-446
            tv = (r2 < q0) ? __builtin_huge_val() : 0.0;
        }
        v += tv - sh;
        p1 += t1;
        p2 += t2;
        p3 += t3;
    }
    v_out = v; p1_out = p1; p2_out = p2; p3_out = p3;
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

// derivatives_ewald with the literal operation order of src/ewald.jl:299-312 (IEEE sqrt and
// divisions, so r -> 0 behaves like the reference) but exp / erfc from ceg_math.h; valid for
// alpha*r <= ERFCX_XMAX.  Used for the few pairs the culled kernel redoes with the reference's
// arithmetic: keeps libm's register-hungry erfc out of that kernel.
__device__ __forceinline__ void ewald_terms_poly(double alpha, double charge, double r2,
                                                 double& v, double& p1, double& p2, double& p3)
{
    const double inv_sqrtpi = 0.56418958354775628695;
    const double r = sqrt(r2);
    const double r3 = r2 * r;
    const double r5 = r3 * r2;
    const double r2a2 = r2 * (alpha * alpha);
    const double E = fast_exp_neg(-r2a2);
    const double er2a2 = 2.0 * alpha * r * E * inv_sqrtpi;
    const double erfar = E * erfcx_poly(alpha * r);
    v = charge * erfar / r;
    p1 = -charge * (er2a2 + erfar) / r3;
    p2 = charge * (er2a2 * (3.0 + 2.0 * r2a2) + 3.0 * erfar) / r5;
    p3 = charge * (-er2a2 * (15.0 + 10.0 * r2a2 + 4.0 * (r2a2 * r2a2)) - 15.0 * erfar) / (r5 * r2);
}

// Out-of-line forms for the exact path of the multi-probe kernels: with three accumulator sets live (48 registers) the inlined
// literal formulas -- IEEE divisions, sqrt, the exp and erfcx polynomials -- pushed the kernel over 128 VGPRs and the exact path
// spilled ~90 registers per pair; as calls their temporaries live in the callee's scratch registers.  Same arithmetic.
__device__ __attribute__((noinline)) double4 ewald_terms_poly_call(double alpha, double charge, double r2)
{
    double v, p1, p2, p3;
    ewald_terms_poly(alpha, charge, r2, v, p1, p2, p3);
    return make_double4(v, p1, p2, p3);
}
__device__ __attribute__((noinline)) double4 lj_terms_call(const DevRule* __restrict__ rules, int rb, int re, double r2)
{
    double v, p1, p2, p3;
    vdw_terms<true, true>(rules, rb, re, r2, v, p1, p2, p3);
    return make_double4(v, p1, p2, p3);
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

// _set_gridpoint! (src/grids.jl:118-135) into 8 registers (same arithmetic as store_gridpoint)
__device__ __forceinline__ void gridpoint8(float r[8], const double* delta, double lambda, double thr, Accum a)
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
    r[0] = (float)__dmul_rn(a.v, lambda);
    r[1] = (float)__dmul_rn(__dmul_rn(a.d1x, D1), lambda);
    r[2] = (float)__dmul_rn(__dmul_rn(a.d1y, D2), lambda);
    r[3] = (float)__dmul_rn(__dmul_rn(a.d1z, D3), lambda);
    r[4] = (float)__dmul_rn(__dmul_rn(a.d2xy, __dmul_rn(D1, D2)), lambda);
    r[5] = (float)__dmul_rn(__dmul_rn(a.d2xz, __dmul_rn(D1, D3)), lambda);
    r[6] = (float)__dmul_rn(__dmul_rn(a.d2yz, __dmul_rn(D2, D3)), lambda);
    r[7] = (float)__dmul_rn(__dmul_rn(a.d3, __dmul_rn(__dmul_rn(D1, D2), D3)), lambda);
}

// 16-byte global store that only needs 4-byte alignment (gfx950 global memory handles dword-aligned
// vector accesses; hipcc would split a packed struct store into dword + dwordx3)
typedef float ceg_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_float4_unaligned(float* dst, float4 v)
{
    const ceg_v4f u = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(dst), "v"(u) : "memory");
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
    return periodic_distance2_literal_m(g.mat, g.invmat, g.ortho, g.safemin2, dx, dy, dz);
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

#ifndef CEG_WAVES_FUSED_BUCK
#define CEG_WAVES_FUSED_BUCK CEG_WAVES
#endif
#ifndef CEG_WAVES_FUSED_LJ
#define CEG_WAVES_FUSED_LJ CEG_WAVES
#endif
#ifndef CEG_WAVES_VDW_LJ
#define CEG_WAVES_VDW_LJ CEG_WAVES
#endif
#ifndef CEG_WAVES_VDW_BUCK
#define CEG_WAVES_VDW_BUCK CEG_WAVES
#endif
#ifndef CEG_WAVES_COULOMB
#define CEG_WAVES_COULOMB CEG_WAVES
#endif
#ifndef CEG_WAVES
#define CEG_WAVES 4      // waves per SIMD the register allocator is asked to allow
#endif
#ifndef CEG_WG
#define CEG_WG 256       // threads per workgroup of k_culled (one tile per wave)
#endif
// pairs closer than this (A^2) take the literal min-image arithmetic in the culled kernel
// waves per SIMD the register allocator plans for, per variant (measured, profiles/r01_variant_waves.txt)
#ifndef CEG_WAVES_MULTI_FUSED
#define CEG_WAVES_MULTI_FUSED 4
#endif
#ifndef CEG_WAVES_MULTI_VDW
#define CEG_WAVES_MULTI_VDW 4
#endif
#ifndef CEG_NW_MULTI_FUSED
#define CEG_NW_MULTI_FUSED 8         // 8 waves x 64 x 96 B + 19.7 KB of tables + 8 KB = 77 KB: two workgroups per CU
#endif
constexpr int culled_waves(int mode, int vdwk, int ewk, int np = 1)
{
    return np > 1 ? (mode == 2 ? CEG_WAVES_MULTI_FUSED : (np == 2 ? CEG_WAVES_MULTI_VDW : 3))
           : (mode == 2 && vdwk >= 2 && ewk) ? CEG_WAVES_FUSED_BUCK
           : (mode == 2 && vdwk == 1 && ewk) ? CEG_WAVES_FUSED_LJ
           : (mode == 0 && vdwk == 1)      ? CEG_WAVES_VDW_LJ
           : (mode == 0 && vdwk >= 2)      ? CEG_WAVES_VDW_BUCK
           : (mode == 1 && ewk)            ? CEG_WAVES_COULOMB
                                           : CEG_WAVES;
}
#ifndef CEG_NW_EW2
#define CEG_NW_EW2 8     // waves per workgroup of the variants that keep the r^2-indexed Ewald tables in LDS
#endif
// waves (= tiles) per workgroup: the r^2-indexed Ewald tables take 18 KB of LDS, shared by 8 waves instead of 4 so that
// two workgroups (16 waves, 4 per SIMD) still fit a CU's 160 KB
constexpr int culled_nw(int mode, int vdwk, int ewk, int np = 1)
{
    // multi-probe records are 48 + 32 np bytes; VdW-only: 4 waves x 64 x (112 .. 176) B, three or four workgroups per CU
    return np > 1 ? (mode == 2 ? CEG_NW_MULTI_FUSED : CEG_WG / 64)
                  : ((mode != 0 && ewk == 2) || (mode != 1 && vdwk == 3)) ? CEG_NW_EW2 : CEG_WG / 64;
}
[[maybe_unused]] constexpr double R_EXACT2 = CEG_R_EXACT2;
static_assert(ERFCX_TAB_N == CEG_ERFCX_TAB_N, "table size mismatch");
constexpr int META_SIMPLE = 1 << 24;      // image is the wrapped one for every point of the tile
constexpr int META_HASVDW = 1 << 25;      // the atom's kind has at least one VdW rule
constexpr int META_BUCK = 1 << 26;        // fast class 2 (Buckingham) instead of 1 (Lennard-Jones)
constexpr int META_KINDMASK = (1 << 24) - 1;

// LDS record of one kept candidate of the culled kernel (80 B)
struct __attribute__((aligned(16))) Quad { double x, y, z, w; };     // double4 would force 32-B alignment (96-B records)
struct __attribute__((aligned(16))) CandRec {
    Quad xyzq;          // lattice-image position, charge
    Quad lj;            // fast VdW class parameters of the image's kind (4 eps, sigma^2, -, shift | A, B, C, shift)
    int32_t meta;       // kind | META_* flags
    int32_t atom;       // index of the original atom (slow path)
    int32_t _pad[2];
};

// ------------------------------------------------------------------ culled kernel
// Pairs the hot loop of k_culled sets aside: for candidate q of the current LDS chunk, s_odd[q] is
// the mask of lanes whose pair with q is "odd" (very close, within 1e-9 of a decision threshold,
// image not provably the wrapped one, stale-vector range); `cands` has bit q set when s_odd[q] is
// valid.  Each such pair is recomputed exactly like the reference does it -- original atom
// position, invmat*d, wrap, mat*f, neighbour search (periodic_distance2_literal) -- and evaluated
// with the literal radial formulas.  Runs after the hot loop so that its registers do not
// overlap the hot loop's.
template <int MODE, bool FASTEW, bool LJSLOW, bool EWSCALED, bool BKSCALED, int NP, bool DOV = true, bool DOC = true, typename Rec>
__device__ __forceinline__ void slow_pairs(const PlanConst* __restrict__ pc, unsigned long long cands, int lane,
                                           double px, double py, double pz,
                                           const Rec* s_rec,
                                           const unsigned long long* s_odd,
                                           Accum* avm, const int32_t* probe_idx, Accum& ac, double& smallest_d2)
{
    Accum& av = avm[0];
    const Geom& g = pc->g;
    while (cands != 0ull) {
        const int q = __builtin_ctzll(cands);
        cands &= cands - 1ull;
        const unsigned long long lanes = s_odd[q];
        if (!((lanes >> lane) & 1ull)) continue;
        const int mt = s_rec[q].meta;
        const double4 O = pc->ib.atoms[s_rec[q].atom];
        double dx = px - O.x, dy = py - O.y, dz = pz - O.z;
        const double r2 = periodic_distance2_literal(g, dx, dy, dz);
        if (r2 >= g.cutoff2) continue;
        if constexpr (NP > 1) {
            if (DOV && (mt & META_HASVDW)) {  // multi-probe plan: the rule run of the image's kind in every probe's table
                const int kd = mt & META_KINDMASK;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const RuleTable& rtp = pc->rtm[probe_idx[p]];
                    const int rb = rtp.offset[kd], re = rtp.offset[kd + 1];
                    if (re == rb) continue;       // no rule for this probe: nothing is added (as in a single-probe launch, where the all-zero
                                                  // record of the hot loop adds exact zeros)
                    const double4 T = lj_terms_call(rtp.rules, rb, re, r2);
                    accum_add(avm[p], T.x, T.y * (-1.0 / 6.0), T.z * (1.0 / 48.0), T.w * (-1.0 / 480.0), dx, dy, dz);
                }
            }
        } else if (MODE != MODE_COULOMB && (mt & META_HASVDW)) {
            const int kd = mt & META_KINDMASK;
            double v, p1, p2, p3;
            vdw_terms<LJSLOW, true>(pc->rt.rules, pc->rt.offset[kd], pc->rt.offset[kd + 1], r2, v, p1, p2, p3);
            if (LJSLOW) {        // the LJ-only hot loop accumulates p1/-6, p2/48, p3/-480 (scaled once per tile at the end)
                p1 *= -1.0 / 6.0; p2 *= 1.0 / 48.0; p3 *= -1.0 / 480.0;
            }
            if (BKSCALED) {      // the single-Buckingham hot loop accumulates v/C, p1/(6C), p2/(-48C), p3/(480C)
                v *= pc->bk_invC; p1 *= pc->bk_s1; p2 *= pc->bk_s2; p3 *= pc->bk_s3;       // (constants from memory, not literals:
                                                                                        //  literals get hoisted into VGPRs)
            }
            accum_add(av, v, p1, p2, p3, dx, dy, dz);
        }
        if (MODE != MODE_VDW && DOC) {
            smallest_d2 = fmin(smallest_d2, r2);
            double v, p1, p2, p3;
            if constexpr (NP > 1) {
                const double4 T = ewald_terms_poly_call(g.alpha, O.w, r2);
                v = T.x; p1 = T.y; p2 = T.z; p3 = T.w;
            } else if (FASTEW)
                ewald_terms_poly(g.alpha, O.w, r2, v, p1, p2, p3);
            else
                ewald_terms(g.alpha, O.w, r2, v, p1, p2, p3);
            if (EWSCALED) {      // the r^2-table hot loop accumulates p2/3 and p3/15 (scaled once per tile at the end)
                p2 *= 1.0 / 3.0; p3 *= 1.0 / 15.0;
            }
            accum_add(ac, v, p1, p2, p3, dx, dy, dz);
        }
    }
}

// a*b + c on 24-bit operands, c in a scalar register (one VALU instruction)
__device__ __forceinline__ int mad_u24(int a, int b, int c)
{
    int r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}

// A wave-uniform FP64 value held in an SGPR pair.  gfx950 has no scalar FP64 ALU, so the compiler keeps
// every FP64 result in VGPRs even when all lanes hold the same number; for values that stay live across
// the hot loop (tile box, thresholds) that costs two VGPRs each and, at 4 waves per SIMD, spills.
// (the halves pass through an empty asm first: the compiler folds __builtin_amdgcn_readfirstlane of a value it can prove
// uniform and then keeps that value in a VGPR pair -- which is what this function exists to avoid; the builtin itself stays,
// so that the compiler knows the instruction it is scheduling)
__device__ __forceinline__ double uniform(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    asm("" : "+v"(lo));
    asm("" : "+v"(hi));
    return __hiloint2double(__builtin_amdgcn_readfirstlane(hi), __builtin_amdgcn_readfirstlane(lo));
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

static_assert(sizeof(CandRec) == 80, "candidate record layout");
// the same without the per-candidate VdW parameters (Coulomb-only builds, generic rules, the single tabulated Buckingham class)
struct __attribute__((aligned(16))) CandRecS {
    Quad xyzq;
    int32_t meta;
    int32_t atom;
    int32_t _pad[2];
};
static_assert(sizeof(CandRecS) == 48, "candidate record layout");
// multi-probe launches: the Lennard-Jones record (4 eps, sigma^2, sigma^6, shift) of the image's kind for each of the NP probes
// (three doubles per probe -- 4 eps, sigma^6, shift; sigma^2 is not used by the hot loop -- so that the two-probe record is 96 B:
// 8 waves x 64 x 96 B + the Ewald tables stay under half a CU's LDS)
template <int NP>
struct __attribute__((aligned(16))) CandRecM {
    Quad xyzq;
    double ljm[3 * NP + (NP & 1)];
    int32_t meta;
    int32_t atom;
    int32_t _pad[2];
};
static_assert(sizeof(CandRecM<2>) == 96 && sizeof(CandRecM<3>) == 128 && sizeof(CandRecM<4>) == 144, "candidate record layout");
// the per-candidate VdW record of probe p, whatever the record type (so that branches the variant never takes still compile)
template <int NP> __device__ __forceinline__ Quad lj_record(const CandRecM<NP>& r, int p) { return Quad{r.ljm[3 * p], 0.0, r.ljm[3 * p + 1], r.ljm[3 * p + 2]}; }
__device__ __forceinline__ Quad lj_record(const CandRec& r, int) { return r.lj; }
__device__ __forceinline__ Quad lj_record(const CandRecS&, int) { return Quad{0.0, 0.0, 0.0, 0.0}; }
template <int NP> __device__ __forceinline__ void set_lj_record(CandRecM<NP>& r, int p, Quad q) { r.ljm[3 * p] = q.x; r.ljm[3 * p + 1] = q.z; r.ljm[3 * p + 2] = q.w; }
__device__ __forceinline__ void set_lj_record(CandRec& r, int, Quad q) { r.lj = q; }
__device__ __forceinline__ void set_lj_record(CandRecS&, int, Quad) {}

// Template flags of k_culled
//   MODE    what is accumulated (VdW / Coulomb / both in one pass)
//   POINTS  arbitrary point list (eval_points) instead of 4x4x4 grid tiles
//   VDWK    0: generic rule runs evaluated with vdw_terms in the hot loop;
//           1: every kind present has at most one rule and it is Lennard-Jones;
//           3: every kind present is none or ONE Buckingham (+ hard sphere inside the exact-path radius): A exp(-B r) from
//              an r^2-indexed table, B, C, shift in scalar registers, no per-candidate parameters
//           2: every kind present is none / LJ / Buckingham (+ hard spheres that lie inside the
//              exact-path radius): the per-kind parameters travel with the candidate through LDS
//              and the pair term uses the shared 1/r (and the table exp for Buckingham)
//   (EWK replaces round 1's FASTEW flag; FASTEW = EWK != 0 below: the erfcx-polynomial slow path is valid)
//           polynomial, no division); otherwise libm-style erfc/exp
//   EWK     real-space Ewald term: 0 libm-style erfc / exp; 1 (alpha*cutoff <= ERFCX_XMAX) one table exp + erfcx table,
//           no division; 2 B0(r^2), C(r^2) from r^2-indexed polynomial tables + the B_n recurrence, no sqrt / exp / erfc
//   NP      probes per launch (multi-probe plans: NP > 1 needs VDWK = 1, grid mode, EWK = 2): NP VdW accumulator sets, the
//           candidate record carries the Lennard-Jones parameters of its kind for each probe; distance, 1/r^2, products and
//           the Coulomb part are shared.  The per-pair arithmetic and the order of the sums are those of the NP = 1 kernel:
//           a multi-probe launch is bit-identical to single-probe launches from the same plan.
template <int MODE, bool POINTS, int VDWK, int EWK, int NP = 1>
__global__ __launch_bounds__(64 * culled_nw(MODE, VDWK, EWK, NP), culled_waves(MODE, VDWK, EWK, NP)) void k_culled(const PlanConst* __restrict__ pc, Output out, Points pts,
                                                              int tiles_j, int tiles_k, int64_t ntiles)
{
    static_assert(NP >= 1 && NP <= CEG_MAX_PROBES && (NP == 1 || (VDWK == 1 && !POINTS && EWK == 2 && MODE != MODE_COULOMB)), "multi-probe variant");
    constexpr bool MULTI = NP > 1;
    constexpr bool FASTEW = EWK != 0;
    constexpr bool EW2 = EWK == 2 && MODE != MODE_VDW;
    constexpr int WG = 64 * culled_nw(MODE, VDWK, EWK, NP);
    constexpr bool BK2 = VDWK == 3 && MODE != MODE_COULOMB;       // one Buckingham class, G0(r^2) tabulated
    // One workgroup = CEG_WG/64 waves; each wave owns one tile and its own slice of the staging
    // arrays (waves never touch each other's slice, so no workgroup barrier inside the loops --
    // LDS operations of one wave complete in order).  The function tables are shared.
    constexpr int NW = WG / 64;
    // one record per kept candidate: image position + charge, VdW parameters of its kind, flags, atom index.
    // A single array so that the hot loop walks ONE LDS address (immediate offsets reach the fields).
    constexpr bool HAS_LJ = !MULTI && (VDWK == 1 || VDWK == 2) && MODE != MODE_COULOMB;      // per-candidate VdW parameters travel in the record
    using Rec = std::conditional_t<MULTI, CandRecM<NP>, std::conditional_t<HAS_LJ, CandRec, CandRecS>>;
    __shared__ __attribute__((aligned(16))) Rec s_rec_all[NW][64];
    __shared__ int32_t s_rowstart_all[NW][64];
    __shared__ int32_t s_rowprefix_all[NW][66];
    __shared__ unsigned long long s_odd_all[NW][64];
    __shared__ __attribute__((aligned(16))) double s_erfcx[EW2 ? 2 : ERFCX_TAB_N * 6];
    // LDS stride of an Ewald interval record, in doubles: the 14 of the table as it is built (112 B = 28 banks: lanes whose intervals are 16
    // apart meet in the same banks, and a wave spans 10-40 intervals -- 30 % of the LDS-active cycles of the fused kernel were conflict
    // cycles) or 15 (120 B = 30 banks: 32 intervals apart; the record is then read as 8-byte pairs).  Single-probe
    // variants only: the multi-probe variants have no 1.4 KB of LDS to spare.
    constexpr int EW2_LS = (NP == 1) ? CEG_EW2_LDS_STRIDE : CEG_EW2_STRIDE;
    __shared__ __attribute__((aligned(16))) double s_ew2[EW2 ? CEG_EW2_NI_MAX * EW2_LS : 2];
    __shared__ __attribute__((aligned(16))) double s_bk2[BK2 ? CEG_BK2_NI_MAX * CEG_BK2_STRIDE : 2];
    __shared__ double s_exp2[64];
    __shared__ int32_t s_org[NW][4];             // tile origins for the output transpose (grid mode)

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Rec* s_rec = s_rec_all[wave];
    int32_t* s_rowstart = s_rowstart_all[wave];
    int32_t* s_rowprefix = s_rowprefix_all[wave];
    unsigned long long* s_odd = s_odd_all[wave];

    constexpr bool FASTVDW = VDWK != 0;
    if (EW2) {
        const int nt = pc->ew2_ni * CEG_EW2_STRIDE;
        for (int t = threadIdx.x; t < nt; t += WG) s_ew2[(t / CEG_EW2_STRIDE) * EW2_LS + t % CEG_EW2_STRIDE] = pc->ew2_tab[t];
    } else if (FASTEW && MODE != MODE_VDW) {
        for (int t = threadIdx.x; t < ERFCX_TAB_N * 6; t += WG) s_erfcx[t] = pc->erfcx_tab[t];
    }
    if (BK2) {
        const int nt = pc->bk2_ni * CEG_BK2_STRIDE;
        for (int t = threadIdx.x; t < nt; t += WG) s_bk2[t] = pc->bk2_tab[t];
    }
    if ((FASTEW && !EW2 && MODE != MODE_VDW) || (VDWK == 2 && MODE != MODE_COULOMB))
        if (threadIdx.x < 64) s_exp2[threadIdx.x] = pc->exp2_tab[threadIdx.x];
    __syncthreads();
    const int64_t tile_raw = (int64_t)blockIdx.x * NW + wave;
    const bool active = tile_raw < ntiles;          // inactive waves idle until the final barrier
    const int64_t tile = active ? tile_raw : ntiles - 1;

    const Geom& g = pc->g;
    const ImageBins& ib = pc->ib;
    const RuleTable& rt = pc->rt;
    const int lane = threadIdx.x & 63;
    const double cutoff2 = g.cutoff2;
    const double r_exact2 = pc->r_exact2;          // >= R_EXACT2, and beyond every hard-sphere radius
    const double alpha = g.alpha;
    const int32_t ortho = g.ortho;
    const double safemin2 = g.safemin2;

    // ---- this lane's point
    int i = 0, j = 0, k = 0;
    int i0 = 0, j0 = 0, k0 = 0;                       // tile origin (grid mode)
    int64_t pidx = 0;
    bool valid;
    double px, py, pz;
    if (POINTS) {
        pidx = tile * 64 + lane;
        valid = active && pidx < pts.n;
        const int64_t tt = valid ? pidx : (pts.n - 1);
        px = pts.xyz[3 * tt]; py = pts.xyz[3 * tt + 1]; pz = pts.xyz[3 * tt + 2];
    } else {
        const int tj = (int)((tile / tiles_k) % tiles_j);
        const int ti = (int)(tile / ((int64_t)tiles_k * tiles_j));
        // Workgroups go to the 8 XCDs round-robin by index: with tiles_k/NW a multiple of 8 every XCD
        // would own fixed z-slabs of the grid and the launch would last as long as the densest slab
        // (measured: a framework with a corner missing ran no faster).  Rotating the z-tiles of each
        // (i, j) column by one workgroup per column makes every XCD sweep all of z, while
        // consecutive workgroups on one XCD still share their candidate bins in that XCD's L2.
        const int nq = (tiles_k + NW - 1) / NW;
        const int tk = (int)((tile % tiles_k + (int64_t)NW * ((ti + (out.i_begin >> 2) + tj) % nq)) % tiles_k);
        i0 = out.i_begin + 4 * ti; j0 = 4 * tj; k0 = 4 * tk;
        if (lane == 0) {                     // read back after the final barrier; written here so that the
            s_org[wave][0] = active ? i0 : -1;   // origin does not have to stay in registers across the loops
            s_org[wave][1] = j0;
            s_org[wave][2] = k0;
        }
        i = i0 + (lane >> 4);
        j = j0 + ((lane >> 2) & 3);
        k = k0 + (lane & 3);
        valid = active && (i < out.i_end) && (j <= g.dims[1]) && (k <= g.dims[2]);
        // out-of-range lanes are clamped onto the last valid plane / row / column of the tile (its first
        // point is always valid), so they stay inside the box of the valid ones and store nothing
        const int ci = i < out.i_end ? i : out.i_end - 1;
        const int cj = j <= g.dims[1] ? j : g.dims[1];
        const int ck = k <= g.dims[2] ? k : g.dims[2];
        px = grid_coord(ci, g.size[0], g.dims[0], g.shift[0]);
        py = grid_coord(cj, g.size[1], g.dims[1], g.shift[1]);
        pz = grid_coord(ck, g.size[2], g.dims[2], g.shift[2]);
    }

    // ---- tile box (exact hull of the lanes' points)
    const double blx = uniform(wave_min(px)), bhx = uniform(wave_max(px));
    const double bly = uniform(wave_min(py)), bhy = uniform(wave_max(py));
    const double blz = uniform(wave_min(pz)), bhz = uniform(wave_max(pz));
    const double cx = uniform(0.5 * (blx + bhx)), cy = uniform(0.5 * (bly + bhy)), cz = uniform(0.5 * (blz + bhz));
    const double hx = uniform(0.5 * (bhx - blx)), hy = uniform(0.5 * (bhy - bly)), hz = uniform(0.5 * (bhz - blz));
    // cutoff with a rounding margin: anything a lane can see with r2 < cutoff2 is kept
    const double rc2 = uniform(cutoff2 * (1.0 + 1e-9) + 1e-9);
    const double rc = uniform(sqrt(rc2));

    // ---- bin rows (bx, by) intersecting the neighbourhood
    auto bin_of = [&](double x, int ax) -> int {
        int b = (int)floor((x - ib.lo[ax]) * ib.inv_bin[ax]);
        b = b < 0 ? 0 : b;
        return b >= ib.nb[ax] ? ib.nb[ax] - 1 : b;
    };
    const int bx0 = bin_of(blx - rc, 0), bx1 = bin_of(bhx + rc, 0);
    const int by0 = bin_of(bly - rc, 1), by1 = bin_of(bhy + rc, 1);
    const int nry = by1 - by0 + 1;
    const int nrows = active ? (bx1 - bx0 + 1) * nry : 0;
    const float inv_nry = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (float)nry)));

    Accum avm[NP], ac;
#pragma unroll
    for (int p = 0; p < NP; ++p) accum_zero(avm[p]);
    Accum& av = avm[0];
    accum_zero(ac);
    double smallest_d2 = __builtin_huge_val();
    // width of the "decide with the reference's arithmetic" band around a threshold
    const double band_cut = 1e-9 * cutoff2;
    const double alpha2 = pc->alpha2;
    const bool stale_possible = !ortho && safemin2 < cutoff2;
    const double cut_hi = uniform(cutoff2 + band_cut);
    // upper end of the regular range: below the cutoff band and, when the stale-vector branch of
    // the reference can fire (safemin2 < cutoff2, src/utils.jl:233-245), below safemin2's band too
    const double reg_hi = uniform(stale_possible ? fmin(cutoff2 - band_cut, safemin2 * (1.0 - 1e-9)) : cutoff2 - band_cut);
    const double erf_inv_h = pc->erfcx_inv_h, erf_mx0 = pc->erfcx_mx0_inv_h;
    // integer forms of the range tests of the hot loop (high words of r_exact2 rounded up, reg_hi rounded down, cut_hi as is)
    const int hi_lo = __builtin_amdgcn_readfirstlane(__double2hiint(r_exact2) + (__double2loint(r_exact2) != 0 ? 1 : 0));
    const int hi_top = __builtin_amdgcn_readfirstlane(__double2hiint(reg_hi));
    const unsigned hi_span = hi_top > hi_lo ? (unsigned)(hi_top - hi_lo) : 0u;
    const int hi_cut = __builtin_amdgcn_readfirstlane(__double2hiint(cut_hi));
    const double int_lo = uniform(r_exact2 * (1.0 + 4e-6)), int_hi = uniform(reg_hi * (1.0 - 4e-6));
    // r^2-indexed tables: record of the interval with key k starts at s_ew2 + (k - ew2_base) * STRIDE
    const int ew2_off = EW2 ? __builtin_amdgcn_readfirstlane(-pc->ew2_base * (EW2_LS * 8)) : 0;
    int ew2_stride = EW2_LS * 8;              // kept in a VGPR: v_mad_u32_u24 reads one scalar operand (ew2_off)
    asm volatile("" : "+v"(ew2_stride));
    const double ew_k3 = pc->ew_k3, ew_k15 = pc->ew_k15;       // 2 alpha^2 / 3, 4 alpha^4 / 15
    const bool all_simple = __builtin_amdgcn_readfirstlane(pc->all_simple) != 0;
    const int bk2_off = BK2 ? __builtin_amdgcn_readfirstlane(-pc->bk2_base * (CEG_BK2_STRIDE * 8)) : 0;
    int bk2_stride = CEG_BK2_STRIDE * 8;
    asm volatile("" : "+v"(bk2_stride));
    const double bk_B = pc->bk_B, bk_nshift = pc->bk_nshift, bk_c1 = pc->bk_c1, bk_c2 = pc->bk_c2, bk_c3 = pc->bk_c3, bk_c4 = pc->bk_c4;

    for (int rbase = 0; rbase < nrows; rbase += 64) {
        // -- one row per lane: image range [start, start+count)
        int count = 0, start = 0;
        const int r = rbase + lane;
        if (r < nrows) {
            // r / nry without the integer-division sequence (whose per-tile magic number the compiler keeps in a VGPR across the
            // hot loops): r < 2^12 and nry < 2^7, so (r + 1/2) / nry in float is at least 1/(2 nry) away from an integer
            const int rq = (int)(((float)r + 0.5f) * inv_nry);
            const int bx = bx0 + rq, by = by0 + (r - rq * nry);
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
        // inclusive scan of counts over the wave.  The lane id is made opaque here: the permute addresses and the two LDS
        // addresses below are loop invariants that the compiler otherwise hoists out of the tile loop and, at the VGPR limit
        // of the hot loops, parks in scratch (stores + reloads per tile that showed up as HBM traffic)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        int incl = count;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __builtin_amdgcn_ds_bpermute((ln - o) << 2, incl);      // lanes < o read a wrapped lane and ignore it
            if (ln >= o) incl += up;
        }
        __builtin_amdgcn_wave_barrier();                       // previous batch's readers are done
        s_rowstart[ln] = start;
        s_rowprefix[ln + 1] = incl;
        if (ln == 0) s_rowprefix[0] = 0;
        __builtin_amdgcn_wave_barrier();
        const int total = __builtin_amdgcn_readfirstlane(s_rowprefix[64]);

        for (int cbase = 0; cbase < total; cbase += 64) {
            // -- stage: lane loads one image of the flattened row list and tests it against the tile
            const int t = cbase + lane;
            bool keep = false, interior = false;
            double4 P = make_double4(0, 0, 0, 0);
            double4 LJ = make_double4(0, 0, 0, 0);
            double4 LJM[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) LJM[p] = make_double4(0, 0, 0, 0);
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
                const int kw = ib.kind ? ib.kind[img] : -1;            // kind | META_HASVDW (set on the host), or -1
                const int kd = kw < 0 ? -1 : (kw & META_KINDMASK);
                const double qx = fmax(0.0, fabs(cx - P.x) - hx);
                const double qy = fmax(0.0, fabs(cy - P.y) - hy);
                const double qz = fmax(0.0, fabs(cz - P.z) - hz);
                const double dmin2 = qx * qx + qy * qy + qz * qz;
                keep = dmin2 < rc2;
                // every point of the tile sees this image at a regular distance (beyond the exact-path radius, below the
                // threshold band of the cutoff, 4e-6 inside both: the hot loop's tests look at the high word of r^2 only):
                // such candidates run without any range test
                const double fx = fabs(cx - P.x) + hx, fy = fabs(cy - P.y) + hy, fz = fabs(cz - P.z) + hz;
                interior = dmin2 > int_lo && (fx * fx + fy * fy + fz * fz) < int_hi;
                bool hasvdw = false;
                int vclass = 0;
                if (MODE != MODE_COULOMB && kd >= 0) {
                    hasvdw = (kw & META_HASVDW) != 0;
                    if constexpr (MULTI) {
                        if (hasvdw && keep) {
#pragma unroll
                            for (int p = 0; p < NP; ++p) {          // a kind without a rule for probe p has an all-zero record: exact zeros
                                const FastVdw F = pc->fastm[out.probe_idx[p]][kd];
                                LJM[p] = make_double4(F.p0, F.p1, F.p2, F.shift);
                            }
                        }
                    } else if (FASTVDW && VDWK != 3 && hasvdw && keep) {
                        const FastVdw F = pc->fastvdw[kd];          // class + parameters of this kind
                        LJ = make_double4(F.p0, F.p1, F.p2, F.shift);
                        vclass = F.cls;
                    }
                }
                if (MODE == MODE_VDW) keep = keep && hasvdw;   // kinds without a rule contribute exact zeros
                bool simple = g.diag != 0 || (!POINTS && all_simple);
                if (!simple && keep) {
                    const double* I = g.invmat;
                    const double ux = cx - P.x, uy = cy - P.y, uz = cz - P.z;
                    const double f0 = I[0] * ux + I[3] * uy + I[6] * uz;
                    const double f1 = I[1] * ux + I[4] * uy + I[7] * uz;
                    const double f2 = I[2] * ux + I[5] * uy + I[8] * uz;
                    // fractional half-extent of the tile box (+ margin)
                    const double e0 = fabs(I[0]) * hx + fabs(I[3]) * hy + fabs(I[6]) * hz + 1e-9;
                    const double e1 = fabs(I[1]) * hx + fabs(I[4]) * hy + fabs(I[7]) * hz + 1e-9;
                    const double e2 = fabs(I[2]) * hx + fabs(I[5]) * hy + fabs(I[8]) * hz + 1e-9;
                    simple = (fabs(f0) + e0 < 0.5) && (fabs(f1) + e1 < 0.5) && (fabs(f2) + e2 < 0.5);
                }
                meta = (kd & META_KINDMASK) | (simple ? META_SIMPLE : 0) | (hasvdw ? META_HASVDW : 0) |
                       (vclass == 2 ? META_BUCK : 0);
            }
#ifdef CEG_SKIP_TESTS           // measurement aid (with CEG_STAGING_ONLY): row scan + image loads only, nothing kept, no compaction writes
            keep = keep && (P.x == 12345.678);
#endif
            // Kept candidates are compacted into five consecutive groups, so that the hot loops below run without a
            // per-candidate class test (no record flag to read, no scalar branch around the VdW part):
            //   [0, nvi)        image provably the wrapped one for the whole tile, kind has a VdW rule, interior (see above)
            //   [nvi, nv)       same, but some points may be out of range or on the exact path: range tests in the loop
            //   [nv, nni)       wrapped image, no VdW rule (fused mode only: Coulomb term alone), interior
            //   [nni, nreg)     same with range tests
            //   [nreg, nkeep)   image near a cell-wrap boundary of this tile: every lane takes the exact path
            const bool simple_c = (meta & META_SIMPLE) != 0;
            const bool withv = MODE != MODE_COULOMB && (meta & META_HASVDW) != 0;
            const bool ks = keep && simple_c;
            const unsigned long long mask_vi = __builtin_amdgcn_ballot_w64(ks && withv && interior);
            const unsigned long long mask_vb = __builtin_amdgcn_ballot_w64(ks && withv && !interior);
            const unsigned long long mask_ni = __builtin_amdgcn_ballot_w64(ks && !withv && interior);
            const unsigned long long mask_nb = __builtin_amdgcn_ballot_w64(ks && !withv && !interior);
            const unsigned long long mask_x = __builtin_amdgcn_ballot_w64(keep && !simple_c);
            const int nvi = __popcll(mask_vi), nv = nvi + __popcll(mask_vb), nni = nv + __popcll(mask_ni), nreg = nni + __popcll(mask_nb),
                      nkeep = nreg + __popcll(mask_x);
            __builtin_amdgcn_wave_barrier();                   // previous chunk's readers are done
            if (keep) {
                const unsigned long long mine = !simple_c ? mask_x : (withv ? (interior ? mask_vi : mask_vb) : (interior ? mask_ni : mask_nb));
                const int first = !simple_c ? nreg : (withv ? (interior ? 0 : nvi) : (interior ? nv : nni));
                const int slot = first + __builtin_amdgcn_mbcnt_hi((unsigned)(mine >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mine, 0));
                s_rec[slot].xyzq = Quad{P.x, P.y, P.z, P.w};
                if constexpr (MULTI) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) set_lj_record(s_rec[slot], p, Quad{LJM[p].x, LJM[p].y, LJM[p].z, LJM[p].w});
                } else if constexpr (HAS_LJ) s_rec[slot].lj = Quad{LJ.x, LJ.y, LJ.z, LJ.w};
                s_rec[slot].meta = meta;
                s_rec[slot].atom = aidx;
            }
            __builtin_amdgcn_wave_barrier();

            // -- every lane against every kept image (LDS broadcast reads).  The hot loop only
            //    handles the regular case -- the image is the wrapped one for the whole tile, the
            //    pair is between R_EXACT and the cutoff and away from every decision threshold.
            //    Anything else is recorded in a per-lane bit mask and redone after the loop with
            //    the reference's literal arithmetic (slow_pairs).
            unsigned long long slow = 0ull;     // wave-uniform: candidates with at least one odd lane
            auto pair_body = [&](const int q, auto with_vdw_tag, auto interior_tag) __attribute__((always_inline)) {
                constexpr bool WITH_VDW = decltype(with_vdw_tag)::value;
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                const Quad A = s_rec[q].xyzq;
                double dx = px - A.x, dy = py - A.y, dz = pz - A.z;
                const double r2 = dx * dx + dy * dy + dz * dz;
                // regular: R_EXACT2 <= r2 < min(cutoff2 - band, stale limit), image provably the
                // wrapped one.  odd: anything else that could contribute.  (bitwise logic on
                // purpose: no short-circuit branches in the hot loop)
                // r2 >= 0, so doubles order like their bit patterns: the range tests look at the high word only
                // (32-bit integer compares issue at twice the FP64 rate).  hi_lo / hi_lo + hi_span are rounded inwards,
                // hi_cut outwards: a pair within 2^-20 relative of a threshold takes the exact path, which decides.
                const int hi = __double2hiint(r2);
                if constexpr (!INTERIOR) {
                    // lane masks straight out of the two compares (ICMP_ULT = 36, ICMP_SLE = 41)
                    const unsigned long long m_in = __builtin_amdgcn_uicmp((unsigned)(hi - hi_lo), hi_span, 36);
                    const unsigned long long oddlanes = __builtin_amdgcn_sicmp(hi, hi_cut, 41) & ~m_in;
                    if (oddlanes != 0ull) {              // scalar branch, rarely taken
                        slow |= 1ull << q;
                        if (lane == 0) s_odd[q] = oddlanes;
                    }
                    if (!((unsigned)(hi - hi_lo) < hi_span)) return;
                }

                // ---- regular pair: 2 A <= r < cutoff
                const double dxy = dx * dy, dxz = dx * dz, dyz = dy * dz;
                const double dxyz = dxz * dy;
                double rr = 0.0, rinv = 0.0, inv = 0.0;
                // sqrt(r2) is needed by Buckingham's exp(-B r) and by the erfcx form of the Ewald term; Lennard-Jones and the
                // r^2-indexed Ewald tables only use 1/r2
                constexpr bool NEED_SQRT = (VDWK >= 2 && MODE != MODE_COULOMB) || (EWK == 1 && MODE != MODE_VDW);
                if (NEED_SQRT) {
                    fast_sqrt_rsqrt(r2, rr, rinv);
                    inv = rinv * rinv;
                } else if (FASTVDW || EW2) {
                    inv = fast_rcp(r2);
                }
                // interval of s = r2 in the r^2-indexed tables: key = exponent + leading mantissa bits, t = s - start of the interval
                int key = 0;
                double t = 0.0;
                static_assert(CEG_BK2_SHIFT == CEG_EW2_SHIFT, "the Buckingham table shares the interval key of the Ewald tables");
                if (EW2 || (BK2 && WITH_VDW)) {
                    key = (int)((unsigned)hi >> CEG_EW2_SHIFT);
                    t = r2 - __hiloint2double(hi & (int)(0xffffffffu << CEG_EW2_SHIFT), 0);
                }
                if constexpr (WITH_VDW && MULTI) {
                    // the LJ branch below, once per probe: 1/r^2 and its powers are shared, the expressions are the same
                    const double inv2 = inv * inv;
                    const double y3 = inv2 * inv;
                    const double inv4 = inv2 * inv2;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const Quad L = lj_record(s_rec[q], p);  // 4 eps, sigma^2, sigma^6, shift
                        const double x6 = L.z * y3;
                        const double w = L.x * x6;
                        const double u = w * x6;
                        const double v = (u - w) - L.w;
                        const double p1 = __builtin_fma(u, 2.0, -w) * inv;                 // / -6
                        const double p2 = fms_vsv(u, 3.5, w) * inv2;                       // / 48
                        const double p3 = fms_vsv(u, 5.6, w) * inv4;                       // / -480
                        Accum& a = avm[p];
                        a.v += v;
                        a.d1x = __builtin_fma(p1, dx, a.d1x);
                        a.d1y = __builtin_fma(p1, dy, a.d1y);
                        a.d1z = __builtin_fma(p1, dz, a.d1z);
                        a.d2xy = __builtin_fma(p2, dxy, a.d2xy);
                        a.d2xz = __builtin_fma(p2, dxz, a.d2xz);
                        a.d2yz = __builtin_fma(p2, dyz, a.d2yz);
                        a.d3 = __builtin_fma(p3, dxyz, a.d3);
                    }
                } else if constexpr (WITH_VDW) {
                    double v, p1, p2, p3;
                    if constexpr (VDWK == 3) {
                        // derivativesGrid, Buckingham branch (src/interactions.jl:447-457) with G0 = A exp(-B r) from the table,
                        // u = G0/r, x6 = C/r^6 and D = (1/r) d/dr:  D G0 = -B u,  D u = -(B G0 + u)/r^2
                        //   v = G0 - x6 - shift,  p1 = -B u + 6 x6/r^2,  p2 = B (B G0 + u)/r^2 - 48 x6/r^4,
                        //   p3 = -B (B^2 u + 3 (B G0 + u)/r^2)/r^2 + 480 x6/r^6
                        const double2* rec = reinterpret_cast<const double2*>(
                            reinterpret_cast<const char*>(s_bk2) + mad_u24(key, bk2_stride, bk2_off));
                        const double2 g01 = rec[0], g23 = rec[1], g45 = rec[2], g67 = rec[3];
                        double g0 = __builtin_fma(g67.y, t, g67.x);      // degree 7 on the Ewald tables' intervals (ceg_internal.h)
                        g0 = __builtin_fma(g0, t, g45.y);
                        g0 = __builtin_fma(g0, t, g45.x);
                        g0 = __builtin_fma(g0, t, g23.y);
                        g0 = __builtin_fma(g0, t, g23.x);
                        g0 = __builtin_fma(g0, t, g01.y);
                        g0 = __builtin_fma(g0, t, g01.x);
                        // The table holds G' = G0/C and the channels are accumulated as v/C, p1/(6C), p2/(-48C), p3/(480C): the
                        // dispersion parts are then bare powers of 1/r^2 and every constant sits in ONE fma per channel
                        // (factors applied once per tile).  With u = G'/r, w = B G' + u:
                        //   v/C = G' - y3 - shift/C,  p1/(6C) = y4 - (B/6) u,  p2/(-48C) = y5 - (B/48) w/r^2,
                        //   p3/(480C) = y6 - (B/160) (w/r^2 + (B^2/3) u)/r^2,     y_n = r^(-2n)
                        const double u = g0 * rinv;
                        const double inv2 = inv * inv, y3 = inv2 * inv;
                        const double w = fma_vsv(g0, bk_B, u);
                        const double wi = w * inv;
                        v = add_sc(g0 - y3, bk_nshift);
                        p1 = fma_vsv(u, bk_c1, y3 * inv);
                        p2 = fma_vsv(wi, bk_c2, y3 * inv2);
                        p3 = fma_vsv(fma_vsv(u, bk_c3, wi) * inv, bk_c4, y3 * y3);
                    } else if constexpr (HAS_LJ) {
                      if (VDWK == 2 && (__builtin_amdgcn_readfirstlane(s_rec[q].meta) & META_BUCK)) {
                        // derivativesGrid, Buckingham branch (src/interactions.jl:447-457); a hard
                        // sphere summed with it is 0 here (its radius lies inside the exact path)
                        const Quad L = lj_record(s_rec[q], 0);  // A, B, C, shift
                        const double Br = L.y * rr;
                        const double xe = L.x * exp_neg_tab(s_exp2, -Br);
                        const double inv2 = inv * inv;
                        const double x6 = L.z * (inv2 * inv);              // C / r^6
                        const double Bxe = L.y * xe;
                        const double rinv3 = rinv * inv;
                        v = (xe - x6) - L.w;
                        p1 = __builtin_fma(mul_sc(x6, 6.0), inv, -Bxe * rinv);
                        p2 = __builtin_fma(mul_sc(x6, -48.0), inv2, (Bxe * rinv3) * (1.0 + Br));
                        p3 = __builtin_fma(mul_sc(x6, 480.0), inv2 * inv,
                                           -((Bxe * rinv3) * inv) * __builtin_fma(Br, add_sc(Br, 3.0), 3.0));
                      } else {
                        // derivativesGrid, LJ branch (src/interactions.jl:434-441), with 1/r2 shared.  In terms of
                        // w = 4 eps x6 and u = 4 eps x6^2 (x6 = sigma^6/r^6, sigma^6 formed once per kind on the host):
                        //   v = u - w - shift,  p1 = -6 (2u - w)/r^2,  p2 = 48 (3.5u - w)/r^4,  p3 = -480 (5.6u - w)/r^8
                        const Quad L = lj_record(s_rec[q], 0);  // 4 eps, sigma^2, sigma^6, shift
                        const double inv2 = inv * inv;
                        const double x6 = L.z * (inv2 * inv);
                        const double w = L.x * x6;
                        const double u = w * x6;
                        v = (u - w) - L.w;
                        p1 = __builtin_fma(u, 2.0, -w) * inv;                                // / -6
                        p2 = fms_vsv(u, 3.5, w) * inv2;                                      // / 48
                        p3 = fms_vsv(u, 5.6, w) * (inv2 * inv2);                             // / -480
                        if (VDWK != 1) {     // mixed classes share the accumulators: scale per pair.  LJ-only plans scale once per tile
                            p1 = mul_sc(p1, -6.0); p2 = mul_sc(p2, 48.0); p3 = mul_sc(p3, -480.0);
                        }
                      }
                    } else {
                        const int kd = __builtin_amdgcn_readfirstlane(s_rec[q].meta) & META_KINDMASK;
                        vdw_terms(rt.rules, rt.offset[kd], rt.offset[kd + 1], r2, v, p1, p2, p3);
                    }
                    av.v += v;
                    av.d1x = __builtin_fma(p1, dx, av.d1x);
                    av.d1y = __builtin_fma(p1, dy, av.d1y);
                    av.d1z = __builtin_fma(p1, dz, av.d1z);
                    av.d2xy = __builtin_fma(p2, dxy, av.d2xy);
                    av.d2xz = __builtin_fma(p2, dxz, av.d2xz);
                    av.d2yz = __builtin_fma(p2, dyz, av.d2yz);
                    av.d3 = __builtin_fma(p3, dxyz, av.d3);
                }
                if (MODE != MODE_VDW) {
                    // (regular pairs have r >= 2 A: they cannot trigger the smallest_d2 < 1 rule)
                    double v, p1, p2, p3;
                    if (EW2) {
                        // derivatives_ewald (src/ewald.jl:299-312) as B_n(r) of the interval polynomials in s = r2:
                        //   B0 = erfc(alpha r)/r,  C = (2 alpha/sqrt(pi)) exp(-alpha^2 s),
                        //   B1 = (B0 + C)/s,  B2 = (3 B1 + 2 alpha^2 C)/s,  B3 = (5 B2 + 4 alpha^4 C)/s
                        //   v = q B0,  p1 = -q B1,  p2 = q B2,  p3 = -q B3
                        const char* recp = reinterpret_cast<const char*>(s_ew2) + mad_u24(key, ew2_stride, ew2_off);
                        double2 a01, a23, a45, a6c0, c12, c34, c56;
                        if (EW2_LS % 2 == 0) {
                            const double2* rec = reinterpret_cast<const double2*>(recp);
                            a01 = rec[0]; a23 = rec[1]; a45 = rec[2]; a6c0 = rec[3]; c12 = rec[4]; c34 = rec[5]; c56 = rec[6];
                        } else {                                            // 8-byte aligned records
                            const double* rec = reinterpret_cast<const double*>(recp);
                            a01 = make_double2(rec[0], rec[1]); a23 = make_double2(rec[2], rec[3]); a45 = make_double2(rec[4], rec[5]);
                            a6c0 = make_double2(rec[6], rec[7]); c12 = make_double2(rec[8], rec[9]); c34 = make_double2(rec[10], rec[11]);
                            c56 = make_double2(rec[12], rec[13]);
                        }
                        double b0 = __builtin_fma(a6c0.x, t, a45.y);
                        double cc = __builtin_fma(c56.y, t, c56.x);
                        b0 = __builtin_fma(b0, t, a45.x);
                        cc = __builtin_fma(cc, t, c34.y);
                        b0 = __builtin_fma(b0, t, a23.y);
                        cc = __builtin_fma(cc, t, c34.x);
                        b0 = __builtin_fma(b0, t, a23.x);
                        cc = __builtin_fma(cc, t, c12.y);
                        b0 = __builtin_fma(b0, t, a01.y);
                        cc = __builtin_fma(cc, t, c12.x);
                        b0 = __builtin_fma(b0, t, a01.x);
                        cc = __builtin_fma(cc, t, a6c0.y);
                        // accumulated as B2/3 and B3/15 (one FMA + one product per step); the factors are applied once per tile
                        const double qb0 = A.w * b0, qc = A.w * cc;
                        const double b1 = (qb0 + qc) * inv;
                        const double b2 = fma_vsv(qc, ew_k3, b1) * inv;        // B2/3  = (B1 + (2 alpha^2/3) C)/s
                        const double b3 = fma_vsv(qc, ew_k15, b2) * inv;       // B3/15 = (B2/3 + (4 alpha^4/15) C)/s
                        v = qb0; p1 = -b1; p2 = b2; p3 = -b3;
                    } else if (FASTEW) {
                        // derivatives_ewald (src/ewald.jl:299-312): erfc(x) = exp(-x^2) erfcx(x)
                        // with E = exp(-x^2), g = erfcx(x), k = 2/sqrt(pi):  erfc = E g,  e = k x E, so
                        //   v  =  (q/r)   E  g
                        //   p1 = -(q/r^3) E (k x + g)
                        //   p2 =  (q/r^5) E (k x (3 + 2 x^2) + 3 g)
                        //   p3 = -(q/r^7) E (k x (15 + 10 x^2 + 4 x^4) + 15 g)
                        const double x = alpha * rr;
                        const double x2 = alpha2 * r2;
                        const double E = exp_neg_tab(s_exp2, -x2);
                        const double gx = erfcx_tab(s_erfcx, x, erf_inv_h, erf_mx0);
                        const double kx = mul_sc(x, 1.1283791670955125739);
                        const double qe1 = (A.w * rinv) * E;
                        const double qe3 = qe1 * inv, qe5 = qe3 * inv, qe7 = qe5 * inv;
                        v = qe1 * gx;
                        p1 = -qe3 * (kx + gx);
                        p2 = qe5 * __builtin_fma(kx, fma2_sc(x2, 3.0), mul_sc(gx, 3.0));
                        p3 = -qe7 * __builtin_fma(kx, fma_sc(fma4_sc(x2, 10.0), x2, 15.0), mul_sc(gx, 15.0));
                    } else {
                        smallest_d2 = min_nonan(smallest_d2, r2);
                        ewald_terms(alpha, A.w, r2, v, p1, p2, p3);
                    }
                    ac.v += v;
                    ac.d1x = __builtin_fma(p1, dx, ac.d1x);
                    ac.d1y = __builtin_fma(p1, dy, ac.d1y);
                    ac.d1z = __builtin_fma(p1, dz, ac.d1z);
                    ac.d2xy = __builtin_fma(p2, dxy, ac.d2xy);
                    ac.d2xz = __builtin_fma(p2, dxz, ac.d2xz);
                    ac.d2yz = __builtin_fma(p2, dyz, ac.d2yz);
                    ac.d3 = __builtin_fma(p3, dxyz, ac.d3);
                }
            };
#ifdef CEG_STAGING_ONLY          // measurement aid: what the kernel costs without its hot loops (results are garbage)
            if (nkeep == 12345) { av.v += s_rec[0].xyzq.x; ac.v += s_rec[1].xyzq.y; }
            continue;
#endif
            if (MODE != MODE_COULOMB) {
                for (int q = 0; q < nvi; ++q) pair_body(q, std::true_type{}, std::true_type{});
                for (int q = nvi; q < nv; ++q) pair_body(q, std::true_type{}, std::false_type{});
            }
            if (MODE != MODE_VDW) {
                for (int q = (MODE == MODE_COULOMB ? 0 : nv); q < nni; ++q) pair_body(q, std::false_type{}, std::true_type{});
                for (int q = nni; q < nreg; ++q) pair_body(q, std::false_type{}, std::false_type{});
            }
            // third group: the lanes within the cutoff of THIS image go to the exact path (which works from the atom and finds
            // its nearest image itself -- two images of one atom are never both inside the cutoff of a point)
            for (int q = nreg; q < nkeep; ++q) {
                const Quad A = s_rec[q].xyzq;
                const double dx = px - A.x, dy = py - A.y, dz = pz - A.z;
                const double r2 = dx * dx + dy * dy + dz * dz;
                const unsigned long long lanes = __builtin_amdgcn_sicmp(__double2hiint(r2), hi_cut, 41);
                if (lanes != 0ull) {
                    slow |= 1ull << q;
                    if (lane == 0) s_odd[q] = lanes;
                }
            }
            // -- the pairs set aside above, one per lane per round
#ifdef CEG_NO_EXACT_PATH          // measurement aid: what the exact path costs (results are wrong near atoms and thresholds)
            slow = 0ull;
#endif
            if (slow != 0ull) {
                __builtin_amdgcn_wave_barrier();
                if constexpr (MULTI && MODE == MODE_FUSED) {
                    // two passes (the literal distance is redone): the registers of the Lennard-Jones and of the Ewald evaluation are
                    // not needed at the same time beside the three accumulator sets
                    slow_pairs<MODE, FASTEW, true, EW2, false, NP, true, false>(pc, slow, lane, px, py, pz, s_rec, s_odd, avm, out.probe_idx, ac, smallest_d2);
                    slow_pairs<MODE, FASTEW, true, EW2, false, NP, false, true>(pc, slow, lane, px, py, pz, s_rec, s_odd, avm, out.probe_idx, ac, smallest_d2);
                } else
                    slow_pairs<MODE, FASTEW, VDWK == 1, EW2, VDWK == 3 && MODE != MODE_COULOMB, NP>(pc, slow, lane, px, py, pz, s_rec, s_odd, avm, out.probe_idx, ac, smallest_d2);
            }
        }
    }
    if (VDWK == 1 && MODE != MODE_COULOMB) {      // constant factors of the LJ derivative channels, deferred out of the hot loop
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            Accum& a = avm[p];
            a.d1x *= -6.0; a.d1y *= -6.0; a.d1z *= -6.0;
            a.d2xy *= 48.0; a.d2xz *= 48.0; a.d2yz *= 48.0;
            a.d3 *= -480.0;
        }
    }
    if (VDWK == 3 && MODE != MODE_COULOMB) {      // constant factors of the single Buckingham class
        const double C = pc->bk_C;
        av.v *= C;
        av.d1x *= 6.0 * C; av.d1y *= 6.0 * C; av.d1z *= 6.0 * C;
        av.d2xy *= -48.0 * C; av.d2xz *= -48.0 * C; av.d2yz *= -48.0 * C;
        av.d3 *= 480.0 * C;
    }
    if (EW2) {                                    // constant factors of the B_n recurrence, deferred out of the hot loop
        ac.d2xy *= 3.0; ac.d2xz *= 3.0; ac.d2yz *= 3.0;
        ac.d3 *= 15.0;
    }
    if (POINTS) {
        if (valid) write_results<MODE>(g, out, true, pidx, i, j, k, av, ac, smallest_d2);
        return;
    }
#ifdef CEG_SKIP_OUTPUT          // measurement aid: no _set_gridpoint! arithmetic, no transpose, no stores (one dummy store keeps the sums alive)
    if (av.v + ac.v == 12345.678 && lane == 0) out.vdw[0] = 1.0f;
    return;
#endif
    // ---- grid mode: the NW tiles of a workgroup are consecutive along z (the fastest array
    // axis), so the workgroup transposes its results through LDS and writes rows of 4*NW
    // contiguous floats (64 B at NW = 4) per (channel, i, j) instead of 16-B fragments.
    // Staging is done: the record slice of a wave (64 x 80 or 64 x 48 B) holds 512 floats per grid -- both grids at once
    // when it is large enough, else one grid after the other.
    constexpr bool TWO_PASS = MODE == MODE_FUSED && sizeof(Rec) * 64 < 2 * 512 * sizeof(float);
    const int64_t nz = g.dims[2] + 1, ny = g.dims[1] + 1;
    constexpr int NSEG = 8 * 16 * NW;                 // 16-byte segments per grid in this workgroup
    if (MODE != MODE_VDW) ac.v = (smallest_d2 < 1.0) ? __builtin_huge_val() : ac.v;      // src/probes.jl:116
    if constexpr (MULTI) {
        // one grid per pass through the wave's record slice: the NP VdW grids, then the Coulomb grid
        constexpr int NGRID = NP + (MODE == MODE_FUSED ? 1 : 0);
        float* sm = reinterpret_cast<float*>(s_rec_all[wave]);
#pragma unroll
        for (int gi = 0; gi < NGRID; ++gi) {
            if (gi == 0) __builtin_amdgcn_wave_barrier(); else __syncthreads();      // readers of the previous contents are done
            float r[8];
            if (gi < NP) gridpoint8(r, g.delta, out.lambda_vdw, out.thr_vdw, avm[gi < NP ? gi : 0]);
            else gridpoint8(r, g.delta, out.lambda_coulomb, out.thr_coulomb, ac);
#pragma unroll
            for (int c = 0; c < 8; ++c) sm[c * 64 + lane] = r[c];
            __syncthreads();
            float* base = gi < NP ? out.vdwm[gi < NP ? gi : 0] : out.coulomb;
            for (int seg = threadIdx.x; seg < NSEG; seg += WG) {
                const int w = seg % NW;
                const int row = seg / NW;                     // (channel, li, lj)
                const int c = row >> 4, li = (row >> 2) & 3, lj = row & 3;
                const int oi = s_org[w][0];
                if (oi < 0) continue;
                const int gi_ = oi + li, gj = s_org[w][1] + lj, gk = s_org[w][2];
                if (gi_ >= out.i_end || gj > g.dims[1] || gk > g.dims[2]) continue;
                const int64_t idx = (int64_t)gk + nz * ((int64_t)gj + ny * (int64_t)(gi_ - out.i_origin)) + (int64_t)c * out.channel_stride;
                const int so = c * 64 + li * 16 + lj * 4;
                const int nvalid = (g.dims[2] + 1 - gk) < 4 ? (g.dims[2] + 1 - gk) : 4;
                const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(s_rec_all[w]) + so);
                float* dst = base + idx;
                if (nvalid == 4) store_float4_unaligned(dst, v);
                else { dst[0] = v.x; if (nvalid > 1) dst[1] = v.y; if (nvalid > 2) dst[2] = v.z; }
            }
        }
        return;
    }
#pragma unroll
    for (int pass = 0; pass < (TWO_PASS ? 2 : 1); ++pass) {
        const bool do_v = MODE != MODE_COULOMB && (!TWO_PASS || pass == 0);
        const bool do_c = MODE != MODE_VDW && (!TWO_PASS || pass == 1);
        const int off_c = (MODE == MODE_FUSED && !TWO_PASS) ? 512 : 0;
        float* sm = reinterpret_cast<float*>(s_rec_all[wave]);
        if (pass == 0) __builtin_amdgcn_wave_barrier(); else __syncthreads();      // readers of the previous contents are done
        if (do_v) {
            float r[8];
            gridpoint8(r, g.delta, out.lambda_vdw, out.thr_vdw, av);
#pragma unroll
            for (int c = 0; c < 8; ++c) sm[c * 64 + lane] = r[c];
        }
        if (do_c) {
            float r[8];
            gridpoint8(r, g.delta, out.lambda_coulomb, out.thr_coulomb, ac);
#pragma unroll
            for (int c = 0; c < 8; ++c) sm[off_c + c * 64 + lane] = r[c];
        }
        __syncthreads();
        for (int seg = threadIdx.x; seg < NSEG; seg += WG) {
            const int w = seg % NW;
            const int row = seg / NW;                     // (channel, li, lj)
            const int c = row >> 4, li = (row >> 2) & 3, lj = row & 3;
            const int oi = s_org[w][0];
            if (oi < 0) continue;
            const int gi = oi + li, gj = s_org[w][1] + lj, gk = s_org[w][2];
            if (gi >= out.i_end || gj > g.dims[1] || gk > g.dims[2]) continue;
            const int64_t idx = (int64_t)gk + nz * ((int64_t)gj + ny * (int64_t)(gi - out.i_origin)) + (int64_t)c * out.channel_stride;
            const int so = c * 64 + li * 16 + lj * 4;
            const int nvalid = (g.dims[2] + 1 - gk) < 4 ? (g.dims[2] + 1 - gk) : 4;
            if (do_v) {
                const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(s_rec_all[w]) + so);
                float* dst = out.vdw + idx;
                if (nvalid == 4) store_float4_unaligned(dst, v);
                else { dst[0] = v.x; if (nvalid > 1) dst[1] = v.y; if (nvalid > 2) dst[2] = v.z; }
            }
            if (do_c) {
                const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(s_rec_all[w]) + off_c + so);
                float* dst = out.coulomb + idx;
                if (nvalid == 4) store_float4_unaligned(dst, v);
                else { dst[0] = v.x; if (nvalid > 1) dst[1] = v.y; if (nvalid > 2) dst[2] = v.z; }
            }
        }
    }
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

template <int MODE, bool POINTS>
static hipError_t launch_cull_flags(int vdwk, int ewk, hipStream_t stream, const PlanConst* pc, const Output& out, const Points& pts,
                                    int tj, int tk, int64_t ntiles)
{
    // Which k_culled<MODE, POINTS, VDWK, EWK> runs for a plan's (vdwk, ewk).  The plan classifies its rules (ceg_api.hip convert_rules:
    // vdwk 1 = every present kind has at most one Lennard-Jones rule; 3 = one Buckingham (+ hard sphere) parameter set, G0(r^2) tabulated;
    // 2 = per-kind Lennard-Jones / Buckingham classes; 0 = anything else, generic rule interpreter) and its Ewald arithmetic (ewk 2 = r^2-indexed
    // tables built; 1 = erfcx table, alpha * cutoff <= 5; 0 = libm-grade).  Flags that cannot matter for a mode are normalised so that fewer
    // variants get instantiated -- every (plan, mode) still gets arithmetic it is entitled to (a lower class is always valid for a higher one):
    //
    //   mode      plan (vdwk, ewk)      kernel <VDWK, EWK>   why
    //   VDW       (v, any)              <v, 1>               no Coulomb term: EWK is dead code, one value instantiated; v = 3 keeps <3, 1>
    //   COULOMB   (any, e)              <1, e>               no VdW term: VDWK is dead code, one value instantiated
    //   FUSED     (0, 2)                <0, 1>               the generic rule interpreter is only instantiated beside the erfcx variant
    //   FUSED     (0, 1) (0, 0)         <0, 1> <0, 0>
    //   FUSED     (1, e)                <1, e>               e = 2, 1, 0
    //   FUSED     (2, e)                <2, e>
    //   FUSED     (3, 2)                <3, 2>               the tabulated Buckingham class shares its interval key with the Ewald tables
    //   FUSED     (3, 1) (3, 0)         <2, 1> <2, 0>        without the r^2-indexed Ewald tables the Buckingham probe runs as class 2 (per-candidate
    //                                                        parameters, table exp): same terms, other arithmetic, within the suite's tolerance
    //   (ewk == 0 and vdwk == 3 in VDW mode cannot meet: MODE_VDW forces ewk = 1 first.)
    // Multi-probe launches (launch_multi_t) are <1, 2, NP> only: Lennard-Jones probes on the r^2-indexed tables.
    if (MODE == MODE_VDW) ewk = 1;
    if (MODE == MODE_COULOMB) vdwk = 1;
    if (ewk == 2 && vdwk == 0) ewk = 1;          // the generic rule interpreter keeps the erfcx variant
    if (vdwk == 3 && MODE == MODE_FUSED && ewk != 2) vdwk = 2;     // the tabulated Buckingham class is instantiated beside EWK = 2 only
    const int nw = culled_nw(MODE, vdwk, ewk);
    const int64_t nblocks = (ntiles + nw - 1) / nw;
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const dim3 grid((unsigned)nblocks), block(64 * nw);
#define CEG_LAUNCH(V, F) hipLaunchKernelGGL((k_culled<MODE, POINTS, V, F>), grid, block, 0, stream, pc, out, pts, tj, tk, ntiles)
    if (ewk == 2 && MODE != MODE_VDW) {
        if (vdwk == 1) CEG_LAUNCH(1, 2); else if (vdwk == 3) CEG_LAUNCH(3, 2); else CEG_LAUNCH(2, 2);
    } else if (ewk) {
        if (vdwk == 1) CEG_LAUNCH(1, 1); else if (vdwk == 2) CEG_LAUNCH(2, 1); else if (vdwk == 3) CEG_LAUNCH(3, 1); else CEG_LAUNCH(0, 1);
    } else {
        if (vdwk == 1) CEG_LAUNCH(1, 0); else if (vdwk == 2) CEG_LAUNCH(2, 0); else if (vdwk == 3) CEG_LAUNCH(2, 0); else CEG_LAUNCH(0, 0);
    }
#undef CEG_LAUNCH
    return hipGetLastError();
}

template <bool POINTS>
static hipError_t launch_cull_t(int mode, const PlanConst* pc, int vdwk, int ewk, const Output& out,
                                const Points& pts, int64_t ntiles, int tj, int tk, hipStream_t stream)
{
    if (ntiles <= 0) return hipSuccess;
    switch (mode) {
    case MODE_VDW: return launch_cull_flags<MODE_VDW, POINTS>(vdwk, ewk, stream, pc, out, pts, tj, tk, ntiles);
    case MODE_COULOMB: return launch_cull_flags<MODE_COULOMB, POINTS>(vdwk, ewk, stream, pc, out, pts, tj, tk, ntiles);
    default: return launch_cull_flags<MODE_FUSED, POINTS>(vdwk, ewk, stream, pc, out, pts, tj, tk, ntiles);
    }
}

template <int MODE, int NP>
static hipError_t launch_multi_t(const PlanConst* pc, const Output& out, int tj, int tk, int64_t ntiles, hipStream_t stream)
{
    const int nw = culled_nw(MODE, 1, 2, NP);
    const int64_t nblocks = (ntiles + nw - 1) / nw;
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_culled<MODE, false, 1, 2, NP>), dim3((unsigned)nblocks), dim3(64 * nw), 0, stream, pc, out, Points{nullptr, 0}, tj, tk, ntiles);
    return hipGetLastError();
}

hipError_t launch_culled_multi(int mode, int np, const PlanConst* d_pc, const Geom& g, const Output& out, hipStream_t stream)
{
    const int ni = out.i_end - out.i_begin;
    const int ti = (ni + 3) / 4, tj = (g.dims[1] + 1 + 3) / 4, tk = (g.dims[2] + 1 + 3) / 4;
    const int64_t ntiles = (int64_t)ti * tj * tk;
    if (ntiles <= 0) return hipSuccess;
    if (mode == MODE_FUSED) {
        static_assert(CEG_MAX_PROBES_FUSED == 2, "fused multi-probe variants");
        if (np == 2) return launch_multi_t<MODE_FUSED, 2>(d_pc, out, tj, tk, ntiles, stream);
        return hipErrorInvalidValue;
    }
    if (mode != MODE_VDW) return hipErrorInvalidValue;
    switch (np) {
    case 2: return launch_multi_t<MODE_VDW, 2>(d_pc, out, tj, tk, ntiles, stream);
    case 3: return launch_multi_t<MODE_VDW, 3>(d_pc, out, tj, tk, ntiles, stream);
    case 4: return launch_multi_t<MODE_VDW, 4>(d_pc, out, tj, tk, ntiles, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_culled(int mode, const PlanConst* d_pc, const Geom& g, int vdwk, int ewk,
                         const Output& out, const Points& pts, hipStream_t stream)
{
    if (pts.xyz) return launch_cull_t<true>(mode, d_pc, vdwk, ewk, out, pts, (pts.n + 63) / 64, 1, 1, stream);
    const int ni = out.i_end - out.i_begin;
    const int ti = (ni + 3) / 4, tj = (g.dims[1] + 1 + 3) / 4, tk = (g.dims[2] + 1 + 3) / 4;
    return launch_cull_t<false>(mode, d_pc, vdwk, ewk, out, pts, (int64_t)ti * tj * tk, tj, tk, stream);
}

}  // namespace ceg
