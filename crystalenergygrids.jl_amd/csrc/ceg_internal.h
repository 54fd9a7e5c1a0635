// ceg_internal.h -- types shared by the HIP kernels (ceg_kernels.hip) and the C-ABI
// layer (ceg_api.hip).  Not installed; the public interface is include/ceg_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ceg_hip.h"

namespace ceg {

// What one launch computes.
enum Mode : int { MODE_VDW = 0, MODE_COULOMB = 1, MODE_FUSED = 2 };

// One pair rule as the kernels want it (derived from ceg_rule_t at plan creation;
// NoInteraction / CoulombEwaldDirect rules are dropped because derivativesGrid returns
// exact zeros for them, src/interactions.jl:458-461).
//   LJ          : p0 = eps, p1 = sigma^2
//   Buckingham  : p0 = A,   p1 = B,  p2 = C
//   HardSphere  : p0 = (R1+R2)^2
struct DevRule {
    int32_t kind;
    int32_t _pad;
    double  p0, p1, p2;
    double  shift;
};

// Geometry + periodic-distance set-up, passed to kernels by value (kernarg -> SGPRs).
struct Geom {
    double mat[9];      // column-major supercell matrix   (src/probes.jl:27)
    double invmat[9];   // its inverse                      (src/probes.jl:28)
    double size[3];     // GridCoordinatesSetup.size        (src/coordinates.jl:34)
    double shift[3];    //                      .shift      (:35)
    double delta[3];    //                      .delta      (:39)
    int32_t dims[3];    //                      .dims       (:36-37)
    int32_t ortho;      // src/utils.jl:148
    double safemin2;    // src/utils.jl:150-152, squared
    double cutoff2;     // src/probes.jl:75
    double alpha;       // ewald.alpha, 1/A                 (src/ewald.jl:203)
    int32_t diag;       // 1 if mat is exactly diagonal (wrapped image == nearest image)
    int32_t _pad;
};

// A multi-probe plan holds the rule tables of up to CEG_MAX_PROBES probe atoms (one VdW grid each) of ONE framework: the K VdW
// grids + the Coulomb grid of a setup_RASPA call (src/raspa.jl:497-520 builds them one after the other) come out of one pass
// over one image list.  (CEG_MAX_PROBES = 4: include/ceg_hip.h)

// Where results go.  Grid mode: 8 float channels per grid, written with
// _set_gridpoint! semantics (src/grids.jl:118-135).  Raw mode (eval_points): the 8 FP64
// numbers of compute_derivatives_* per point, before clamping/scaling.
struct Output {
    float*  vdw;            // device, may be null
    float*  coulomb;        // device, may be null
    double* raw_vdw;        // device [8*npoints], may be null
    double* raw_coulomb;
    int64_t channel_stride; // floats between channels
    int32_t i_origin;       // x-plane stored at offset 0
    int32_t i_begin, i_end; // x-planes computed by this launch
    double  lambda_vdw, thr_vdw;
    double  lambda_coulomb, thr_coulomb;
    // multi-probe launches (k_culled<..., NP > 1>): grid p of the launch is written to vdwm[p] with the rules of the plan's
    // probe probe_idx[p]
    float*  vdwm[CEG_MAX_PROBES];
    int32_t probe_idx[CEG_MAX_PROBES];
};

// Arbitrary point list (eval_points) or null for grid mode.
struct Points {
    const double* xyz;      // device [3*n]
    int64_t n;
};

// Atom table for the brute-force kernels (one record per ProbeSystem atom).
struct AtomTable {
    const double4* xyzq;    // x, y, z, charge (0 if no charges given)
    const int32_t* kind;    // 0-based force-field index, -1 if no kinds given
    int64_t n;
};

// Rule table on the device.
struct RuleTable {
    const DevRule* rules;
    const int32_t* offset;  // [nkinds+1]
    int32_t nkinds;
};

// Lattice-image list + bins for the culled kernels.
struct ImageBins {
    const double4* xyzq;    // image position (atom + lattice vector), charge
    const int32_t* kind;    // 0-based kind, -1 when the plan has no rules
    const int32_t* atom;    // index of the ProbeSystem atom this image belongs to
    const double4* atoms;   // the ProbeSystem atoms themselves (exact path for very close pairs)
    const int32_t* bin_start; // [nbx*nby*nbz + 1], bins ordered (bx, by, bz) with bz fastest
    double lo[3];           // lower corner of the binned region
    double inv_bin[3];      // 1 / bin edge
    double bin[3];          // bin edge
    int32_t nb[3];
    int32_t nimages;
};

// Plan constants resident in device memory (uploaded once per plan): the culled kernel reads
// them through a pointer so that only what the hot loop needs lives in scalar registers.
// Per-kind record of the fast VdW classes of k_culled: cls 0 none, 1 Lennard-Jones
// {4 eps, sigma^2, sigma^6, shift}, 2 Buckingham {A, B, C, shift} (hard spheres inside r_exact ignored).
struct FastVdw {
    double p0, p1, p2, shift;
    int32_t cls;
    int32_t _pad;
};

struct PlanConst {
    Geom g;
    ImageBins ib;
    RuleTable rt;
    const FastVdw* fastvdw;    // [nkinds]
    double r_exact2;           // pairs closer than this (A^2) take the exact path
    // function tables of the fast real-space Ewald term (ceg_math.h), built per plan
    const double* erfcx_tab;   // [ERFCX_TAB_N * 6]
    const double* exp2_tab;    // [64]  2^(j/64)
    double erfcx_inv_h;        // 1/h
    double erfcx_mx0_inv_h;    // -x0/h
    double alpha2;             // alpha^2
    // r^2-indexed tables of the real-space Ewald radial functions (EWK = 2, see ceg_math.h): [ew2_ni][CEG_EW2_STRIDE]
    const double* ew2_tab;
    int32_t ew2_ni;            // intervals in the table (<= CEG_EW2_NI_MAX)
    int32_t ew2_base;          // key of the first interval: hi32(r_exact2) >> CEG_EW2_SHIFT
    double ew_k3, ew_k15;      // 2 alpha^2 / 3, 4 alpha^4 / 15: constants of the B_n recurrence in the r^2-table loop
    // single Buckingham class (VDWK = 3): G0(s) = A exp(-B sqrt(s)) on r^2 intervals of its own (CEG_BK2_LOGM), [bk2_ni][CEG_BK2_STRIDE]
    const double* bk2_tab;
    int32_t bk2_ni, bk2_base;
    double bk_B, bk_C, bk_invC;                       // single tabulated Buckingham class: B, C, 1/C and the constants of the
    double bk_nshift, bk_c1, bk_c2, bk_c3, bk_c4;      // scaled hot loop: -shift/C, -B/6, -B/48, B^2/3, -B/160
    double bk_s1, bk_s2, bk_s3;                        // exact path: 1/(6C), -1/(48C), 1/(480C)
    int32_t all_simple;        // grid mode: every image a tile can keep is provably the fractionally wrapped one (no per-candidate test)
    int32_t nprobes;           // multi-probe plans: number of probes (0 = ordinary plan)
    // multi-probe plans (all probes Lennard-Jones-only): per probe the rule table (exact path) and the per-kind fast records
    RuleTable rtm[CEG_MAX_PROBES];
    const FastVdw* fastm[CEG_MAX_PROBES];
};

// shared between the host table builder and the kernels
constexpr int CEG_ERFCX_TAB_N = 128;     // pieces of the erfcx table (= ERFCX_TAB_N of ceg_math.h)
#ifndef CEG_R_EXACT2_VALUE
#define CEG_R_EXACT2_VALUE 4.0
#endif
constexpr double CEG_R_EXACT2 = CEG_R_EXACT2_VALUE;     // pairs closer than this (A^2) take the exact path

// r^2-indexed Ewald tables: the interval of s = r^2 is read off the bits of s -- exponent + the CEG_EW2_LOGM leading
// mantissa bits, i.e. 2^LOGM intervals per octave -- and each interval holds two degree-6 polynomials in t = s - s_lo:
//   B0(s) = erfc(alpha sqrt(s)) / sqrt(s),   C(s) = (2 alpha / sqrt(pi)) exp(-alpha^2 s)
// from which derivatives_ewald (src/ewald.jl:299-312) follows by the recurrence B_{n+1} = ((2n+1) B_n + (2 alpha^2)^n C) / s.
// Record = 7 + 7 coefficients = 112 B: with 16-byte LDS reads, 16 consecutive intervals fall into 16 different bank
// groups (a 96-byte record of two degree-5 polynomials would use 8).  Degree 5 left 7e-10 of the term itself in the last
// octave before the cutoff (visible at 1e-9 in sums that hold far pairs only); degree 6: <= 1e-11 there, 2e-15 below 8 A.
constexpr int CEG_EW2_LOGM = 5;
constexpr int CEG_EW2_SHIFT = 20 - CEG_EW2_LOGM;               // bits of the high word below the interval key
constexpr int CEG_EW2_STRIDE = 14;                             // doubles per interval record
#ifndef CEG_EW2_LDS_STRIDE
#define CEG_EW2_LDS_STRIDE 14                                   // LDS stride of the record in the single-probe kernels (15: see k_culled)
#endif
constexpr int CEG_EW2_NI_MAX = 176;                            // cutoff 12 A from r_exact 2 A: 165 intervals
// Single Buckingham class: G0(s)/C = (A/C) exp(-B sqrt(s)) on the SAME intervals as the Ewald pair (one key and one t per
// candidate serve both tables), one degree-7 polynomial per interval (64 B).  Round 2 used degree 5 on 64 intervals per octave
// (3e-13 of the pair energy: up to 77 ULP in stored values where the attractive and repulsive sums of a channel cancel);
// degree 7 on 32 per octave reaches 2e-15 (fit in long double, checked per plan against CEG_BK2_TOL) with 2/3 of the LDS
// footprint and half the distinct records per wave.
constexpr int CEG_BK2_ND = 8;                                  // coefficients per record (degree 7)
// doubles per record: 80 B, not 64 -- the lanes of a wave read 10-40 CONSECUTIVE intervals, and with a 64-byte stride the records
// start on only 4 distinct bank groups (16 k mod 64 dwords): every ds_read_b128 of the table was a 4-way conflict
// (SQ_LDS_BANK_CONFLICT 3.3 x the LDS instruction cycles of the Na kernels); 20 k mod 64 visits all 16 groups
constexpr int CEG_BK2_STRIDE = 10;
constexpr int CEG_BK2_LOGM = CEG_EW2_LOGM;
constexpr int CEG_BK2_SHIFT = 20 - CEG_BK2_LOGM;
constexpr int CEG_BK2_NI_MAX = CEG_EW2_NI_MAX;
constexpr double CEG_BK2_TOL = 2e-14;                          // of G0 + C/s^3, the pair energy the table value is added into

// launchers (ceg_kernels.hip)
hipError_t launch_bruteforce(int mode, const Geom& g, const AtomTable& atoms, const RuleTable& rt,
                             const Output& out, const Points& pts, hipStream_t stream);
// ewk: real-space Ewald arithmetic of the hot loop -- 0 libm-grade erfc / exp, 1 erfcx table + exp (alpha*cutoff <= 5),
// 2 r^2-indexed tables
hipError_t launch_culled(int mode, const PlanConst* d_pc, const Geom& g, int vdwk, int ewk,
                         const Output& out, const Points& pts, hipStream_t stream);
// multi-probe grid build: np (2..CEG_MAX_PROBES; with mode == MODE_FUSED at most CEG_MAX_PROBES_FUSED) Lennard-Jones probes of a
// multi-probe plan in one pass, out.vdwm / out.probe_idx say which; mode MODE_VDW or MODE_FUSED (+ the Coulomb grid, EWK = 2)
constexpr int CEG_MAX_PROBES_FUSED = 2;
hipError_t launch_culled_multi(int mode, int np, const PlanConst* d_pc, const Geom& g, const Output& out, hipStream_t stream);

}  // namespace ceg
