// ceg_pairfrac.h -- the guest-guest pair sum of one trial placement on FRACTIONAL coordinates, for one wave64: what k_pairs_frac
// (ceg_pairs.hip) and k_mcw_pairs_frac (ceg_mc.hip, BASELINE config 5) share.  single_contribution_vdw,
// src/energy.jl:407-427, with unsafe_periodic_distance2! (src/utils.jl:294-302) and the rule energies of src/interactions.jl:367-406.
//
// Valid on the fast path only (rules inside the domain of ceg_math.h, every perpendicular width of the cell above two cutoffs -- what
// the reference demands of an MC cell).  The guest atoms are also kept as f = invmat * pos.  A pair test is d = f_trial - f_atom, wrap
// as fract(d + 1/2) - 1/2, v = mat * d, r2 = |v|^2: 16 FP64 instructions in an upper-triangular cell where the literal form has 47
// separate flops.  A pair whose wrapped difference is within rounding of +-1/2 lies beyond the cutoff under either image; candidates
// within 1e-9 of the cutoff are re-measured in the reference's operation order from the Cartesian positions when the queue is worked
// off, so the cutoff decision (energy.jl:422) is the reference's.  The loop has no branch per test: the trial atoms are taken K at a time
// (independent dependency chains), candidates go to a 384-entry LDS queue (one 16-byte write each) that is looked at once per block of
// 64 atoms and worked off in full batches of 64 (the remainder stays queued), with one branch-free Lennard-Jones + CoulombEwaldDirect
// record per pair-table entry.
// MM: the molecule has exactly MM atoms (1-4: fractional coordinates in registers); 0: any size, four at a time from LDS.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>

#include "ceg_internal.h"
#include "ceg_math.h"
#include "ceg_consumers.h"

namespace ceg_pairfrac {

using ceg::DevRule;

constexpr int FQCAP = 384;

// waves per SIMD asked of the kernels built on this header (3: 168 VGPRs, no scratch; 4 measured: see profiles/r04_consumers_pairs.txt)
#ifndef CEG_PAIRFRAC_WAVES
#define CEG_PAIRFRAC_WAVES 3
#endif

// a pair-table entry whose rules are at most one Lennard-Jones and one CoulombEwaldDirect term (+ NoInteraction) as one branch-free record
struct __attribute__((aligned(16))) PairFast {
    double c4eps, sigma2, qq, alpha, shift;        // 4 eps, sigma^2, coulombic q1 q2, alpha, sum of the shifts
    int32_t cls, _pad;                             // 1: this record is the whole entry; 0: walk the rules
};

struct __attribute__((aligned(16))) FracHit {      // a queued candidate pair
    double r2;
    int32_t t, ia;                                 // pair-table index; atom index << 4 | trial atom (for the band around the cutoff)
};

// (erfc(alpha r)/r records, see ErfcTable below)
constexpr int ERFC_SHIFT = 15;                  // bits of the high word below the interval key
constexpr int ERFC_REC = 10;                    // doubles per record: a0 ... a6 + padding to 80 B = 5 x 16 B (an odd multiple of 16 B sends 16
                                                // consecutive intervals to 16 different bank groups; at 64 B only four groups were in use)

// The pair table as the kernels stage it in LDS -- only the entries (kind of a guest atom, trial atom a) the molecule on trial can meet,
// entry kind * m + a: the PairFast records, then the rules of these entries (for those that are not one record), then their offsets.
// (The whole table of the fixture force field -- 20 kinds: 26 KB -- beside the queue left room for two workgroups per CU only.)
struct FracTable {
    const PairFast* fast;      // [nkinds * m]
    const DevRule* rules;      // [nrules]
    const int32_t* off;        // [nkinds * m + 1]
    int32_t nrules, nentries;
    // erfc(alpha r)/r of the CoulombEwaldDirect terms as a function of s = r^2 (ErfcTable below): [eni][ERFC_REC] doubles, nullptr / 0 when the
    // entries do not share one alpha (the records then take exp(-x^2) erfcx(x) by the polynomials of ceg_math.h)
    const double* etab;        // [eni][ERFC_REC]
    int32_t ebase, eni;
};
__host__ __device__ inline size_t frac_table_bytes(int nentries, int nrules, int eni = 0)
{
    const size_t b = sizeof(PairFast) * (size_t)nentries + sizeof(DevRule) * (size_t)(nrules > 0 ? nrules : 1) + sizeof(int32_t) * ((size_t)nentries + 1);
    return ((b + 15) & ~(size_t)15) + sizeof(double) * ERFC_REC * (size_t)eni;
}

// ---- erfc(alpha sqrt(s))/sqrt(s) on [s_min, s_max] for the rule arithmetic of the queue: intervals whose key is the exponent and the top
// five mantissa bits of s (32 per octave: clearing the low bits of s gives the interval's lower end), per interval the degree-6 interpolant
// at the Chebyshev nodes in t = s - s_lo, fitted in long double (the construction of the grid kernels' r^2-indexed tables, csrc/ceg_api.hip).
// One record = ERFC_REC doubles (a0 ... a6, padding): four 16-byte LDS reads and six FMAs replace square root, reciprocal square root, exp and the
// erfcx polynomial (~60 instructions per queued pair).  Checked here against erfcl at 17 points per interval with the kernel's Horner
// form: used only if every error is below 1e-13 of the value + 1e-15 of the function at s_min (towards the cutoff erfc has decayed by
// five and more decades and the degree-6 fit holds ~1e-11 of those values: 1e-16 of the terms that make up the sum).
struct ErfcTable {
    std::vector<double> rec;                    // [ni][ERFC_REC]
    int32_t base = 0, ni = 0;
    double worst = 0.0;
};
inline bool build_erfc_table(double alpha, double s_min, double s_max, ErfcTable& out)
{
    auto key_of = [](double s) { uint64_t b; memcpy(&b, &s, 8); return (int32_t)((uint32_t)(b >> 32) >> ERFC_SHIFT); };
    out = ErfcTable{};
    if (!(alpha > 0.0) || !(s_min > 0.0) || !(s_max > s_min) || !std::isfinite(s_max)) return false;
    const int32_t base = key_of(s_min), last = key_of(s_max * (1.0 + 4e-9));
    const int32_t ni = last - base + 1;
    if (ni < 1 || ni > 512) return false;
    constexpr int ND = 7;
    const long double PI = 3.14159265358979323846264338327950288L, a = alpha;
    long double node[ND];
    for (int k = 0; k < ND; ++k) node[k] = cosl(PI * (k + 0.5L) / (long double)ND);
    auto B0 = [&](long double s) { const long double r = sqrtl(s); return erfcl(a * r) / r; };
    out.rec.assign((size_t)ni * ERFC_REC, 0.0);
    const long double top = B0((long double)s_min);
    double worst = 0.0;
    for (int32_t i = 0; i < ni; ++i) {
        const uint64_t lo_bits = (uint64_t)(uint32_t)((base + i) << ERFC_SHIFT) << 32;
        const uint64_t hi_bits = (uint64_t)(uint32_t)((base + i + 1) << ERFC_SHIFT) << 32;
        double s_lo, s_hi;
        memcpy(&s_lo, &lo_bits, 8); memcpy(&s_hi, &hi_bits, 8);
        const long double hh = 0.5L * ((long double)s_hi - (long double)s_lo), s_mid = (long double)s_lo + hh;
        long double V[ND][ND + 1];
        for (int r = 0; r < ND; ++r) {
            long double pw = 1.0L;
            for (int c = 0; c < ND; ++c) { V[r][c] = pw; pw *= node[r]; }
            V[r][ND] = B0(s_mid + node[r] * hh);
        }
        for (int c = 0; c < ND; ++c) {                      // Gauss-Jordan with partial pivoting
            int piv = c;
            for (int r = c + 1; r < ND; ++r) if (fabsl(V[r][c]) > fabsl(V[piv][c])) piv = r;
            for (int q = 0; q <= ND; ++q) { const long double tmp = V[c][q]; V[c][q] = V[piv][q]; V[piv][q] = tmp; }
            const long double d = V[c][c];
            for (int q = 0; q <= ND; ++q) V[c][q] /= d;
            for (int r = 0; r < ND; ++r) if (r != c) {
                const long double g = V[r][c];
                for (int q = 0; q <= ND; ++q) V[r][q] -= g * V[c][q];
            }
        }
        // P(u), u = (t - hh)/hh with t = s - s_lo  ->  coefficients in t
        long double cu[ND], ct[ND] = {0, 0, 0, 0, 0, 0, 0};
        long double sc = 1.0L;
        for (int c = 0; c < ND; ++c) { cu[c] = V[c][ND] * sc; sc /= hh; }
        for (int c = 0; c < ND; ++c) {
            long double binom = 1.0L;
            for (int k = 0; k <= c; ++k) {
                ct[k] += cu[c] * binom * powl(-hh, c - k);
                binom = binom * (c - k) / (k + 1);
            }
        }
        double* co = &out.rec[(size_t)i * ERFC_REC];
        for (int c = 0; c < ND; ++c) co[c] = (double)ct[c];
        for (int q = 0; q <= 16; ++q) {
            const double t = (double)(((long double)q / 16.0L) * 2.0L * hh * (1.0L - 1e-12L));
            double pv = co[ND - 1];
            for (int c = ND - 2; c >= 0; --c) pv = std::fma(pv, t, co[c]);
            const long double ref = B0((long double)s_lo + (long double)t);
            worst = std::fmax(worst, (double)(fabsl((long double)pv - ref) / (fabsl(ref) + 1e-2L * top)));
        }
    }
    out.base = base; out.ni = ni; out.worst = worst;
    if (!(worst < 1e-13)) { out.rec.clear(); out.ni = 0; return false; }
    return true;
}
// the alpha the CoulombEwaldDirect terms of the records share (0: none or several)
inline double shared_alpha(const PairFast* fast, size_t n)
{
    double alpha = 0.0;
    for (size_t i = 0; i < n; ++i)
        if (fast[i].cls && fast[i].qq != 0.0) {
            if (alpha == 0.0) alpha = fast[i].alpha;
            else if (alpha != fast[i].alpha) return 0.0;
        }
    return alpha;
}

// host: the record of one pair-table entry (rules [q0, q1) of `rules`); `walked` rules of other kinds make it cls = 0
inline PairFast make_pair_fast(const DevRule* rules, int32_t q0, int32_t q1, double coulombic)
{
    // (v - shift summed in another order than the rule loop: inside the 1e-9 of the pair sum, like the rest of the fast path)
    PairFast P{0.0, 0.0, 0.0, 0.0, 0.0, 1, 0};
    int nlj = 0, nced = 0;
    for (int32_t q = q0; q < q1; ++q) {
        const DevRule& R = rules[q];
        if (R.kind == CEG_LENNARDJONES && nlj == 0) { P.c4eps = 4.0 * R.p0; P.sigma2 = R.p1 * R.p1; P.shift += R.shift; ++nlj; }
        else if (R.kind == CEG_COULOMB_EWALD_DIRECT && nced == 0) { P.alpha = R.p0; P.qq = coulombic * R.p1 * R.p2; P.shift += R.shift; ++nced; }
        else if (R.kind == CEG_NOINTERACTION) P.shift += R.shift;
        else P.cls = 0;
    }
    return P;
}

// the libm-grade rule energies behind a call: inlined, their exp / erfc / pow temporaries set the register count of the whole kernel
// while they serve the pairs closer than 0.5 A only
__device__ __attribute__((noinline)) inline double rule_energy_call(const DevRule* R, double r2, double coulombic)
{
    return ceg_consumers::rule_energy(*R, r2, coulombic);
}

// the workgroup copies the table into LDS (the caller synchronises); mat12 <- mat[9], -mat * (1/2, 1/2, 1/2)
__device__ __forceinline__ void stage(unsigned char* s_table, const FracTable& tab, const double* mat, double* mat12, int tid, int nthreads,
                                      PairFast*& fastrec, DevRule*& rules, int32_t*& offset, const double*& etab)
{
    fastrec = reinterpret_cast<PairFast*>(s_table);
    rules = reinterpret_cast<DevRule*>(fastrec + tab.nentries);
    offset = reinterpret_cast<int32_t*>(rules + (tab.nrules > 0 ? tab.nrules : 1));
    double* et = reinterpret_cast<double*>(s_table + frac_table_bytes(tab.nentries, tab.nrules, 0));
    etab = et;
    for (int t = tid; t < ERFC_REC * tab.eni; t += nthreads) et[t] = tab.etab[t];
    for (int t = tid; t < tab.nentries; t += nthreads) fastrec[t] = tab.fast[t];
    for (int t = tid; t < tab.nrules; t += nthreads) rules[t] = tab.rules[t];
    for (int t = tid; t <= tab.nentries; t += nthreads) offset[t] = tab.off[t];
    if (tid < 9) mat12[tid] = mat[tid];
    if (tid >= 9 && tid < 12) mat12[tid] = -0.5 * (mat[tid - 9] + mat[tid - 6] + mat[tid - 3]);
}

// One wave, one trial placement at a time.  The kernel fills the members, then per placement: load() -> {scan(), flush(false)}* -> sum().
template <int MM, bool TRI>
struct FracWave {
    const PairFast* fastrec;       // LDS: the staged table
    const DevRule* rules;
    const int32_t* offset;
    const double* etab;            // LDS: erfc(alpha r)/r records (eni > 0), key of the first one
    int32_t ebase, eni;
    const double* s_mat;           // LDS: stage()'s mat12
    double* t3;                    // LDS of this wave: Cartesian positions of the trial atoms [3 m] (band, cell range)
    double* ft;                    // LDS of this wave: their fractional coordinates + 1/2
    FracHit* hq;                   // LDS of this wave: FQCAP entries
    const double4* frac;           // guest atoms: fx, fy, fz, (kind | molecule << 32) bits
    const double4* cart;           // the same atoms, Cartesian, same indices (band only)
    const double* geom;            // mat[9], invmat[9] in device memory (pair_distance2_literal_call)
    double cutoff2, band, cutoff2_band, coulombic;
    int m, exclude, lane;
    int qn;                        // wave-uniform
    double e;
    int a_next;

    // positions of the trial atoms -> LDS (pos(i): element i of the 3 m coordinates)
    template <class Pos>
    __device__ __forceinline__ void load(const double* invmat, Pos&& pos)
    {
        if (lane < 3 * m) t3[lane] = pos(lane);
        __builtin_amdgcn_wave_barrier();
        if (lane < 3 * m) {
            const int a = lane / 3, ax = lane - 3 * a;
            ft[lane] = __builtin_fma(invmat[6 + ax], t3[3 * a + 2], __builtin_fma(invmat[3 + ax], t3[3 * a + 1], invmat[ax] * t3[3 * a])) + 0.5;       // (+ 1/2: see `tests`)
        }
        __builtin_amdgcn_wave_barrier();
        qn = 0;
        e = 0.0;
        a_next = 0;
    }

    // Lennard-Jones + erfc(alpha r)/r of one pair from the r^2-indexed records: no square root, no exp, no erfcx, no branch (the interval
    // index is clamped to the table: outside [1, cutoff^2] the value is meaningless and the caller does not use it)
    __device__ __forceinline__ double record_value(const double r2, const PairFast& P) const
    {
        const int hi = __double2hiint(r2);
        const double tt = r2 - __hiloint2double(hi & (int)(0xffffffffu << ERFC_SHIFT), 0);
        int k = (int)((unsigned)hi >> ERFC_SHIFT) - ebase;
        k = k < 0 ? 0 : (k > eni - 1 ? eni - 1 : k);
        const double2* rec = reinterpret_cast<const double2*>(etab + ERFC_REC * (size_t)k);
        const double2 a01 = rec[0], a23 = rec[1], a45 = rec[2], a6 = rec[3];
        double b0 = __builtin_fma(a6.x, tt, a45.y);
        b0 = __builtin_fma(b0, tt, a45.x);
        b0 = __builtin_fma(b0, tt, a23.y);
        b0 = __builtin_fma(b0, tt, a23.x);
        b0 = __builtin_fma(b0, tt, a01.y);
        b0 = __builtin_fma(b0, tt, a01.x);
        const double q2 = P.sigma2 * ceg::fast_rcp(r2);
        const double x6 = q2 * q2 * q2;
        const double v = __builtin_fma(P.c4eps * x6, x6 - 1.0, -P.shift);
        return __builtin_fma(P.qq, b0, v);
    }

    // full batches of 64 queued pairs (everything when `all`); the remainder moves to the front
    __device__ __forceinline__ void flush(const bool all)
    {
        __builtin_amdgcn_wave_barrier();
        const int nfull = all ? qn : (qn & ~63);
        // one queued pair, every case: the band around the cutoff re-measured, records / polynomials / rule walk / libm-grade call
        auto one = [&](const int i) -> double {
            const FracHit H = hq[i];
            double r2 = H.r2;
            const int t = H.t;
            if (__builtin_expect(r2 >= cutoff2 - band, 0)) {             // the cutoff decision is the reference's (utils.jl:294-302 as written)
                const int ia = H.ia, a = ia & 15;
                const double4 A = cart[ia >> 4];
                r2 = ceg_consumers::pair_distance2_literal_call(geom, t3[3 * a] - A.x, t3[3 * a + 1] - A.y, t3[3 * a + 2] - A.z);
                if (!(r2 < cutoff2)) return 0.0;                         // energy.jl:422
            }
            double acc = 0.0;
            if (eni > 0 && r2 >= 1.0 && fastrec[t].cls) {
                acc = record_value(r2, fastrec[t]);
            } else if (r2 >= 0.25) {
                double r, rinv;
                ceg::fast_sqrt_rsqrt(r2, r, rinv);
                const PairFast P = fastrec[t];
                if (P.cls) {
                    const double q2 = P.sigma2 * (rinv * rinv);
                    const double x6 = q2 * q2 * q2;
                    double v = __builtin_fma(P.c4eps * x6, x6 - 1.0, -P.shift);
                    if (P.qq != 0.0) {
                        const double x = P.alpha * r;
                        v = __builtin_fma(P.qq * rinv, ceg::fast_exp_neg(-(x * x)) * ceg::erfcx_poly(x), v);
                    }
                    acc = v;
                } else {
                    for (int q = offset[t]; q < offset[t + 1]; ++q) acc += ceg_consumers::rule_energy_fast(rules[q], r2, r, rinv, coulombic);
                }
            } else {
                for (int q = offset[t]; q < offset[t + 1]; ++q) acc += rule_energy_call(&rules[q], r2, coulombic);
            }
            return acc;
        };
        int i = lane;
        if (eni > 0) {
            // two batches at a time: the record arithmetic of both pairs runs unconditionally (two independent chains; the table index is
            // clamped, so a pair the records do not serve computes a value nobody uses), the exceptions go through `one`
            const double lim = cutoff2 - band;
            for (; i + 64 < nfull; i += 128) {
                const FracHit H0 = hq[i], H1 = hq[i + 64];
                const PairFast P0 = fastrec[H0.t], P1 = fastrec[H1.t];
                const bool ok0 = H0.r2 >= 1.0 && H0.r2 < lim && P0.cls != 0, ok1 = H1.r2 >= 1.0 && H1.r2 < lim && P1.cls != 0;
                const double v0 = record_value(H0.r2, P0), v1 = record_value(H1.r2, P1);
                e += ok0 ? v0 : 0.0;
                e += ok1 ? v1 : 0.0;
                if (__builtin_expect(!ok0, 0)) e += one(i);
                if (__builtin_expect(!ok1, 0)) e += one(i + 64);
            }
        }
        for (; i < nfull; i += 64) e += one(i);
        const int rest = qn - nfull;              // < 64
        FracHit Hm{0.0, 0, 0};
        if (lane < rest) Hm = hq[nfull + lane];
        __builtin_amdgcn_wave_barrier();
        if (lane < rest) hq[lane] = Hm;
        __builtin_amdgcn_wave_barrier();
        qn = rest;
    }

    // The scan of the atoms and the rule arithmetic alternate: `scan` tests blocks of 64 atoms from entry l0 until the queue could
    // overflow in the next block (then the caller works the queue off).  Everything the scan keeps in registers (cell matrix, fractional
    // coordinates of the trial atoms) is re-read at the start of each scan, so that it is NOT live across the rule arithmetic -- with
    // both sets live the kernel needs 150-170 VGPRs and spills.  locate(l) -> index into frac / cart of entry l (any valid index for
    // l >= total); returns the entry the next scan starts from.
    template <class Locate>
    __device__ __forceinline__ int scan(Locate&& locate, const int total, int l0)
    {
        constexpr int QROOM = FQCAP - 64 * (MM > 0 ? MM : 4);               // the queue has room for one more step of the scan
        asm volatile("" ::: "memory");
        // the cell matrix in VGPRs (wave-uniform, but the kernel has more uniform values than scalar registers: see k_pairs)
        double Mv[12];
#pragma unroll
        for (int a = 0; a < 12; ++a) Mv[a] = s_mat[a];
        int exclude_v = exclude;                                        // (in a VGPR: as a scalar it was re-read from the kernel arguments every block)
        asm volatile("" : "+v"(exclude_v));
        double ftr[MM > 0 ? MM : 1][3];
        if (MM > 0) {
#pragma unroll
            for (int a = 0; a < MM; ++a) {
                ftr[a][0] = ft[3 * a]; ftr[a][1] = ft[3 * a + 1]; ftr[a][2] = ft[3 * a + 2];
            }
        }
        // K trial atoms (a0 ... a0 + K - 1) against the atom of this lane
        // The wrapped difference d - rint(d) is taken as fract(d + 1/2) - 1/2 (the 1/2 is part of the stored trial coordinates, the
        // -1/2 is folded into the matrix product as -mat * (1/2, 1/2, 1/2)): two instructions per component instead of three.  Lanes
        // without an atom (or with an atom of the excluded molecule) carry NaN in F.x and never compare inside.
        auto tests = [&](auto ktag, const int a0, const double4 F, const int kbase, const int idx) __attribute__((always_inline)) {
            constexpr int K = decltype(ktag)::value;
            double r2[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double f0 = __builtin_amdgcn_fract((MM > 0 ? ftr[k][0] : ft[3 * (a0 + k)]) - F.x);
                const double f1 = __builtin_amdgcn_fract((MM > 0 ? ftr[k][1] : ft[3 * (a0 + k) + 1]) - F.y);
                const double f2 = __builtin_amdgcn_fract((MM > 0 ? ftr[k][2] : ft[3 * (a0 + k) + 2]) - F.z);
                double vx, vy, vz;
                if (TRI) {
                    vx = __builtin_fma(Mv[0], f0, __builtin_fma(Mv[3], f1, __builtin_fma(Mv[6], f2, Mv[9])));
                    vy = __builtin_fma(Mv[4], f1, __builtin_fma(Mv[7], f2, Mv[10]));
                    vz = __builtin_fma(Mv[8], f2, Mv[11]);
                } else {
                    vx = __builtin_fma(Mv[0], f0, __builtin_fma(Mv[3], f1, __builtin_fma(Mv[6], f2, Mv[9])));
                    vy = __builtin_fma(Mv[1], f0, __builtin_fma(Mv[4], f1, __builtin_fma(Mv[7], f2, Mv[10])));
                    vz = __builtin_fma(Mv[2], f0, __builtin_fma(Mv[5], f1, __builtin_fma(Mv[8], f2, Mv[11])));
                }
                r2[k] = __builtin_fma(vz, vz, __builtin_fma(vy, vy, vx * vx));
            }
            // (all K distances before the first queue entry: K independent chains for the scheduler, not K chains one after the other)
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("" : "+v"(r2[k]));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                // candidates: inside the cutoff or in the band around it (those are re-measured when the queue is worked off)
                const bool inside = r2[k] <= cutoff2_band;
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(inside);        // (the ballot of a compare is its scalar result as it stands)
                if (inside) {
                    const int slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                    hq[slot] = FracHit{r2[k], kbase + (a0 + k), (idx << 4) | (a0 + k)};
                }
                qn += __popcll(mask);
            }
        };
        auto masked = [&](double4 F, const bool live) __attribute__((always_inline)) -> double4 {
            F.x = live ? F.x : __builtin_nan("");
            return F;
        };
        int a0 = a_next;                                                // (MM == 0: the molecule is taken four atoms at a time)
        if (MM > 0) {
            // Three blocks in registers, the loop body written out three times: the block fetched after a step is used two steps later
            // (a block is ~0.3 us of work, an L2 hit takes longer), and no register is copied -- a copy would wait for its load.
            int ixa = locate(l0 + lane), ixb = locate(l0 + 64 + lane), ixc = locate(l0 + 128 + lane);
            double4 Fa = frac[ixa], Fb = frac[ixb], Fc = frac[ixc];
            auto step = [&](double4& F, int& ix) __attribute__((always_inline)) {
                const long long bits = __double_as_longlong(F.w);
                const int kind1 = (int)(bits & 0xffffffffll), mol = (int)(bits >> 32);
                const bool live = l0 + lane < total && mol != exclude_v && mol >= 0;
                tests(std::integral_constant<int, (MM > 0 ? MM : 1)>{}, 0, masked(F, live), kind1 * m, ix);
                ix = locate(l0 + 192 + lane);
                F = frac[ix];
                l0 += 64;
            };
            while (true) {
                if (!(l0 < total && qn <= QROOM)) break;
                step(Fa, ixa);
                if (!(l0 < total && qn <= QROOM)) break;
                step(Fb, ixb);
                if (!(l0 < total && qn <= QROOM)) break;
                step(Fc, ixc);
            }
        } else {
            int ix = locate(l0 + lane), ix1 = locate(l0 + 64 + lane);
            double4 F = frac[ix], F1 = frac[ix1];
            while (l0 < total && qn <= QROOM) {
                const long long bits = __double_as_longlong(F.w);
                const int kind1 = (int)(bits & 0xffffffffll), mol = (int)(bits >> 32);
                const bool live = l0 + lane < total && mol != exclude_v && mol >= 0;
                const int kbase = kind1 * m;
                const double4 Fm = masked(F, live);
                switch (m - a0 < 4 ? m - a0 : 4) {
                    case 4: tests(std::integral_constant<int, 4>{}, a0, Fm, kbase, ix); break;
                    case 3: tests(std::integral_constant<int, 3>{}, a0, Fm, kbase, ix); break;
                    case 2: tests(std::integral_constant<int, 2>{}, a0, Fm, kbase, ix); break;
                    default: tests(std::integral_constant<int, 1>{}, a0, Fm, kbase, ix); break;
                }
                a0 += 4;
                if (a0 >= m) {
                    a0 = 0;
                    l0 += 64;
                    F = F1; ix = ix1;
                    ix1 = locate(l0 + 64 + lane);
                    F1 = frac[ix1];
                }
            }
        }
        a_next = a0;
        return l0;
    }

    // the pair sum of the placement (every lane)
    __device__ __forceinline__ double sum()
    {
        flush(true);
        double s = e;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        return s;
    }
};

}  // namespace ceg_pairfrac
