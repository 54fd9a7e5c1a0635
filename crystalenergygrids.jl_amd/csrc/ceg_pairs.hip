// ceg_pairs.hip -- guest-guest pair energy of a rigid molecule for many trial placements
// (SURVEY 8f row f3): single_contribution_vdw_noneighbour, src/energy.jl:407-427, with
// unsafe_periodic_distance2! (src/utils.jl:294-302) and the rule energies of
// src/interactions.jl:367-406 / :589-595.
//
// One wave64 per placement: lanes stride over the guest atoms of the system (coalesced 32-B reads),
// each lane loops over the <= 16 atoms of the molecule, wave reduction at the end.  In MC cells much larger than
// the cutoff sphere the atoms are kept sorted by neighbour cell (ceg_consumers.h; the reference's CellListMap
// branch, energy.jl:399-404) and the lanes stride over the cell rows the molecule can reach instead.  The pair table
// is small (kinds^2 rule runs) and staged in LDS.  FP64 throughout.  Two kernels: k_pairs_frac (round 4; handles on the fast path:
// pair tests on fractional coordinates, ceg_math.h arithmetic) and k_pairs (everything else: literal wrap, libm-grade exp / erfc,
// tables beyond the LDS budget).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>
#include <vector>

#include "../../include/ceg_hip.h"
#include "ceg_internal.h"
#include "ceg_math.h"
#include "ceg_consumers.h"
#include "ceg_pairfrac.h"

using ceg::DevRule;
using ceg_consumers::rule_energy;
using ceg_consumers::rule_energy_fast;
using ceg_pairfrac::FQCAP;
using ceg_pairfrac::FracHit;
using ceg_pairfrac::FracTable;
using ceg_pairfrac::PairFast;
using ceg_pairfrac::frac_table_bytes;

extern "C" void ceg_set_last_error_(const char* msg);

namespace {

constexpr int PAIRS_MAX_ATOMS = 16;
constexpr int PAIRS_WAVES = 4;

struct PairsGeom {
    double mat[9], invmat[9];
    double cutoff2, coulombic;
    int32_t nkinds, m, exclude;
    int32_t kinds[PAIRS_MAX_ATOMS];
    const double* geom;                  // mat[9], invmat[9] in device memory (the literal fall-back of the fast pair distance)
    int32_t nb[3];                       // neighbour cells (CELLS): bins per fractional axis, z fastest in cell_start
    double hfrac[3];
};

// the libm-grade rule energies behind a call: inlined, their exp / erfc / pow temporaries set the register count of the whole kernel
// (168 VGPRs, up to 44 B of scratch) while they serve the pairs closer than 0.5 A only (and the handles whose exp / erfc arguments leave
// the domain of ceg_math.h)
__device__ __attribute__((noinline)) double pairs_rule_energy_call(const DevRule* R, double r2, double coulombic) { return rule_energy(*R, r2, coulombic); }

// WRAP: 0 the reference's operation order for every pair; 1 / 2: ceg_consumers::pair_distance2_fast (2: upper-triangular cell)
template <bool FAST, bool TABLE_IN_LDS, bool CELLS, int WRAP>
__global__ __launch_bounds__(64 * PAIRS_WAVES, 3) void k_pairs(PairsGeom g, const DevRule* __restrict__ g_rules,
                                                             const int32_t* __restrict__ g_offset, int32_t nrules,
                                                             const double4* __restrict__ atoms,      // x, y, z, (kind | molecule) bits
                                                             const int32_t* __restrict__ cell_start, // CELLS: atoms sorted by cell, [ncells + 1]
                                                             int64_t natoms, const double* __restrict__ trial, int64_t n,
                                                             double* __restrict__ out)
{
    // the pair table is read once per in-cutoff pair with lane-dependent indices: from global memory that
    // is a dependent chain of vector loads inside a divergent branch, so it is staged in LDS when it fits
    extern __shared__ __attribute__((aligned(16))) unsigned char s_table[];
    const DevRule* rules = g_rules;
    const int32_t* offset = g_offset;
    if (TABLE_IN_LDS) {
        DevRule* lr = reinterpret_cast<DevRule*>(s_table);
        int32_t* lo = reinterpret_cast<int32_t*>(s_table + sizeof(DevRule) * (size_t)(nrules > 0 ? nrules : 1));
        const int nt = g.nkinds * g.nkinds + 1;
        for (int t = threadIdx.x; t < nrules; t += 64 * PAIRS_WAVES) lr[t] = g_rules[t];
        for (int t = threadIdx.x; t < nt; t += 64 * PAIRS_WAVES) lo[t] = g_offset[t];
        __syncthreads();
        rules = lr;
        offset = lo;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * PAIRS_WAVES + wave;
    if (p >= n) return;
    __shared__ double s_trial[PAIRS_WAVES][PAIRS_MAX_ATOMS * 3];
    __shared__ int32_t s_tk[PAIRS_WAVES][PAIRS_MAX_ATOMS];      // row offsets of the trial atoms' kinds in the pair table (not 16 scalar registers)
    double* t3 = s_trial[wave];
    if (lane < 3 * g.m) t3[lane] = trial[(size_t)p * g.m * 3 + lane];
    if (lane < g.m) s_tk[wave][lane] = g.kinds[lane];
    __builtin_amdgcn_wave_barrier();
    const int32_t* tk = s_tk[wave];
    const int m = g.m, nkinds = g.nkinds, exclude = g.exclude;
    const double cutoff2 = g.cutoff2;
    // The cell matrices of the fast pair distance live in VGPRs (the same value in every lane): as kernel arguments they are scalar
    // registers, the kernel has more wave-uniform values than scalar registers, and the compiler spilled them into VGPR lanes -- 16
    // v_readlane reloads per pair test, a third of the hot loop's VALU instructions (round 4).  The registers are there: 3 waves per SIMD.
    double Mv[9], Iv[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        Mv[a] = g.mat[a];
        Iv[a] = g.invmat[a];
        if (WRAP != 0) {
            asm volatile("" : "+v"(Mv[a]));
            asm volatile("" : "+v"(Iv[a]));
        }
    }
    const double* M = Mv;
    const double* I = Iv;
    // Only ~10 % of the tested pairs are inside the cutoff (MC cells are 2-4 cutoffs wide), but a wave almost
    // always contains one: evaluating the rules under the lane mask would cost every lane the full exp / erfc
    // price per test.  Hits are therefore compacted into a per-wave LDS queue (r2, pair-table index) and the
    // rules are evaluated on dense batches.
    constexpr int QCAP = 128;
    __shared__ double s_qr2[PAIRS_WAVES][QCAP];
    __shared__ int32_t s_qt[PAIRS_WAVES][QCAP];
    double* qr2 = s_qr2[wave];
    int32_t* qt = s_qt[wave];
    int qn = 0;                                   // wave-uniform
    double e = 0.0;
    const double band = 1e-9 * g.cutoff2;
    const double coulombic = g.coulombic;
    auto flush = [&]() {
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < qn; i += 64) {
            const double r2 = qr2[i];
            const int t = qt[i];
            // (the FAST arithmetic stays inline: behind a call -- tried -- the kernel fits four waves per SIMD but spills 200-400 B per
            //  lane around the call and runs 8x slower)
            if (FAST && r2 >= 0.25) {
                double r, rinv;
                ceg::fast_sqrt_rsqrt(r2, r, rinv);
                for (int q = offset[t]; q < offset[t + 1]; ++q) e += rule_energy_fast(rules[q], r2, r, rinv, coulombic);
            } else {
                for (int q = offset[t]; q < offset[t + 1]; ++q) e += pairs_rule_energy_call(&rules[q], r2, coulombic);
            }
        }
        __builtin_amdgcn_wave_barrier();
        qn = 0;
    };
    auto process = [&](const double4 A, const bool have) __attribute__((always_inline)) {
        const long long bits = __double_as_longlong(A.w);
        const int kind1 = (int)(bits & 0xffffffffll), mol = (int)(bits >> 32);
        const bool live = have && mol != exclude;
        for (int a = 0; a < m; ++a) {
            // buffer = pos2 - pos1 (energy.jl:420); invmat * buffer; wrap; mat * frac; norm2
            const double dx = t3[3 * a] - A.x, dy = t3[3 * a + 1] - A.y, dz = t3[3 * a + 2] - A.z;
            const double r2 = WRAP == 0 ? ceg_consumers::pair_distance2_literal(M, I, dx, dy, dz)
                                        : ceg_consumers::pair_distance2_fast<WRAP == 2>(M, I, g.geom, dx, dy, dz, cutoff2, band);
            const bool hit = live && (r2 < cutoff2);               // energy.jl:422 (a NaN distance is no hit there either)
            const unsigned long long mask = __ballot(hit);
            const int cnt = __popcll(mask);
            if (cnt == 0) continue;
            if (qn + cnt > QCAP) flush();
            if (hit) {
                const int slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                qr2[slot] = r2;
                qt[slot] = kind1 * nkinds + tk[a];
            }
            qn += cnt;
        }
    };
    if (!CELLS) {
        // the next 64 atoms are fetched while the current ones are worked on: a block is ~140 instructions of work against a
        // microsecond of load latency, which three waves per SIMD do not cover by themselves (VALU issue 0.67 without the prefetch)
        double4 A = atoms[lane < natoms ? lane : 0];
        for (int64_t l0 = 0; l0 < natoms; l0 += 64) {
            const int64_t ln = l0 + 64 + lane;
            const double4 An = atoms[ln < natoms ? ln : 0];
            process(A, l0 + lane < natoms);
            A = An;
        }
    } else {
        // The atoms are sorted by cell with z fastest: the cells (c0, c1, z-range) the molecule can reach are one or (across the
        // periodic boundary) two contiguous runs of the atom array per (c0, c1).  Lane r takes run r, a wave scan turns the run
        // lengths into offsets, then the lanes stride over the concatenation of the runs.
        __shared__ int32_t s_run[PAIRS_WAVES][64], s_off[PAIRS_WAVES][65];
        int32_t* run = s_run[wave];
        int32_t* off = s_off[wave];
        int first = 0, nbin = 1;
        if (lane < 3) ceg_consumers::cell_range(I, t3, g.m, lane, g.nb[lane], g.hfrac[lane], first, nbin);
        const int b0 = __shfl(first, 0), b1 = __shfl(first, 1), b2 = __shfl(first, 2);
        const int n0 = __shfl(nbin, 0), n1 = __shfl(nbin, 1), n2 = __shfl(nbin, 2);
        const int wrapped = b2 + n2 > g.nb[2] ? 1 : 0;                 // the z range crosses the boundary: two runs per row
        const int nruns = n0 * n1 * (1 + wrapped);
        for (int rbase = 0; rbase < nruns; rbase += 64) {
            const int r = rbase + lane;
            int start = 0, len = 0;
            if (r < nruns) {
                const int row = wrapped ? (r >> 1) : r, part = wrapped ? (r & 1) : 0;
                int c0 = b0 + row / n1, c1 = b1 + row % n1;
                if (c0 >= g.nb[0]) c0 -= g.nb[0];
                if (c1 >= g.nb[1]) c1 -= g.nb[1];
                const int rowbase = (c0 * g.nb[1] + c1) * g.nb[2];
                const int zlo = part ? 0 : b2, zhi = part ? b2 + n2 - g.nb[2] : (wrapped ? g.nb[2] : b2 + n2);   // [zlo, zhi)
                start = cell_start[rowbase + zlo];
                len = cell_start[rowbase + zhi] - start;
            }
            int incl = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            const int total = __shfl(incl, 63);
            __builtin_amdgcn_wave_barrier();
            run[lane] = start;
            off[lane] = incl - len;
            __builtin_amdgcn_wave_barrier();
            auto fetch = [&](int l) -> double4 {                       // entry l of the concatenated runs (the first atom for l >= total)
                int j = 0;                                             // last run whose first entry is <= l
#pragma unroll
                for (int step = 32; step > 0; step >>= 1)
                    if (off[j + step] <= l) j += step;
                return atoms[l < total ? run[j] + (l - off[j]) : 0];
            };
            double4 A = fetch(lane);
            for (int l0 = 0; l0 < total; l0 += 64) {
                const double4 An = fetch(l0 + 64 + lane);              // one block ahead, as in the exhaustive loop
                process(A, l0 + lane < total);
                A = An;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    flush();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
    if (lane == 0) out[p] = e;
}

// ---- round 4: the pair tests on FRACTIONAL coordinates (ceg_pairfrac.h) ---------------------------------------------------------
template <int MM, bool CELLS, bool TRI>
__global__ __launch_bounds__(64 * PAIRS_WAVES, CEG_PAIRFRAC_WAVES) void k_pairs_frac(PairsGeom g, FracTable tab,
                                                                  const double4* __restrict__ frac,     // fx, fy, fz, (kind | molecule) bits
                                                                  const double4* __restrict__ atoms,                    // Cartesian, same order (band only)
                                                                  const int32_t* __restrict__ cell_start, int64_t natoms,
                                                                  const double* __restrict__ trial, int64_t n, double* __restrict__ out, int per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_table[];
    __shared__ double s_mat[12];
    __shared__ double s_trial[PAIRS_WAVES][PAIRS_MAX_ATOMS * 3];
    __shared__ double s_ft[PAIRS_WAVES][PAIRS_MAX_ATOMS * 3];
    __shared__ FracHit s_q[PAIRS_WAVES][FQCAP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ceg_pairfrac::FracWave<MM, TRI> w;
    {
        PairFast* fastrec; DevRule* rules; int32_t* offset; const double* etab;
        ceg_pairfrac::stage(s_table, tab, g.mat, s_mat, threadIdx.x, 64 * PAIRS_WAVES, fastrec, rules, offset, etab);
        __syncthreads();
        w.fastrec = fastrec; w.rules = rules; w.offset = offset; w.etab = etab; w.ebase = tab.ebase; w.eni = tab.eni;
    }
    w.s_mat = s_mat; w.t3 = s_trial[wave]; w.ft = s_ft[wave]; w.hq = s_q[wave];
    w.frac = frac; w.cart = atoms; w.geom = g.geom;
    w.cutoff2 = g.cutoff2; w.band = 1e-9 * g.cutoff2; w.cutoff2_band = w.cutoff2 + w.band; w.coulombic = g.coulombic;
    const int m = MM > 0 ? MM : g.m;
    w.m = m; w.exclude = g.exclude; w.lane = lane;
    // (per_wave consecutive placements per wave: a wave that lives for one placement only is ~40 us of work)
    const int64_t p_first = ((int64_t)blockIdx.x * PAIRS_WAVES + wave) * per_wave;
    const int64_t p_last = p_first + per_wave < n ? p_first + per_wave : n;
    for (int64_t p = p_first; p < p_last; ++p) {
        w.load(g.invmat, [&](int i) { return trial[(size_t)p * m * 3 + i]; });
        if (!CELLS) {
            const int total = (int)natoms;
            auto locate = [&](int l) -> int { return l < total ? l : 0; };
            for (int l0 = 0; l0 < total;) {
                l0 = w.scan(locate, total, l0);
                w.flush(false);
            }
        } else {
            // (the run construction of k_pairs: lane r takes run r of the reachable cells, a wave scan turns run lengths into offsets)
            __shared__ int32_t s_run[PAIRS_WAVES][64], s_off[PAIRS_WAVES][65];
            int32_t* run = s_run[wave];
            int32_t* off = s_off[wave];
            int first = 0, nbin = 1;
            if (lane < 3) ceg_consumers::cell_range(g.invmat, w.t3, m, lane, g.nb[lane], g.hfrac[lane], first, nbin);
            const int b0 = __shfl(first, 0), b1 = __shfl(first, 1), b2 = __shfl(first, 2);
            const int n0 = __shfl(nbin, 0), n1 = __shfl(nbin, 1), n2 = __shfl(nbin, 2);
            const int wrapped = b2 + n2 > g.nb[2] ? 1 : 0;
            const int nruns = n0 * n1 * (1 + wrapped);
            for (int rbase = 0; rbase < nruns; rbase += 64) {
                const int r = rbase + lane;
                int start = 0, len = 0;
                if (r < nruns) {
                    const int row = wrapped ? (r >> 1) : r, part = wrapped ? (r & 1) : 0;
                    int c0 = b0 + row / n1, c1 = b1 + row % n1;
                    if (c0 >= g.nb[0]) c0 -= g.nb[0];
                    if (c1 >= g.nb[1]) c1 -= g.nb[1];
                    const int rowbase = (c0 * g.nb[1] + c1) * g.nb[2];
                    const int zlo = part ? 0 : b2, zhi = part ? b2 + n2 - g.nb[2] : (wrapped ? g.nb[2] : b2 + n2);
                    start = cell_start[rowbase + zlo];
                    len = cell_start[rowbase + zhi] - start;
                }
                int incl = len;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(incl, o);
                    if (lane >= o) incl += up;
                }
                const int total = __shfl(incl, 63);
                __builtin_amdgcn_wave_barrier();
                run[lane] = start;
                off[lane] = incl - len;
                __builtin_amdgcn_wave_barrier();
                auto locate = [&](int l) -> int {                          // index of entry l of the concatenated runs (0 beyond the end)
                    int j = 0;
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1)
                        if (off[j + step] <= l) j += step;
                    return l < total ? run[j] + (l - off[j]) : 0;
                };
                for (int l0 = 0; l0 < total;) {
                    l0 = w.scan(locate, total, l0);
                    w.flush(false);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        const double s = w.sum();
        if (lane == 0) out[p] = s;
        __builtin_amdgcn_wave_barrier();                             // t3 / ft are rewritten for the next placement
    }
}

int perr(int code, const char* msg)
{
    ceg_set_last_error_(msg);
    return code;
}

}  // namespace

struct ceg_pairs {
    int device = 0;
    double mat[9], invmat[9];
    double cutoff2 = 0.0, coulombic = 0.0;
    int32_t nkinds = 0;
    bool fast = false;          // every exp / erfc argument inside the domain of the ceg_math.h functions
    int32_t nrules = 0;
    DevRule* d_rules = nullptr;
    int32_t* d_offset = nullptr;
    std::vector<DevRule> h_rules;       // host copies: the per-molecule tables of k_pairs_frac are cut from them
    std::vector<int32_t> h_offset;
    struct Compact { PairFast* d_fast = nullptr; DevRule* d_rules = nullptr; int32_t* d_off = nullptr; int32_t nrules = 0, nentries = 0; };
    std::map<std::vector<int32_t>, Compact> compact;      // by the kinds of the molecule on trial
    ceg_consumers::HostIo io;           // ceg_pairs_energy
    double* d_etab = nullptr;           // erfc(alpha r)/r records of ceg_pairfrac.h (when the CoulombEwaldDirect rules share one alpha)
    int32_t ebase = 0, eni = 0;
    double4* d_atoms = nullptr;
    double4* d_frac = nullptr;          // invmat * position of the same atoms in the same order (k_pairs_frac)
    int64_t natoms = 0, cap = 0;
    ceg_consumers::CellBins bins{};     // neighbour cells: the atoms are uploaded sorted by cell when bins.on
    int32_t* d_cell_start = nullptr;
    double* d_geom = nullptr;           // mat[9], invmat[9]
};

extern "C" int ceg_pairs_create(ceg_pairs_t** handle, int32_t device, const double mat[9], const double invmat[9], double cutoff2,
                                const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds, double coulombic)
{
    if (!handle || !mat || !invmat || !rule_offset || nkinds < 1 || nkinds > 4096 || !(cutoff2 > 0.0))
        return perr(CEG_ERR_INVALID, "bad argument");
    *handle = nullptr;
    const int64_t nt = (int64_t)nkinds * nkinds;
    if (rule_offset[0] != 0) return perr(CEG_ERR_INVALID, "rule_offset[0] must be 0");
    for (int64_t t = 0; t < nt; ++t)
        if (rule_offset[t + 1] < rule_offset[t]) return perr(CEG_ERR_INVALID, "rule_offset must be non-decreasing");
    const int32_t nr = rule_offset[nt];
    if (nr > 0 && !rules) return perr(CEG_ERR_INVALID, "rules missing");
    std::vector<DevRule> dr((size_t)(nr > 0 ? nr : 1));
    bool fast = true;
    const double cutoff = std::sqrt(cutoff2);
    for (int32_t q = 0; q < nr; ++q) {
        const ceg_rule_t& r = rules[q];
        if (r.kind < CEG_HARDSPHERE || r.kind > CEG_NOINTERACTION) return perr(CEG_ERR_INVALID, "unknown rule kind");
        if (r.kind == CEG_UNDEFINED_INTERACTION) return perr(CEG_ERR_RULE, "Undefined interaction");     // interactions.jl:386-387
        dr[q].kind = r.kind; dr[q]._pad = 0;
        dr[q].p0 = r.p[0]; dr[q].p1 = r.p[1]; dr[q].p2 = r.p[2]; dr[q].shift = r.shift;
        if (r.kind == CEG_COULOMB_EWALD_DIRECT && !(r.p[0] >= 0.0 && r.p[0] * cutoff <= 5.0 * (1.0 - 1e-9))) fast = false;
        if ((r.kind == CEG_BUCKINGHAM || r.kind == CEG_EXPONENTIAL) && !(r.p[1] >= 0.0 && r.p[1] * cutoff <= 700.0)) fast = false;
    }
    if (ceg_device_count() <= 0) return perr(CEG_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ceg_device_count()) return perr(CEG_ERR_NO_DEVICE, "device not present");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return perr(CEG_ERR_HIP, "hipSetDevice failed");
    ceg_pairs* h = new ceg_pairs();
    h->device = device;
    for (int a = 0; a < 9; ++a) { h->mat[a] = mat[a]; h->invmat[a] = invmat[a]; }
    h->cutoff2 = cutoff2; h->coulombic = coulombic; h->nkinds = nkinds; h->fast = fast; h->nrules = nr;
    h->bins = ceg_consumers::choose_cell_bins(invmat, cutoff);
    double geom[18];
    for (int a = 0; a < 9; ++a) { geom[a] = mat[a]; geom[9 + a] = invmat[a]; }
    h->h_rules = dr;
    h->h_offset.assign(rule_offset, rule_offset + nt + 1);
    {   // the r^2-indexed erfc(alpha r)/r records of the fractional-coordinate kernel
        double alpha = 0.0;
        bool shared = true;
        for (int32_t q = 0; q < nr; ++q)
            if (dr[q].kind == CEG_COULOMB_EWALD_DIRECT) {
                if (alpha == 0.0) alpha = dr[q].p0;
                else if (alpha != dr[q].p0) shared = false;
            }
        ceg_pairfrac::ErfcTable et;
        if (fast && shared && alpha > 0.0 && cutoff2 > 1.0 && ceg_pairfrac::build_erfc_table(alpha, 1.0, cutoff2, et)) {
            if (hipMalloc((void**)&h->d_etab, et.rec.size() * sizeof(double)) == hipSuccess &&
                hipMemcpy(h->d_etab, et.rec.data(), et.rec.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess) {
                h->ebase = et.base; h->eni = et.ni;
            } else {
                (void)hipFree(h->d_etab);
                h->d_etab = nullptr;
            }
        }
    }
    bool ok = hipMalloc((void**)&h->d_rules, dr.size() * sizeof(DevRule)) == hipSuccess &&
              hipMalloc((void**)&h->d_offset, (size_t)(nt + 1) * sizeof(int32_t)) == hipSuccess &&
              hipMalloc((void**)&h->d_geom, sizeof geom) == hipSuccess &&
              hipMemcpy(h->d_geom, geom, sizeof geom, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(h->d_rules, dr.data(), dr.size() * sizeof(DevRule), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(h->d_offset, rule_offset, (size_t)(nt + 1) * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
    if (prev >= 0) (void)hipSetDevice(prev);
    if (!ok) {
        ceg_pairs_destroy(h);
        return perr(CEG_ERR_HIP, "could not upload the pair table");
    }
    *handle = h;
    return CEG_OK;
}

extern "C" int ceg_pairs_destroy(ceg_pairs_t* h)
{
    if (!h) return CEG_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(h->device) == hipSuccess) {
        (void)hipFree(h->d_rules);
        (void)hipFree(h->d_offset);
        for (auto& kv : h->compact) { (void)hipFree(kv.second.d_fast); (void)hipFree(kv.second.d_rules); (void)hipFree(kv.second.d_off); }
        (void)hipFree(h->d_atoms);
        (void)hipFree(h->d_frac);
        (void)hipFree(h->d_cell_start);
        (void)hipFree(h->d_geom);
        (void)hipFree(h->d_etab);
        h->io.release();
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete h;
    return CEG_OK;
}

extern "C" int ceg_pairs_set_atoms(ceg_pairs_t* h, const double* positions, const int32_t* kinds, const int32_t* molecule,
                                   int64_t natoms)
{
    if (!h || natoms < 0 || (natoms > 0 && (!positions || !kinds || !molecule))) return perr(CEG_ERR_INVALID, "bad argument");
    std::vector<double4> host((size_t)(natoms > 0 ? natoms : 1));
    for (int64_t l = 0; l < natoms; ++l) {
        if (kinds[l] < 0 || kinds[l] >= h->nkinds) return perr(CEG_ERR_INVALID, "atom kind outside the pair table");
        if (molecule[l] < 0) return perr(CEG_ERR_INVALID, "molecule ids must be non-negative");
        const long long bits = ((long long)molecule[l] << 32) | (long long)(uint32_t)kinds[l];
        double w;
        memcpy(&w, &bits, sizeof(w));
        host[l] = make_double4(positions[3 * l], positions[3 * l + 1], positions[3 * l + 2], w);
    }
    std::vector<int32_t> cell_start;
    if (h->bins.on) {                    // counting sort by cell (z fastest): the kernel walks contiguous runs of cells
        const int ncells = h->bins.nb[0] * h->bins.nb[1] * h->bins.nb[2];
        cell_start.assign((size_t)ncells + 1, 0);
        std::vector<int32_t> cell((size_t)(natoms > 0 ? natoms : 1));
        for (int64_t l = 0; l < natoms; ++l) {
            cell[l] = ceg_consumers::cell_of_position(h->bins, h->invmat, positions + 3 * l);
            ++cell_start[(size_t)cell[l] + 1];
        }
        for (int c = 0; c < ncells; ++c) cell_start[(size_t)c + 1] += cell_start[c];
        std::vector<int32_t> next(cell_start.begin(), cell_start.end() - 1);
        std::vector<double4> sorted(host.size());
        for (int64_t l = 0; l < natoms; ++l) sorted[(size_t)next[cell[l]]++] = host[l];
        host.swap(sorted);
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(h->device) != hipSuccess) return perr(CEG_ERR_HIP, "hipSetDevice failed");
    bool ok = true;
    if (natoms > h->cap) {
        (void)hipFree(h->d_atoms);
        (void)hipFree(h->d_frac);
        h->d_atoms = h->d_frac = nullptr;
        h->cap = natoms + natoms / 2 + 64;
        ok = hipMalloc((void**)&h->d_atoms, (size_t)h->cap * sizeof(double4)) == hipSuccess &&
             hipMalloc((void**)&h->d_frac, (size_t)h->cap * sizeof(double4)) == hipSuccess;
        if (!ok) h->cap = 0;
    }
    if (ok && natoms > 0) ok = hipMemcpy(h->d_atoms, host.data(), (size_t)natoms * sizeof(double4), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && natoms > 0) {              // the fractional coordinates of k_pairs_frac (approximate path only: any rounding will do)
        const double* I = h->invmat;
        for (int64_t l = 0; l < natoms; ++l) {
            const double4 A = host[l];
            host[l] = make_double4(I[0] * A.x + I[3] * A.y + I[6] * A.z, I[1] * A.x + I[4] * A.y + I[7] * A.z, I[2] * A.x + I[5] * A.y + I[8] * A.z, A.w);
        }
        ok = hipMemcpy(h->d_frac, host.data(), (size_t)natoms * sizeof(double4), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok && h->bins.on) {
        if (!h->d_cell_start) ok = hipMalloc((void**)&h->d_cell_start, cell_start.size() * sizeof(int32_t)) == hipSuccess;
        ok = ok && hipMemcpy(h->d_cell_start, cell_start.data(), cell_start.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (!ok) return perr(CEG_ERR_HIP, "could not upload the guest atoms");
    h->natoms = natoms;
    return CEG_OK;
}

namespace {
// the pair-table entries a molecule with these kinds can meet, entry kind1 * m + a (built once per distinct molecule, kept with the handle)
const ceg_pairs::Compact* pairs_compact_table(ceg_pairs* h, const int32_t* kinds, int m)
{
    std::vector<int32_t> key(kinds, kinds + m);
    auto it = h->compact.find(key);
    if (it != h->compact.end()) return &it->second;
    const int nkinds = h->nkinds;
    std::vector<int32_t> off((size_t)nkinds * m + 1, 0);
    std::vector<DevRule> rules;
    std::vector<PairFast> fast((size_t)nkinds * m);
    for (int k1 = 0; k1 < nkinds; ++k1)
        for (int a = 0; a < m; ++a) {
            const size_t t = (size_t)k1 * nkinds + kinds[a];
            for (int32_t q = h->h_offset[t]; q < h->h_offset[t + 1]; ++q) rules.push_back(h->h_rules[(size_t)q]);
            const PairFast P = ceg_pairfrac::make_pair_fast(h->h_rules.data(), h->h_offset[t], h->h_offset[t + 1], h->coulombic);
            fast[(size_t)k1 * m + a] = P;
            off[(size_t)k1 * m + a + 1] = (int32_t)rules.size();
        }
    ceg_pairs::Compact c;
    c.nrules = (int32_t)rules.size();
    c.nentries = nkinds * m;
    if (rules.empty()) rules.resize(1);
    const bool ok = hipMalloc((void**)&c.d_fast, fast.size() * sizeof(PairFast)) == hipSuccess &&
                    hipMalloc((void**)&c.d_rules, rules.size() * sizeof(DevRule)) == hipSuccess &&
                    hipMalloc((void**)&c.d_off, off.size() * sizeof(int32_t)) == hipSuccess &&
                    hipMemcpy(c.d_fast, fast.data(), fast.size() * sizeof(PairFast), hipMemcpyHostToDevice) == hipSuccess &&
                    hipMemcpy(c.d_rules, rules.data(), rules.size() * sizeof(DevRule), hipMemcpyHostToDevice) == hipSuccess &&
                    hipMemcpy(c.d_off, off.data(), off.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        (void)hipFree(c.d_fast); (void)hipFree(c.d_rules); (void)hipFree(c.d_off);
        return nullptr;
    }
    return &h->compact.emplace(std::move(key), c).first->second;
}
}  // namespace

extern "C" int ceg_pairs_energy_device(ceg_pairs_t* h, const double* d_trial, const int32_t* trial_kinds, int32_t m, int64_t n,
                                       int32_t exclude_molecule, double* d_out, void* stream)
{
    if (!h || n < 0 || m < 1 || !trial_kinds || (n > 0 && (!d_trial || !d_out))) return perr(CEG_ERR_INVALID, "bad argument");
    if (m > PAIRS_MAX_ATOMS) return perr(CEG_ERR_UNSUPPORTED, "molecule has more atoms than the kernel holds in registers (16)");
    if (n == 0) return CEG_OK;
    PairsGeom g{};
    for (int a = 0; a < 9; ++a) { g.mat[a] = h->mat[a]; g.invmat[a] = h->invmat[a]; }
    g.cutoff2 = h->cutoff2; g.coulombic = h->coulombic; g.nkinds = h->nkinds; g.m = m; g.exclude = exclude_molecule;
    for (int a = 0; a < m; ++a) {
        if (trial_kinds[a] < 0 || trial_kinds[a] >= h->nkinds) return perr(CEG_ERR_INVALID, "trial atom kind outside the pair table");
        g.kinds[a] = trial_kinds[a];
    }
    for (int i = 0; i < 3; ++i) { g.nb[i] = h->bins.nb[i]; g.hfrac[i] = h->bins.hfrac[i]; }
    g.geom = h->d_geom;
    const bool cells = h->bins.on && h->d_cell_start && h->natoms > 0;
    const int64_t nblocks = (n + PAIRS_WAVES - 1) / PAIRS_WAVES;
    if (nblocks > 0x7fffffffLL) return perr(CEG_ERR_INVALID, "too many placements");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(h->device) != hipSuccess) return perr(CEG_ERR_HIP, "hipSetDevice failed");
    const size_t table_bytes = sizeof(DevRule) * (size_t)(h->nrules > 0 ? h->nrules : 1) + sizeof(int32_t) * ((size_t)h->nkinds * h->nkinds + 1);
    const bool in_lds = table_bytes <= 48 * 1024;
    const size_t lds = in_lds ? table_bytes : 0;
    const dim3 grid((unsigned)nblocks), block(64 * PAIRS_WAVES);
    hipStream_t st = (hipStream_t)stream;
    int wrap = ceg_consumers::wrap_mode(h->mat, h->invmat, h->bins.hfrac);
    if (const char* env = getenv("CEG_HIP_PAIRS_WRAP")) wrap = std::min(wrap, std::max(0, atoi(env)));      // measurement aid: 0 forces the literal form
    bool use_frac = h->fast && wrap >= 1 && h->natoms > 0 && h->natoms < (1 << 27);       // (a queued candidate carries atom index << 4 | trial atom in 32 bits)
    if (const char* env = getenv("CEG_HIP_PAIRS_FRAC")) use_frac = use_frac && atoi(env) != 0;               // measurement aid: 0 = the Cartesian kernel
    const ceg_pairs::Compact* ctab = use_frac ? pairs_compact_table(h, g.kinds, m) : nullptr;
    if (use_frac && !ctab) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return perr(CEG_ERR_HIP, "could not upload the pair-table rows of the molecule");
    }
    const size_t ftab_bytes = ctab ? frac_table_bytes(ctab->nentries, ctab->nrules, h->eni) : 0;
    use_frac = use_frac && ftab_bytes + sizeof(FracHit) * FQCAP * PAIRS_WAVES + 4096 <= 60 * 1024;
    if (use_frac) {
        const FracTable ftab{ctab->d_fast, ctab->d_rules, ctab->d_off, ctab->nrules, ctab->nentries, h->d_etab, h->ebase, h->eni};
        // placements per wave: as many as leave >= 16 workgroups per CU (4096 on the chip)
        int per_wave = 1;
        if (const char* env = getenv("CEG_HIP_PAIRS_PER_WAVE")) per_wave = std::max(1, std::min(64, atoi(env)));
        else while (per_wave < 8 && n / ((int64_t)PAIRS_WAVES * per_wave * 2) >= 4096) per_wave *= 2;
        const dim3 fgrid((unsigned)((n + (int64_t)PAIRS_WAVES * per_wave - 1) / ((int64_t)PAIRS_WAVES * per_wave)));
#define CEG_PAIRS_F(MMv, CL, TR) hipLaunchKernelGGL((k_pairs_frac<MMv, CL, TR>), fgrid, block, ftab_bytes, st, g, ftab, h->d_frac, h->d_atoms, h->d_cell_start, h->natoms, d_trial, n, d_out, per_wave)
#define CEG_PAIRS_FM(CL, TR)                  \
    do {                                      \
        switch (m) {                          \
            case 1: CEG_PAIRS_F(1, CL, TR); break; \
            case 2: CEG_PAIRS_F(2, CL, TR); break; \
            case 3: CEG_PAIRS_F(3, CL, TR); break; \
            case 4: CEG_PAIRS_F(4, CL, TR); break; \
            default: CEG_PAIRS_F(0, CL, TR); break; \
        }                                     \
    } while (0)
        if (cells) { if (wrap == 2) CEG_PAIRS_FM(true, true); else CEG_PAIRS_FM(true, false); }
        else { if (wrap == 2) CEG_PAIRS_FM(false, true); else CEG_PAIRS_FM(false, false); }
#undef CEG_PAIRS_FM
#undef CEG_PAIRS_F
        const hipError_t ef = hipGetLastError();
        if (prev >= 0) (void)hipSetDevice(prev);
        if (ef != hipSuccess) return perr(CEG_ERR_HIP, hipGetErrorString(ef));
        return CEG_OK;
    }
#define CEG_PAIRS_LAUNCH(F, L, CL, WR) hipLaunchKernelGGL((k_pairs<F, L, CL, WR>), grid, block, lds, st, g, h->d_rules, h->d_offset, h->nrules, h->d_atoms, h->d_cell_start, h->natoms, d_trial, n, d_out)
#define CEG_PAIRS_WRAP(F, L, CL)                              \
    do {                                                      \
        if (wrap == 2) CEG_PAIRS_LAUNCH(F, L, CL, 2);         \
        else if (wrap == 1) CEG_PAIRS_LAUNCH(F, L, CL, 1);    \
        else CEG_PAIRS_LAUNCH(F, L, CL, 0);                   \
    } while (0)
#define CEG_PAIRS_PICK(CL)                                    \
    do {                                                      \
        if (h->fast && in_lds) CEG_PAIRS_WRAP(true, true, CL);   \
        else if (h->fast) CEG_PAIRS_WRAP(true, false, CL);       \
        else if (in_lds) CEG_PAIRS_WRAP(false, true, CL);        \
        else CEG_PAIRS_WRAP(false, false, CL);                   \
    } while (0)
    if (cells) CEG_PAIRS_PICK(true);
    else CEG_PAIRS_PICK(false);
#undef CEG_PAIRS_PICK
#undef CEG_PAIRS_WRAP
#undef CEG_PAIRS_LAUNCH
    const hipError_t e = hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return perr(CEG_ERR_HIP, hipGetErrorString(e));
    return CEG_OK;
}

extern "C" int ceg_pairs_neighbour_cells(ceg_pairs_t* h, int32_t nb[3])
{
    if (!h) return perr(CEG_ERR_INVALID, "bad argument");
    for (int i = 0; i < 3; ++i)
        if (nb) nb[i] = h->bins.on ? h->bins.nb[i] : 0;
    return h->bins.on ? 1 : 0;
}

extern "C" int ceg_pairs_energy(ceg_pairs_t* h, const double* trial, const int32_t* trial_kinds, int32_t m, int64_t n,
                                int32_t exclude_molecule, double* out)
{
    if (!h || n < 0 || m < 1 || m > PAIRS_MAX_ATOMS || !trial_kinds || (n > 0 && (!trial || !out)))
        return perr(m > PAIRS_MAX_ATOMS ? CEG_ERR_UNSUPPORTED : CEG_ERR_INVALID, "bad argument");
    if (n == 0) return CEG_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(h->device) != hipSuccess) return perr(CEG_ERR_HIP, "hipSetDevice failed");
    const size_t np = (size_t)n * m * 3;
    int rc = CEG_OK;
    if (!h->io.ensure(sizeof(double) * np, sizeof(double) * (size_t)n)) rc = perr(CEG_ERR_HIP, "hipMalloc failed");
    double *d_p = h->io.d_in, *d_o = h->io.d_out;
    if (!rc && hipMemcpy(d_p, trial, sizeof(double) * np, hipMemcpyHostToDevice) != hipSuccess) rc = perr(CEG_ERR_HIP, "H2D failed");
    if (!rc) rc = ceg_pairs_energy_device(h, d_p, trial_kinds, m, n, exclude_molecule, d_o, nullptr);
    // (the copy back runs on the null stream behind the kernel and reports its failure)
    if (!rc && hipMemcpy(out, d_o, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = perr(CEG_ERR_HIP, "kernel execution or D2H failed");
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}
