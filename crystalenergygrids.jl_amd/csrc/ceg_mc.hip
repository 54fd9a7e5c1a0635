// ceg_mc.hip -- device-resident energy state of a Monte-Carlo run (BASELINE config 5; SURVEY 8f rows f1 + f2 + f3 fused):
//
//   movement_energy            src/montecarlo.jl:563-579   one launch per batch of trial placements of one molecule:
//     framework_interactions     :490-504                     tricubic interpolation of the VdW grid of every atom + charge x Coulomb grid
//     single_contribution_vdw    src/energy.jl:397-427        guest-guest pair rules against all other molecules
//     single_contribution_ewald  src/ewald.jl:704-738         2 sum kf Re(conj(rest) S) + sum kf |S|^2, rest = framework + all guests - this one
//   update_mc! / update_ewald_context!   src/montecarlo.jl:615-628, src/ewald.jl:757-773
//                                                            on acceptance: positions, per-molecule and total structure factors
//                                                            updated ON THE DEVICE (no host-built `rest`, no upload between moves)
//   compute_ewald(::IncrementalEwaldContext) structure factors   src/ewald.jl:630-652 (sums[:,1], sums[:,ij+1])
//
// The guest atoms, the pair table, the k-space tables, the framework structure factor, the per-molecule structure
// factors sums[:, ij+1] and their total sums[:, 1] live in device memory.  One workgroup (4 waves) evaluates one
// placement: wave lanes 0..2m-1 interpolate, all threads fill the e^{2 pi i m f} tables and stride over the k-vectors,
// all threads stride over the guest atoms for the pair sum; block 0 evaluates the molecule where it currently is
// (with its stored structure factor, like the reference's `positions === nothing` branch).  Small batches travel
// through a pinned, device-mapped host buffer: a batch-1 trial is ONE kernel launch and one stream synchronisation.
// The MC driver (move proposal, acceptance rule, GCMC swaps) stays on the host, as in SURVEY 8f.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#include "../../include/ceg_hip.h"
#include "ceg_consumers.h"
#include "ceg_rows.h"
#include "ceg_pairfrac.h"

using ceg::DevRule;
using ceg_consumers::InterpGeom;
using ceg_consumers::interp_point;
using ceg_consumers::rule_energy;
using ceg_consumers::rule_energy_fast;

extern "C" void ceg_set_last_error_(const char* msg);

namespace {

constexpr int MC_MAX_ATOMS = 16;
constexpr int MC_THREADS = 256;
#ifndef MC_TRIAL_WAVES
#define MC_TRIAL_WAVES 3          // waves per SIMD asked of the trial kernel (the neighbour-cell code had pushed it to 169 VGPRs = 2 waves)
#endif
constexpr int MC_MAX_TAB = 400;                // (kx+1) + (2ky+1) + (2kz+1)
constexpr size_t MC_MAPPED_BYTES = 1 << 20;    // batches up to this size go through the pinned, device-mapped buffers

struct McGrid {
    InterpGeom g;
    const float* grid;        // node-major [x][y][z][8] in K; nullptr: zero grid (interpolate_grid returns 0, grids.jl:213)
};

// everything a kernel needs, by value (kernarg)
struct McView {
    double mat[9], invmat[9];                  // MC cell (pair distances, src/utils.jl:294-302)
    double ew_invmat[9];                       // inverse of the Ewald supercell matrix
    double cutoff2, coulombic;
    int32_t nkinds, nrules, fast, table_in_lds;
    int32_t ks[3], nk;
    int32_t natoms, nmol;                      // natoms: high-water mark of the atom slots
    const McGrid* vdw;                         // [nkinds]
    McGrid coulomb;
    const double* kind_charge;                 // [nkinds]
    const DevRule* rules;
    const int32_t* rule_offset;                // [nkinds*nkinds + 1]
    const int32_t* ijk;                        // [3 nk]
    const double* kf;                          // [nk]
    const double2* sf_fw;                      // [nk] StoreRigidChargeFramework
    double2* sf_tot;                           // [nk] sums[:, 1]
    double2* sf_mol;                           // [nmol][nk] sums[:, ij+1]
    double4* atoms;                            // x, y, z, (molecule << 32 | kind); molecule < 0: free slot
    double4* fatoms;                           // the same slots with invmat * position (the pair tests of ceg_pairfrac.h)
    int2* mol;                                 // [nmol] atoms of molecule j: slots [mol[j].x, mol[j].x + mol[j].y)
    // neighbour cells of the guest atoms (what the reference gets from CellListMap, src/energy.jl:341-349,399-404):
    // fractional bins of the MC cell, fixed capacity, each holding COPIES of its atoms' records
    int32_t use_cells, cell_cap;
    int32_t nb[3];
    double hfrac[3];                           // cutoff / perpendicular width: fractional half-extent of the cutoff sphere
    double4* cells;                            // [nb0*nb1*nb2][cell_cap]
    double4* fcells;                           // the same entries, fractional
    int32_t* cell_count;                       // [nb0*nb1*nb2]
    // the k-vectors as rows cut into segments and dealt to 64 lanes in rounds (ceg_rows.h)
    int32_t nrounds, ns;
    int32_t fastwrap, _pad1;                   // ceg_consumers::wrap_mode: 0 literal pair distances, 1 fast wrap, 2 fast wrap in an upper-triangular cell
    const double* geom;                        // mat[9], invmat[9] in device memory (literal fall-back of the fast pair distance)
    const int32_t* desc;                       // [nrounds * 64]
    const int32_t* qof;                        // [ns * 64] k-vector of (slot, lane), -1 in the padding slots
};

// what an update does to the cells, worked out on the host mirror of the cell lists: cells[dst[i]] = atoms[src[i]] once the
// atom records are current, then cell_count[cell[i]] = count[i]
constexpr int MC_MAX_CELL_OPS = 2 * 16;
struct McCellOps {
    int32_t nops, ncnt;
    int32_t dst[MC_MAX_CELL_OPS], src[MC_MAX_CELL_OPS], cell[MC_MAX_CELL_OPS], count[MC_MAX_CELL_OPS];
};

// an atom record into its slot, Cartesian and fractional
__device__ __forceinline__ void put_atom(const McView& v, int slot, const double4 A)
{
    const double* I = v.invmat;
    v.atoms[slot] = A;
    v.fatoms[slot] = make_double4(__builtin_fma(I[6], A.z, __builtin_fma(I[3], A.y, I[0] * A.x)), __builtin_fma(I[7], A.z, __builtin_fma(I[4], A.y, I[1] * A.x)),
                                  __builtin_fma(I[8], A.z, __builtin_fma(I[5], A.y, I[2] * A.x)), A.w);
}

__device__ __forceinline__ void apply_cell_ops(const McView& v, const McCellOps& ops, int tid)
{
    if (tid < ops.nops) {
        v.cells[ops.dst[tid]] = v.atoms[ops.src[tid]];
        v.fcells[ops.dst[tid]] = v.fatoms[ops.src[tid]];
    }
    if (tid < ops.ncnt) v.cell_count[ops.cell[tid]] = ops.count[tid];
}

// a molecule that is not (yet) in the system: kinds of its atoms (single_contribution_ewald with ij < 0, ewald.jl:704-728)
struct McMolecule { int32_t m; int32_t kinds[MC_MAX_ATOMS]; };

struct McPositions { double xyz[MC_MAX_ATOMS * 3]; };

// the molecule on trial as the host knows it (k_mc_trial: no dependent loads of slot, kinds and charges in front of a batch-1 call)
struct McLocal { int32_t first, m; int32_t kinds[MC_MAX_ATOMS]; double q[MC_MAX_ATOMS]; };

__device__ __forceinline__ void unpack(double w, int& kind, int& mol)
{
    const long long bits = __double_as_longlong(w);
    kind = (int)(bits & 0xffffffffll);
    mol = (int)(bits >> 32);
}

// e^{2 pi i m f} tables of `m_atoms` atoms at s_pos (setup_Eik / move_one_system!, src/ewald.jl:109-146,352-366),
// by sine / cosine of the exact angle (ceg_math.h sincos_2pi); entry t of atom a at tab[a * stride + t]: t in [0, kx] -> x, then y (m = -ky..ky), then z
// s_q != nullptr: the z entries carry the atom's charge as a factor (what the row-wise walk of ceg_rows.h expects)
__device__ __forceinline__ void fill_tables(const McView& v, const double* s_pos, int m_atoms, double2* tab, int stride, int tid, int nthreads,
                                            const double* s_q = nullptr)
{
    const int kx = v.ks[0], ky = v.ks[1], kz = v.ks[2];
    const int nxp = kx + 1, nyp = 2 * ky + 1;
    const double* I = v.ew_invmat;
    for (int e = tid; e < m_atoms * stride; e += nthreads) {
        const int a = e / stride, t = e - a * stride;
        const double x = s_pos[3 * a], y = s_pos[3 * a + 1], z = s_pos[3 * a + 2];
        double f;
        int mm;
        if (t < nxp) { f = I[0] * x + I[3] * y + I[6] * z; mm = t; }
        else if (t < nxp + nyp) { f = I[1] * x + I[4] * y + I[7] * z; mm = t - nxp - ky; }
        else { f = I[2] * x + I[5] * y + I[8] * z; mm = t - nxp - nyp - kz; }
        const double ff = f - rint(f);
        double s, c;
        ceg::sincos_2pi((double)mm * ff, s, c);
        const double w = (s_q && t >= nxp + nyp) ? s_q[a] : 1.0;
        tab[e] = make_double2(w * c, w * s);
    }
}

// the same tables filled by ONE wave (k_mcw_ewald): entry t per lane, the atoms in an inner loop -- no division by the stride, the axis
// decoded once per entry
__device__ __forceinline__ void fill_tables_wave(const McView& v, const double* s_pos, int m_atoms, double2* tab, int stride, int lane, const double* s_q)
{
    const int kx = v.ks[0], ky = v.ks[1], kz = v.ks[2];
    const int nxp = kx + 1, nyp = 2 * ky + 1;
    const double* I = v.ew_invmat;
    for (int t = lane; t < stride; t += 64) {
        const int ax = t < nxp ? 0 : (t < nxp + nyp ? 1 : 2);
        const int mm = ax == 0 ? t : (ax == 1 ? t - nxp - ky : t - nxp - nyp - kz);
        const double i0 = I[ax], i1 = I[ax + 3], i2 = I[ax + 6];
        for (int a = 0; a < m_atoms; ++a) {
            const double f = i0 * s_pos[3 * a] + i1 * s_pos[3 * a + 1] + i2 * s_pos[3 * a + 2];
            const double ff = f - rint(f);
            double s, c;
            ceg::sincos_2pi((double)mm * ff, s, c);
            const double w = ax == 2 ? s_q[a] : 1.0;
            tab[a * stride + t] = make_double2(w * c, w * s);
        }
    }
}

// The structure factor of the molecule whose tables (charge on z) are `tab`, k-vector by k-vector in the row-wise order of ceg_rows.h:
// wave `wave` of `nwaves` takes the rounds wave, wave + nwaves, ...; sink(q, re, im) for every real k-vector.
template <class Sink>
__device__ __forceinline__ void rows_structure_factor(const McView& v, const double2* tab, int stride, int m_atoms, int wave, int nwaves, int lane, Sink&& sink)
{
    const int nxp = v.ks[0] + 1, nyp = 2 * v.ks[1] + 1;
    int slot = 0;
    for (int r = 0; r < v.nrounds; ++r) {
        const int d = v.desc[r * 64 + lane];
        const int L = __builtin_amdgcn_readfirstlane(d >> 27);
        if (r % nwaves == wave) {
            const int at = slot * 64 + lane;
            ceg_rows::round_dispatch(L, m_atoms, tab, stride, nxp, nyp, d & 0x1ff, (d >> 9) & 0x1ff, (d >> 18) & 0x1ff, [&](int sidx, double sr, double si) {
                const int q = v.qof[at + sidx * 64];
                if (q >= 0) sink(q, sr, si);
            });
        }
        slot += L;
    }
}

// structure factor of the molecule at k-vector q from the tables: sum_a q_a Ex[i] Ey[j] Ez[k]   (src/ewald.jl:148-185)
__device__ __forceinline__ double2 molecule_sf(const McView& v, const double2* tab, int stride, const double* s_q, int m_atoms, int64_t q)
{
    const int ky = v.ks[1], kz = v.ks[2];
    const int nxp = v.ks[0] + 1, nyp = 2 * ky + 1;
    const int i = v.ijk[3 * q], j = v.ijk[3 * q + 1], k = v.ijk[3 * q + 2];
    double sr = 0.0, si = 0.0;
    for (int a = 0; a < m_atoms; ++a) {
        const double2 ex = tab[a * stride + i];
        const double2 ey = tab[a * stride + nxp + ky + j];
        const double2 ez = tab[a * stride + nxp + nyp + kz + k];
        const double yr = ey.x * ez.x - ey.y * ez.y, yi = ey.x * ez.y + ey.y * ez.x;
        const double cr = s_q[a] * yr, ci = s_q[a] * yi;
        sr += ex.x * cr - ex.y * ci;
        si += ex.x * ci + ex.y * cr;
    }
    return make_double2(sr, si);
}

// INSERT: the molecule is described by `nm` and is not in the system -- no current-position row, nothing excluded from the
// pair sum, rest = framework + sums[:, 1]
template <bool FAST, bool INSERT, bool CELLS>
__global__ __launch_bounds__(MC_THREADS, MC_TRIAL_WAVES) void k_mc_trial(McView v, int32_t molecule, McLocal L, const double* __restrict__ trial, int64_t n,
                                                          double* __restrict__ out, int stride, unsigned* done, unsigned long long* flag,
                                                          unsigned long long seq)
{
    // dynamic LDS: [m][stride] double2 tables, then (table_in_lds) the pair table
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ double s_pos[MC_MAX_ATOMS * 3];
    __shared__ double s_q[MC_MAX_ATOMS];
    __shared__ int32_t s_kind[MC_MAX_ATOMS];
    __shared__ double s_red[MC_THREADS / 64][5];
    __shared__ int s_bin0[3], s_nbin[3], s_wtot[MC_THREADS / 64], s_first[MC_THREADS], s_cell[MC_THREADS];
    const int tid = threadIdx.x;
    // gridDim.y == 3: the three terms of a row on three workgroups (blockIdx.y = 0 framework grids, 1 reciprocal sum, 2 guest-guest pairs) --
    // the latency of a small batch is the longest term, not their sum; gridDim.y == 1: one workgroup does all three
    const int term = gridDim.y == 1 ? -1 : (int)blockIdx.y;
    const bool do_frame = term < 0 || term == 0, do_ewald = term < 0 || term == 1, do_pairs = term < 0 || term == 2;
    const int64_t b = INSERT ? (int64_t)blockIdx.x + 1 : (int64_t)blockIdx.x;     // 0: where the molecule is now; b >= 1: trial b - 1
    const int first = L.first, m = L.m;
    double2* tab = reinterpret_cast<double2*>(s_raw);
    const DevRule* rules = v.rules;
    const int32_t* offset = v.rule_offset;
    if (v.table_in_lds && do_pairs) {
        DevRule* lr = reinterpret_cast<DevRule*>(s_raw + sizeof(double2) * (size_t)m * stride);
        int32_t* lo = reinterpret_cast<int32_t*>(lr + (v.nrules > 0 ? v.nrules : 1));
        for (int t = tid; t < v.nrules; t += MC_THREADS) lr[t] = v.rules[t];
        for (int t = tid; t < v.nkinds * v.nkinds + 1; t += MC_THREADS) lo[t] = v.rule_offset[t];
        rules = lr;
        offset = lo;
    }
    if (tid < 3 * m) {
        if (b == 0) {
            const double4 A = v.atoms[first + tid / 3];
            s_pos[tid] = (tid % 3 == 0) ? A.x : ((tid % 3 == 1) ? A.y : A.z);
        } else {
            s_pos[tid] = trial[(size_t)(b - 1) * m * 3 + tid];
        }
    }
    if (tid < m) {
        s_kind[tid] = L.kinds[tid];
        s_q[tid] = L.q[tid];
    }
    // the k-space constants of the first k-vectors of this thread do not depend on the positions: fetched before anything else
    constexpr int R = 3;
    const double2* mine = v.sf_mol + (size_t)(INSERT ? 0 : molecule) * v.nk;
    struct KChunk { double2 old[R], f[R], t[R]; double kf[R]; };
    auto load_chunk = [&](const int64_t q0, KChunk& c) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t q = q0 + (int64_t)r * MC_THREADS;
            const bool in = q < v.nk;
            const int64_t qq = in ? q : 0;
            c.old[r] = (INSERT || !in) ? make_double2(0.0, 0.0) : mine[qq];
            c.f[r] = in ? v.sf_fw[qq] : make_double2(0.0, 0.0);
            c.t[r] = in ? v.sf_tot[qq] : make_double2(0.0, 0.0);
            c.kf[r] = in ? v.kf[qq] : 0.0;
        }
    };
    KChunk cur;
    if (v.nk > 0 && do_ewald) load_chunk(tid, cur);
    double4 A_first = make_double4(0.0, 0.0, 0.0, 0.0);             // likewise the first guest atom of this thread
    if (do_pairs && !CELLS) A_first = v.atoms[tid < v.natoms ? tid : 0];
    __syncthreads();

    double fv = 0.0, fd = 0.0, inter = 0.0, rs = 0.0, ss = 0.0;
    // ---- framework_interactions (montecarlo.jl:490-504): thread 16a + 8g + corner, g = 0 the VdW grid of atom a, g = 1 the Coulomb grid --
    // one grid CORNER per thread, three shuffles per sum (the 64-term polynomial on one thread was the longest chain of a batch-1 call)
    if (do_frame) {
        static_assert(16 * MC_MAX_ATOMS <= MC_THREADS, "one thread per corner");
        const int a = tid >> 4, gsel = (tid >> 3) & 1, corner = tid & 7;
        double part = 0.0;
        bool blocked = false, have = false, isvdw = false;
        if (tid < 16 * m) {
            const double px = s_pos[3 * a], py = s_pos[3 * a + 1], pz = s_pos[3 * a + 2];
            if (gsel == 0) {
                const McGrid* G = v.vdw + s_kind[a];
                if (G->grid) { part = ceg_consumers::interp_corner(G->g, G->grid, px, py, pz, corner, blocked); have = true; isvdw = G->g.is_vdw != 0; }
            } else if (v.coulomb.grid) {
                part = ceg_consumers::interp_corner(v.coulomb.g, v.coulomb.grid, px, py, pz, corner, blocked);
                have = true;
                isvdw = v.coulomb.g.is_vdw != 0;
            }
        }
        int blk = blocked ? 1 : 0;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            part += __shfl_xor(part, o);
            blk |= __shfl_xor(blk, o);
        }
        if (have && corner == 0) {
            const double val = (isvdw && blk) ? 1e100 : part;         // grids.jl:245-248
            if (gsel == 0) fv = val;
            else fd = (val == 1e100) ? val : s_q[a] * val;            // montecarlo.jl:500
        }
    }
    // ---- single_contribution_ewald (ewald.jl:704-738)
    if (v.nk > 0 && do_ewald) {
        if (b != 0) fill_tables(v, s_pos, m, tab, stride, tid, MC_THREADS);
        __syncthreads();
        // three k-vectors per thread at a time, the constants of the next three fetched while these are worked on: one L2 round trip for
        // the whole walk where a plain loop pays one per round (5-6 rounds of 256 k-vectors; the latency of a small batch was this loop)
        for (int64_t q0 = tid; q0 < v.nk; q0 += (int64_t)R * MC_THREADS) {
            KChunk nxt;
            load_chunk(q0 + (int64_t)R * MC_THREADS, nxt);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t q = q0 + (int64_t)r * MC_THREADS;
                const int64_t qq = q < v.nk ? q : q0;
                const double2 S = (b == 0) ? cur.old[r] : molecule_sf(v, tab, stride, s_q, m, qq);
                const double rr = cur.f[r].x + (cur.t[r].x - cur.old[r].x), ri = cur.f[r].y + (cur.t[r].y - cur.old[r].y);      // rest = framework + (sums[:,1] - sums[:,ij+1])
                rs += cur.kf[r] * (rr * S.x + ri * S.y);
                ss += cur.kf[r] * (S.x * S.x + S.y * S.y);
            }
            cur = nxt;
        }
    }
    // ---- single_contribution_vdw (energy.jl:407-427)
    if (v.table_in_lds) __syncthreads();
    if (do_pairs) {
        const double* M = v.mat;
        const double* I = v.invmat;
        auto pairs_with = [&](const double4 A) __attribute__((always_inline)) {
            int kind1, mol;
            unpack(A.w, kind1, mol);
            if (mol < 0 || (!INSERT && mol == molecule)) return;            // :419 (and free slots)
            for (int a = 0; a < m; ++a) {
                double r2;
                {
#pragma clang fp contract(off)
                    const double dx = s_pos[3 * a] - A.x, dy = s_pos[3 * a + 1] - A.y, dz = s_pos[3 * a + 2] - A.z;
                    double f0 = I[0] * dx + I[3] * dy + I[6] * dz;
                    double f1 = I[1] * dx + I[4] * dy + I[7] * dz;
                    double f2 = I[2] * dx + I[5] * dy + I[8] * dz;
                    f0 = ((f0 + 0.5) - floor(f0 + 0.5)) - 0.5;
                    f1 = ((f1 + 0.5) - floor(f1 + 0.5)) - 0.5;
                    f2 = ((f2 + 0.5) - floor(f2 + 0.5)) - 0.5;
                    const double vx = M[0] * f0 + M[3] * f1 + M[6] * f2;
                    const double vy = M[1] * f0 + M[4] * f1 + M[7] * f2;
                    const double vz = M[2] * f0 + M[5] * f1 + M[8] * f2;
                    r2 = vx * vx + vy * vy + vz * vz;
                }
                if (!(r2 < v.cutoff2)) continue;                            // :422
                const int t = kind1 * v.nkinds + s_kind[a];
                if (FAST && r2 >= 0.25) {
                    double r, rinv;
                    ceg::fast_sqrt_rsqrt(r2, r, rinv);
                    for (int q = offset[t]; q < offset[t + 1]; ++q) inter += rule_energy_fast(rules[q], r2, r, rinv, v.coulombic);
                } else {
                    for (int q = offset[t]; q < offset[t + 1]; ++q) inter += rule_energy(rules[q], r2, v.coulombic);
                }
            }
        };
        if (!CELLS) {
            for (int l = tid; l < v.natoms; l += MC_THREADS) pairs_with(l == tid ? A_first : v.atoms[l]);
        } else {
            // only the cells the cutoff spheres of the molecule's atoms can reach (ceg_consumers.h)
            if (tid < 3) ceg_consumers::cell_range(I, s_pos, m, tid, v.nb[tid], v.hfrac[tid], s_bin0[tid], s_nbin[tid]);
            __syncthreads();
            const int n0 = s_nbin[0], n1 = s_nbin[1], n2 = s_nbin[2];
            const int ncell = n0 * n1 * n2;
            const int wave = tid >> 6, lane = tid & 63;
            for (int base = 0; base < ncell; base += MC_THREADS) {
                const int e = base + tid;
                int cnt = 0, cell = 0;
                if (e < ncell) {
                    const int j2 = e % n2, j1 = (e / n2) % n1, j0 = e / (n2 * n1);
                    int c0 = s_bin0[0] + j0, c1 = s_bin0[1] + j1, c2 = s_bin0[2] + j2;
                    if (c0 >= v.nb[0]) c0 -= v.nb[0];
                    if (c1 >= v.nb[1]) c1 -= v.nb[1];
                    if (c2 >= v.nb[2]) c2 -= v.nb[2];
                    cell = (c0 * v.nb[1] + c1) * v.nb[2] + c2;
                    cnt = v.cell_count[cell];
                }
                int incl = cnt;                                            // inclusive scan over the workgroup
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(incl, o);
                    if (lane >= o) incl += up;
                }
                if (lane == 63) s_wtot[wave] = incl;
                __syncthreads();
                int before = 0, total = 0;
#pragma unroll
                for (int w = 0; w < MC_THREADS / 64; ++w) {
                    if (w < wave) before += s_wtot[w];
                    total += s_wtot[w];
                }
                s_first[tid] = before + incl - cnt;
                s_cell[tid] = cell;
                __syncthreads();
                for (int l = tid; l < total; l += MC_THREADS) {
                    int j = 0;                                             // last cell whose first entry is <= l
#pragma unroll
                    for (int step = MC_THREADS / 2; step > 0; step >>= 1)
                        if (s_first[j + step] <= l) j += step;
                    pairs_with(v.cells[(size_t)s_cell[j] * v.cell_cap + (l - s_first[j])]);
                }
                __syncthreads();
            }
        }
    }
    // ---- block reduction
    double vals[5] = {fv, fd, inter, rs, ss};
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vals[c] += __shfl_xor(vals[c], o);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0)
        for (int c = 0; c < 5; ++c) s_red[wave][c] = vals[c];
    __syncthreads();
    if (tid == 0) {
        double tot[5] = {0, 0, 0, 0, 0};
        for (int w = 0; w < MC_THREADS / 64; ++w)
            for (int c = 0; c < 5; ++c) tot[c] += s_red[w][c];
        double* o = out + 4 * (size_t)blockIdx.x;
        if (do_frame) { o[0] = tot[0]; o[1] = tot[1]; }
        if (do_pairs) o[2] = tot[2];
        if (do_ewald) o[3] = 2.0 * tot[3] + tot[4];
        // small batches: the rows sit in mapped host memory and the host polls `flag` instead of going through
        // hipStreamSynchronize (whose wake-up costs about as much as this kernel); the last workgroup to finish raises it
        if (flag) {
            __threadfence_system();
            if (atomicAdd(done, 1u) == gridDim.x * gridDim.y - 1) {
                *done = 0u;
                __threadfence_system();
                __atomic_store_n(flag, seq, __ATOMIC_RELEASE);
            }
        }
    }
}

// ---- the same rows for LARGE batches: one WAVE per placement, one kernel per term (round 4).
// k_mc_trial gives every placement a workgroup of four waves that meet at five barriers and read the k-space constants (56 B per
// k-vector) from L2 for every placement: right for the latency of a batch-1 call, 0.064 of the FP64 peak at batch 65 536.  From
// `wave_kernel_min_rows` rows on a batch goes through three launches instead, each with the registers its term needs and nothing
// else (fused into one kernel the three terms took 225 VGPRs -- 400 B of scratch at four waves per SIMD):
//   k_mcw_frame  framework_interactions: one grid CORNER per lane (16 lanes per atom: VdW grid + Coulomb grid), three shuffles per sum;
//   k_mcw_ewald  single_contribution_ewald: the workgroup stages kf Re(rest), kf Im(rest), kf once in the [slot][lane] planes of
//                ceg_rows.h (rest = framework + sums[:,1] - sums[:,ij+1], ewald.jl:722-728), every wave then walks its own
//                placements: tables by sincos_2pi with the charge on z, row-wise k-vector walk (8 FMAs per atom and k-vector, no LDS
//                access), one wave reduction -- no workgroup barrier after the staging;
//   k_mcw_pairs  single_contribution_vdw: the rows of the pair table that belong to the kinds of THIS molecule staged in LDS, lanes
//                stride over the guest atoms (or over the reachable neighbour cells).
// Row r of `out` (4 doubles) receives columns 0-1 from the first, 3 from the second, 2 from the third.
using McFastPair = ceg_pairfrac::PairFast;      // an entry whose rules are at most one Lennard-Jones and one CoulombEwaldDirect term (+ NoInteraction)
struct McCompact {                 // the pair-table rows of one molecule's kinds: entry (kind1, a) -> rules [off[kind1 * m + a], off[.. + 1])
    const DevRule* rules;
    const int32_t* off;
    const McFastPair* fast;        // [nkinds * m]
    int32_t nrules, in_lds;
};

// the libm-grade rule energies behind a call: inlined, their exp / erfc / pow temporaries would set the register count of the whole
// kernel (k_pairs: 168 VGPRs) while they serve the pairs closer than 0.5 A only
__device__ __attribute__((noinline)) double rule_energy_call(const DevRule* R, double r2, double coulombic) { return rule_energy(*R, r2, coulombic); }

// positions of row `row` into pos[3 m] (LDS of the wave): row 0 of a displacement batch is the molecule where it is now
template <bool INSERT>
__device__ __forceinline__ void mcw_load_row(const McView& v, int first, int m, const double* __restrict__ trial, int64_t row, int lane, double* pos)
{
    if (lane < 3 * m) {
        if (!INSERT && row == 0) {
            const double4 A = v.atoms[first + lane / 3];
            pos[lane] = (lane % 3 == 0) ? A.x : ((lane % 3 == 1) ? A.y : A.z);
        } else {
            pos[lane] = trial[(size_t)(INSERT ? row : row - 1) * m * 3 + lane];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

constexpr int MCW_WAVES = 4;       // waves per workgroup of the frame and pairs kernels (nothing large is staged there)
constexpr int MCW_FRAME_ROWS = 4;  // placements a wave of k_mcw_frame takes at most (their positions and per-atom values sit in LDS)

template <bool INSERT>
__global__ __launch_bounds__(64 * MCW_WAVES) void k_mcw_frame(McView v, int32_t molecule, McMolecule nm, const double* __restrict__ trial, int64_t nrows,
                                                             double* __restrict__ out, int per_wave)
{
    __shared__ double s_pos[MCW_WAVES][MCW_FRAME_ROWS][MC_MAX_ATOMS * 3];
    __shared__ double s_val[MCW_WAVES][MCW_FRAME_ROWS][MC_MAX_ATOMS][2];
    __shared__ double s_q[MC_MAX_ATOMS];
    __shared__ int32_t s_kind[MC_MAX_ATOMS];
    __shared__ McGrid s_grid[MC_MAX_ATOMS];        // geometry + pointer of every atom's VdW grid: read per lane from global memory they were a
                                                   // dependent 230-byte fetch in front of every interpolation
    static_assert(sizeof(McGrid) % 8 == 0, "McGrid is copied in 8-byte words");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int first = INSERT ? 0 : v.mol[molecule].x, m = INSERT ? nm.m : v.mol[molecule].y;
    if (tid < m) {
        int kind, mol;
        if (INSERT) kind = nm.kinds[tid];
        else unpack(v.atoms[first + tid].w, kind, mol);
        s_kind[tid] = kind;
        s_q[tid] = v.kind_charge[kind];
    }
    __syncthreads();
    {
        constexpr int W = (int)(sizeof(McGrid) / 8);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(s_grid);
        for (int t = tid; t < m * W; t += 64 * MCW_WAVES)
            dst[t] = reinterpret_cast<const unsigned long long*>(v.vdw + s_kind[t / W])[t % W];
    }
    __syncthreads();
    const int64_t p0 = ((int64_t)blockIdx.x * MCW_WAVES + wave) * per_wave;
    const int64_t p1 = p0 + per_wave < nrows ? p0 + per_wave : nrows;
    const int np = p1 > p0 ? (int)(p1 - p0) : 0;             // placements of this wave (<= MCW_FRAME_ROWS)
    // The 16 m corner evaluations of a placement (montecarlo.jl:490-504: lane 16a + 8g + corner, g = 0 the VdW grid of atom a, g = 1 the
    // Coulomb grid) fill 48 of 64 lanes for a three-atom molecule: the evaluations of ALL placements of the wave are numbered through and
    // taken 64 at a time (four placements of CO2: three passes instead of four), the per-atom values meet in LDS and are summed per
    // placement in the order of the atoms.
    double* pos = s_pos[wave][0];
    double* val = s_val[wave][0][0];
    const int per_row = 16 * m;
    for (int t = lane; t < np * 3 * m; t += 64) {
        const int r = t / (3 * m), c = t - r * 3 * m;
        const int64_t row = p0 + r;
        double x;
        if (!INSERT && row == 0) {
            const double4 A = v.atoms[first + c / 3];
            x = (c % 3 == 0) ? A.x : ((c % 3 == 1) ? A.y : A.z);
        } else {
            x = trial[(size_t)(INSERT ? row : row - 1) * m * 3 + c];
        }
        pos[r * (MC_MAX_ATOMS * 3) + c] = x;
    }
    __builtin_amdgcn_wave_barrier();
    const int total = np * per_row;
    for (int base = 0; base < total; base += 64) {
        const int l = base + lane;
        const int r = l / per_row, j = l - r * per_row;
        const int a = j >> 4, gsel = (j >> 3) & 1, corner = j & 7;
        double part = 0.0;
        bool blocked = false, have = false, isvdw = false;
        if (l < total) {
            const double* q3 = pos + r * (MC_MAX_ATOMS * 3) + 3 * a;
            const double px = q3[0], py = q3[1], pz = q3[2];
            if (gsel == 0) {
                const McGrid* G = s_grid + a;
                if (G->grid) { part = ceg_consumers::interp_corner(G->g, G->grid, px, py, pz, corner, blocked); have = true; isvdw = G->g.is_vdw != 0; }
            } else if (v.coulomb.grid) {
                part = ceg_consumers::interp_corner(v.coulomb.g, v.coulomb.grid, px, py, pz, corner, blocked);
                have = true;
                isvdw = v.coulomb.g.is_vdw != 0;
            }
        }
        int blk = blocked ? 1 : 0;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            part += __shfl_xor(part, o);
            blk |= __shfl_xor(blk, o);
        }
        if (l < total && corner == 0) {
            double out_v = 0.0;
            if (have) {
                const double vv = (isvdw && blk) ? 1e100 : part;      // grids.jl:245-248
                out_v = gsel == 0 ? vv : ((vv == 1e100) ? vv : s_q[a] * vv);        // montecarlo.jl:500
            }
            val[(r * MC_MAX_ATOMS + a) * 2 + gsel] = out_v;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 2 * np) {
        const int r = lane >> 1, gsel = lane & 1;
        double sum = 0.0;
        for (int a = 0; a < m; ++a) sum += val[(r * MC_MAX_ATOMS + a) * 2 + gsel];
        out[4 * (size_t)(p0 + r) + gsel] = sum;
    }
}

template <bool INSERT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 4) void k_mcw_ewald(McView v, int32_t molecule, McMolecule nm, const double* __restrict__ trial, int64_t nrows,
                                                              double* __restrict__ out, int stride, int per_wave)
{
    // dynamic LDS: [3 ns 64] doubles (planes A, B, kf) | [nrounds 64] int32 (padded to 16 B) | [WAVES][m][stride] double2 tables
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ double s_pos[WAVES][MC_MAX_ATOMS * 3];
    __shared__ double s_q[MC_MAX_ATOMS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int first = INSERT ? 0 : v.mol[molecule].x, m = INSERT ? nm.m : v.mol[molecule].y;
    const size_t plane = (size_t)v.ns * 64;
    double* cA = reinterpret_cast<double*>(s_raw);
    double* cB = cA + plane;
    double* ckf = cB + plane;
    int32_t* s_desc = reinterpret_cast<int32_t*>(ckf + plane);
    const size_t ndesc = ((size_t)v.nrounds * 64 + 3) & ~(size_t)3;
    double2* tab = reinterpret_cast<double2*>(s_desc + ndesc) + (size_t)wave * m * stride;
    if (tid < m) {
        int kind, mol;
        if (INSERT) kind = nm.kinds[tid];
        else unpack(v.atoms[first + tid].w, kind, mol);
        s_q[tid] = v.kind_charge[kind];
    }
    {   // staged once per workgroup
        const double2* mine = v.sf_mol + (size_t)(INSERT ? 0 : molecule) * v.nk;
        for (size_t t = tid; t < plane; t += 64 * WAVES) {
            const int q = v.qof[t];
            double a = 0.0, b = 0.0, k = 0.0;
            if (q >= 0) {
                const double2 f = v.sf_fw[q], tot = v.sf_tot[q];
                const double2 old = INSERT ? make_double2(0.0, 0.0) : mine[q];
                k = v.kf[q];
                a = k * (f.x + (tot.x - old.x));          // rest = framework + (sums[:,1] - sums[:,ij+1])
                b = k * (f.y + (tot.y - old.y));
            }
            cA[t] = a; cB[t] = b; ckf[t] = k;
        }
        for (int t = tid; t < v.nrounds * 64; t += 64 * WAVES) s_desc[t] = v.desc[t];
    }
    __syncthreads();
    const int nxp = v.ks[0] + 1, nyp = 2 * v.ks[1] + 1;
    const int64_t p0 = ((int64_t)blockIdx.x * WAVES + wave) * per_wave;
    const int64_t p1 = p0 + per_wave < nrows ? p0 + per_wave : nrows;
    double* pos = s_pos[wave];
    for (int64_t row = p0; row < p1; ++row) {
        double rs = 0.0, ss = 0.0;
        if (!INSERT && row == 0) {
            // positions === nothing (ewald.jl:731): the stored sums[:, ij+1] of the molecule
            const double2* mine = v.sf_mol + (size_t)molecule * v.nk;
            for (int q = lane; q < v.nk; q += 64) {
                const double2 old = mine[q], f = v.sf_fw[q], t = v.sf_tot[q];
                const double rr = f.x + (t.x - old.x), ri = f.y + (t.y - old.y);
                const double kf = v.kf[q];
                rs += kf * (rr * old.x + ri * old.y);
                ss += kf * (old.x * old.x + old.y * old.y);
            }
        } else {
            mcw_load_row<INSERT>(v, first, m, trial, row, lane, pos);
            fill_tables_wave(v, pos, m, tab, stride, lane, s_q);
            __builtin_amdgcn_wave_barrier();
            int slot = 0;
            for (int r = 0; r < v.nrounds; ++r) {
                const int d = s_desc[r * 64 + lane];
                const int L = __builtin_amdgcn_readfirstlane(d >> 27);
                const size_t at = (size_t)slot * 64 + lane;
                ceg_rows::round_dispatch(L, m, tab, stride, nxp, nyp, d & 0x1ff, (d >> 9) & 0x1ff, (d >> 18) & 0x1ff, [&](int sidx, double sr, double si) {
                    const size_t idx = at + (size_t)sidx * 64;
                    rs += cA[idx] * sr + cB[idx] * si;
                    ss += ckf[idx] * (sr * sr + si * si);
                });
                slot += L;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            rs += __shfl_xor(rs, o);
            ss += __shfl_xor(ss, o);
        }
        if (lane == 0) out[4 * (size_t)row + 3] = 2.0 * rs + ss;
        __builtin_amdgcn_wave_barrier();                           // pos / tab are rewritten for the next placement
    }
}

// WRAP (ceg_consumers::wrap_mode) is a template parameter so that an upper-triangular cell keeps twelve matrix elements live, not
// eighteen; whether the molecule is in the system (insert = false: row 0 is its current position, its own atoms are excluded) is a
// run-time flag
template <bool FAST, bool CELLS, int WRAP>
__global__ __launch_bounds__(64 * MCW_WAVES, (FAST && CELLS) ? 3 : 4) void k_mcw_pairs(McView v, McCompact ct, int32_t molecule, int32_t insert, McMolecule nm, const double* __restrict__ trial,
                                                                 int64_t nrows, double* __restrict__ out, int per_wave)
{
    // dynamic LDS (ct.in_lds): the compact pair table: fast records, rules, offsets
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ double s_pos[MCW_WAVES][MC_MAX_ATOMS * 3];
    __shared__ int s_first[CELLS ? MCW_WAVES : 1][64], s_cell[CELLS ? MCW_WAVES : 1][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int first = insert ? 0 : v.mol[molecule].x, m = insert ? nm.m : v.mol[molecule].y;
    const int excluded = insert ? -2 : molecule;                       // (free slots carry -1 and are skipped before this is looked at)
    const DevRule* rules = ct.rules;
    const int32_t* offset = ct.off;
    const McFastPair* fastp = ct.fast;
    if (ct.in_lds) {
        McFastPair* s_fast = reinterpret_cast<McFastPair*>(s_raw);
        DevRule* s_rules = reinterpret_cast<DevRule*>(s_fast + (size_t)v.nkinds * m);
        int32_t* s_off = reinterpret_cast<int32_t*>(s_rules + (ct.nrules > 0 ? ct.nrules : 1));
        for (int t = tid; t < v.nkinds * m; t += 64 * MCW_WAVES) s_fast[t] = ct.fast[t];
        for (int t = tid; t < ct.nrules; t += 64 * MCW_WAVES) s_rules[t] = ct.rules[t];
        for (int t = tid; t < v.nkinds * m + 1; t += 64 * MCW_WAVES) s_off[t] = ct.off[t];
        rules = s_rules;
        offset = s_off;
        fastp = s_fast;
    }
    __syncthreads();
    // the pair distance: ceg_consumers::pair_distance2_fast when every perpendicular width exceeds 2 cutoff (30 / 24 instructions
    // instead of 50), the reference's operation order otherwise and for pairs within 1e-9 of the cutoff
    const double band = 1e-9 * v.cutoff2;
    const int64_t p0 = ((int64_t)blockIdx.x * MCW_WAVES + wave) * per_wave;
    const int64_t p1 = p0 + per_wave < nrows ? p0 + per_wave : nrows;
    double* pos = s_pos[wave];
    const double* M = v.mat;
    const double* I = v.invmat;
    for (int64_t row = p0; row < p1; ++row) {
        if (lane < 3 * m) {
            if (!insert && row == 0) {
                const double4 A = v.atoms[first + lane / 3];
                pos[lane] = (lane % 3 == 0) ? A.x : ((lane % 3 == 1) ? A.y : A.z);
            } else {
                pos[lane] = trial[(size_t)(insert ? row : row - 1) * m * 3 + lane];
            }
        }
        __builtin_amdgcn_wave_barrier();
        double inter = 0.0;
        auto pairs_with = [&](const double4 A) __attribute__((always_inline)) {
            int kind1, mol;
            unpack(A.w, kind1, mol);
            if (mol < 0 || mol == excluded) return;                         // energy.jl:419 (and free slots)
            for (int a = 0; a < m; ++a) {
                const double dx = pos[3 * a] - A.x, dy = pos[3 * a + 1] - A.y, dz = pos[3 * a + 2] - A.z;
                const double r2 = WRAP == 0 ? ceg_consumers::pair_distance2_literal(M, I, dx, dy, dz)
                                            : ceg_consumers::pair_distance2_fast<WRAP == 2>(M, I, v.geom, dx, dy, dz, v.cutoff2, band);
                if (!(r2 < v.cutoff2)) continue;                            // :422
                const int t = kind1 * m + a;
                if (FAST && r2 >= 0.25) {
                    double r, rinv;
                    ceg::fast_sqrt_rsqrt(r2, r, rinv);
                    const McFastPair P = fastp[t];
                    if (P.cls) {
                        const double q2 = P.sigma2 * (rinv * rinv);
                        const double x6 = q2 * q2 * q2;
                        double e = __builtin_fma(P.c4eps * x6, x6 - 1.0, -P.shift);
                        if (P.qq != 0.0) {
                            const double x = P.alpha * r;
                            e = __builtin_fma(P.qq * rinv, ceg::fast_exp_neg(-(x * x)) * ceg::erfcx_poly(x), e);
                        }
                        inter += e;
                    } else {
                        for (int q = offset[t]; q < offset[t + 1]; ++q) inter += rule_energy_fast(rules[q], r2, r, rinv, v.coulombic);
                    }
                } else {
                    for (int q = offset[t]; q < offset[t + 1]; ++q) inter += rule_energy_call(&rules[q], r2, v.coulombic);
                }
            }
        };
        if (!CELLS) {
            // (the next 64 atoms are fetched while the current ones are worked on; free slots carry molecule -1 and are skipped)
            const double4 none = make_double4(0.0, 0.0, 0.0, __longlong_as_double(-1ll));
            double4 A = lane < v.natoms ? v.atoms[lane] : none;
            for (int l0 = 0; l0 < v.natoms; l0 += 64) {
                const int ln = l0 + 64 + lane;
                const double4 An = ln < v.natoms ? v.atoms[ln] : none;
                pairs_with(A);
                A = An;
            }
        } else {
            // only the cells the cutoff spheres of the molecule's atoms can reach (ceg_consumers.h); every lane works the range out
            int bin0[3], nbin[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) ceg_consumers::cell_range(I, pos, m, ax, v.nb[ax], v.hfrac[ax], bin0[ax], nbin[ax]);
            const int n1 = nbin[1], n2 = nbin[2];
            const int ncell = nbin[0] * n1 * n2;
            for (int base = 0; base < ncell; base += 64) {
                const int e = base + lane;
                int cnt = 0, cell = 0;
                if (e < ncell) {
                    const int j2 = e % n2, j1 = (e / n2) % n1, j0 = e / (n2 * n1);
                    int c0 = bin0[0] + j0, c1 = bin0[1] + j1, c2 = bin0[2] + j2;
                    if (c0 >= v.nb[0]) c0 -= v.nb[0];
                    if (c1 >= v.nb[1]) c1 -= v.nb[1];
                    if (c2 >= v.nb[2]) c2 -= v.nb[2];
                    cell = (c0 * v.nb[1] + c1) * v.nb[2] + c2;
                    cnt = v.cell_count[cell];
                }
                int incl = cnt;                                            // inclusive scan over the wave
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(incl, o);
                    if (lane >= o) incl += up;
                }
                const int total = __shfl(incl, 63);
                s_first[CELLS ? wave : 0][lane] = incl - cnt;
                s_cell[CELLS ? wave : 0][lane] = cell;
                __builtin_amdgcn_wave_barrier();
                for (int l = lane; l < total; l += 64) {
                    int j = 0;                                             // last cell whose first entry is <= l
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1)
                        if (s_first[CELLS ? wave : 0][j + step] <= l) j += step;
                    pairs_with(v.cells[(size_t)s_cell[CELLS ? wave : 0][j] * v.cell_cap + (l - s_first[CELLS ? wave : 0][j])]);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) inter += __shfl_xor(inter, o);
        if (lane == 0) out[4 * (size_t)row + 2] = inter;
        __builtin_amdgcn_wave_barrier();                           // pos is rewritten for the next placement
    }
}

// single_contribution_vdw on the fast path (ceg_pairfrac.h: pair tests on fractional coordinates, candidates queued, rules on full waves;
// round 4, after k_pairs_frac): FAST rules and every perpendicular width of the MC cell above two cutoffs.  MM: exact molecule size 1-4
// (0: any); TRI: upper-triangular cell.  The guest atoms come from fatoms / fcells (kept beside atoms / cells by every update kernel).
template <int MM, bool CELLS, bool TRI>
__global__ __launch_bounds__(64 * MCW_WAVES, CEG_PAIRFRAC_WAVES) void k_mcw_pairs_frac(McView v, ceg_pairfrac::FracTable tab, int32_t molecule, int32_t insert, McMolecule nm,
                                                                      const double* __restrict__ trial, int64_t nrows, double* __restrict__ out, int per_wave)
{
    using ceg_pairfrac::FQCAP;
    using ceg_pairfrac::FracHit;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ double s_mat[12];
    __shared__ double s_trial[MCW_WAVES][MC_MAX_ATOMS * 3];
    __shared__ double s_ft[MCW_WAVES][MC_MAX_ATOMS * 3];
    __shared__ FracHit s_q[MCW_WAVES][FQCAP];
    __shared__ int s_first[CELLS ? MCW_WAVES : 1][64], s_cell[CELLS ? MCW_WAVES : 1][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int first = insert ? 0 : v.mol[molecule].x;
    const int m = MM > 0 ? MM : (insert ? nm.m : v.mol[molecule].y);
    ceg_pairfrac::FracWave<MM, TRI> w;
    {
        ceg_pairfrac::PairFast* fastrec; DevRule* rules; int32_t* offset; const double* etab;
        ceg_pairfrac::stage(s_raw, tab, v.mat, s_mat, tid, 64 * MCW_WAVES, fastrec, rules, offset, etab);
        __syncthreads();
        w.fastrec = fastrec; w.rules = rules; w.offset = offset; w.etab = etab; w.ebase = tab.ebase; w.eni = tab.eni;
    }
    w.s_mat = s_mat; w.t3 = s_trial[wave]; w.ft = s_ft[wave]; w.hq = s_q[wave];
    w.frac = CELLS ? v.fcells : v.fatoms;
    w.cart = CELLS ? v.cells : v.atoms;
    w.geom = v.geom;
    w.cutoff2 = v.cutoff2; w.band = 1e-9 * v.cutoff2; w.cutoff2_band = w.cutoff2 + w.band; w.coulombic = v.coulombic;
    w.m = m; w.exclude = insert ? -2 : molecule; w.lane = lane;      // (free slots carry molecule -1: never live)
    const int64_t p0 = ((int64_t)blockIdx.x * MCW_WAVES + wave) * per_wave;
    const int64_t p1 = p0 + per_wave < nrows ? p0 + per_wave : nrows;
    for (int64_t row = p0; row < p1; ++row) {
        // row 0 of a displacement batch is the molecule where it is now
        w.load(v.invmat, [&](int i) -> double {
            if (!insert && row == 0) {
                const double4 A = v.atoms[first + i / 3];
                return (i % 3 == 0) ? A.x : ((i % 3 == 1) ? A.y : A.z);
            }
            return trial[(size_t)(insert ? row : row - 1) * m * 3 + i];
        });
        if (!CELLS) {
            const int total = v.natoms;
            auto locate = [&](int l) -> int { return l < total ? l : 0; };
            for (int l0 = 0; l0 < total;) {
                l0 = w.scan(locate, total, l0);
                w.flush(false);
            }
        } else {
            // only the cells the cutoff spheres of the molecule's atoms can reach (ceg_consumers.h): lane ax works out the range of axis ax
            int first_bin = 0, nbins = 1;
            if (lane < 3) ceg_consumers::cell_range(v.invmat, w.t3, m, lane, v.nb[lane], v.hfrac[lane], first_bin, nbins);
            const int b0 = __shfl(first_bin, 0), b1 = __shfl(first_bin, 1), b2 = __shfl(first_bin, 2);
            const int n1 = __shfl(nbins, 1), n2 = __shfl(nbins, 2);
            const int ncell = __shfl(nbins, 0) * n1 * n2;
            int* cfirst = s_first[CELLS ? wave : 0];
            int* ccell = s_cell[CELLS ? wave : 0];
            for (int base = 0; base < ncell; base += 64) {
                const int e = base + lane;
                int cnt = 0, cell = 0;
                if (e < ncell) {
                    const int j2 = e % n2, j1 = (e / n2) % n1, j0 = e / (n2 * n1);
                    int c0 = b0 + j0, c1 = b1 + j1, c2 = b2 + j2;
                    if (c0 >= v.nb[0]) c0 -= v.nb[0];
                    if (c1 >= v.nb[1]) c1 -= v.nb[1];
                    if (c2 >= v.nb[2]) c2 -= v.nb[2];
                    cell = (c0 * v.nb[1] + c1) * v.nb[2] + c2;
                    cnt = v.cell_count[cell];
                }
                int incl = cnt;                                            // inclusive scan over the wave
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(incl, o);
                    if (lane >= o) incl += up;
                }
                const int total = __shfl(incl, 63);
                __builtin_amdgcn_wave_barrier();
                cfirst[lane] = incl - cnt;
                ccell[lane] = cell;
                __builtin_amdgcn_wave_barrier();
                const int cap = v.cell_cap;
                auto locate = [&](int l) -> int {                          // entry l of the concatenated cells (0 beyond the end)
                    int j = 0;                                             // last cell whose first entry is <= l
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1)
                        if (cfirst[j + step] <= l) j += step;
                    return l < total ? ccell[j] * cap + (l - cfirst[j]) : 0;
                };
                for (int l0 = 0; l0 < total;) {
                    l0 = w.scan(locate, total, l0);
                    w.flush(false);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        const double sum = w.sum();
        if (lane == 0) out[4 * (size_t)row + 2] = sum;
        __builtin_amdgcn_wave_barrier();                           // t3 / ft are rewritten for the next placement
    }
}

// update_mc! for a displacement (montecarlo.jl:615-628): positions; sums[:,1] += new - sums[:,ij+1]; sums[:,ij+1] = new
__global__ __launch_bounds__(MC_THREADS) void k_mc_accept(McView v, int32_t molecule, McPositions np, McCellOps ops, int stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ double s_pos[MC_MAX_ATOMS * 3];
    __shared__ double s_q[MC_MAX_ATOMS];
    const int tid = threadIdx.x;
    const int first = v.mol[molecule].x, m = v.mol[molecule].y;
    if (tid < 3 * m) s_pos[tid] = np.xyz[tid];
    if (tid < m) {
        double4 A = v.atoms[first + tid];
        int kind, mol;
        unpack(A.w, kind, mol);
        s_q[tid] = v.kind_charge[kind];
        A.x = np.xyz[3 * tid]; A.y = np.xyz[3 * tid + 1]; A.z = np.xyz[3 * tid + 2];
        put_atom(v, first + tid, A);
    }
    __syncthreads();
    if (v.use_cells) apply_cell_ops(v, ops, tid);
    if (v.nk == 0) return;
    double2* tab = reinterpret_cast<double2*>(s_raw);
    fill_tables(v, s_pos, m, tab, stride, tid, MC_THREADS, s_q);
    __syncthreads();
    double2* mine = v.sf_mol + (size_t)molecule * v.nk;
    rows_structure_factor(v, tab, stride, m, tid >> 6, MC_THREADS / 64, tid & 63, [&](int q, double sr, double si) {
        const double2 old = mine[q];
        double2 t = v.sf_tot[q];
        t.x += sr - old.x;
        t.y += si - old.y;
        v.sf_tot[q] = t;
        mine[q] = make_double2(sr, si);
    });
}

// sums[:, ij+1] of every molecule from its current positions (one workgroup per molecule)
__global__ __launch_bounds__(MC_THREADS) void k_mc_sf_molecules(McView v, int stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ double s_pos[MC_MAX_ATOMS * 3];
    __shared__ double s_q[MC_MAX_ATOMS];
    const int tid = threadIdx.x, molecule = blockIdx.x;
    const int first = v.mol[molecule].x, m = v.mol[molecule].y;
    if (tid < m) {
        const double4 A = v.atoms[first + tid];
        int kind, mol;
        unpack(A.w, kind, mol);
        s_q[tid] = v.kind_charge[kind];
        s_pos[3 * tid] = A.x; s_pos[3 * tid + 1] = A.y; s_pos[3 * tid + 2] = A.z;
    }
    __syncthreads();
    double2* tab = reinterpret_cast<double2*>(s_raw);
    fill_tables(v, s_pos, m, tab, stride, tid, MC_THREADS, s_q);
    __syncthreads();
    double2* mine = v.sf_mol + (size_t)molecule * v.nk;
    rows_structure_factor(v, tab, stride, m, tid >> 6, MC_THREADS / 64, tid & 63,
                          [&](int q, double sr, double si) { mine[q] = make_double2(sr, si); });
}

// sums[:, 1] = sum over the molecules, in molecule order
__global__ void k_mc_sf_total(McView v)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= v.nk) return;
    double sr = 0.0, si = 0.0;
    for (int j = 0; j < v.nmol; ++j) {
        const double2 s = v.sf_mol[(size_t)j * v.nk + q];
        sr += s.x;
        si += s.y;
    }
    v.sf_tot[q] = make_double2(sr, si);
}

// add_one_system! (ewald.jl:775-792, montecarlo.jl:615-621): new molecule `molecule` (= old nmol) in atom slots [first, first + m)
__global__ __launch_bounds__(MC_THREADS) void k_mc_insert(McView v, int32_t molecule, int32_t first, McMolecule nm, McPositions np, McCellOps ops, int stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ double s_pos[MC_MAX_ATOMS * 3];
    __shared__ double s_q[MC_MAX_ATOMS];
    const int tid = threadIdx.x, m = nm.m;
    if (tid < 3 * m) s_pos[tid] = np.xyz[tid];
    if (tid < m) {
        const long long bits = ((long long)molecule << 32) | (long long)(uint32_t)nm.kinds[tid];
        put_atom(v, first + tid, make_double4(np.xyz[3 * tid], np.xyz[3 * tid + 1], np.xyz[3 * tid + 2], __longlong_as_double(bits)));
        s_q[tid] = v.kind_charge[nm.kinds[tid]];
    }
    if (tid == 0) v.mol[molecule] = make_int2(first, m);
    __syncthreads();
    if (v.use_cells) apply_cell_ops(v, ops, tid);
    if (v.nk == 0) return;
    double2* tab = reinterpret_cast<double2*>(s_raw);
    fill_tables(v, s_pos, m, tab, stride, tid, MC_THREADS, s_q);
    __syncthreads();
    double2* mine = v.sf_mol + (size_t)molecule * v.nk;
    rows_structure_factor(v, tab, stride, m, tid >> 6, MC_THREADS / 64, tid & 63, [&](int q, double sr, double si) {
        double2 t = v.sf_tot[q];
        t.x += sr;
        t.y += si;
        v.sf_tot[q] = t;
        mine[q] = make_double2(sr, si);
    });
}

// remove_one_system! (ewald.jl:794-810, :404-413): sums[:,1] -= sums[:,ij+1]; the LAST molecule takes index `molecule`
// (its structure factor column and the molecule id of its atoms); the atom slots of the removed molecule become free
__global__ __launch_bounds__(MC_THREADS) void k_mc_remove(McView v, int32_t molecule, int32_t last, McCellOps ops)
{
    const int tid = threadIdx.x;
    const int2 gone = v.mol[molecule], moved = v.mol[last];
    double2* mine = v.sf_mol + (size_t)molecule * v.nk;
    const double2* lastsf = v.sf_mol + (size_t)last * v.nk;
    for (int64_t q = tid; q < v.nk; q += MC_THREADS) {
        const double2 old = mine[q];
        double2 t = v.sf_tot[q];
        t.x -= old.x;
        t.y -= old.y;
        v.sf_tot[q] = t;
        if (last != molecule) mine[q] = lastsf[q];
    }
    if (tid < gone.y) {
        double4 A = v.atoms[gone.x + tid];
        int kind, mol;
        unpack(A.w, kind, mol);
        const long long bits = (long long)(0xffffffff00000000ull | (unsigned long long)(uint32_t)kind);      // molecule id -1: free slot
        A.w = __longlong_as_double(bits);
        put_atom(v, gone.x + tid, A);
    }
    if (last != molecule && tid >= 64 && tid < 64 + moved.y) {
        double4 A = v.atoms[moved.x + tid - 64];
        int kind, mol;
        unpack(A.w, kind, mol);
        const long long bits = ((long long)molecule << 32) | (long long)(uint32_t)kind;
        A.w = __longlong_as_double(bits);
        put_atom(v, moved.x + tid - 64, A);
    }
    __syncthreads();
    if (v.use_cells) apply_cell_ops(v, ops, tid);
    if (tid == 0 && last != molecule) v.mol[molecule] = moved;
}

// cells[i] = atoms[map[i]] for every occupied entry (map[i] >= 0): the whole structure from the host's cell lists
__global__ void k_mc_cells_fill(McView v, const int32_t* __restrict__ map, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && map[i] >= 0) {
        v.cells[i] = v.atoms[map[i]];
        v.fcells[i] = v.fatoms[map[i]];
    }
}

// fatoms[] from atoms[] for every slot (after an upload, after the arrays grew)
__global__ void k_mc_frac_fill(McView v, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) put_atom(v, (int)i, v.atoms[i]);
}

int merr(int code, const char* msg)
{
    ceg_set_last_error_(msg);
    return code;
}

struct Guard {
    int prev = -1;
    bool ok;
    explicit Guard(int device)
    {
        (void)hipGetDevice(&prev);
        ok = hipSetDevice(device) == hipSuccess;
    }
    ~Guard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <typename T>
bool upload(T** dst, const T* src, size_t n)
{
    const size_t m = n > 0 ? n : 1;
    if (hipMalloc((void**)dst, m * sizeof(T)) != hipSuccess) return false;
    return n == 0 || hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}

// Host mirror of the cell lists (which atom slot sits where); the device holds the records themselves.  Every update is
// worked out here first and shipped to the device as a handful of "copy atom record `src` to cell entry `dst`" operations
// inside the update kernel's arguments, so the bookkeeping costs no extra launch, no upload and no device-side search.
struct CellMirror {
    bool on = false;
    int nb[3] = {1, 1, 1};
    int cap = 0;
    double invmat[9];
    ceg_consumers::CellBins bins{};
    std::vector<std::vector<int32_t>> members;          // [ncells] atom slots
    std::vector<int32_t> cell_of, idx_of;               // per atom slot; cell_of < 0: not in any cell
    std::vector<std::pair<int32_t, int32_t>> touched;   // (cell, entry) whose content changed
    std::vector<int32_t> touched_cells;
    int max_fill = 0;

    int ncells() const { return nb[0] * nb[1] * nb[2]; }
    int bin_of(const double* p) const { return ceg_consumers::cell_of_position(bins, invmat, p); }
    void begin() { touched.clear(); touched_cells.clear(); }
    void touch(int32_t c, int32_t i)
    {
        for (const auto& t : touched)
            if (t.first == c && t.second == i) return;
        touched.emplace_back(c, i);
    }
    void touch_cell(int32_t c)
    {
        if (std::find(touched_cells.begin(), touched_cells.end(), c) == touched_cells.end()) touched_cells.push_back(c);
    }
    void ensure_slot(int64_t slot)
    {
        if ((int64_t)cell_of.size() <= slot) { cell_of.resize((size_t)slot + 1, -1); idx_of.resize((size_t)slot + 1, -1); }
    }
    void take_out(int32_t slot)
    {
        const int32_t c = cell_of[slot], i = idx_of[slot];
        if (c < 0) return;
        std::vector<int32_t>& mem = members[c];
        const int32_t moved = mem.back();
        mem[i] = moved;
        idx_of[moved] = i;
        mem.pop_back();
        cell_of[slot] = -1; idx_of[slot] = -1;
        touch(c, i);
        touch_cell(c);
    }
    void put_in(int32_t slot, int32_t c)
    {
        ensure_slot(slot);
        std::vector<int32_t>& mem = members[c];
        mem.push_back(slot);
        cell_of[slot] = c; idx_of[slot] = (int32_t)mem.size() - 1;
        max_fill = std::max(max_fill, (int)mem.size());
        touch(c, idx_of[slot]);
        touch_cell(c);
    }
    void refresh(int32_t slot)
    {
        if (cell_of[slot] >= 0) touch(cell_of[slot], idx_of[slot]);
    }
    // false: an entry beyond the capacity is in use (or too many operations): the caller rebuilds the device arrays
    bool finish(McCellOps& ops) const
    {
        ops.nops = 0; ops.ncnt = 0;
        if (max_fill > cap) return false;
        for (const auto& t : touched) {
            const std::vector<int32_t>& mem = members[t.first];
            if (t.second >= (int32_t)mem.size()) continue;           // the entry fell off the end of its list
            if (ops.nops == MC_MAX_CELL_OPS) return false;
            ops.dst[ops.nops] = t.first * cap + t.second;
            ops.src[ops.nops] = mem[t.second];
            ++ops.nops;
        }
        for (int32_t c : touched_cells) {
            if (ops.ncnt == MC_MAX_CELL_OPS) return false;
            ops.cell[ops.ncnt] = c;
            ops.count[ops.ncnt] = (int32_t)members[c].size();
            ++ops.ncnt;
        }
        return true;
    }
};

}  // namespace

struct ceg_mc {
    int device = 0;
    McView v{};
    hipStream_t stream = nullptr;
    // owned device arrays
    McGrid* d_vdw = nullptr;
    double* d_charge = nullptr;
    DevRule* d_rules = nullptr;
    int32_t* d_offset = nullptr;
    int32_t* d_ijk = nullptr;
    double* d_kf = nullptr;
    double2 *d_fw = nullptr, *d_tot = nullptr, *d_mol = nullptr;
    double4* d_atoms = nullptr;
    int2* d_molidx = nullptr;
    int64_t atoms_cap = 0, mol_cap = 0;
    std::vector<int2> h_mol;                     // host copy of (start, count) per molecule
    std::vector<std::vector<int32_t>> free_runs; // free_runs[m]: starts of free runs of m atom slots
    CellMirror cm;                               // neighbour cells of the guest atoms (when the MC cell is large enough to gain)
    double4* d_cells = nullptr;
    int32_t* d_cell_count = nullptr;
    int stride = 0;
    // row-wise k-vector layout (ceg_rows.h) and, per distinct molecule (tuple of atom kinds), the pair-table rows of its kinds
    int32_t *d_desc = nullptr, *d_qof = nullptr;
    double* d_geom = nullptr;
    std::vector<DevRule> h_rules;
    std::vector<int32_t> h_offset;
    std::vector<int32_t> h_kind;                 // kind per atom slot (host copy)
    std::vector<double> h_charge;                // charge per kind (host copy)
    double* d_etab = nullptr;                    // erfc(alpha r)/r records of ceg_pairfrac.h (CoulombEwaldDirect rules sharing one alpha)
    int32_t ebase = 0, eni = 0;
    struct Compact { DevRule* d_rules = nullptr; int32_t* d_off = nullptr; void* d_fast = nullptr; int32_t nrules = 0; };
    std::map<std::vector<int32_t>, Compact> compact;
    // pinned, device-mapped staging for small batches; device scratch for large ones
    double *h_in = nullptr, *h_out = nullptr, *dm_in = nullptr, *dm_out = nullptr;
    double *d_in = nullptr, *d_out = nullptr;
    size_t d_in_cap = 0, d_out_cap = 0;
    // completion flag of the mapped-buffer path (polled by the host) and the device-side count of finished workgroups
    unsigned long long *h_flag = nullptr, *dm_flag = nullptr;
    unsigned* d_done = nullptr;
    unsigned long long seq = 0;
    // set when a state-changing call failed after it had started to change the host mirror (counts, slot lists, cell lists) or the
    // device state: host and device may then disagree, so every later call fails until ceg_mc_set_guests rebuilds both
    bool poisoned = false;
};

namespace {
int rebuild_cells(ceg_mc* h);

int poison(ceg_mc* h, int rc)
{
    h->poisoned = true;
    return rc;
}

int refuse_poisoned()
{
    return merr(CEG_ERR_HIP, "the handle is inconsistent after an earlier failure of accept / insert / remove: call ceg_mc_set_guests");
}

// test hook: CEG_HIP_MC_INJECT_FAILURE=accept|insert|remove|set_guests makes the next such call fail after its host-side bookkeeping
bool injected_failure(const char* what)
{
    const char* e = std::getenv("CEG_HIP_MC_INJECT_FAILURE");
    return e && std::strcmp(e, what) == 0;
}
}

extern "C" int ceg_mc_create(ceg_mc_t** handle, int32_t device, ceg_interp_t* const* vdw_grids, ceg_interp_t* coulomb_grid,
                             const double* kind_charge, int32_t nkinds, const double mat[9], const double invmat[9], double cutoff2,
                             const ceg_rule_t* rules, const int32_t* rule_offset, double coulombic, const int32_t* kvec_ijk,
                             const double* kfactors, const double* sf_re, const double* sf_im, int64_t nk, const int32_t ks[3],
                             const double ewald_invmat[9])
{
    if (!handle || !kind_charge || nkinds < 1 || nkinds > 4096 || !mat || !invmat || !rule_offset || !(cutoff2 > 0.0) || nk < 0)
        return merr(CEG_ERR_INVALID, "bad argument");
    *handle = nullptr;
    if (nk > 0 && (!kvec_ijk || !kfactors || !sf_re || !sf_im || !ks || !ewald_invmat)) return merr(CEG_ERR_INVALID, "k-space tables missing");
    if (nk > 0 && (ks[0] < 0 || ks[1] < 0 || ks[2] < 0 || ks[0] + 1 + 2 * ks[1] + 1 + 2 * ks[2] + 1 > MC_MAX_TAB))
        return merr(CEG_ERR_UNSUPPORTED, "k-space box too large for the LDS tables");
    for (int64_t q = 0; q < nk; ++q)
        if (kvec_ijk[3 * q] < 0 || kvec_ijk[3 * q] > ks[0] || abs(kvec_ijk[3 * q + 1]) > ks[1] || abs(kvec_ijk[3 * q + 2]) > ks[2])
            return merr(CEG_ERR_INVALID, "k-vector outside the (kx, ky, kz) box");
    const int64_t nt = (int64_t)nkinds * nkinds;
    if (rule_offset[0] != 0) return merr(CEG_ERR_INVALID, "rule_offset[0] must be 0");
    for (int64_t t = 0; t < nt; ++t)
        if (rule_offset[t + 1] < rule_offset[t]) return merr(CEG_ERR_INVALID, "rule_offset must be non-decreasing");
    const int32_t nr = rule_offset[nt];
    if (nr > 0 && !rules) return merr(CEG_ERR_INVALID, "rules missing");
    std::vector<DevRule> dr((size_t)(nr > 0 ? nr : 1));
    bool fast = true;
    const double cutoff = std::sqrt(cutoff2);
    for (int32_t q = 0; q < nr; ++q) {
        const ceg_rule_t& r = rules[q];
        if (r.kind < CEG_HARDSPHERE || r.kind > CEG_NOINTERACTION) return merr(CEG_ERR_INVALID, "unknown rule kind");
        if (r.kind == CEG_UNDEFINED_INTERACTION) return merr(CEG_ERR_RULE, "Undefined interaction");
        dr[q].kind = r.kind; dr[q]._pad = 0;
        dr[q].p0 = r.p[0]; dr[q].p1 = r.p[1]; dr[q].p2 = r.p[2]; dr[q].shift = r.shift;
        if (r.kind == CEG_COULOMB_EWALD_DIRECT && !(r.p[0] >= 0.0 && r.p[0] * cutoff <= 5.0 * (1.0 - 1e-9))) fast = false;
        if ((r.kind == CEG_BUCKINGHAM || r.kind == CEG_EXPONENTIAL) && !(r.p[1] >= 0.0 && r.p[1] * cutoff <= 700.0)) fast = false;
    }
    if (ceg_device_count() <= 0) return merr(CEG_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ceg_device_count()) return merr(CEG_ERR_NO_DEVICE, "device not present");
    Guard guard(device);
    if (!guard.ok) return merr(CEG_ERR_HIP, "hipSetDevice failed");
    ceg_mc* h = new ceg_mc();
    h->device = device;
    McView& v = h->v;
    for (int a = 0; a < 9; ++a) { v.mat[a] = mat[a]; v.invmat[a] = invmat[a]; v.ew_invmat[a] = nk > 0 ? ewald_invmat[a] : 0.0; }
    v.cutoff2 = cutoff2; v.coulombic = coulombic; v.nkinds = nkinds; v.nrules = nr; v.fast = fast ? 1 : 0;
    v.nk = (int32_t)nk;
    for (int a = 0; a < 3; ++a) v.ks[a] = nk > 0 ? ks[a] : 0;
    h->stride = nk > 0 ? ks[0] + 1 + 2 * ks[1] + 1 + 2 * ks[2] + 1 : 1;
    std::vector<McGrid> grids((size_t)nkinds);
    for (int32_t k = 0; k < nkinds; ++k) {
        grids[k] = McGrid{};
        if (vdw_grids && vdw_grids[k]) {
            if (vdw_grids[k]->device != device) { delete h; return merr(CEG_ERR_INVALID, "grid handle lives on another device"); }
            grids[k].g = vdw_grids[k]->g;
            grids[k].grid = vdw_grids[k]->d_grid;
        }
    }
    v.coulomb = McGrid{};
    if (coulomb_grid) {
        if (coulomb_grid->device != device) { delete h; return merr(CEG_ERR_INVALID, "grid handle lives on another device"); }
        v.coulomb.g = coulomb_grid->g;
        v.coulomb.grid = coulomb_grid->d_grid;
    }
    std::vector<double2> fw((size_t)(nk > 0 ? nk : 1));
    for (int64_t q = 0; q < nk; ++q) fw[q] = make_double2(sf_re[q], sf_im[q]);
    const size_t table_bytes = sizeof(DevRule) * dr.size() + sizeof(int32_t) * (size_t)(nt + 1);
    v.table_in_lds = table_bytes <= 32 * 1024 ? 1 : 0;
    h->h_charge.assign(kind_charge, kind_charge + nkinds);
    bool ok = upload(&h->d_vdw, grids.data(), grids.size()) && upload(&h->d_charge, kind_charge, (size_t)nkinds) &&
              upload(&h->d_rules, dr.data(), dr.size()) && upload(&h->d_offset, rule_offset, (size_t)(nt + 1)) &&
              upload(&h->d_ijk, kvec_ijk, (size_t)(3 * nk)) && upload(&h->d_kf, kfactors, (size_t)nk) &&
              upload(&h->d_fw, fw.data(), (size_t)nk) && upload(&h->d_tot, fw.data(), (size_t)nk);
    ok = ok && hipMemset(h->d_tot, 0, sizeof(double2) * (size_t)(nk > 0 ? nk : 1)) == hipSuccess;
    if (ok && v.fast) {   // the r^2-indexed erfc(alpha r)/r records of the fractional-coordinate pair kernel (ceg_pairfrac.h)
        double alpha = 0.0;
        bool shared = true;
        for (const DevRule& R : dr)
            if (R.kind == CEG_COULOMB_EWALD_DIRECT) {
                if (alpha == 0.0) alpha = R.p0;
                else if (alpha != R.p0) shared = false;
            }
        ceg_pairfrac::ErfcTable et;
        if (shared && alpha > 0.0 && cutoff2 > 1.0 && ceg_pairfrac::build_erfc_table(alpha, 1.0, cutoff2, et) && upload(&h->d_etab, et.rec.data(), et.rec.size())) {
            h->ebase = et.base; h->eni = et.ni;
        }
    }
    {
        const ceg_rows::Layout lay = ceg_rows::choose_layout(kvec_ijk, nk, nk > 0 ? ks : v.ks);
        std::vector<int32_t> qof((size_t)std::max(lay.ns, 1) * 64, -1);
        for (int64_t q = 0; q < nk; ++q) qof[(size_t)lay.slot_of[(size_t)q]] = (int32_t)q;
        v.nrounds = lay.nrounds; v.ns = lay.ns;
        ok = ok && upload(&h->d_desc, lay.desc.data(), lay.desc.size()) && upload(&h->d_qof, qof.data(), qof.size());
    }
    {
        double geom[18];
        for (int a = 0; a < 9; ++a) { geom[a] = mat[a]; geom[9 + a] = invmat[a]; }
        ok = ok && upload(&h->d_geom, geom, 18);
    }
    h->h_rules.assign(dr.begin(), dr.begin() + (nr > 0 ? nr : 0));
    h->h_offset.assign(rule_offset, rule_offset + nt + 1);
    ok = ok && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&h->h_in, MC_MAPPED_BYTES, hipHostMallocMapped) == hipSuccess &&
         hipHostMalloc((void**)&h->h_out, MC_MAPPED_BYTES, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
         hipHostGetDevicePointer((void**)&h->dm_in, h->h_in, 0) == hipSuccess &&
         hipHostGetDevicePointer((void**)&h->dm_out, h->h_out, 0) == hipSuccess &&
         hipHostMalloc((void**)&h->h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
         hipHostGetDevicePointer((void**)&h->dm_flag, h->h_flag, 0) == hipSuccess &&
         hipMalloc((void**)&h->d_done, sizeof(unsigned)) == hipSuccess && hipMemset(h->d_done, 0, sizeof(unsigned)) == hipSuccess;
    if (ok) *h->h_flag = 0ull;
    if (!ok) {
        ceg_mc_destroy(h);
        return merr(CEG_ERR_HIP, "could not allocate the Monte-Carlo state on the device");
    }
    v.vdw = h->d_vdw; v.kind_charge = h->d_charge; v.rules = h->d_rules; v.rule_offset = h->d_offset;
    v.ijk = h->d_ijk; v.kf = h->d_kf; v.sf_fw = h->d_fw; v.sf_tot = h->d_tot;
    v.desc = h->d_desc; v.qof = h->d_qof; v.geom = h->d_geom;
    {
        CellMirror& cm = h->cm;
        cm.bins = ceg_consumers::choose_cell_bins(invmat, cutoff);
        for (int i = 0; i < 3; ++i) { cm.nb[i] = cm.bins.nb[i]; v.nb[i] = cm.bins.nb[i]; v.hfrac[i] = cm.bins.hfrac[i]; }
        for (int a = 0; a < 9; ++a) cm.invmat[a] = invmat[a];
        cm.on = cm.bins.on != 0;
        v.use_cells = cm.on ? 1 : 0;
        v.fastwrap = ceg_consumers::wrap_mode(mat, invmat, cm.bins.hfrac);
        if (cm.on) {
            cm.members.assign((size_t)cm.ncells(), {});
            if (int rc = rebuild_cells(h)) { ceg_mc_destroy(h); return rc; }
        }
    }
    *handle = h;
    return CEG_OK;
}

extern "C" int ceg_mc_destroy(ceg_mc_t* h)
{
    if (!h) return CEG_OK;
    Guard guard(h->device);
    if (guard.ok) {
        if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
        for (void* p : {(void*)h->d_vdw, (void*)h->d_charge, (void*)h->d_rules, (void*)h->d_offset, (void*)h->d_ijk, (void*)h->d_kf,
                        (void*)h->d_fw, (void*)h->d_tot, (void*)h->d_mol, (void*)h->d_atoms, (void*)h->d_molidx, (void*)h->d_in, (void*)h->d_out, (void*)h->d_cells, (void*)h->d_cell_count, (void*)h->d_etab})
            if (p) (void)hipFree(p);
        if (h->h_in) (void)hipHostFree(h->h_in);
        if (h->h_out) (void)hipHostFree(h->h_out);
        if (h->h_flag) (void)hipHostFree(h->h_flag);
        if (h->d_done) (void)hipFree(h->d_done);
        if (h->d_desc) (void)hipFree(h->d_desc);
        if (h->d_geom) (void)hipFree(h->d_geom);
        if (h->d_qof) (void)hipFree(h->d_qof);
        for (auto& kv : h->compact) {
            if (kv.second.d_rules) (void)hipFree(kv.second.d_rules);
            if (kv.second.d_off) (void)hipFree(kv.second.d_off);
            if (kv.second.d_fast) (void)hipFree(kv.second.d_fast);
        }
    }
    delete h;
    return CEG_OK;
}

namespace {

// grow the device arrays (contents kept); the stream is idle when this returns
int ensure_capacity(ceg_mc* h, int64_t natoms, int64_t nmol)
{
    if (natoms <= h->atoms_cap && nmol <= h->mol_cap) return CEG_OK;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return merr(CEG_ERR_HIP, "stream synchronisation failed");
    const size_t nk = (size_t)(h->v.nk > 0 ? h->v.nk : 1);
    const bool regrown = natoms > h->atoms_cap;
    if (natoms > h->atoms_cap) {
        const int64_t cap = natoms + natoms / 2 + 64;
        double4* p = nullptr;
        if (hipMalloc((void**)&p, 2 * sizeof(double4) * (size_t)cap) != hipSuccess) return merr(CEG_ERR_HIP, "could not allocate the guest atoms");      // Cartesian, then fractional
        if (h->d_atoms && h->v.natoms > 0 &&
            hipMemcpy(p, h->d_atoms, sizeof(double4) * (size_t)h->v.natoms, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipFree(p); return merr(CEG_ERR_HIP, "device copy failed"); }
        if (h->d_atoms) (void)hipFree(h->d_atoms);
        h->d_atoms = p;
        h->atoms_cap = cap;
    }
    if (nmol > h->mol_cap) {
        const int64_t cap = nmol + nmol / 2 + 16;
        double2* pm = nullptr;
        int2* pi = nullptr;
        if (hipMalloc((void**)&pm, sizeof(double2) * (size_t)cap * nk) != hipSuccess || hipMalloc((void**)&pi, sizeof(int2) * (size_t)cap) != hipSuccess) {
            if (pm) (void)hipFree(pm);
            return merr(CEG_ERR_HIP, "could not allocate the per-molecule arrays");
        }
        bool ok = true;
        if (h->d_mol && h->v.nmol > 0) ok = hipMemcpy(pm, h->d_mol, sizeof(double2) * (size_t)h->v.nmol * nk, hipMemcpyDeviceToDevice) == hipSuccess;
        if (ok && h->d_molidx && h->v.nmol > 0) ok = hipMemcpy(pi, h->d_molidx, sizeof(int2) * (size_t)h->v.nmol, hipMemcpyDeviceToDevice) == hipSuccess;
        if (!ok) { (void)hipFree(pm); (void)hipFree(pi); return merr(CEG_ERR_HIP, "device copy failed"); }
        if (h->d_mol) (void)hipFree(h->d_mol);
        if (h->d_molidx) (void)hipFree(h->d_molidx);
        h->d_mol = pm;
        h->d_molidx = pi;
        h->mol_cap = cap;
    }
    h->v.atoms = h->d_atoms; h->v.fatoms = h->d_atoms + h->atoms_cap; h->v.mol = h->d_molidx; h->v.sf_mol = h->d_mol;
    if (regrown && h->v.natoms > 0) {
        hipLaunchKernelGGL(k_mc_frac_fill, dim3((unsigned)((h->v.natoms + 255) / 256)), dim3(256), 0, h->stream, h->v, (int64_t)h->v.natoms);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return merr(CEG_ERR_HIP, "could not convert the guest atoms");
    }
    return CEG_OK;
}

// the device cell arrays from the host lists (first fill, or after a cell outgrew the capacity); atoms[] must be current in
// stream order.  The stream is idle when this returns.
int rebuild_cells(ceg_mc* h)
{
    CellMirror& cm = h->cm;
    if (!cm.on) return CEG_OK;
    const int ncells = cm.ncells();
    int cap = std::max(cm.cap, 8);
    while (cap < cm.max_fill + cm.max_fill / 2 + 2) cap *= 2;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return merr(CEG_ERR_HIP, "stream synchronisation failed");
    if (cap != cm.cap || !h->d_cells) {
        if (h->d_cells) (void)hipFree(h->d_cells);
        h->d_cells = nullptr;
        if (hipMalloc((void**)&h->d_cells, 2 * sizeof(double4) * (size_t)ncells * cap) != hipSuccess) return merr(CEG_ERR_HIP, "could not allocate the neighbour cells");
        cm.cap = cap;
    }
    if (!h->d_cell_count && hipMalloc((void**)&h->d_cell_count, sizeof(int32_t) * (size_t)ncells) != hipSuccess)
        return merr(CEG_ERR_HIP, "could not allocate the neighbour cells");
    std::vector<int32_t> map((size_t)ncells * cap, -1), count((size_t)ncells);
    for (int c = 0; c < ncells; ++c) {
        count[c] = (int32_t)cm.members[c].size();
        std::copy(cm.members[c].begin(), cm.members[c].end(), map.begin() + (size_t)c * cap);
    }
    int32_t* d_map = nullptr;
    if (hipMalloc((void**)&d_map, sizeof(int32_t) * map.size()) != hipSuccess) return merr(CEG_ERR_HIP, "hipMalloc failed");
    bool ok = hipMemcpy(d_map, map.data(), sizeof(int32_t) * map.size(), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(h->d_cell_count, count.data(), sizeof(int32_t) * count.size(), hipMemcpyHostToDevice) == hipSuccess;
    h->v.cells = h->d_cells; h->v.fcells = h->d_cells + (size_t)ncells * cm.cap; h->v.cell_count = h->d_cell_count; h->v.cell_cap = cm.cap;
    if (ok) {
        const int64_t n = (int64_t)map.size();
        hipLaunchKernelGGL(k_mc_cells_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->v, d_map, n);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess;
    }
    (void)hipFree(d_map);
    return ok ? CEG_OK : merr(CEG_ERR_HIP, "could not fill the neighbour cells");
}

size_t tables_bytes(const ceg_mc* h, int m) { return sizeof(double2) * (size_t)m * (size_t)h->stride; }

// the pair-table rows of the kinds of one molecule: entry (kind1, a) -> the rules of (kind1, kinds[a]); built once per distinct
// molecule and kept on the device (a handful per run: one per species)
const ceg_mc::Compact* compact_table(ceg_mc* h, const int32_t* kinds, int m)
{
    std::vector<int32_t> key(kinds, kinds + m);
    auto it = h->compact.find(key);
    if (it != h->compact.end()) return &it->second;
    const int nkinds = h->v.nkinds;
    std::vector<int32_t> off((size_t)nkinds * m + 1, 0);
    std::vector<DevRule> rules;
    std::vector<McFastPair> fast((size_t)nkinds * m);
    for (int k1 = 0; k1 < nkinds; ++k1)
        for (int a = 0; a < m; ++a) {
            const size_t t = (size_t)k1 * nkinds + kinds[a];
            // at most one Lennard-Jones and one CoulombEwaldDirect term (and NoInteraction): the whole entry is one branch-free record
            // (v - shift summed in another order than the rule loop: inside the 1e-9 of the pair sum, like the rest of the fast path)
            McFastPair P{0.0, 0.0, 0.0, 0.0, 0.0, 1, 0};
            int nlj = 0, nced = 0;
            for (int32_t q = h->h_offset[t]; q < h->h_offset[t + 1]; ++q) {
                const DevRule& R = h->h_rules[(size_t)q];
                rules.push_back(R);
                if (R.kind == CEG_LENNARDJONES && nlj == 0) { P.c4eps = 4.0 * R.p0; P.sigma2 = R.p1 * R.p1; P.shift += R.shift; ++nlj; }
                else if (R.kind == CEG_COULOMB_EWALD_DIRECT && nced == 0) { P.alpha = R.p0; P.qq = h->v.coulombic * R.p1 * R.p2; P.shift += R.shift; ++nced; }
                else if (R.kind == CEG_NOINTERACTION) P.shift += R.shift;
                else P.cls = 0;
            }
            fast[(size_t)k1 * m + a] = P;
            off[(size_t)k1 * m + a + 1] = (int32_t)rules.size();
        }
    ceg_mc::Compact c;
    c.nrules = (int32_t)rules.size();
    McFastPair* d_fast = nullptr;
    if (!upload(&c.d_rules, rules.data(), rules.size()) || !upload(&c.d_off, off.data(), off.size()) || !upload(&d_fast, fast.data(), fast.size())) {
        if (c.d_rules) (void)hipFree(c.d_rules);
        if (c.d_off) (void)hipFree(c.d_off);
        if (d_fast) (void)hipFree(d_fast);
        return nullptr;
    }
    c.d_fast = d_fast;
    return &h->compact.emplace(std::move(key), c).first->second;
}

// rows from which a batch goes to the wave-per-placement kernels (CEG_HIP_MC_WAVE_MIN overrides; 0 = always, a huge value = never)
int64_t wave_kernel_min_rows()
{
    if (const char* e = std::getenv("CEG_HIP_MC_WAVE_MIN")) return std::atoll(e);
    return 1024;        // measured cross-over (64 CO2 in CHA, 1368 k-vectors): 512 rows 40 vs 47 us, 1024 rows 55 vs 50 us, 2048 rows 77 vs 62 us, 65 536 rows 1520 vs 545 us
}

// placements per wave: amortise what a workgroup stages, but keep every CU busy (>= ~2048 workgroups when the batch allows it)
int rows_per_wave(int64_t rows, int waves, int most)
{
    int per_wave = most;
    while (per_wave > 1 && rows / ((int64_t)per_wave * waves) < 2048) per_wave >>= 1;
    return per_wave;
}

// large batch: one wave per placement, one launch per term (k_mcw_frame, k_mcw_ewald, k_mcw_pairs)
int launch_wave_kernels(ceg_mc* h, bool insert, int32_t molecule, const McMolecule& nm, int m, const double* d_in, int64_t rows, double* d_out)
{
    const McView& v = h->v;
    const int32_t* kinds = insert ? nm.kinds : h->h_kind.data() + h->h_mol[molecule].x;
    const ceg_mc::Compact* ctab = compact_table(h, kinds, m);
    if (!ctab) return merr(CEG_ERR_HIP, "could not upload the pair-table rows of the molecule");
    {   // framework_interactions
        const int per_wave = rows_per_wave(rows, MCW_WAVES, MCW_FRAME_ROWS);
        const int64_t nb = (rows + (int64_t)MCW_WAVES * per_wave - 1) / ((int64_t)MCW_WAVES * per_wave);
        if (insert) hipLaunchKernelGGL(k_mcw_frame<true>, dim3((unsigned)nb), dim3(64 * MCW_WAVES), 0, h->stream, v, molecule, nm, d_in, rows, d_out, per_wave);
        else hipLaunchKernelGGL(k_mcw_frame<false>, dim3((unsigned)nb), dim3(64 * MCW_WAVES), 0, h->stream, v, molecule, nm, d_in, rows, d_out, per_wave);
    }
    if (v.nk > 0) {   // single_contribution_ewald
        const size_t c_bytes = sizeof(double) * 3 * (size_t)v.ns * 64 + sizeof(int32_t) * ((((size_t)v.nrounds * 64) + 3) & ~(size_t)3);
        int waves = 8;
        while (waves > 1 && c_bytes + (size_t)waves * tables_bytes(h, m) > 72 * 1024) waves >>= 1;
        const size_t lds = c_bytes + (size_t)waves * tables_bytes(h, m);
        if (lds > 150 * 1024) return merr(CEG_ERR_UNSUPPORTED, "k-space tables do not fit in LDS");
        const int per_wave = rows_per_wave(rows, waves, 8);
        const int64_t nb = (rows + (int64_t)waves * per_wave - 1) / ((int64_t)waves * per_wave);
#define CEG_MCW_E(I, W) hipLaunchKernelGGL((k_mcw_ewald<I, W>), dim3((unsigned)nb), dim3(64 * W), lds, h->stream, v, molecule, nm, d_in, rows, d_out, h->stride, per_wave)
#define CEG_MCW_EW(I)                        \
    do {                                     \
        if (waves == 8) CEG_MCW_E(I, 8);     \
        else if (waves == 4) CEG_MCW_E(I, 4); \
        else if (waves == 2) CEG_MCW_E(I, 2); \
        else CEG_MCW_E(I, 1);                \
    } while (0)
        if (insert) CEG_MCW_EW(true);
        else CEG_MCW_EW(false);
#undef CEG_MCW_EW
#undef CEG_MCW_E
    } else {
        // no Ewald summation: column 3 is zero
        if (hipMemset2DAsync(d_out + 3, 4 * sizeof(double), 0, sizeof(double), (size_t)rows, h->stream) != hipSuccess) return merr(CEG_ERR_HIP, "memset failed");
    }
    // single_contribution_vdw
    bool frac_pairs = v.fast && v.fastwrap >= 1 && v.natoms > 0 && v.natoms < (1 << 27) &&
                      (!v.use_cells || (int64_t)v.nb[0] * v.nb[1] * v.nb[2] * v.cell_cap < (1 << 27));      // (atom index << 4 | trial atom in 32 bits)
    if (const char* e = std::getenv("CEG_HIP_MC_FRAC")) frac_pairs = frac_pairs && std::atoi(e) != 0;          // measurement aid: 0 = the Cartesian kernel
    const int nentries = v.nkinds * m;
    const size_t ftab_bytes = ceg_pairfrac::frac_table_bytes(nentries, ctab->nrules, h->eni);
    if (ftab_bytes + sizeof(ceg_pairfrac::FracHit) * ceg_pairfrac::FQCAP * MCW_WAVES + 4096 > 60 * 1024) frac_pairs = false;
    if (frac_pairs) {
        const ceg_pairfrac::FracTable ftab{static_cast<const McFastPair*>(ctab->d_fast), ctab->d_rules, ctab->d_off, ctab->nrules, nentries, h->d_etab, h->ebase, h->eni};
        const int per_wave = rows_per_wave(rows, MCW_WAVES, 8);
        const int64_t nb = (rows + (int64_t)MCW_WAVES * per_wave - 1) / ((int64_t)MCW_WAVES * per_wave);
#define CEG_MCW_F(MMv, CL, TR) hipLaunchKernelGGL((k_mcw_pairs_frac<MMv, CL, TR>), dim3((unsigned)nb), dim3(64 * MCW_WAVES), ftab_bytes, h->stream, v, ftab, molecule, insert ? 1 : 0, nm, d_in, rows, d_out, per_wave)
#define CEG_MCW_FM(CL, TR)                         \
    do {                                           \
        switch (m) {                               \
            case 1: CEG_MCW_F(1, CL, TR); break;   \
            case 2: CEG_MCW_F(2, CL, TR); break;   \
            case 3: CEG_MCW_F(3, CL, TR); break;   \
            case 4: CEG_MCW_F(4, CL, TR); break;   \
            default: CEG_MCW_F(0, CL, TR); break;  \
        }                                          \
    } while (0)
        if (v.use_cells) { if (v.fastwrap == 2) CEG_MCW_FM(true, true); else CEG_MCW_FM(true, false); }
        else { if (v.fastwrap == 2) CEG_MCW_FM(false, true); else CEG_MCW_FM(false, false); }
#undef CEG_MCW_FM
#undef CEG_MCW_F
    } else {
        const size_t ct_full = sizeof(McFastPair) * (size_t)v.nkinds * m + sizeof(DevRule) * (size_t)std::max(ctab->nrules, 1) +
                               sizeof(int32_t) * ((size_t)v.nkinds * m + 1);
        McCompact ct{ctab->d_rules, ctab->d_off, static_cast<const McFastPair*>(ctab->d_fast), ctab->nrules, ct_full <= 32 * 1024 ? 1 : 0};
        const size_t lds = ct.in_lds ? ct_full : 0;
        const int per_wave = rows_per_wave(rows, MCW_WAVES, 4);
        const int64_t nb = (rows + (int64_t)MCW_WAVES * per_wave - 1) / ((int64_t)MCW_WAVES * per_wave);
#define CEG_MCW_P(F, CL, WR) hipLaunchKernelGGL((k_mcw_pairs<F, CL, WR>), dim3((unsigned)nb), dim3(64 * MCW_WAVES), lds, h->stream, v, ct, molecule, insert ? 1 : 0, nm, d_in, rows, d_out, per_wave)
#define CEG_MCW_WR(F, CL)                                  \
    do {                                                   \
        if (v.fastwrap == 2) CEG_MCW_P(F, CL, 2);          \
        else if (v.fastwrap == 1) CEG_MCW_P(F, CL, 1);     \
        else CEG_MCW_P(F, CL, 0);                          \
    } while (0)
#define CEG_MCW_PICK(CL)                                                                      \
    do {                                                                                      \
        if (v.fast) CEG_MCW_WR(true, CL); else CEG_MCW_WR(false, CL);                         \
    } while (0)
        if (v.use_cells) CEG_MCW_PICK(true);
        else CEG_MCW_PICK(false);
#undef CEG_MCW_PICK
#undef CEG_MCW_WR
#undef CEG_MCW_P
    }
    return hipGetLastError() == hipSuccess ? CEG_OK : merr(CEG_ERR_HIP, "trial kernel launch failed");
}

// launch the trial kernel for n placements (INSERT: of a molecule that is not in the system) and bring the rows back
int run_trial(ceg_mc* h, bool insert, int32_t molecule, const McMolecule& nm, int m, const double* trial, int64_t n, double* out)
{
    const int64_t rows = insert ? n : n + 1;
    if (rows <= 0) return CEG_OK;
    if (rows > 0x7fffffffLL) return merr(CEG_ERR_INVALID, "too many placements");
    if (tables_bytes(h, m) > 64 * 1024) return merr(CEG_ERR_UNSUPPORTED, "k-space tables of the molecule do not fit in LDS");
    const size_t in_bytes = sizeof(double) * 3 * (size_t)m * (size_t)n, out_bytes = sizeof(double) * 4 * (size_t)rows;
    Guard guard(h->device);
    if (!guard.ok) return merr(CEG_ERR_HIP, "hipSetDevice failed");
    const bool mapped = in_bytes <= MC_MAPPED_BYTES && out_bytes <= MC_MAPPED_BYTES;
    const double* d_in;
    double* d_out;
    if (mapped) {
        if (n > 0) memcpy(h->h_in, trial, in_bytes);
        d_in = h->dm_in;
        d_out = h->dm_out;
    } else {
        if (in_bytes > h->d_in_cap) {
            if (h->d_in) (void)hipFree(h->d_in);
            h->d_in = nullptr; h->d_in_cap = 0;
            if (hipMalloc((void**)&h->d_in, in_bytes) != hipSuccess) return merr(CEG_ERR_HIP, "hipMalloc failed");
            h->d_in_cap = in_bytes;
        }
        if (out_bytes > h->d_out_cap) {
            if (h->d_out) (void)hipFree(h->d_out);
            h->d_out = nullptr; h->d_out_cap = 0;
            if (hipMalloc((void**)&h->d_out, out_bytes) != hipSuccess) return merr(CEG_ERR_HIP, "hipMalloc failed");
            h->d_out_cap = out_bytes;
        }
        if (hipMemcpyAsync(h->d_in, trial, in_bytes, hipMemcpyHostToDevice, h->stream) != hipSuccess) return merr(CEG_ERR_HIP, "H2D failed");
        d_in = h->d_in;
        d_out = h->d_out;
    }
    if (rows >= wave_kernel_min_rows()) {
        if (int rc = launch_wave_kernels(h, insert, molecule, nm, m, d_in, rows, d_out)) return rc;
        if (!mapped && hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return merr(CEG_ERR_HIP, "D2H failed");
        if (hipStreamSynchronize(h->stream) != hipSuccess) return merr(CEG_ERR_HIP, "trial kernel failed");
        if (mapped) memcpy(out, h->h_out, out_bytes);
        return CEG_OK;
    }
    McView v = h->v;
    size_t table_bytes = v.table_in_lds ? sizeof(DevRule) * (size_t)(v.nrules > 0 ? v.nrules : 1) + sizeof(int32_t) * ((size_t)v.nkinds * v.nkinds + 1) : 0;
    if (tables_bytes(h, m) + table_bytes > 64 * 1024) { v.table_in_lds = 0; table_bytes = 0; }   // pair table from global memory then
    const size_t lds = tables_bytes(h, m) + table_bytes;
    // small batches: the three terms of a row on three workgroups (latency = the longest term); CEG_HIP_MC_SPLIT_MAX moves the limit (0: never)
    int64_t split_max = 256;           // measured (64 CO2 in CHA): 1 row 32 -> 24 us per call, 64 rows 36 -> 29, 256 rows 38 = 38, 512 rows 40 -> 46
    if (const char* e = std::getenv("CEG_HIP_MC_SPLIT_MAX")) split_max = std::atoll(e);
    const dim3 grid((unsigned)rows, rows <= split_max ? 3u : 1u), block(MC_THREADS);
    unsigned long long* flag = nullptr;
    // (only for the latency-bound small batches: with ~1000 workgroups the fences and the shared counter cost more than the wake-up)
    if (mapped && rows <= 64 && !getenv("CEG_HIP_MC_NO_POLL")) { flag = h->dm_flag; ++h->seq; }
    McLocal L{};
    L.first = insert ? 0 : h->h_mol[molecule].x;
    L.m = m;
    for (int a = 0; a < m; ++a) {
        L.kinds[a] = insert ? nm.kinds[a] : h->h_kind[(size_t)L.first + a];
        L.q[a] = h->h_charge[(size_t)L.kinds[a]];
    }
#define CEG_MC_LAUNCH(F, I, CL) hipLaunchKernelGGL((k_mc_trial<F, I, CL>), grid, block, lds, h->stream, v, molecule, L, d_in, n, d_out, h->stride, h->d_done, flag, h->seq)
#define CEG_MC_PICK(CL)                                                                       \
    do {                                                                                      \
        if (insert) { if (v.fast) CEG_MC_LAUNCH(true, true, CL); else CEG_MC_LAUNCH(false, true, CL); }   \
        else { if (v.fast) CEG_MC_LAUNCH(true, false, CL); else CEG_MC_LAUNCH(false, false, CL); }        \
    } while (0)
    if (v.use_cells) CEG_MC_PICK(true);
    else CEG_MC_PICK(false);
#undef CEG_MC_PICK
#undef CEG_MC_LAUNCH
    if (hipGetLastError() != hipSuccess) return merr(CEG_ERR_HIP, "trial kernel launch failed");
    if (!mapped && hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return merr(CEG_ERR_HIP, "D2H failed");
    bool seen = false;
    if (flag) {                      // poll the completion flag; after ~20 ms without it fall back to the stream (a failed launch
                                     // never raises the flag, and hipStreamSynchronize is what reports the error)
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0;; ++spin) {
            if (__atomic_load_n(h->h_flag, __ATOMIC_ACQUIRE) == h->seq) { seen = true; break; }
            if ((spin & 1023u) == 1023u &&
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 20.0) break;
            __builtin_ia32_pause();
        }
    }
    if (!seen && hipStreamSynchronize(h->stream) != hipSuccess) return merr(CEG_ERR_HIP, "trial kernel failed");
    if (mapped) memcpy(out, h->h_out, out_bytes);
    return CEG_OK;
}

int check_molecule(const ceg_mc* h, const int32_t* kinds, int32_t m, McMolecule* nm)
{
    if (!kinds || m < 1) return merr(CEG_ERR_INVALID, "bad argument");
    if (m > MC_MAX_ATOMS) return merr(CEG_ERR_UNSUPPORTED, "molecule has more atoms than the kernels hold in LDS (16)");
    nm->m = m;
    for (int a = 0; a < m; ++a) {
        if (kinds[a] < 0 || kinds[a] >= h->v.nkinds) return merr(CEG_ERR_INVALID, "atom kind outside the pair table");
        nm->kinds[a] = kinds[a];
    }
    return CEG_OK;
}

}  // namespace

extern "C" int ceg_mc_set_guests(ceg_mc_t* h, const double* positions, const int32_t* kinds, const int32_t* mol_first, int32_t nmol)
{
    if (!h || nmol < 0 || !mol_first || mol_first[0] != 0) return merr(CEG_ERR_INVALID, "bad argument");
    int max_m = 1;
    for (int32_t j = 0; j < nmol; ++j) {
        const int32_t m = mol_first[j + 1] - mol_first[j];
        if (m < 1) return merr(CEG_ERR_INVALID, "empty molecule");
        if (m > MC_MAX_ATOMS) return merr(CEG_ERR_UNSUPPORTED, "molecule has more atoms than the kernels hold in LDS (16)");
        max_m = std::max(max_m, m);
    }
    if (tables_bytes(h, max_m) > 64 * 1024) return merr(CEG_ERR_UNSUPPORTED, "k-space tables of a molecule do not fit in LDS");
    const int64_t natoms = mol_first[nmol];
    if (natoms > 0 && (!positions || !kinds)) return merr(CEG_ERR_INVALID, "bad argument");
    std::vector<double4> host((size_t)(natoms > 0 ? natoms : 1));
    std::vector<int2> idx((size_t)(nmol > 0 ? nmol : 1));
    for (int32_t j = 0; j < nmol; ++j) {
        idx[j] = make_int2(mol_first[j], mol_first[j + 1] - mol_first[j]);
        for (int32_t l = mol_first[j]; l < mol_first[j + 1]; ++l) {
            if (kinds[l] < 0 || kinds[l] >= h->v.nkinds) return merr(CEG_ERR_INVALID, "atom kind outside the pair table");
            const long long bits = ((long long)j << 32) | (long long)(uint32_t)kinds[l];
            double w;
            memcpy(&w, &bits, sizeof(w));
            host[l] = make_double4(positions[3 * l], positions[3 * l + 1], positions[3 * l + 2], w);
        }
    }
    Guard guard(h->device);
    if (!guard.ok) return merr(CEG_ERR_HIP, "hipSetDevice failed");
    if (hipStreamSynchronize(h->stream) != hipSuccess) return poison(h, merr(CEG_ERR_HIP, "stream synchronisation failed"));
    h->poisoned = true;                                    // until this call has rebuilt host and device state completely
    h->v.natoms = 0; h->v.nmol = 0;                        // nothing worth copying when the arrays grow
    h->h_mol.clear();
    if (int rc = ensure_capacity(h, std::max<int64_t>(natoms, 1), std::max<int64_t>(nmol, 1))) return rc;
    bool ok = !injected_failure("set_guests");
    if (ok && natoms > 0) ok = hipMemcpy(h->d_atoms, host.data(), sizeof(double4) * (size_t)natoms, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && nmol > 0) ok = hipMemcpy(h->d_molidx, idx.data(), sizeof(int2) * (size_t)nmol, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) return merr(CEG_ERR_HIP, "could not upload the guest atoms");
    h->h_mol.assign(idx.begin(), idx.begin() + nmol);
    h->h_kind.assign(kinds, kinds + natoms);
    h->free_runs.assign(MC_MAX_ATOMS + 1, {});
    McView& v = h->v;
    v.natoms = (int32_t)natoms; v.nmol = nmol;
    if (natoms > 0) {
        hipLaunchKernelGGL(k_mc_frac_fill, dim3((unsigned)((natoms + 255) / 256)), dim3(256), 0, h->stream, v, (int64_t)natoms);
        if (hipGetLastError() != hipSuccess) return merr(CEG_ERR_HIP, "could not convert the guest atoms");
    }
    if (h->cm.on) {
        CellMirror& cm = h->cm;
        cm.members.assign((size_t)cm.ncells(), {});
        cm.cell_of.assign((size_t)natoms, -1);
        cm.idx_of.assign((size_t)natoms, -1);
        cm.max_fill = 0;
        cm.begin();
        for (int64_t l = 0; l < natoms; ++l) cm.put_in((int32_t)l, cm.bin_of(positions + 3 * l));
        if (int rc = rebuild_cells(h)) return rc;
    }
    if (v.nk > 0) {
        if (nmol > 0)
            hipLaunchKernelGGL(k_mc_sf_molecules, dim3((unsigned)nmol), dim3(MC_THREADS), tables_bytes(h, max_m), h->stream, v, h->stride);
        hipLaunchKernelGGL(k_mc_sf_total, dim3((unsigned)((v.nk + 255) / 256)), dim3(256), 0, h->stream, v);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
            return merr(CEG_ERR_HIP, "structure-factor kernels failed");
    }
    h->poisoned = false;
    return CEG_OK;
}

namespace {
// trial placements and result rows in DEVICE memory, enqueued on the caller's stream (after what this handle has enqueued so far):
// the wave-per-placement kernels whatever the batch size -- no host copies, no synchronisation
int run_trial_device(ceg_mc* h, bool insert, int32_t molecule, const McMolecule& nm, int m, const double* d_trial, int64_t n, double* d_out, void* stream)
{
    const int64_t rows = insert ? n : n + 1;
    if (rows <= 0) return CEG_OK;
    if (rows > 0x7fffffffLL) return merr(CEG_ERR_INVALID, "too many placements");
    if (tables_bytes(h, m) > 64 * 1024) return merr(CEG_ERR_UNSUPPORTED, "k-space tables of the molecule do not fit in LDS");
    Guard guard(h->device);
    if (!guard.ok) return merr(CEG_ERR_HIP, "hipSetDevice failed");
    hipStream_t user = (hipStream_t)stream;
    // order the caller's stream behind the handle's own (accept / insert / remove are asynchronous on it), run there, and make the
    // handle's stream wait for the trial in turn so that a later accept does not overtake it
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return merr(CEG_ERR_HIP, "event creation failed");
    bool ok = hipEventRecord(ev, h->stream) == hipSuccess && hipStreamWaitEvent(user, ev, 0) == hipSuccess;
    int rc = CEG_OK;
    if (ok) {
        hipStream_t own = h->stream;
        h->stream = user;
        rc = launch_wave_kernels(h, insert, molecule, nm, m, d_trial, rows, d_out);
        h->stream = own;
    }
    ok = ok && !rc && hipEventRecord(ev, user) == hipSuccess && hipStreamWaitEvent(h->stream, ev, 0) == hipSuccess;
    (void)hipEventDestroy(ev);
    if (rc) return rc;
    return ok ? CEG_OK : merr(CEG_ERR_HIP, "stream ordering failed");
}
}  // namespace

extern "C" int ceg_mc_trial_device(ceg_mc_t* h, int32_t molecule, const double* d_trial, int64_t n, double* d_out, void* stream)
{
    if (!h || n < 0 || !d_out || (n > 0 && !d_trial)) return merr(CEG_ERR_INVALID, "bad argument");
    if (h->poisoned) return refuse_poisoned();
    if (molecule < 0 || molecule >= h->v.nmol) return merr(CEG_ERR_INVALID, "no such molecule");
    return run_trial_device(h, false, molecule, McMolecule{}, h->h_mol[molecule].y, d_trial, n, d_out, stream);
}

extern "C" int ceg_mc_trial_insert_device(ceg_mc_t* h, const int32_t* kinds, int32_t m, const double* d_trial, int64_t n, double* d_out, void* stream)
{
    if (!h || n < 0 || (n > 0 && (!d_trial || !d_out))) return merr(CEG_ERR_INVALID, "bad argument");
    if (h->poisoned) return refuse_poisoned();
    McMolecule nm{};
    if (int rc = check_molecule(h, kinds, m, &nm)) return rc;
    return run_trial_device(h, true, -1, nm, m, d_trial, n, d_out, stream);
}

extern "C" int ceg_mc_trial(ceg_mc_t* h, int32_t molecule, const double* trial, int64_t n, double* out)
{
    if (!h || n < 0 || !out || (n > 0 && !trial)) return merr(CEG_ERR_INVALID, "bad argument");
    if (h->poisoned) return refuse_poisoned();
    if (molecule < 0 || molecule >= h->v.nmol) return merr(CEG_ERR_INVALID, "no such molecule");
    return run_trial(h, false, molecule, McMolecule{}, h->h_mol[molecule].y, trial, n, out);
}

extern "C" int ceg_mc_trial_insert(ceg_mc_t* h, const int32_t* kinds, int32_t m, const double* trial, int64_t n, double* out)
{
    if (!h || n < 0 || (n > 0 && (!trial || !out))) return merr(CEG_ERR_INVALID, "bad argument");
    if (h->poisoned) return refuse_poisoned();
    McMolecule nm{};
    if (int rc = check_molecule(h, kinds, m, &nm)) return rc;
    return run_trial(h, true, -1, nm, m, trial, n, out);
}

extern "C" int ceg_mc_accept(ceg_mc_t* h, int32_t molecule, const double* positions)
{
    if (!h || !positions) return merr(CEG_ERR_INVALID, "bad argument");
    if (h->poisoned) return refuse_poisoned();
    if (molecule < 0 || molecule >= h->v.nmol) return merr(CEG_ERR_INVALID, "no such molecule");
    const int m = h->h_mol[molecule].y;
    McPositions np{};
    for (int t = 0; t < 3 * m; ++t) np.xyz[t] = positions[t];
    Guard guard(h->device);
    if (!guard.ok) return merr(CEG_ERR_HIP, "hipSetDevice failed");
    McCellOps ops{};
    bool rebuild = false;
    if (h->cm.on) {
        CellMirror& cm = h->cm;
        const int first = h->h_mol[molecule].x;
        cm.begin();
        for (int a = 0; a < m; ++a) {
            const int c = cm.bin_of(positions + 3 * a);
            if (c == cm.cell_of[first + a]) { cm.refresh(first + a); continue; }     // same cell: new coordinates in place
            cm.take_out(first + a);
            cm.put_in(first + a, c);
        }
        rebuild = !cm.finish(ops);
    }
    // (from here on the cell mirror already describes the accepted state: any failure leaves host and device out of step)
    if (!injected_failure("accept"))
        hipLaunchKernelGGL(k_mc_accept, dim3(1), dim3(MC_THREADS), tables_bytes(h, m), h->stream, h->v, molecule, np, ops, h->stride);
    if (injected_failure("accept") || hipGetLastError() != hipSuccess) return poison(h, merr(CEG_ERR_HIP, "accept kernel launch failed"));
    if (rebuild) { if (int rc = rebuild_cells(h)) return poison(h, rc); }
    return CEG_OK;                      // asynchronous: the next call on this handle is ordered behind it
}

extern "C" int ceg_mc_insert(ceg_mc_t* h, const int32_t* kinds, int32_t m, const double* positions, int32_t* molecule_out)
{
    if (!h || !positions) return merr(CEG_ERR_INVALID, "bad argument");
    if (h->poisoned) return refuse_poisoned();
    McMolecule nm{};
    if (int rc = check_molecule(h, kinds, m, &nm)) return rc;
    if (tables_bytes(h, m) > 64 * 1024) return merr(CEG_ERR_UNSUPPORTED, "k-space tables of the molecule do not fit in LDS");
    Guard guard(h->device);
    if (!guard.ok) return merr(CEG_ERR_HIP, "hipSetDevice failed");
    if (h->free_runs.empty()) h->free_runs.assign(MC_MAX_ATOMS + 1, {});
    int32_t first;
    // capacity first (it may fail and has changed nothing yet), the host mirror afterwards
    const bool reuse = !h->free_runs[m].empty();  // the slots of a removed molecule of the same size
    if (int rc = ensure_capacity(h, (int64_t)h->v.natoms + (reuse ? 0 : m), (int64_t)h->v.nmol + 1)) return poison(h, rc);
    if (reuse) {
        first = h->free_runs[m].back();
        h->free_runs[m].pop_back();
    } else {
        first = h->v.natoms;
        h->v.natoms += m;
    }
    const int32_t molecule = h->v.nmol;
    McPositions np{};
    for (int t = 0; t < 3 * m; ++t) np.xyz[t] = positions[t];
    h->v.nmol += 1;
    h->h_mol.push_back(make_int2(first, m));
    if ((int64_t)h->h_kind.size() < (int64_t)first + m) h->h_kind.resize((size_t)first + m, 0);
    for (int a = 0; a < m; ++a) h->h_kind[(size_t)first + a] = nm.kinds[a];
    McCellOps ops{};
    bool rebuild = false;
    if (h->cm.on) {
        CellMirror& cm = h->cm;
        cm.begin();
        for (int a = 0; a < m; ++a) cm.put_in(first + a, cm.bin_of(positions + 3 * a));
        rebuild = !cm.finish(ops);
    }
    if (!injected_failure("insert"))
        hipLaunchKernelGGL(k_mc_insert, dim3(1), dim3(MC_THREADS), tables_bytes(h, m), h->stream, h->v, molecule, first, nm, np, ops, h->stride);
    if (injected_failure("insert") || hipGetLastError() != hipSuccess) return poison(h, merr(CEG_ERR_HIP, "insert kernel launch failed"));
    if (molecule_out) *molecule_out = molecule;
    if (rebuild) { if (int rc = rebuild_cells(h)) return poison(h, rc); }
    return CEG_OK;
}

extern "C" int ceg_mc_remove(ceg_mc_t* h, int32_t molecule, int32_t* moved_out)
{
    if (!h) return merr(CEG_ERR_INVALID, "bad argument");
    if (h->poisoned) return refuse_poisoned();
    if (molecule < 0 || molecule >= h->v.nmol) return merr(CEG_ERR_INVALID, "no such molecule");
    Guard guard(h->device);
    if (!guard.ok) return merr(CEG_ERR_HIP, "hipSetDevice failed");
    const int32_t last = h->v.nmol - 1;
    McCellOps ops{};
    bool rebuild = false;
    if (h->cm.on) {
        CellMirror& cm = h->cm;
        cm.begin();
        for (int a = 0; a < h->h_mol[molecule].y; ++a) cm.take_out(h->h_mol[molecule].x + a);
        if (last != molecule)                                  // the records of the renumbered molecule carry its new index
            for (int a = 0; a < h->h_mol[last].y; ++a) cm.refresh(h->h_mol[last].x + a);
        rebuild = !cm.finish(ops);
    }
    if (!injected_failure("remove"))
        hipLaunchKernelGGL(k_mc_remove, dim3(1), dim3(MC_THREADS), 0, h->stream, h->v, molecule, last, ops);
    if (injected_failure("remove") || hipGetLastError() != hipSuccess) return poison(h, merr(CEG_ERR_HIP, "remove kernel launch failed"));
    if (rebuild) { if (int rc = rebuild_cells(h)) return poison(h, rc); }
    if (h->free_runs.empty()) h->free_runs.assign(MC_MAX_ATOMS + 1, {});
    h->free_runs[h->h_mol[molecule].y].push_back(h->h_mol[molecule].x);
    if (last != molecule) h->h_mol[molecule] = h->h_mol[last];
    h->h_mol.pop_back();
    h->v.nmol = last;
    if (moved_out) *moved_out = last;     // like remove_one_system! (ewald.jl:404-413): the molecule that was `last` is now `molecule`
    return CEG_OK;
}

extern "C" int ceg_mc_neighbour_cells(ceg_mc_t* h, int32_t nb[3], int32_t* capacity)
{
    if (!h) return merr(CEG_ERR_INVALID, "bad argument");
    for (int i = 0; i < 3; ++i)
        if (nb) nb[i] = h->cm.on ? h->cm.nb[i] : 0;
    if (capacity) *capacity = h->cm.on ? h->cm.cap : 0;
    return h->cm.on ? 1 : 0;
}

extern "C" int ceg_mc_get_state(ceg_mc_t* h, double* positions, double* sf_total_re, double* sf_total_im)
{
    if (!h) return merr(CEG_ERR_INVALID, "bad argument");
    if (h->poisoned) return refuse_poisoned();
    Guard guard(h->device);
    if (!guard.ok) return merr(CEG_ERR_HIP, "hipSetDevice failed");
    if (hipStreamSynchronize(h->stream) != hipSuccess) return merr(CEG_ERR_HIP, "stream synchronisation failed");
    if (positions && h->v.natoms > 0) {          // molecule order: atoms of molecule 0, then 1, ...
        std::vector<double4> host((size_t)h->v.natoms);
        if (hipMemcpy(host.data(), h->d_atoms, sizeof(double4) * host.size(), hipMemcpyDeviceToHost) != hipSuccess) return merr(CEG_ERR_HIP, "D2H failed");
        size_t o = 0;
        for (const int2& mj : h->h_mol)
            for (int a = 0; a < mj.y; ++a, ++o) {
                positions[3 * o] = host[mj.x + a].x; positions[3 * o + 1] = host[mj.x + a].y; positions[3 * o + 2] = host[mj.x + a].z;
            }
    }
    if ((sf_total_re || sf_total_im) && h->v.nk > 0) {
        std::vector<double2> t((size_t)h->v.nk);
        if (hipMemcpy(t.data(), h->d_tot, sizeof(double2) * t.size(), hipMemcpyDeviceToHost) != hipSuccess) return merr(CEG_ERR_HIP, "D2H failed");
        for (size_t q = 0; q < t.size(); ++q) {
            if (sf_total_re) sf_total_re[q] = t[q].x;
            if (sf_total_im) sf_total_im[q] = t[q].y;
        }
    }
    return CEG_OK;
}
