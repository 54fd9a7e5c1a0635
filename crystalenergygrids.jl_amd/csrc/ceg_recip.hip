// ceg_recip.hip -- batched reciprocal-space Ewald energy of one rigid molecule:
// compute_ewald(ctx), src/ewald.jl:555-577, with the structure-factor loop of ewald_main_loop!
// (:148-185) and the e^{2 pi i m f} tables of setup_Eik / move_one_system! (:109-146, 352-366).
//
// One wave64 per placement of the molecule.  The wave first fills, in LDS, the three tables
// e^{2 pi i m f_x} (m = 0..kx), e^{2 pi i m f_y} (m = -ky..ky), e^{2 pi i m f_z} (m = -kz..kz) of
// every atom -- by sine and cosine of the exact angle (ceg_math.h sincos_2pi) instead of the reference's repeated complex
// multiplication (same values, smaller rounding error).  The k-vectors are then walked the way the
// reference stores them (kspace.kindices, src/ewald.jl:213-236): as ROWS (j, k) x (i = i0 .. i0+len-1).
// ceg_recip_create regroups the flat list into such rows, cuts them into segments of <= SEG k-vectors and
// deals the segments, longest first, to the 64 lanes; a lane forms Ey[j] Ez[k] q ONCE per segment and atom
// (the charge rides on the z table) and steps along i with Ex[i+1] = Ex[i] Ex[1] -- per k-vector and atom 8 FMAs and no LDS access, where the
// first version of this kernel (three per-lane table reads + a 32-byte constant record per k-vector) was
// bound by the LDS pipe at 0.19 of the FP64 peak.  The per-k-vector constants (kf Re S_f, kf Im S_f, kf)
// sit in [slot][lane] planes: conflict-free 8-byte reads, zero in the padding slots, so that a round runs
// to its longest segment without lane masks.  A wave reduction finishes the two sums.  No MFMA: complex
// products with per-lane operands, not a contraction with a shared operand.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ceg_hip.h"
#include "ceg_math.h"
#include "ceg_consumers.h"
#include "ceg_rows.h"

extern "C" void ceg_set_last_error_(const char* msg);

namespace {

constexpr int MAX_ATOMS = 16;       // atoms per molecule held in LDS
constexpr int MAX_TAB = 400;        // (kx+1) + (2ky+1) + (2kz+1) per atom
constexpr int MAX_WAVES = 8;        // placements in flight per workgroup (the constants in LDS are shared: two workgroups = 4 waves per SIMD);
                                    // fewer (4, 2, 1) when the tables of a large molecule / k-space would not fit
using ceg_rows::Layout;
using ceg_rows::choose_layout;

struct RecipGeom {
    double invmat[9];
    int32_t ks[3];
    int32_t natoms;
    double q[MAX_ATOMS];
    double energy_net_charges, static_contribution;
};

// g_desc[r * 64 + lane]: segment of `lane` in round r: i0 | (j + ky) << 9 | (k + kz) << 18 | L_r << 27 (nine bits each: MAX_TAB < 512; L_r = longest segment of the
// round, the same in every lane).  g_c: three planes [ns * 64] of the constants A = kf Re S_f, B = kf Im S_f, kf by (slot, lane):
// the slots of round r are the L_r following those of round r - 1.
template <bool C_IN_LDS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_recip(RecipGeom g, const int32_t* __restrict__ g_desc, const double* __restrict__ g_c,
                                                       int nrounds, int ns, const double* __restrict__ pos, int64_t n,
                                                       double* __restrict__ out, int tab_stride, int per_wave)
{
    // dynamic LDS: [C_IN_LDS: 3 ns 64 doubles, nrounds 64 int32 (padded to 16 B)], then [WAVES][natoms][tab_stride] double2
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const size_t cdoubles = C_IN_LDS ? (size_t)3 * ns * 64 : 0;
    const size_t dints = C_IN_LDS ? (((size_t)nrounds * 64 + 3) & ~(size_t)3) : 0;
    double* s_c = reinterpret_cast<double*>(s_raw);
    int32_t* s_desc = reinterpret_cast<int32_t*>(s_raw + cdoubles * sizeof(double));
    double2* s_tab = reinterpret_cast<double2*>(s_raw + cdoubles * sizeof(double) + dints * sizeof(int32_t));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (C_IN_LDS) {
        // every placement of the workgroup reads all constants: stage them once
        for (size_t t = threadIdx.x; t < cdoubles; t += 64 * WAVES) s_c[t] = g_c[t];
        for (int t = threadIdx.x; t < nrounds * 64; t += 64 * WAVES) s_desc[t] = g_desc[t];
        __syncthreads();
    }
    const double* cA = C_IN_LDS ? s_c : g_c;
    const double* cB = cA + (size_t)ns * 64;
    const double* ckf = cB + (size_t)ns * 64;
    const int32_t* desc = C_IN_LDS ? s_desc : g_desc;
    const int kx = g.ks[0], ky = g.ks[1], kz = g.ks[2];
    const int nxp = kx + 1, nyp = 2 * ky + 1, nzp = 2 * kz + 1;
    double2* tab = s_tab + (size_t)wave * g.natoms * tab_stride;
    const double* I = g.invmat;
    const int64_t p0 = ((int64_t)blockIdx.x * WAVES + wave) * per_wave;
    const int64_t p1 = p0 + per_wave < n ? p0 + per_wave : n;
    // the coordinates of a placement (3 natoms <= 48 doubles) are fetched by the lanes one placement ahead and handed over through LDS
    __shared__ double s_pos[MAX_WAVES][3 * MAX_ATOMS];
    __shared__ double s_q[MAX_ATOMS];
    if (threadIdx.x < MAX_ATOMS) s_q[threadIdx.x] = (int)threadIdx.x < g.natoms ? g.q[threadIdx.x] : 0.0;
    __syncthreads();
    const int nc = 3 * g.natoms;
    double next = (lane < nc && p0 < p1) ? pos[(size_t)p0 * nc + lane] : 0.0;
    for (int64_t p = p0; p < p1; ++p) {
        if (lane < nc) s_pos[wave][lane] = next;
        __builtin_amdgcn_wave_barrier();
        if (lane < nc && p + 1 < p1) next = pos[(size_t)(p + 1) * nc + lane];
        // ---- tables: entry t of atom a = exp(2 pi i m f), m and the axis decoded from t
        for (int t = lane; t < nxp + nyp + nzp; t += 64) {
            const int ax = t < nxp ? 0 : (t < nxp + nyp ? 1 : 2);
            const int m = ax == 0 ? t : (ax == 1 ? t - nxp - ky : t - nxp - nyp - kz);
            const double i0 = I[ax], i1 = I[ax + 3], i2 = I[ax + 6];
            for (int a = 0; a < g.natoms; ++a) {
                const double* r = s_pos[wave] + 3 * a;
                const double f = i0 * r[0] + i1 * r[1] + i2 * r[2];
                const double ff = f - rint(f);                   // exp(2 pi i m f) is periodic in f
                double sn, cs;
                ceg::sincos_2pi((double)m * ff, sn, cs);
                const double w = ax == 2 ? s_q[a] : 1.0;          // the charge rides on the z factor
                tab[a * tab_stride + t] = make_double2(w * cs, w * sn);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- rounds: one segment (j, k, i0 .. i0 + L - 1) per lane
        double fa = 0.0, aa = 0.0;
        int slot = 0;
        for (int r = 0; r < nrounds; ++r) {
            const int d = desc[r * 64 + lane];
            const int L = __builtin_amdgcn_readfirstlane(d >> 27);
            const int i0 = d & 0x1ff, jj = (d >> 9) & 0x1ff, kk = (d >> 18) & 0x1ff;
            const size_t at = (size_t)slot * 64 + lane;
            ceg_rows::round_dispatch(L, g.natoms, tab, tab_stride, nxp, nyp, i0, jj, kk, [&](int s, double sr, double si) {
                const size_t idx = at + (size_t)s * 64;
                fa += cA[idx] * sr + cB[idx] * si;
                aa += ckf[idx] * (sr * sr + si * si);
            });
            slot += L;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            fa += __shfl_xor(fa, o);
            aa += __shfl_xor(aa, o);
        }
        if (lane == 0) out[p] = 2.0 * (fa + g.energy_net_charges) + (aa + g.static_contribution);
        __builtin_amdgcn_wave_barrier();                         // the tables are rewritten for the next placement
    }
}

int rerr(int code, const char* msg)
{
    ceg_set_last_error_(msg);
    return code;
}

}  // namespace

struct ceg_recip {
    int device = 0;
    int64_t nk = 0;
    int32_t ks[3] = {0, 0, 0};
    double invmat[9];
    // the k-vectors regrouped into rows (j, k) x (i0 .. i0 + len - 1), cut into segments and dealt to the lanes (see the kernel)
    int nrounds = 0, ns = 0;
    std::vector<int64_t> slot_of;       // plane index (slot * 64 + lane) of k-vector q
    std::vector<double> h_kf;
    int32_t* d_desc = nullptr;
    double* d_c = nullptr;              // planes A, B, kf: [3][ns * 64]
    ceg_consumers::HostIo io;           // ceg_recip_energy
};

namespace {

int upload_constants(ceg_recip* h, const double* sf_re, const double* sf_im)
{
    const size_t plane = (size_t)h->ns * 64;
    std::vector<double> c(3 * std::max<size_t>(plane, 1), 0.0);
    for (int64_t q = 0; q < h->nk; ++q) {
        const double t = h->h_kf[(size_t)q];
        const size_t at = (size_t)h->slot_of[(size_t)q];
        c[at] = t * sf_re[q];
        c[plane + at] = t * sf_im[q];
        c[2 * plane + at] = t;
    }
    return hipMemcpy(h->d_c, c.data(), 3 * plane * sizeof(double), hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}

}  // namespace

static int check_kspace(const int32_t* kvec_ijk, int64_t nk, const int32_t ks[3])
{
    if (ks[0] < 0 || ks[1] < 0 || ks[2] < 0 || ks[0] + 1 + 2 * ks[1] + 1 + 2 * ks[2] + 1 > MAX_TAB)
        return rerr(CEG_ERR_UNSUPPORTED, "k-space box too large for the LDS tables");
    for (int64_t q = 0; q < nk; ++q)
        if (kvec_ijk[3 * q] < 0 || kvec_ijk[3 * q] > ks[0] || abs(kvec_ijk[3 * q + 1]) > ks[1] || abs(kvec_ijk[3 * q + 2]) > ks[2])
            return rerr(CEG_ERR_INVALID, "k-vector outside the (kx, ky, kz) box");
    return CEG_OK;
}

extern "C" int ceg_recip_layout(const int32_t* kvec_ijk, int64_t nk, const int32_t ks[3], int32_t* nrounds, int32_t* nslots,
                                int64_t* slot_of, int32_t* desc)
{
    if (!ks || nk < 0 || (nk > 0 && !kvec_ijk) || !nrounds || !nslots) return rerr(CEG_ERR_INVALID, "bad argument");
    if (int rc = check_kspace(kvec_ijk, nk, ks)) return rc;
    const Layout l = choose_layout(kvec_ijk, nk, ks);
    *nrounds = l.nrounds;
    *nslots = l.ns;
    if (slot_of) std::copy(l.slot_of.begin(), l.slot_of.end(), slot_of);
    if (desc) std::copy(l.desc.begin(), l.desc.end(), desc);
    return CEG_OK;
}

extern "C" int ceg_recip_create(ceg_recip_t** handle, int32_t device, const int32_t* kvec_ijk, const double* kfactors,
                                const double* sf_re, const double* sf_im, int64_t nk, const int32_t ks[3],
                                const double invmat[9])
{
    if (!handle || !ks || !invmat || nk < 0 || (nk > 0 && (!kvec_ijk || !kfactors || !sf_re || !sf_im)))
        return rerr(CEG_ERR_INVALID, "bad argument");
    *handle = nullptr;
    if (int rc = check_kspace(kvec_ijk, nk, ks)) return rc;
    if (ceg_device_count() <= 0) return rerr(CEG_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ceg_device_count()) return rerr(CEG_ERR_NO_DEVICE, "device not present");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return rerr(CEG_ERR_HIP, "hipSetDevice failed");
    ceg_recip* h = new ceg_recip();
    h->device = device;
    h->nk = nk;
    for (int a = 0; a < 3; ++a) h->ks[a] = ks[a];
    for (int a = 0; a < 9; ++a) h->invmat[a] = invmat[a];
    Layout best = choose_layout(kvec_ijk, nk, ks);
    h->nrounds = best.nrounds;
    h->ns = best.ns;
    h->slot_of = std::move(best.slot_of);
    h->h_kf.assign(kfactors, kfactors + nk);
    const size_t plane = std::max<size_t>((size_t)h->ns * 64, 1), nd = std::max<size_t>((size_t)h->nrounds * 64, 1);
    bool ok = hipMalloc((void**)&h->d_desc, nd * sizeof(int32_t)) == hipSuccess &&
              hipMalloc((void**)&h->d_c, 3 * plane * sizeof(double)) == hipSuccess;
    if (ok && nk > 0)
        ok = hipMemcpy(h->d_desc, best.desc.data(), best.desc.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess &&
             upload_constants(h, sf_re, sf_im) == 0;
    if (prev >= 0) (void)hipSetDevice(prev);
    if (!ok) {
        ceg_recip_destroy(h);
        return rerr(CEG_ERR_HIP, "could not upload the k-space tables");
    }
    *handle = h;
    return CEG_OK;
}

extern "C" int ceg_recip_set_structure_factor(ceg_recip_t* h, const double* sf_re, const double* sf_im)
{
    if (!h || (h->nk > 0 && (!sf_re || !sf_im))) return rerr(CEG_ERR_INVALID, "bad argument");
    if (h->nk == 0) return CEG_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(h->device) != hipSuccess) return rerr(CEG_ERR_HIP, "hipSetDevice failed");
    // hipMemcpy from pageable memory is synchronous with respect to earlier work on the null stream;
    // launches on other streams must be ordered by the caller (documented in INTEGRATION.md)
    const bool ok = upload_constants(h, sf_re, sf_im) == 0;
    if (prev >= 0) (void)hipSetDevice(prev);
    return ok ? CEG_OK : rerr(CEG_ERR_HIP, "could not upload the structure factor");
}

extern "C" int ceg_recip_destroy(ceg_recip_t* h)
{
    if (!h) return CEG_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(h->device) == hipSuccess) {
        (void)hipFree(h->d_desc);
        (void)hipFree(h->d_c);
        h->io.release();
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete h;
    return CEG_OK;
}

extern "C" int ceg_recip_energy_device(ceg_recip_t* h, const double* d_positions, const double* charges, int32_t natoms,
                                       int64_t n, double energy_net_charges, double static_contribution, double* d_out,
                                       void* stream)
{
    if (!h || n < 0 || natoms < 1 || !charges || (n > 0 && (!d_positions || !d_out))) return rerr(CEG_ERR_INVALID, "bad argument");
    if (natoms > MAX_ATOMS) return rerr(CEG_ERR_UNSUPPORTED, "molecule has more atoms than the kernel holds in LDS (16)");
    if (n == 0) return CEG_OK;
    RecipGeom g{};
    for (int a = 0; a < 9; ++a) g.invmat[a] = h->invmat[a];
    for (int a = 0; a < 3; ++a) g.ks[a] = h->ks[a];
    g.natoms = natoms;
    for (int a = 0; a < natoms; ++a) g.q[a] = charges[a];
    g.energy_net_charges = energy_net_charges;
    g.static_contribution = static_contribution;
    const int tab_stride = h->ks[0] + 1 + 2 * h->ks[1] + 1 + 2 * h->ks[2] + 1;
    int waves = MAX_WAVES;
    while (waves > 1 && sizeof(double2) * (size_t)waves * natoms * tab_stride > 40 * 1024) waves >>= 1;
    const size_t tab_bytes = sizeof(double2) * (size_t)waves * natoms * tab_stride;
    if (tab_bytes > 64 * 1024) return rerr(CEG_ERR_UNSUPPORTED, "tables do not fit in LDS");
    // k-vector constants in LDS when they fit beside the tables (every placement reads all of them)
    const size_t c_bytes = sizeof(double) * 3 * (size_t)h->ns * 64 + sizeof(int32_t) * ((((size_t)h->nrounds * 64) + 3) & ~(size_t)3);
    const bool c_in_lds = tab_bytes + c_bytes <= 64 * 1024;
    const size_t lds = tab_bytes + (c_in_lds ? c_bytes : 0);
    // placements per wave: amortise the staging of the constants, but keep >= ~4 workgroups per CU in flight
    int per_wave = 1;
    if (c_in_lds) {
        per_wave = 8;
        while (per_wave > 1 && n / ((int64_t)per_wave * waves) < 2048) per_wave >>= 1;
    }
    const int64_t nblocks = (n + (int64_t)waves * per_wave - 1) / ((int64_t)waves * per_wave);
    if (nblocks > 0x7fffffffLL) return rerr(CEG_ERR_INVALID, "too many placements");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(h->device) != hipSuccess) return rerr(CEG_ERR_HIP, "hipSetDevice failed");
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)nblocks), dim3(64 * waves), lds, (hipStream_t)stream, g, h->d_desc, h->d_c,
                           h->nrounds, h->ns, d_positions, n, d_out, tab_stride, per_wave);
    };
    if (c_in_lds) {
        if (waves == 8) launch(k_recip<true, 8>);
        else if (waves == 4) launch(k_recip<true, 4>);
        else if (waves == 2) launch(k_recip<true, 2>);
        else launch(k_recip<true, 1>);
    } else {
        if (waves == 8) launch(k_recip<false, 8>);
        else if (waves == 4) launch(k_recip<false, 4>);
        else if (waves == 2) launch(k_recip<false, 2>);
        else launch(k_recip<false, 1>);
    }
    const hipError_t e = hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return rerr(CEG_ERR_HIP, hipGetErrorString(e));
    return CEG_OK;
}

extern "C" int ceg_recip_energy(ceg_recip_t* h, const double* positions, const double* charges, int32_t natoms, int64_t n,
                                double energy_net_charges, double static_contribution, double* out)
{
    if (!h || n < 0 || natoms < 1 || !charges || (n > 0 && (!positions || !out))) return rerr(CEG_ERR_INVALID, "bad argument");
    if (natoms > MAX_ATOMS) return rerr(CEG_ERR_UNSUPPORTED, "molecule has more atoms than the kernel holds in LDS (16)");
    if (n == 0) return CEG_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(h->device) != hipSuccess) return rerr(CEG_ERR_HIP, "hipSetDevice failed");
    const size_t np = (size_t)n * natoms * 3;
    int rc = CEG_OK;
    if (!h->io.ensure(sizeof(double) * np, sizeof(double) * (size_t)n)) rc = rerr(CEG_ERR_HIP, "hipMalloc failed");
    double *d_p = h->io.d_in, *d_o = h->io.d_out;
    if (!rc && hipMemcpy(d_p, positions, sizeof(double) * np, hipMemcpyHostToDevice) != hipSuccess) rc = rerr(CEG_ERR_HIP, "H2D failed");
    if (!rc) rc = ceg_recip_energy_device(h, d_p, charges, natoms, n, energy_net_charges, static_contribution, d_o, nullptr);
    // (the copy back runs on the null stream behind the kernel and reports its failure)
    if (!rc && hipMemcpy(out, d_o, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = rerr(CEG_ERR_HIP, "kernel execution or D2H failed");
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}
