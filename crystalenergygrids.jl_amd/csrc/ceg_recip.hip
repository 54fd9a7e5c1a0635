// ceg_recip.hip -- batched reciprocal-space Ewald energy of one rigid molecule:
// compute_ewald(ctx), src/ewald.jl:555-577, with the structure-factor loop of ewald_main_loop!
// (:148-185) and the e^{2 pi i m f} tables of setup_Eik / move_one_system! (:109-146, 352-366).
//
// One wave64 per placement of the molecule.  The wave first fills, in LDS, the three tables
// e^{2 pi i m f_x} (m = 0..kx), e^{2 pi i m f_y} (m = -ky..ky), e^{2 pi i m f_z} (m = -kz..kz) of
// every atom -- by sincospi of the exact angle instead of the reference's repeated complex
// multiplication (same values, smaller rounding error) -- then the lanes stride over the k-vectors
// (coalesced reads of ijk / kfactor / framework structure factor), each forming
// S_a(k) = sum_atoms q Ex[i] Ey[j] Ez[k] from the tables, and a wave reduction finishes the two
// sums.  ~nk * natoms * 20 flops per placement; no MFMA (complex products with per-lane table
// lookups, not a contraction with a shared operand).
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ceg_hip.h"

extern "C" void ceg_set_last_error_(const char* msg);

namespace {

constexpr int MAX_ATOMS = 16;       // atoms per molecule held in LDS
constexpr int MAX_TAB = 400;        // (kx+1) + (2ky+1) + (2kz+1) per atom
constexpr int WAVES = 4;            // placements per workgroup

struct RecipGeom {
    double invmat[9];
    int32_t ks[3];
    int32_t natoms;
    double q[MAX_ATOMS];
    double energy_net_charges, static_contribution;
};

// k-space constants of one k-vector as the kernel reads them from LDS: packed (i, j, k) and the three
// products the energy needs -- E = 2 sum(A sr + B si) + sum(kf (sr^2 + si^2)), A = kf Re S_f, B = kf Im S_f
struct KPack {
    double A, B, kf;
    int32_t ijk;          // i | (j + 128) << 8 | (k + 128) << 16
    int32_t _pad;
};

__global__ __launch_bounds__(64 * WAVES) void k_recip(RecipGeom g, const int32_t* __restrict__ ijk,
                                                       const double* __restrict__ kf, const double* __restrict__ sfre,
                                                       const double* __restrict__ sfim, int64_t nk,
                                                       const double* __restrict__ pos, int64_t n, double* __restrict__ out,
                                                       int tab_stride, int per_wave, int k_in_lds)
{
    // dynamic LDS: [k_in_lds ? nk : 0] KPack, then [WAVES][natoms][tab_stride] double2
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    KPack* s_k = reinterpret_cast<KPack*>(s_raw);
    double2* s_tab = reinterpret_cast<double2*>(s_raw + (k_in_lds ? sizeof(KPack) * (size_t)nk : 0));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k_in_lds) {
        // every placement of the workgroup reads all k-vectors: stage them once (32 B per k-vector)
        for (int64_t q = threadIdx.x; q < nk; q += 64 * WAVES) {
            KPack kp;
            const double t = kf[q];
            kp.A = t * sfre[q]; kp.B = t * sfim[q]; kp.kf = t;
            kp.ijk = ijk[3 * q] | ((ijk[3 * q + 1] + 128) << 8) | ((ijk[3 * q + 2] + 128) << 16);
            kp._pad = 0;
            s_k[q] = kp;
        }
        __syncthreads();
    }
    const int kx = g.ks[0], ky = g.ks[1], kz = g.ks[2];
    const int nxp = kx + 1, nyp = 2 * ky + 1, nzp = 2 * kz + 1;
    double2* tab = s_tab + (size_t)wave * g.natoms * tab_stride;
    const double* I = g.invmat;
    const int64_t p0 = ((int64_t)blockIdx.x * WAVES + wave) * per_wave;
    for (int64_t p = p0; p < p0 + per_wave && p < n; ++p) {
        // ---- tables: entry t of atom a = exp(2 pi i m f), m and the axis decoded from t
        for (int a = 0; a < g.natoms; ++a) {
            const double* r = pos + ((size_t)p * g.natoms + a) * 3;
            const double fx = I[0] * r[0] + I[3] * r[1] + I[6] * r[2];
            const double fy = I[1] * r[0] + I[4] * r[1] + I[7] * r[2];
            const double fz = I[2] * r[0] + I[5] * r[1] + I[8] * r[2];
            for (int t = lane; t < nxp + nyp + nzp; t += 64) {
                double f;
                int m;
                if (t < nxp) { f = fx; m = t; }
                else if (t < nxp + nyp) { f = fy; m = t - nxp - ky; }
                else { f = fz; m = t - nxp - nyp - kz; }
                const double ff = f - rint(f);                   // exp(2 pi i m f) is periodic in f
                double s, c;
                sincospi(2.0 * (double)m * ff, &s, &c);
                tab[a * tab_stride + t] = make_double2(c, s);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- k-vector loop
        double fa = 0.0, aa = 0.0;
        for (int64_t q = lane; q < nk; q += 64) {
            int i, j, k;
            double A, B, t;
            if (k_in_lds) {
                const KPack kp = s_k[q];
                i = kp.ijk & 0xff; j = ((kp.ijk >> 8) & 0xff) - 128; k = ((kp.ijk >> 16) & 0xff) - 128;
                A = kp.A; B = kp.B; t = kp.kf;
            } else {
                i = ijk[3 * q]; j = ijk[3 * q + 1]; k = ijk[3 * q + 2];
                t = kf[q]; A = t * sfre[q]; B = t * sfim[q];
            }
            double sr = 0.0, si = 0.0;
            for (int a = 0; a < g.natoms; ++a) {
                const double2 ex = tab[a * tab_stride + i];
                const double2 ey = tab[a * tab_stride + nxp + ky + j];
                const double2 ez = tab[a * tab_stride + nxp + nyp + kz + k];
                const double yr = ey.x * ez.x - ey.y * ez.y, yi = ey.x * ez.y + ey.y * ez.x;      // Eiky*Eikz
                const double cr = g.q[a] * yr, ci = g.q[a] * yi;                                  // c*Eik_yz
                sr += ex.x * cr - ex.y * ci;
                si += ex.x * ci + ex.y * cr;
            }
            fa += A * sr + B * si;
            aa += t * (sr * sr + si * si);
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
    int32_t* d_ijk = nullptr;
    double *d_kf = nullptr, *d_re = nullptr, *d_im = nullptr;
};

extern "C" int ceg_recip_create(ceg_recip_t** handle, int32_t device, const int32_t* kvec_ijk, const double* kfactors,
                                const double* sf_re, const double* sf_im, int64_t nk, const int32_t ks[3],
                                const double invmat[9])
{
    if (!handle || !ks || !invmat || nk < 0 || (nk > 0 && (!kvec_ijk || !kfactors || !sf_re || !sf_im)))
        return rerr(CEG_ERR_INVALID, "bad argument");
    *handle = nullptr;
    if (ks[0] < 0 || ks[1] < 0 || ks[2] < 0 || ks[0] + 1 + 2 * ks[1] + 1 + 2 * ks[2] + 1 > MAX_TAB)
        return rerr(CEG_ERR_UNSUPPORTED, "k-space box too large for the LDS tables");
    for (int64_t q = 0; q < nk; ++q)
        if (kvec_ijk[3 * q] < 0 || kvec_ijk[3 * q] > ks[0] || abs(kvec_ijk[3 * q + 1]) > ks[1] || abs(kvec_ijk[3 * q + 2]) > ks[2])
            return rerr(CEG_ERR_INVALID, "k-vector outside the (kx, ky, kz) box");
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
    const size_t m = nk > 0 ? (size_t)nk : 1;
    bool ok = hipMalloc((void**)&h->d_ijk, m * 3 * sizeof(int32_t)) == hipSuccess &&
              hipMalloc((void**)&h->d_kf, m * sizeof(double)) == hipSuccess &&
              hipMalloc((void**)&h->d_re, m * sizeof(double)) == hipSuccess &&
              hipMalloc((void**)&h->d_im, m * sizeof(double)) == hipSuccess;
    if (ok && nk > 0)
        ok = hipMemcpy(h->d_ijk, kvec_ijk, nk * 3 * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(h->d_kf, kfactors, nk * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(h->d_re, sf_re, nk * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(h->d_im, sf_im, nk * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
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
    const bool ok = hipMemcpy(h->d_re, sf_re, h->nk * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
                    hipMemcpy(h->d_im, sf_im, h->nk * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (prev >= 0) (void)hipSetDevice(prev);
    return ok ? CEG_OK : rerr(CEG_ERR_HIP, "could not upload the structure factor");
}

extern "C" int ceg_recip_destroy(ceg_recip_t* h)
{
    if (!h) return CEG_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(h->device) == hipSuccess) {
        (void)hipFree(h->d_ijk);
        (void)hipFree(h->d_kf);
        (void)hipFree(h->d_re);
        (void)hipFree(h->d_im);
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
    const size_t tab_bytes = sizeof(double2) * (size_t)WAVES * natoms * tab_stride;
    if (tab_bytes > 64 * 1024) return rerr(CEG_ERR_UNSUPPORTED, "tables do not fit in LDS");
    // k-vector constants in LDS when they fit beside the tables (every placement reads all of them)
    const size_t k_bytes = sizeof(KPack) * (size_t)h->nk;
    const bool small_k = h->ks[1] < 128 && h->ks[2] < 128 && h->ks[0] < 256;
    const int k_in_lds = (small_k && tab_bytes + k_bytes <= 60 * 1024) ? 1 : 0;
    const size_t lds = tab_bytes + (k_in_lds ? k_bytes : 0);
    // placements per wave: amortise the staging of the k-vectors, but keep >= ~4 workgroups per CU in flight
    int per_wave = 1;
    if (k_in_lds) {
        per_wave = 8;
        while (per_wave > 1 && n / ((int64_t)per_wave * WAVES) < 2048) per_wave >>= 1;
    }
    const int64_t nblocks = (n + (int64_t)WAVES * per_wave - 1) / ((int64_t)WAVES * per_wave);
    if (nblocks > 0x7fffffffLL) return rerr(CEG_ERR_INVALID, "too many placements");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(h->device) != hipSuccess) return rerr(CEG_ERR_HIP, "hipSetDevice failed");
    hipLaunchKernelGGL(k_recip, dim3((unsigned)nblocks), dim3(64 * WAVES), lds, (hipStream_t)stream, g, h->d_ijk, h->d_kf,
                       h->d_re, h->d_im, h->nk, d_positions, n, d_out, tab_stride, per_wave, k_in_lds);
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
    double *d_p = nullptr, *d_o = nullptr;
    const size_t np = (size_t)n * natoms * 3;
    int rc = CEG_OK;
    if (hipMalloc((void**)&d_p, sizeof(double) * np) != hipSuccess || hipMalloc((void**)&d_o, sizeof(double) * n) != hipSuccess)
        rc = rerr(CEG_ERR_HIP, "hipMalloc failed");
    if (!rc && hipMemcpy(d_p, positions, sizeof(double) * np, hipMemcpyHostToDevice) != hipSuccess) rc = rerr(CEG_ERR_HIP, "H2D failed");
    if (!rc) rc = ceg_recip_energy_device(h, d_p, charges, natoms, n, energy_net_charges, static_contribution, d_o, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = rerr(CEG_ERR_HIP, "kernel execution failed");
    if (!rc && hipMemcpy(out, d_o, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = rerr(CEG_ERR_HIP, "D2H failed");
    if (d_p) (void)hipFree(d_p);
    if (d_o) (void)hipFree(d_o);
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}
