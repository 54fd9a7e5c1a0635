// ceg_block.hip -- blocking masks on the grid lattice:
//   BlockFile(g::EnergyGrid)  src/grids.jl:188-204     every lattice cell whose first corner value
//                                                      exceeds 5e6 K blocks its 8 corners
//   parse_blockfile           src/coordinates.jl:112-167  every lattice point within a blocking sphere
//                                                      (minimum-image distance in the unit cell)
// Both write a uint8 mask [nx][ny][nz] (z fastest, like the grid).  HBM-bound: the first reads 4 B and
// writes 1 B per point (neighbour reads hit L2), the second is a few hundred flops per point.
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/ceg_hip.h"
#include "ceg_minimage.h"

extern "C" void ceg_set_last_error_(const char* msg);

namespace {

int berr(int code, const char* msg)
{
    ceg_set_last_error_(msg);
    return code;
}

// point (i, j, k) is blocked iff one of the cells it is a corner of -- (i-di, j-dj, k-dk), di, dj, dk in
// {0, 1}, cell indices in [0, n-2] -- has value > threshold (grids.jl:190-201, gathered instead of scattered)
__global__ __launch_bounds__(256) void k_block_from_grid(const float* __restrict__ value, int nx, int ny, int nz, float threshold,
                                                         uint8_t* __restrict__ block)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)nx * ny * nz;
    if (t >= total) return;
    const int k = (int)(t % nz), j = (int)((t / nz) % ny), i = (int)(t / ((int64_t)nz * ny));
    bool b = false;
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                const int ci = i - di, cj = j - dj, ck = k - dk;
                if (ci < 0 || cj < 0 || ck < 0 || ci > nx - 2 || cj > ny - 2 || ck > nz - 2) continue;
                b = b || (value[(int64_t)ck + nz * ((int64_t)cj + (int64_t)ny * ci)] > threshold);
            }
    block[t] = b ? 1 : 0;
}

struct SphereGeom {
    double mat[9], invmat[9];
    double delta[3], shift[3];
    double safemin2;
    int32_t ortho, nx, ny, nz, ns;
};

// coordinates.jl:139-152: first sphere whose minimum-image distance to the lattice point is < radius
__global__ __launch_bounds__(256) void k_block_spheres(SphereGeom g, const double4* __restrict__ spheres /* cx, cy, cz, r2 */,
                                                       uint8_t* __restrict__ block)
{
    extern __shared__ double4 s_sph[];
    for (int t = threadIdx.x; t < g.ns; t += 256) s_sph[t] = spheres[t];
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)g.nx * g.ny * g.nz;
    if (t >= total) return;
    const int k = (int)(t % g.nz), j = (int)((t / g.nz) % g.ny), i = (int)(t / ((int64_t)g.nz * g.ny));
    double px, py, pz;
    {
#pragma clang fp contract(off)
        // inverse_offsetpoint (coordinates.jl:68-70): (ipoint - 1) .* Δ .+ shift, 1-based ipoint
        px = (double)i * g.delta[0] + g.shift[0];
        py = (double)j * g.delta[1] + g.shift[1];
        pz = (double)k * g.delta[2] + g.shift[2];
    }
    bool b = false;
    for (int s = 0; s < g.ns && !b; ++s) {
        const double4 S = s_sph[s];
        double dx = S.x - px, dy = S.y - py, dz = S.z - pz;          // center .- point (:146)
        b = ceg::periodic_distance2_literal_m(g.mat, g.invmat, g.ortho, g.safemin2, dx, dy, dz) < S.w;
    }
    block[t] = b ? 1 : 0;
}

struct DevGuard {
    int prev = -1;
    bool ok = false;
    explicit DevGuard(int d)
    {
        (void)hipGetDevice(&prev);
        ok = hipSetDevice(d) == hipSuccess;
    }
    ~DevGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

extern "C" int ceg_block_from_grid(int32_t device, const float* value, int32_t value_on_device, const int32_t dims[3],
                                   double threshold, uint8_t* block)
{
    if (!value || !dims || !block || dims[0] < 1 || dims[1] < 1 || dims[2] < 1) return berr(CEG_ERR_INVALID, "bad argument");
    if (ceg_device_count() <= 0) return berr(CEG_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ceg_device_count()) return berr(CEG_ERR_NO_DEVICE, "device not present");
    const int nx = dims[0] + 1, ny = dims[1] + 1, nz = dims[2] + 1;
    const int64_t total = (int64_t)nx * ny * nz;
    DevGuard guard(device);
    if (!guard.ok) return berr(CEG_ERR_HIP, "hipSetDevice failed");
    float* d_v = nullptr;
    uint8_t* d_b = nullptr;
    int rc = CEG_OK;
    if (hipMalloc((void**)&d_b, (size_t)total) != hipSuccess) rc = berr(CEG_ERR_HIP, "hipMalloc failed");
    if (!rc && !value_on_device) {
        if (hipMalloc((void**)&d_v, sizeof(float) * (size_t)total) != hipSuccess ||
            hipMemcpy(d_v, value, sizeof(float) * (size_t)total, hipMemcpyHostToDevice) != hipSuccess)
            rc = berr(CEG_ERR_HIP, "upload of the value channel failed");
    }
    if (!rc) {
        hipLaunchKernelGGL(k_block_from_grid, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr,
                           value_on_device ? value : d_v, nx, ny, nz, (float)threshold, d_b);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = berr(CEG_ERR_HIP, "kernel failed");
    }
    if (!rc && hipMemcpy(block, d_b, (size_t)total, hipMemcpyDeviceToHost) != hipSuccess) rc = berr(CEG_ERR_HIP, "D2H failed");
    if (d_v) (void)hipFree(d_v);
    if (d_b) (void)hipFree(d_b);
    return rc;
}

extern "C" int ceg_block_spheres(int32_t device, const int32_t dims[3], const double delta[3], const double shift[3],
                                 const double mat[9], const double invmat[9], int32_t ortho, double safemin2,
                                 const double* centers, const double* radius2, int32_t nspheres, uint8_t* block)
{
    if (!dims || !delta || !shift || !mat || !invmat || !block || nspheres < 0 || (nspheres > 0 && (!centers || !radius2)) ||
        dims[0] < 1 || dims[1] < 1 || dims[2] < 1)
        return berr(CEG_ERR_INVALID, "bad argument");
    if (nspheres > 2048) return berr(CEG_ERR_UNSUPPORTED, "more than 2048 blocking spheres");
    if (ceg_device_count() <= 0) return berr(CEG_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ceg_device_count()) return berr(CEG_ERR_NO_DEVICE, "device not present");
    SphereGeom g{};
    for (int a = 0; a < 9; ++a) { g.mat[a] = mat[a]; g.invmat[a] = invmat[a]; }
    for (int a = 0; a < 3; ++a) { g.delta[a] = delta[a]; g.shift[a] = shift[a]; }
    g.safemin2 = safemin2; g.ortho = ortho;
    g.nx = dims[0] + 1; g.ny = dims[1] + 1; g.nz = dims[2] + 1; g.ns = nspheres;
    const int64_t total = (int64_t)g.nx * g.ny * g.nz;
    std::vector<double4> sph((size_t)std::max(nspheres, 1));
    for (int s = 0; s < nspheres; ++s) sph[s] = make_double4(centers[3 * s], centers[3 * s + 1], centers[3 * s + 2], radius2[s]);
    DevGuard guard(device);
    if (!guard.ok) return berr(CEG_ERR_HIP, "hipSetDevice failed");
    double4* d_s = nullptr;
    uint8_t* d_b = nullptr;
    int rc = CEG_OK;
    if (hipMalloc((void**)&d_b, (size_t)total) != hipSuccess || hipMalloc((void**)&d_s, sph.size() * sizeof(double4)) != hipSuccess ||
        hipMemcpy(d_s, sph.data(), sph.size() * sizeof(double4), hipMemcpyHostToDevice) != hipSuccess)
        rc = berr(CEG_ERR_HIP, "allocation / upload failed");
    if (!rc) {
        hipLaunchKernelGGL(k_block_spheres, dim3((unsigned)((total + 255) / 256)), dim3(256), sizeof(double4) * sph.size(), nullptr, g,
                           d_s, d_b);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = berr(CEG_ERR_HIP, "kernel failed");
    }
    if (!rc && hipMemcpy(block, d_b, (size_t)total, hipMemcpyDeviceToHost) != hipSuccess) rc = berr(CEG_ERR_HIP, "D2H failed");
    if (d_s) (void)hipFree(d_s);
    if (d_b) (void)hipFree(d_b);
    return rc;
}
