// ceg_interp.hip -- batched tricubic interpolation of a device-resident energy grid
// (SURVEY 8f row f1): interpolate_grid, src/grids.jl:212-273 of CrystalEnergyGrids.jl.
//
// One thread per point.  Per point: wrap into the unit cell and convert to the fractional grid
// index exactly like offsetpoint/wrap_atom (src/coordinates.jl:58-66, operation order kept, no
// FMA contraction, so the cell (p0) a point falls in is the reference's); gather the 8 corners x
// 8 channels (32 8-byte loads when the z neighbours are adjacent, z being the fastest axis);
// apply the VdW blocking rule; evaluate the tricubic interpolant.  The reference multiplies the
// 64 data by the 64x64 integer matrix COEFF and evaluates the resulting monomials; the same
// polynomial is the tensor product of the four cubic Hermite basis functions per axis, which
// needs 64 weighted terms instead of 4096 multiply-adds.  Gather-latency bound, not ALU bound.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/ceg_hip.h"
#include "ceg_consumers.h"

using ceg_consumers::InterpGeom;
using ceg_consumers::interp_point;

namespace {

__global__ __launch_bounds__(256) void k_interpolate(InterpGeom g, const float* __restrict__ grid,
                                                      const double* __restrict__ pts, int64_t n,
                                                      double* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    out[t] = interp_point(g, grid, pts[3 * t], pts[3 * t + 1], pts[3 * t + 2]);
}

// [c][x][y][z] (the reference's array) -> [x][y][z][c]
__global__ void k_to_node_major(const float* __restrict__ in, float* __restrict__ out, int64_t nodes)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nodes * 8; i += stride) {
        const int64_t node = i >> 3;
        const int c = (int)(i & 7);
        out[i] = in[c * nodes + node];
    }
}

__global__ void k_scale(float* __restrict__ x, int64_t n, double scale)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        x[i] = (float)((double)x[i] * scale);          // Float32(grid[i] * GRID_TO_KELVIN)
}

thread_local std::string g_ierr = "";
int ifail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_ierr = buf;
    return code;
}

}  // namespace

// ceg_last_error() lives in ceg_api.hip; interpolation errors are reported through this hook
extern "C" void ceg_set_last_error_(const char* msg);

#define IHIP(expr)                                                                             \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            int rc_ = ifail(CEG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));       \
            ceg_set_last_error_(g_ierr.c_str());                                               \
            return rc_;                                                                        \
        }                                                                                      \
    } while (0)

static int ierr(int code, const char* msg)
{
    ceg_set_last_error_(msg);
    return code;
}

extern "C" int ceg_interp_create(ceg_interp_t** handle, int32_t device, const float* grid, int32_t grid_on_device,
                                 const int32_t dims[3], const double size[3], const double shift[3],
                                 const double mat[9], const double invmat[9], int32_t is_vdw)
{
    if (!handle || !grid || !dims || !size || !shift || !mat || !invmat) return ierr(CEG_ERR_INVALID, "NULL argument");
    *handle = nullptr;
    for (int a = 0; a < 3; ++a)
        if (dims[a] < 1) return ierr(CEG_ERR_INVALID, "dims < 1");
    if (ceg_device_count() <= 0) return ierr(CEG_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ceg_device_count()) return ierr(CEG_ERR_NO_DEVICE, "device not present");
    int prev = -1;
    (void)hipGetDevice(&prev);
    IHIP(hipSetDevice(device));
    ceg_interp* h = new ceg_interp();
    h->device = device;
    for (int a = 0; a < 9; ++a) { h->g.mat[a] = mat[a]; h->g.invmat[a] = invmat[a]; }
    for (int a = 0; a < 3; ++a) { h->g.size[a] = size[a]; h->g.shift[a] = shift[a]; h->g.dims[a] = dims[a]; }
    h->g.is_vdw = is_vdw ? 1 : 0;
    const size_t nodes = (size_t)(dims[0] + 1) * (dims[1] + 1) * (dims[2] + 1);
    const size_t n = 8 * nodes;
    // the handle owns a node-major copy [x][y][z][8]: one interpolation touches 8 x 64 contiguous
    // bytes instead of 32 scattered 8-byte pairs of the channel-major array
    float* staged = nullptr;
    const float* src = grid;
    bool ok = hipMalloc((void**)&h->owned, n * sizeof(float)) == hipSuccess;
    if (ok && !grid_on_device) {
        ok = hipMalloc((void**)&staged, n * sizeof(float)) == hipSuccess &&
             hipMemcpy(staged, grid, n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
        src = staged;
    }
    if (ok) {
        hipLaunchKernelGGL(k_to_node_major, dim3(4096), dim3(256), 0, nullptr, src, h->owned, (int64_t)nodes);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    }
    if (staged) (void)hipFree(staged);
    if (!ok) {
        if (h->owned) (void)hipFree(h->owned);
        delete h;
        if (prev >= 0) (void)hipSetDevice(prev);
        return ierr(CEG_ERR_HIP, "could not stage the grid on the device");
    }
    h->d_grid = h->owned;
    if (prev >= 0) (void)hipSetDevice(prev);
    *handle = h;
    return CEG_OK;
}

extern "C" int ceg_interp_set_higherorder(ceg_interp_t* h, int32_t higherorder)
{
    if (!h) return ierr(CEG_ERR_INVALID, "bad argument");
    h->g.trilinear = higherorder ? 0 : 1;
    return CEG_OK;
}

extern "C" int ceg_interp_destroy(ceg_interp_t* h)
{
    if (!h) return CEG_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(h->device) == hipSuccess) {
        if (h->owned) (void)hipFree(h->owned);
        h->io.release();
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete h;
    return CEG_OK;
}

extern "C" int ceg_interp_points_device(ceg_interp_t* h, const double* d_points, int64_t n, double* d_out, void* stream)
{
    if (!h || (n > 0 && (!d_points || !d_out)) || n < 0) return ierr(CEG_ERR_INVALID, "bad argument");
    if (n == 0) return CEG_OK;
    const int64_t nblocks = (n + 255) / 256;
    if (nblocks > 0x7fffffffLL) return ierr(CEG_ERR_INVALID, "too many points");
    int prev = -1;
    (void)hipGetDevice(&prev);
    IHIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_interpolate, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, h->g, h->d_grid, d_points, n, d_out);
    const hipError_t e = hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return ierr(CEG_ERR_HIP, hipGetErrorString(e));
    return CEG_OK;
}

extern "C" int ceg_interp_points(ceg_interp_t* h, const double* points, int64_t n, double* out)
{
    if (!h || (n > 0 && (!points || !out)) || n < 0) return ierr(CEG_ERR_INVALID, "bad argument");
    if (n == 0) return CEG_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    IHIP(hipSetDevice(h->device));
    int rc = CEG_OK;
    if (!h->io.ensure(sizeof(double) * 3 * (size_t)n, sizeof(double) * (size_t)n)) rc = ierr(CEG_ERR_HIP, "hipMalloc failed");
    double *d_p = h->io.d_in, *d_o = h->io.d_out;
    if (!rc && hipMemcpy(d_p, points, sizeof(double) * 3 * n, hipMemcpyHostToDevice) != hipSuccess) rc = ierr(CEG_ERR_HIP, "H2D failed");
    if (!rc) rc = ceg_interp_points_device(h, d_p, n, d_o, nullptr);
    // (the copy back runs on the null stream behind the kernel and reports its failure)
    if (!rc && hipMemcpy(out, d_o, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = ierr(CEG_ERR_HIP, "kernel execution or D2H failed");
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

extern "C" int ceg_scale_grid_device(float* d_grid, int64_t nfloats, double scale, int32_t device, void* stream)
{
    if (!d_grid || nfloats < 0) return ierr(CEG_ERR_INVALID, "bad argument");
    if (nfloats == 0) return CEG_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    IHIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_scale, dim3(2048), dim3(256), 0, (hipStream_t)stream, d_grid, nfloats, scale);
    const hipError_t e = hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return ierr(CEG_ERR_HIP, hipGetErrorString(e));
    return CEG_OK;
}
