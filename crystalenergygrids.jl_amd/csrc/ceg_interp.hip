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

namespace {

struct InterpGeom {
    double mat[9], invmat[9];
    double size[3], shift[3];
    int32_t dims[3];
    int32_t is_vdw;
};

// 1-D cubic Hermite basis on [0,1]: value at 0, value at 1, slope at 0, slope at 1
__device__ __forceinline__ void hermite(double t, double w[2][2])
{
    const double t2 = t * t, t3 = t2 * t;
    w[0][0] = 2.0 * t3 - 3.0 * t2 + 1.0;     // f(0)
    w[0][1] = -2.0 * t3 + 3.0 * t2;          // f(1)
    w[1][0] = t3 - 2.0 * t2 + t;             // f'(0)
    w[1][1] = t3 - t2;                       // f'(1)
}

__global__ __launch_bounds__(256) void k_interpolate(InterpGeom g, const float* __restrict__ grid,
                                                      const double* __restrict__ pts, int64_t n,
                                                      double* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const double px = pts[3 * t], py = pts[3 * t + 1], pz = pts[3 * t + 2];
    double sh[3];
    {
#pragma clang fp contract(off)
        // wrap_atom: abc = invmat * p;  newpoint = mat * (abc - floor(abc))      coordinates.jl:58-61
        const double* I = g.invmat;
        const double* M = g.mat;
        double a0 = (I[0] * px + I[3] * py) + I[6] * pz;
        double a1 = (I[1] * px + I[4] * py) + I[7] * pz;
        double a2 = (I[2] * px + I[5] * py) + I[8] * pz;
        a0 -= floor(a0); a1 -= floor(a1); a2 -= floor(a2);
        const double q0 = (M[0] * a0 + M[3] * a1) + M[6] * a2;
        const double q1 = (M[1] * a0 + M[4] * a1) + M[7] * a2;
        const double q2 = (M[2] * a0 + M[5] * a1) + M[8] * a2;
        // offsetpoint: (newpoint - shift)*dims/size + 1                           coordinates.jl:63-66
        sh[0] = (q0 - g.shift[0]) * (double)g.dims[0] / g.size[0] + 1.0;
        sh[1] = (q1 - g.shift[1]) * (double)g.dims[1] / g.size[1] + 1.0;
        sh[2] = (q2 - g.shift[2]) * (double)g.dims[2] / g.size[2] + 1.0;
    }
    const int nx = g.dims[0] + 1, ny = g.dims[1] + 1, nz = g.dims[2] + 1;
    // p0 = floor.(Int, shifted);  p1 = p0 .+ (p0 != extent)   (1-based)           grids.jl:216-218
    int p0[3], p1[3];
    double r[3];
    const int ext[3] = {nx, ny, nz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double f = floor(sh[a]);
        int i0 = (int)f;
        r[a] = sh[a] - f;
        // memory safety only: a wrapped point always lands in [1, extent]
        i0 = i0 < 1 ? 1 : (i0 > ext[a] ? ext[a] : i0);
        p0[a] = i0;
        p1[a] = i0 + (i0 != ext[a] ? 1 : 0);
    }
    // node-major layout [x][y][z][8 channels]: the 8 channels of a corner are 32 contiguous bytes and
    // the two z neighbours of an (x, y) row 64 contiguous bytes
    const int64_t sx = (int64_t)ny * nz, sy = nz;
    const int64_t bx[2] = {(int64_t)(p0[0] - 1) * sx, (int64_t)(p1[0] - 1) * sx};
    const int64_t by[2] = {(int64_t)(p0[1] - 1) * sy, (int64_t)(p1[1] - 1) * sy};
    const int z0 = p0[2] - 1, z1 = p1[2] - 1;

    double wx[2][2], wy[2][2], wz[2][2];
    hermite(r[0], wx);
    hermite(r[1], wy);
    hermite(r[2], wz);

    double ret = 0.0;
    bool blocked = false;
    const float4* g4 = reinterpret_cast<const float4*>(grid);
#pragma unroll
    for (int ax = 0; ax < 2; ++ax)
#pragma unroll
        for (int ay = 0; ay < 2; ++ay) {
            const int64_t node0 = bx[ax] + by[ay] + z0, node1 = bx[ax] + by[ay] + z1;
            const float4 a0 = g4[2 * node0], b0 = g4[2 * node0 + 1];      // channels 0-3, 4-7 at z0
            const float4 a1 = g4[2 * node1], b1 = g4[2 * node1 + 1];      // ... at z1
            blocked = blocked || (a0.x > 5e6f) || (a1.x > 5e6f);
            // channels: value, dx, dy, dz, dxy, dxz, dyz, dxyz (derivatives pre-scaled by the grid step)
            const double v0[8] = {a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, b0.z, b0.w};
            const double v1[8] = {a1.x, a1.y, a1.z, a1.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ox = (c == 1 || c == 4 || c == 5 || c == 7) ? 1 : 0;
                const int oy = (c == 2 || c == 4 || c == 6 || c == 7) ? 1 : 0;
                const int oz = (c == 3 || c == 5 || c == 6 || c == 7) ? 1 : 0;
                const double wxy = wx[ox][ax] * wy[oy][ay];
                ret += wxy * (v0[c] * wz[oz][0] + v1[c] * wz[oz][1]);
            }
        }
    // VdW grid with any corner value > 5e6 -> 1e100 K                             grids.jl:245-248
    out[t] = (g.is_vdw && blocked) ? 1e100 : ret;
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

struct ceg_interp {
    int device = 0;
    InterpGeom g{};
    const float* d_grid = nullptr;
    float* owned = nullptr;
};

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

extern "C" int ceg_interp_destroy(ceg_interp_t* h)
{
    if (!h) return CEG_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(h->device) == hipSuccess && h->owned) (void)hipFree(h->owned);
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
    double *d_p = nullptr, *d_o = nullptr;
    int rc = CEG_OK;
    if (hipMalloc((void**)&d_p, sizeof(double) * 3 * n) != hipSuccess || hipMalloc((void**)&d_o, sizeof(double) * n) != hipSuccess)
        rc = ierr(CEG_ERR_HIP, "hipMalloc failed");
    if (!rc && hipMemcpy(d_p, points, sizeof(double) * 3 * n, hipMemcpyHostToDevice) != hipSuccess) rc = ierr(CEG_ERR_HIP, "H2D failed");
    if (!rc) rc = ceg_interp_points_device(h, d_p, n, d_o, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = ierr(CEG_ERR_HIP, "kernel execution failed");
    if (!rc && hipMemcpy(out, d_o, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = ierr(CEG_ERR_HIP, "D2H failed");
    if (d_p) (void)hipFree(d_p);
    if (d_o) (void)hipFree(d_o);
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
