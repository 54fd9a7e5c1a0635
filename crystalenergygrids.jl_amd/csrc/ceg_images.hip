// ceg_images.hip -- the lattice-image list of a plan built ON THE DEVICE (round 4).
//
// ceg_api.hip build_images() expands every framework atom into the lattice images that can be within the cutoff of a grid point
// (the grid's bounding box grown by the cutoff), bins them on a cartesian lattice and sorts them by bin -- the host-side analogue of
// the reference's ProbeSystem tiling (src/probes.jl:37-53).  On the host that is 1.0-1.4 ms for the 11 664-atom roofline framework:
// the whole compute of one rank at N = 8, paid by the FIRST call on a framework (later calls hit the image cache).  Here the same
// list comes out of a few launches on arrays that are already on the device (first form; the lean form further down is the default):
//   k_img_count   per atom: how many of its images fall into the box (the fractional hull of the box corners bounds the lattice loop);
//   exclusive scan of the counts (hipcub);
//   k_img_emit    per atom: its images in lattice order at the atom's offset -- the order the host loop emits them in;
//   stable radix sort of (bin, emit index) by bin (hipcub): inside a bin the emit order survives = the host's counting sort;
//   k_img_gather + k_bin_start.
// Same FP64 expressions in the same order as the host code (no contraction), so the lists are BYTE-identical to the host build
// (tests/test_gpu_parity.py::test_image_list_built_on_the_device) and the kernels' sums do not change by a bit.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <initializer_list>

#include "ceg_internal.h"

namespace ceg_host {
hipError_t pool_malloc(void** out, size_t bytes);      // the plan-table block cache of ceg_api.hip
void pool_free(void* ptr);
}

namespace ceg {

struct ImgBox {
    double mat[9], invmat[9];
    double lo[3], hi[3], bin[3];
    int32_t nb[3];
    int32_t has_rules, has_charge, vdw_only;      // vdw_only: atoms whose kind has no rule are left out (create_grid_vdw plans)
    int32_t nkinds;
    uint64_t hasbits[16];                         // nkinds <= 1024: "kind has a VdW rule" as a bit mask in the kernel arguments (no upload, no sync)
};

// the lattice loop of build_images for one atom; emit(P, bin) per image inside the box, in the host's order (nx, ny, nz ascending)
template <class Emit>
__device__ __forceinline__ void for_each_image(const ImgBox& B, double pax, double pay, double paz, Emit&& emit)
{
#pragma clang fp contract(off)
    const double* M = B.mat;
    const double* I = B.invmat;
    double fmin[3] = {1e300, 1e300, 1e300}, fmax[3] = {-1e300, -1e300, -1e300};
    for (int c = 0; c < 8; ++c) {
        const double d0 = ((c & 1) ? B.hi[0] : B.lo[0]) - pax, d1 = ((c & 2) ? B.hi[1] : B.lo[1]) - pay, d2 = ((c & 4) ? B.hi[2] : B.lo[2]) - paz;
        for (int q = 0; q < 3; ++q) {
            const double f = I[q] * d0 + I[q + 3] * d1 + I[q + 6] * d2;
            fmin[q] = f < fmin[q] ? f : fmin[q];          // std::min / std::max of the host code
            fmax[q] = fmax[q] < f ? f : fmax[q];
        }
    }
    int n0[3], n1[3];
    for (int q = 0; q < 3; ++q) {
        n0[q] = (int)ceil(fmin[q] - 1e-9);
        n1[q] = (int)floor(fmax[q] + 1e-9);
    }
    for (int nx = n0[0]; nx <= n1[0]; ++nx)
        for (int ny = n0[1]; ny <= n1[1]; ++ny)
            for (int nz = n0[2]; nz <= n1[2]; ++nz) {
                const double P0 = pax + (nx * M[0] + ny * M[3] + nz * M[6]);
                const double P1 = pay + (nx * M[1] + ny * M[4] + nz * M[7]);
                const double P2 = paz + (nx * M[2] + ny * M[5] + nz * M[8]);
                if (P0 < B.lo[0] || P0 > B.hi[0] || P1 < B.lo[1] || P1 > B.hi[1] || P2 < B.lo[2] || P2 > B.hi[2]) continue;
                int b[3];
                const double P[3] = {P0, P1, P2};
                for (int q = 0; q < 3; ++q) {
                    b[q] = (int)floor((P[q] - B.lo[q]) / B.bin[q]);
                    b[q] = b[q] < 0 ? 0 : (b[q] > B.nb[q] - 1 ? B.nb[q] - 1 : b[q]);      // std::min(std::max(b, 0), nb - 1)
                }
                emit(P0, P1, P2, (b[0] * B.nb[1] + b[1]) * B.nb[2] + b[2]);
            }
}

__device__ __forceinline__ bool img_kind_has_rule(const ImgBox& B, const int32_t* __restrict__ has, int32_t k)
{
    if (k < 0 || k >= B.nkinds) return false;
    return has ? has[k] != 0 : ((B.hasbits[k >> 6] >> (k & 63)) & 1ull) != 0;
}

__device__ __forceinline__ bool img_atom_listed(const ImgBox& B, const int32_t* __restrict__ kind, const int32_t* __restrict__ has, int64_t a)
{
    if (!(B.has_rules && B.vdw_only)) return true;
    return img_kind_has_rule(B, has, kind[a]);
}

__global__ void k_img_count(ImgBox B, const double4* __restrict__ atoms, const int32_t* __restrict__ kind, const int32_t* __restrict__ has,
                            int64_t natoms, int32_t* __restrict__ count)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= natoms) return;
    int n = 0;
    if (img_atom_listed(B, kind, has, a)) {
        const double4 A = atoms[a];
        for_each_image(B, A.x, A.y, A.z, [&](double, double, double, int) { ++n; });
    }
    count[a] = n;
}

__global__ void k_img_emit(ImgBox B, const double4* __restrict__ atoms, const int32_t* __restrict__ kind, const int32_t* __restrict__ has,
                           int64_t natoms, const int32_t* __restrict__ offset, double4* __restrict__ xyzq, int32_t* __restrict__ imgkind,
                           int32_t* __restrict__ imgatom, int32_t* __restrict__ imgbin, int32_t* __restrict__ iota)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= natoms) return;
    if (!img_atom_listed(B, kind, has, a)) return;
    const double4 A = atoms[a];
    int32_t kw = -1;
    if (B.has_rules) {
        const int32_t k = kind[a];
        kw = k < 0 ? -1 : (k | (img_kind_has_rule(B, has, k) ? (1 << 25) : 0));
    }
    const double q = B.has_charge ? A.w : 0.0;
    int32_t at = offset[a];
    for_each_image(B, A.x, A.y, A.z, [&](double x, double y, double z, int bin) {
        xyzq[at] = make_double4(x, y, z, q);
        imgkind[at] = kw;
        imgatom[at] = (int32_t)a;
        imgbin[at] = bin;
        iota[at] = at;
        ++at;
    });
}

__global__ void k_img_gather(const int32_t* __restrict__ perm, const double4* __restrict__ xyzq_in, const int32_t* __restrict__ kind_in,
                             const int32_t* __restrict__ atom_in, int32_t n, double4* __restrict__ xyzq, int32_t* __restrict__ kind,
                             int32_t* __restrict__ atom)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int32_t src = perm[s];
    xyzq[s] = xyzq_in[src];
    kind[s] = kind_in[src];
    atom[s] = atom_in[src];
}

// bin_start[b] = first position whose bin is >= b (b = 0 .. nbins)
__global__ void k_bin_start(const int32_t* __restrict__ sorted_bin, int32_t n, int32_t nbins, int32_t* __restrict__ bin_start)
{
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nbins) return;
    int32_t lo = 0, hi = n;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (sorted_bin[mid] < b) lo = mid + 1; else hi = mid;
    }
    bin_start[b] = lo;
}

// ---- the lean form of the same build: at 11 664 atoms every kernel is a few microseconds, and what the first form spends is launches
// (thirteen), two sleeps in hipStreamSynchronize and a radix sort of six passes.  Here the counting kernel also counts per BIN (atomics);
// two exclusive scans (hipcub) give the offsets of the atoms' images in emit order and the starts of the bins (= the bin_start result);
// the total reaches the host through a polled word of mapped memory; the emitting kernel claims a slot of its bin
// per image (atomics: any order inside a bin) and records the emit index there; one thread per bin sorts its slots by emit index --
// the order the host's stable counting sort leaves them in -- and gathers the records.  Byte-identical to the radix-sort form.
__global__ void k_img_count_bins(ImgBox B, const double4* __restrict__ atoms, const int32_t* __restrict__ kind, const int32_t* __restrict__ has,
                                 int64_t natoms, int32_t* __restrict__ count, int32_t* __restrict__ bin_count)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= natoms) return;
    int n = 0;
    if (img_atom_listed(B, kind, has, a)) {
        const double4 A = atoms[a];
        for_each_image(B, A.x, A.y, A.z, [&](double, double, double, int bin) {
            ++n;
            atomicAdd(&bin_count[bin], 1);
        });
    }
    count[a] = n;
}

// total = offset[natoms] into page-locked host memory the host polls (a stream synchronisation that has gone to sleep wakes up tens of
// microseconds late: more than every kernel of this build together)
__global__ void k_img_publish(const int32_t* __restrict__ last_offset, volatile int32_t* __restrict__ mapped)
{
    mapped[0] = *last_offset;
    __threadfence_system();
    mapped[1] = 1;
}

__global__ void k_img_emit_slots(ImgBox B, const double4* __restrict__ atoms, const int32_t* __restrict__ kind, const int32_t* __restrict__ has,
                                 int64_t natoms, const int32_t* __restrict__ offset, const int32_t* __restrict__ bin_start, int32_t* __restrict__ cursor,
                                 double4* __restrict__ xyzq, int32_t* __restrict__ imgkind, int32_t* __restrict__ imgatom, int32_t* __restrict__ slot_emit)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= natoms) return;
    if (!img_atom_listed(B, kind, has, a)) return;
    const double4 A = atoms[a];
    int32_t kw = -1;
    if (B.has_rules) {
        const int32_t k = kind[a];
        kw = k < 0 ? -1 : (k | (img_kind_has_rule(B, has, k) ? (1 << 25) : 0));
    }
    const double q = B.has_charge ? A.w : 0.0;
    int32_t at = offset[a];
    for_each_image(B, A.x, A.y, A.z, [&](double x, double y, double z, int bin) {
        xyzq[at] = make_double4(x, y, z, q);
        imgkind[at] = kw;
        imgatom[at] = (int32_t)a;
        slot_emit[bin_start[bin] + atomicAdd(&cursor[bin], 1)] = at;
        ++at;
    });
}

// one thread per bin: its slots sorted by emit index (insertion sort: a bin holds a handful of images), records gathered
__global__ void k_img_sort_gather(const int32_t* __restrict__ bin_start, int64_t nbins, int32_t* __restrict__ slot_emit, const double4* __restrict__ xyzq_in,
                                  const int32_t* __restrict__ kind_in, const int32_t* __restrict__ atom_in, double4* __restrict__ xyzq,
                                  int32_t* __restrict__ kind, int32_t* __restrict__ atom, unsigned* __restrict__ done, volatile int32_t* __restrict__ mapped)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t s0 = b < nbins ? bin_start[b] : 0, s1 = b < nbins ? bin_start[b + 1] : 0;
    for (int32_t i = s0 + 1; i < s1; ++i) {
        const int32_t v = slot_emit[i];
        int32_t j = i - 1;
        while (j >= s0 && slot_emit[j] > v) { slot_emit[j + 1] = slot_emit[j]; --j; }
        slot_emit[j + 1] = v;
    }
    for (int32_t i = s0; i < s1; ++i) {
        const int32_t src = slot_emit[i];
        xyzq[i] = xyzq_in[src];
        kind[i] = kind_in[src];
        atom[i] = atom_in[src];
    }
    // the last workgroup to finish raises the completion word the host polls
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(done, 1u) == gridDim.x - 1) {
            __threadfence_system();
            mapped[2] = 1;
        }
    }
}

// three words of page-locked, device-mapped host memory per thread (total, "total is there", "the build is complete"): the host polls
// them instead of sleeping in hipStreamSynchronize (20 ms without the word: fall back to the stream, which also reports a failed launch)
struct MappedWords {
    volatile int32_t* h = nullptr;
    int32_t* d = nullptr;
    MappedWords()
    {
        struct Slot {
            int32_t* h = nullptr;
            int32_t* d = nullptr;
            int device = -1;          // (64 bytes per thread, never freed: the HIP runtime may be gone when thread-local destructors run)
        };
        static thread_local Slot slot;
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return;
        if (slot.h && slot.device != dev) { (void)hipHostFree(slot.h); slot.h = nullptr; }
        if (!slot.h) {
            if (hipHostMalloc((void**)&slot.h, 64, hipHostMallocMapped) != hipSuccess) { slot.h = nullptr; return; }
            if (hipHostGetDevicePointer((void**)&slot.d, slot.h, 0) != hipSuccess) { (void)hipHostFree(slot.h); slot.h = nullptr; return; }
            slot.device = dev;
        }
        h = slot.h;
        d = slot.d;
    }
    bool ok() const { return h != nullptr; }
    bool wait(int word, hipStream_t st) const
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0;; ++spin) {
            if (__atomic_load_n(const_cast<const int32_t*>(h) + word, __ATOMIC_ACQUIRE) == 1) return true;
            if ((spin & 1023u) == 1023u && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 20.0) break;
            __builtin_ia32_pause();
        }
        return hipStreamSynchronize(st) == hipSuccess && __atomic_load_n(const_cast<const int32_t*>(h) + word, __ATOMIC_ACQUIRE) == 1;
    }
};

// Returns hipSuccess and the four arrays of an image set (allocated from the plan-table pool; the caller owns them), or an error with
// nothing allocated.  d_kind: 0-based kind per atom or nullptr; d_has: per kind "has a VdW rule" (with the plan's probe / any probe).
hipError_t build_images_device(const ImgBox& B, const double4* d_atoms, const int32_t* d_kind, const int32_t* d_has, int64_t natoms,
                               double4** out_xyzq, int32_t** out_kind, int32_t** out_atom, int32_t** out_binstart, int64_t* out_n)
{
    *out_xyzq = nullptr; *out_kind = nullptr; *out_atom = nullptr; *out_binstart = nullptr; *out_n = 0;
    const int64_t nbins = (int64_t)B.nb[0] * B.nb[1] * B.nb[2];
    if (natoms > 0x3fffffffLL || nbins > 0x3fffffffLL) return hipErrorInvalidValue;
    hipStream_t st = nullptr;
    void* blocks[10] = {nullptr};
    int nblocks_held = 0;
    auto get = [&](size_t bytes) -> void* {
        void* p = nullptr;
        if (ceg_host::pool_malloc(&p, bytes > 0 ? bytes : 1) != hipSuccess) return nullptr;
        blocks[nblocks_held++] = p;
        return p;
    };
    auto fail = [&](hipError_t e) {
        (void)hipDeviceSynchronize();
        for (int i = 0; i < nblocks_held; ++i) ceg_host::pool_free(blocks[i]);
        return e == hipSuccess ? hipErrorUnknown : e;
    };
    const size_t na = (size_t)(natoms > 0 ? natoms : 1);
    int32_t* d_count = (int32_t*)get(sizeof(int32_t) * (na + 1));
    int32_t* d_offset = (int32_t*)get(sizeof(int32_t) * (na + 1));
    if (!d_count || !d_offset) return fail(hipErrorOutOfMemory);
    const unsigned ga = (unsigned)((natoms + 127) / 128);
    // the lean form (CEG_HIP_IMAGES_SORT=radix keeps the radix-sort form)
    const char* sort_env = getenv("CEG_HIP_IMAGES_SORT");
    MappedWords words;
    if (!(sort_env && sort_env[0] == 'r') && words.ok()) {
        int32_t* r_start = nullptr;
        if (ceg_host::pool_malloc((void**)&r_start, sizeof(int32_t) * (size_t)(nbins + 1)) != hipSuccess) return fail(hipErrorOutOfMemory);
        int32_t* d_bincount = (int32_t*)get(sizeof(int32_t) * (2 * (size_t)nbins + 2));      // per-bin counts (+ 1), per-bin cursors, completion counter
        auto fail_lean = [&](hipError_t e, std::initializer_list<void*> more) {
            const hipError_t r = fail(e);
            ceg_host::pool_free(r_start);
            for (void* p : more) if (p) ceg_host::pool_free(p);
            return r;
        };
        if (!d_bincount) return fail_lean(hipErrorOutOfMemory, {});
        int32_t* d_cursor = d_bincount + nbins + 1;
        unsigned* d_done = reinterpret_cast<unsigned*>(d_cursor + nbins);
        size_t tmp_a = 0, tmp_b = 0;
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_a, d_count, d_offset, (int)(natoms + 1), st);
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_b, d_bincount, r_start, (int)(nbins + 1), st);
        void* d_tmp = get(tmp_a > tmp_b ? tmp_a : tmp_b);
        if (!d_tmp) return fail_lean(hipErrorOutOfMemory, {});
        words.h[0] = 0; words.h[1] = 0; words.h[2] = 0;
        if (hipMemsetAsync(d_bincount, 0, sizeof(int32_t) * (2 * (size_t)nbins + 2), st) != hipSuccess ||
            hipMemsetAsync(d_count + natoms, 0, sizeof(int32_t), st) != hipSuccess)
            return fail_lean(hipGetLastError(), {});
        if (natoms > 0) hipLaunchKernelGGL(k_img_count_bins, dim3(ga), dim3(128), 0, st, B, d_atoms, d_kind, d_has, natoms, d_count, d_bincount);
        if (hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_a, d_count, d_offset, (int)(natoms + 1), st) != hipSuccess) return fail_lean(hipGetLastError(), {});
        hipLaunchKernelGGL(k_img_publish, dim3(1), dim3(1), 0, st, d_offset + natoms, words.d);
        if (hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_b, d_bincount, r_start, (int)(nbins + 1), st) != hipSuccess) return fail_lean(hipGetLastError(), {});
        if (hipGetLastError() != hipSuccess || !words.wait(1, st)) return fail_lean(hipErrorUnknown, {});
        const int32_t total = words.h[0];
        if (total < 0 || (int64_t)total > 0x7ffffff0LL) return fail_lean(hipErrorInvalidValue, {});
        const size_t n = (size_t)(total > 0 ? total : 1);
        double4* t_xyzq = (double4*)get(sizeof(double4) * n);
        int32_t* t_kind = (int32_t*)get(sizeof(int32_t) * 3 * n);                      // kind, atom, slot -> emit index
        if (!t_xyzq || !t_kind) return fail_lean(hipErrorOutOfMemory, {});
        int32_t *t_atom = t_kind + n, *t_slot = t_kind + 2 * n;
        double4* r_xyzq = nullptr;
        int32_t *r_kind = nullptr, *r_atom = nullptr;
        if (ceg_host::pool_malloc((void**)&r_xyzq, sizeof(double4) * n) != hipSuccess || ceg_host::pool_malloc((void**)&r_kind, sizeof(int32_t) * n) != hipSuccess ||
            ceg_host::pool_malloc((void**)&r_atom, sizeof(int32_t) * n) != hipSuccess)
            return fail_lean(hipErrorOutOfMemory, {(void*)r_xyzq, (void*)r_kind, (void*)r_atom});
        if (natoms > 0 && total > 0) {
            hipLaunchKernelGGL(k_img_emit_slots, dim3(ga), dim3(128), 0, st, B, d_atoms, d_kind, d_has, natoms, d_offset, r_start, d_cursor, t_xyzq, t_kind, t_atom, t_slot);
            hipLaunchKernelGGL(k_img_sort_gather, dim3((unsigned)((nbins + 127) / 128)), dim3(128), 0, st, r_start, nbins, t_slot, t_xyzq, t_kind, t_atom, r_xyzq, r_kind, r_atom,
                               d_done, words.d);
            if (hipGetLastError() != hipSuccess || !words.wait(2, st)) return fail_lean(hipErrorUnknown, {(void*)r_xyzq, (void*)r_kind, (void*)r_atom});
        } else if (hipStreamSynchronize(st) != hipSuccess) {
            return fail_lean(hipErrorUnknown, {(void*)r_xyzq, (void*)r_kind, (void*)r_atom});
        }
        for (int i = 0; i < nblocks_held; ++i) ceg_host::pool_free(blocks[i]);
        *out_xyzq = r_xyzq; *out_kind = r_kind; *out_atom = r_atom; *out_binstart = r_start; *out_n = total;
        return hipSuccess;
    }
    if (natoms > 0) hipLaunchKernelGGL(k_img_count, dim3(ga), dim3(128), 0, st, B, d_atoms, d_kind, d_has, natoms, d_count);
    if (hipMemsetAsync(d_count + natoms, 0, sizeof(int32_t), st) != hipSuccess) return fail(hipGetLastError());
    size_t tmp_scan = 0, tmp_sort = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, d_count, d_offset, (int)(natoms + 1), st);
    void* d_tmp = get(tmp_scan);
    if (!d_tmp) return fail(hipErrorOutOfMemory);
    if (hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_scan, d_count, d_offset, (int)(natoms + 1), st) != hipSuccess) return fail(hipGetLastError());
    int32_t total = 0;
    if (hipMemcpyAsync(&total, d_offset + natoms, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(hipGetLastError());
    if (total < 0 || (int64_t)total > 0x7ffffff0LL) return fail(hipErrorInvalidValue);
    const size_t n = (size_t)(total > 0 ? total : 1);
    // unsorted (emit order) arrays + keys; the results
    double4* t_xyzq = (double4*)get(sizeof(double4) * n);
    int32_t* t_kind = (int32_t*)get(sizeof(int32_t) * n);
    int32_t* t_atom = (int32_t*)get(sizeof(int32_t) * n);
    int32_t* t_bin = (int32_t*)get(sizeof(int32_t) * 4 * n);       // bin, iota, sorted bin, permutation
    if (!t_xyzq || !t_kind || !t_atom || !t_bin) return fail(hipErrorOutOfMemory);
    int32_t *t_iota = t_bin + n, *s_bin = t_bin + 2 * n, *s_perm = t_bin + 3 * n;
    double4* r_xyzq = nullptr;
    int32_t *r_kind = nullptr, *r_atom = nullptr, *r_start = nullptr;
    if (ceg_host::pool_malloc((void**)&r_xyzq, sizeof(double4) * n) != hipSuccess || ceg_host::pool_malloc((void**)&r_kind, sizeof(int32_t) * n) != hipSuccess ||
        ceg_host::pool_malloc((void**)&r_atom, sizeof(int32_t) * n) != hipSuccess ||
        ceg_host::pool_malloc((void**)&r_start, sizeof(int32_t) * (size_t)(nbins + 1)) != hipSuccess) {
        for (void* p : {(void*)r_xyzq, (void*)r_kind, (void*)r_atom, (void*)r_start})
            if (p) ceg_host::pool_free(p);
        return fail(hipErrorOutOfMemory);
    }
    auto fail_all = [&](hipError_t e) {
        const hipError_t r = fail(e);
        for (void* p : {(void*)r_xyzq, (void*)r_kind, (void*)r_atom, (void*)r_start}) ceg_host::pool_free(p);
        return r;
    };
    if (natoms > 0) hipLaunchKernelGGL(k_img_emit, dim3(ga), dim3(128), 0, st, B, d_atoms, d_kind, d_has, natoms, d_offset, t_xyzq, t_kind, t_atom, t_bin, t_iota);
    if (total > 0) {
        int bits = 1;
        while (bits < 31 && (1ll << bits) < nbins) ++bits;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, t_bin, s_bin, t_iota, s_perm, total, 0, bits, st);
        void* d_tmp2 = get(tmp_sort);
        if (!d_tmp2) return fail_all(hipErrorOutOfMemory);
        if (hipcub::DeviceRadixSort::SortPairs(d_tmp2, tmp_sort, t_bin, s_bin, t_iota, s_perm, total, 0, bits, st) != hipSuccess) return fail_all(hipGetLastError());
        hipLaunchKernelGGL(k_img_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s_perm, t_xyzq, t_kind, t_atom, total, r_xyzq, r_kind, r_atom);
    }
    hipLaunchKernelGGL(k_bin_start, dim3((unsigned)((nbins + 1 + 255) / 256)), dim3(256), 0, st, s_bin, total, (int32_t)nbins, r_start);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail_all(hipErrorUnknown);
    for (int i = 0; i < nblocks_held; ++i) ceg_host::pool_free(blocks[i]);
    *out_xyzq = r_xyzq; *out_kind = r_kind; *out_atom = r_atom; *out_binstart = r_start; *out_n = total;
    return hipSuccess;
}

}  // namespace ceg
