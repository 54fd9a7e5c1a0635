// ceg_api.hip -- C ABI of libceg_hip.so (include/ceg_hip.h): plan management, host-side
// preparation of the lattice-image list and bins, slab scheduling over devices.
//
// Replaces the loop nests of create_grid_vdw / create_grid_coulomb
// (src/grids.jl:144-150, 171-177 of CrystalEnergyGrids.jl).  No CPU compute path exists
// here: without a HIP device every build entry point fails with CEG_ERR_NO_DEVICE.
#include "ceg_internal.h"

#include <algorithm>
#include <atomic>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <string>
#include <thread>
#include <vector>

using namespace ceg;

// ------------------------------------------------------------------ errors
static thread_local std::string g_err = "";

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(CEG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

// used by the other translation units (ceg_interp.hip) to report through ceg_last_error()
extern "C" void ceg_set_last_error_(const char* msg) { g_err = msg ? msg : ""; }

extern "C" int ceg_abi_version(void) { return CEG_ABI_VERSION; }

extern "C" const char* ceg_last_error(void) { return g_err.c_str(); }

extern "C" int ceg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ------------------------------------------------------------------ plan
struct ceg_plan {
    int device = 0;
    Geom g{};
    int64_t natoms = 0;
    bool has_rules = false, has_charge = false;
    bool can_cull = false;
    // host copies (needed to (re)build the image list)
    std::vector<double> h_pos, h_charge;
    std::vector<int32_t> h_kind;           // 0-based, -1 if none
    std::vector<DevRule> h_rules;
    std::vector<int32_t> h_offset;
    std::vector<int32_t> h_offset_union;   // multi-probe plans: offsets of a table with one entry per kind that has a rule with ANY probe
    int32_t nkinds = 0;
    // device
    double4* d_atoms = nullptr;
    int32_t* d_kind = nullptr;
    DevRule* d_rules = nullptr;
    int32_t* d_offset = nullptr;
    double4* d_images = nullptr;    // (views into `images`, which owns them)
    int32_t* d_imgkind = nullptr;
    int32_t* d_imgatom = nullptr;
    int32_t* d_binstart = nullptr;
    std::shared_ptr<void> images;          // owner of the lattice-image list + bins on the device (an ImageSet, shared between plans: image cache)
    ImageBins ib{};
    bool images_built = false;
    bool images_from_cache = false;
    PlanConst* d_pc = nullptr;   // device copy of {g, ib, rt, tables}
    double* d_erfcx = nullptr;
    double* d_exp2 = nullptr;
    int vdwk = 0;                // hot-loop VdW variant: 0 generic, 1 LJ-only, 2 LJ/Buckingham classes, 3 one tabulated Buckingham class
    bool single_buck = false;    // every present kind with a VdW rule is the same Buckingham (+ hard sphere) -> candidate for vdwk 3
    double bk[4] = {0, 0, 0, 0}; // its A, B, C, shift
    double* d_bk2 = nullptr;
    double r_exact2 = CEG_R_EXACT2;
    std::vector<FastVdw> h_fast;
    FastVdw* d_fast = nullptr;
    bool fast_ewald = false;     // alpha*cutoff within the erfcx polynomial's domain
    bool ew2 = false;            // r^2-indexed Ewald tables built (hot-loop variant EWK = 2)
    double* d_ew2 = nullptr;
    // multi-probe plans (ceg_plan_create_multi): the rule tables of every probe; d_pc holds probe 0 in its single-probe slots and
    // all of them in rtm / fastm, d_pc_probe[p] is the same block with probe p in the single-probe slots (launches of one probe)
    struct ProbeTab {
        std::vector<DevRule> rules;
        std::vector<int32_t> offset;
        std::vector<FastVdw> fast;
        DevRule* d_rules = nullptr;
        int32_t* d_offset = nullptr;
        FastVdw* d_fast = nullptr;
        PlanConst* d_pc = nullptr;
        // the probe's own hot-loop class (round 4: probes of several classes in one plan): 1 Lennard-Jones-only -- shares launches
        // with other such probes --, 3 one tabulated Buckingham class, 2 / 0 as in an ordinary plan: single-probe launches
        int vdwk = 0;
        bool single_buck = false;
        double bk[4] = {0, 0, 0, 0};
        double* d_bk2 = nullptr;
        int32_t* d_imgkind = nullptr;   // per image: kind | "the kind has a rule with THIS probe" (the shared list carries the union)
    };
    int nprobes = 0;             // 0: ordinary plan
    std::vector<ProbeTab> probes;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

inline void matvec(const double* m, const double v[3], double o[3])
{
    for (int i = 0; i < 3; ++i) o[i] = m[i] * v[0] + m[i + 3] * v[1] + m[i + 6] * v[2];
}

// perpendicular widths of the cell (src/utils.jl:10-29)
void perpendicular_widths(const double* m, double w[3])
{
    const double* a = m;
    const double* b = m + 3;
    const double* c = m + 6;
    auto cross = [](const double* u, const double* v, double* o) {
        o[0] = u[1] * v[2] - u[2] * v[1];
        o[1] = u[2] * v[0] - u[0] * v[2];
        o[2] = u[0] * v[1] - u[1] * v[0];
    };
    auto norm = [](const double* u) { return std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]); };
    double axb[3], bxc[3], cxa[3];
    cross(a, b, axb);
    cross(b, c, bxc);
    cross(c, a, cxa);
    const double vol = std::fabs(a[0] * bxc[0] + a[1] * bxc[1] + a[2] * bxc[2]);
    w[0] = vol / norm(bxc);
    w[1] = vol / norm(cxa);
    w[2] = vol / norm(axb);
}

// Translate the public rule table into device rules.  Only kinds that occur in the atom list
// are validated, mirroring the reference where derivativesGrid raises lazily
// (src/interactions.jl:442-443,462-467).
int convert_rules(ceg_plan* p, const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds)
{
    std::vector<char> present(nkinds, 0);
    for (int64_t a = 0; a < p->natoms; ++a) present[p->h_kind[a]] = 1;
    p->h_offset.assign(nkinds + 1, 0);
    p->h_rules.clear();
    for (int32_t k = 0; k < nkinds; ++k) {
        p->h_offset[k] = (int32_t)p->h_rules.size();
        for (int32_t t = rule_offset[k]; t < rule_offset[k + 1]; ++t) {
            const ceg_rule_t& r = rules[t];
            DevRule d{};
            d.kind = r.kind;
            d.shift = r.shift;
            switch (r.kind) {
            case CEG_NOINTERACTION:
            case CEG_COULOMB_EWALD_DIRECT:
                continue;  // exact zeros in a VdW grid
            case CEG_LENNARDJONES:
                d.p0 = r.p[0];
                d.p1 = r.p[1] * r.p[1];
                break;
            case CEG_BUCKINGHAM:
                d.p0 = r.p[0];
                d.p1 = r.p[1];
                d.p2 = r.p[2];
                break;
            case CEG_HARDSPHERE: {
                const double rr = r.p[0] + r.p[1];
                d.p0 = rr * rr;
                break;
            }
            default:
                if (present[k])
                    return fail(CEG_ERR_RULE,
                                "interaction kind %d (force-field index %d) is not valid in a VdW grid "
                                "(src/interactions.jl:442-467)", r.kind, k + 1);
                continue;
            }
            p->h_rules.push_back(d);
        }
    }
    p->h_offset[nkinds] = (int32_t)p->h_rules.size();
    p->nkinds = nkinds;
    // classify every kind for the hot loop of k_culled
    p->h_fast.assign(nkinds, FastVdw{});
    bool all_lj = true, all_fast = true;
    double hs_max2 = 0.0;
    for (int32_t k = 0; k < nkinds; ++k) {
        const int32_t b = p->h_offset[k], e = p->h_offset[k + 1];
        FastVdw f{};
        int nlj = 0, nbuck = 0, nhs = 0, nother = 0;
        for (int32_t t = b; t < e; ++t) {
            const DevRule& r = p->h_rules[t];
            if (r.kind == CEG_LENNARDJONES) { ++nlj; f.p0 = 4.0 * r.p0; f.p1 = r.p1; f.p2 = r.p1 * r.p1 * r.p1; f.shift += r.shift; }
            else if (r.kind == CEG_BUCKINGHAM) { ++nbuck; f.p0 = r.p0; f.p1 = r.p1; f.p2 = r.p2; f.shift += r.shift; }
            else if (r.kind == CEG_HARDSPHERE) { ++nhs; f.shift += r.shift; if (present[k]) hs_max2 = std::max(hs_max2, r.p0); }
            else ++nother;
        }
        if (e == b) f.cls = 0;
        else if (nlj == 1 && nbuck == 0 && nhs == 0 && nother == 0) f.cls = 1;
        else if (nbuck == 1 && nlj == 0 && nother == 0) f.cls = 2;
        else f.cls = 3;
        p->h_fast[k] = f;
        if (!present[k]) continue;
        if (f.cls > 1) all_lj = false;
        if (f.cls > 2) all_fast = false;
    }
    p->vdwk = all_lj ? 1 : (all_fast ? 2 : 0);
    // one Buckingham parameter set for every present VdW-active kind (the Na probe of the fixture force field: Na-O only)
    if (p->vdwk == 2) {
        bool first = true, same = true;
        for (int32_t k = 0; k < nkinds && same; ++k) {
            const FastVdw& f = p->h_fast[k];
            if (!present[k] || f.cls == 0) continue;
            if (f.cls != 2) { same = false; break; }
            if (first) { p->bk[0] = f.p0; p->bk[1] = f.p1; p->bk[2] = f.p2; p->bk[3] = f.shift; first = false; }
            else same = f.p0 == p->bk[0] && f.p1 == p->bk[1] && f.p2 == p->bk[2] && f.shift == p->bk[3];
        }
        p->single_buck = same && !first && p->bk[1] > 0.0 && p->bk[2] != 0.0 && std::isfinite(1.0 / p->bk[2]);
    }
    // hard spheres must lie inside the exact-path radius for the fast Buckingham class
    p->r_exact2 = std::max(CEG_R_EXACT2, hs_max2 * (1.0 + 1e-9) + 1e-9);
    if (p->r_exact2 >= p->g.cutoff2) p->vdwk = 0, p->r_exact2 = CEG_R_EXACT2;
    return CEG_OK;
}

// ---- small-block cache for the plan tables.  A plan is a dozen small device arrays; hipMalloc +
// hipFree of those cost ~3 ms per plan, a third of the one-shot VdW build.  Blocks are kept by
// (device, power-of-two size class) and handed back by ceg_plan_destroy after a device
// synchronisation (hipFree's implicit one), so a cached block is never still in use by a kernel.
struct BlockCache {
    std::mutex m;
    std::map<std::pair<int, size_t>, std::vector<void*>> idle;
    std::unordered_map<void*, std::pair<int, size_t>> owner;
    size_t idle_bytes = 0;
};
BlockCache g_blocks;
constexpr size_t BLOCK_CACHE_LIMIT = 256ull << 20;

hipError_t cached_malloc(void** out, size_t bytes)
{
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    size_t cls = 512;
    while (cls < bytes) cls <<= 1;
    std::lock_guard<std::mutex> lock(g_blocks.m);
    auto& list = g_blocks.idle[{dev, cls}];
    if (!list.empty()) {
        *out = list.back();
        list.pop_back();
        g_blocks.idle_bytes -= cls;
        return hipSuccess;
    }
    if (hipError_t e = hipMalloc(out, cls); e != hipSuccess) return e;
    g_blocks.owner[*out] = {dev, cls};
    return hipSuccess;
}

void cached_free(void* ptr)         // caller has synchronised the device
{
    if (!ptr) return;
    std::lock_guard<std::mutex> lock(g_blocks.m);
    auto it = g_blocks.owner.find(ptr);
    if (it == g_blocks.owner.end()) { (void)hipFree(ptr); return; }
    const size_t cls = it->second.second;
    if (g_blocks.idle_bytes + cls > BLOCK_CACHE_LIMIT) {
        g_blocks.owner.erase(it);
        (void)hipFree(ptr);
        return;
    }
    g_blocks.idle[it->second].push_back(ptr);
    g_blocks.idle_bytes += cls;
}

void block_cache_release()
{
    std::lock_guard<std::mutex> lock(g_blocks.m);
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (auto& kv : g_blocks.idle) {
        if (hipSetDevice(kv.first.first) != hipSuccess) continue;
        for (void* ptr : kv.second) {
            g_blocks.owner.erase(ptr);
            (void)hipFree(ptr);
        }
        kv.second.clear();
    }
    g_blocks.idle_bytes = 0;
    if (prev >= 0) (void)hipSetDevice(prev);
}

}  // namespace

namespace ceg_host {          // the same pool for the other translation units of the library (ceg_images.hip)
hipError_t pool_malloc(void** out, size_t bytes) { return cached_malloc(out, bytes); }
void pool_free(void* ptr) { cached_free(ptr); }
}

namespace ceg {
struct ImgBox {
    double mat[9], invmat[9];
    double lo[3], hi[3], bin[3];
    int32_t nb[3];
    int32_t has_rules, has_charge, vdw_only;
    int32_t nkinds;
    uint64_t hasbits[16];
};
hipError_t build_images_device(const ImgBox& B, const double4* d_atoms, const int32_t* d_kind, const int32_t* d_has, int64_t natoms,
                               double4** out_xyzq, int32_t** out_kind, int32_t** out_atom, int32_t** out_binstart, int64_t* out_n);
}

namespace {

// The tables of a plan are a dozen uploads of a few hundred bytes to a few hundred kilobytes; a synchronous hipMemcpy from pageable memory
// costs 15-20 us each whatever its size (a fifth of a plan creation on a new framework).  They go through a page-locked staging area
// instead -- a host memcpy and an asynchronous copy on the null stream each -- and ONE synchronisation at the end of the creation
// (flush_uploads; also before the staging area wraps around).  Larger arrays keep the synchronous copy.  The areas live in a small
// process-wide pool: a plan creation borrows one for its duration (StagingLease), so threads that come and go do not each leave
// page-locked memory behind.
struct UploadStaging {
    char* h = nullptr;
    size_t cap = 0, used = 0;
    bool pending = false;
};
constexpr size_t UPLOAD_STAGING_BYTES = 4u << 20, UPLOAD_STAGED_MAX = 1u << 20;
struct StagingPool {
    std::mutex m;
    std::vector<char*> idle;           // (kept until the process ends: the HIP runtime may be gone when static destructors run)
};
StagingPool g_staging_pool;
thread_local UploadStaging* g_staging = nullptr;      // the area of the plan creation running on this thread, if any

struct StagingLease {
    UploadStaging st;
    UploadStaging* outer;
    StagingLease() : outer(g_staging)
    {
        if (!std::getenv("CEG_HIP_SYNC_UPLOADS")) {
            {
                std::lock_guard<std::mutex> lock(g_staging_pool.m);
                if (!g_staging_pool.idle.empty()) { st.h = g_staging_pool.idle.back(); g_staging_pool.idle.pop_back(); }
            }
            if (!st.h && hipHostMalloc((void**)&st.h, UPLOAD_STAGING_BYTES, hipHostMallocPortable) != hipSuccess) { st.h = nullptr; (void)hipGetLastError(); }
            if (st.h) st.cap = UPLOAD_STAGING_BYTES;
        }
        g_staging = &st;
    }
    ~StagingLease()
    {
        if (st.pending) (void)hipStreamSynchronize(nullptr);
        g_staging = outer;
        if (st.h) {
            std::lock_guard<std::mutex> lock(g_staging_pool.m);
            if (g_staging_pool.idle.size() < 8) g_staging_pool.idle.push_back(st.h);
            else (void)hipHostFree(st.h);
        }
    }
};

inline hipError_t flush_uploads()
{
    UploadStaging* st = g_staging;
    if (!st || !st->pending) return hipSuccess;
    st->pending = false;
    st->used = 0;
    return hipStreamSynchronize(nullptr);
}

template <class T>
int upload(T** dst, const T* src, size_t n)
{
    *dst = nullptr;
    if (n == 0) n = 1;  // keep pointers valid
    HIP_TRY(cached_malloc((void**)dst, n * sizeof(T)));
    if (!src) return CEG_OK;
    const size_t bytes = n * sizeof(T);
    UploadStaging* st = g_staging;
    if (st && st->h && bytes <= UPLOAD_STAGED_MAX) {
        if (st->used + bytes > st->cap) HIP_TRY(flush_uploads());
        memcpy(st->h + st->used, src, bytes);
        HIP_TRY(hipMemcpyAsync(*dst, st->h + st->used, bytes, hipMemcpyHostToDevice, nullptr));
        st->used += (bytes + 255) & ~(size_t)255;
        st->pending = true;
        return CEG_OK;
    }
    HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return CEG_OK;
}

// ---- image cache.  The lattice-image list of a plan depends on the framework (positions, cell), the cutoff, the grid's
// bounding box and, per atom, on what travels with an image: its kind + "has a VdW rule" flag, its charge, and whether atoms
// without a rule are left out (VdW-only plans).  setup_RASPA builds several grids of ONE framework one after the other
// (src/raspa.jl:497-520) and every one-shot call used to enumerate, sort and upload the same images again (1.0-1.4 ms of host time
// for the 11 664-atom workload: as long as a whole rank's compute at N = 8).  Finished lists are kept on the device, keyed by a
// 128-bit hash of everything they depend on, and shared by reference between plans; the cache holds the last few (CEG_HIP_IMAGE_CACHE
// = number of entries, 0 switches it off).
struct ImageSet {
    int device = 0;
    uint64_t key[2] = {0, 0};
    int64_t natoms = 0;             // compared on a hit together with the 128-bit (non-cryptographic) key: a collision must also
    double cutoff2 = 0.0;           // reproduce the atom count, the cutoff and the box to be taken for a hit
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    double4* d_images = nullptr;
    int32_t* d_imgkind = nullptr;
    int32_t* d_imgatom = nullptr;
    int32_t* d_binstart = nullptr;
    ImageBins ib{};                 // geometry of the bins + the pointers above (ib.atoms is per plan)
    size_t bytes = 0;
    ~ImageSet()
    {
        DeviceGuard guard(device);
        (void)hipDeviceSynchronize();          // no kernel still reads the arrays
        for (void* ptr : {(void*)d_images, (void*)d_imgkind, (void*)d_imgatom, (void*)d_binstart}) cached_free(ptr);
    }
};
std::mutex g_image_mutex;
// (heap-allocated and never destroyed: a static destructor would run ~ImageSet -- HIP calls -- after the runtime may be gone at exit)
std::vector<std::shared_ptr<ImageSet>>& g_image_cache = *new std::vector<std::shared_ptr<ImageSet>>();       // most recently used last
std::atomic<long> g_image_hits{0}, g_image_misses{0};

struct Hash128 {
    uint64_t a = 0x243F6A8885A308D3ull, b = 0x13198A2E03707344ull;
    void word(uint64_t w)
    {
        a = (a ^ w) * 0x9E3779B97F4A7C15ull; a ^= a >> 29;
        b = (b + w) * 0xC2B2AE3D27D4EB4Full; b ^= b >> 31; b += a;
    }
    void bytes(const void* ptr, size_t n)
    {
        const unsigned char* c = static_cast<const unsigned char*>(ptr);
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, c + i, 8); word(w); }
        uint64_t w = 0;
        if (i < n) memcpy(&w, c + i, n - i);
        word(w ^ ((uint64_t)n << 56));
    }
    template <class T> void pod(const T& v) { bytes(&v, sizeof v); }
};

int image_cache_capacity()
{
    if (const char* e = std::getenv("CEG_HIP_IMAGE_CACHE")) return std::max(0, atoi(e));
    return 6;
}

void image_cache_release()
{
    std::lock_guard<std::mutex> lock(g_image_mutex);
    g_image_cache.clear();
}

// Expand the atoms into every lattice image that can be within the cutoff of a grid point
// (grid bounding box grown by the cutoff), bin them on a cartesian lattice and upload.
#ifndef CEG_BIN_Z
#define CEG_BIN_Z 1.5
#endif
int build_images(ceg_plan* p)
{
    const Geom& g = p->g;
    const double cutoff = std::sqrt(g.cutoff2);
    const double margin = cutoff * (1.0 + 1e-6) + 1e-6;
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = g.shift[a] - margin;
        hi[a] = g.shift[a] + g.size[a] + margin;
    }
    // bins: ~4.5 A edge in x and y (a tile's neighbourhood is then <= 64 bin rows (bx, by): one row per lane); z is the
    // contiguous direction of a row, where a finer bin only tightens the [bz0, bz1] range a row contributes
    double target[3] = {4.5, 4.5, CEG_BIN_Z};
    if (const char* e = std::getenv("CEG_HIP_BIN_XY")) target[0] = target[1] = std::max(0.5, atof(e));
    if (const char* e = std::getenv("CEG_HIP_BIN_Z")) target[2] = std::max(0.25, atof(e));
    int nb[3];
    double bin[3];
    for (int a = 0; a < 3; ++a) {
        nb[a] = std::max(1, (int)std::floor((hi[a] - lo[a]) / target[a]));
        bin[a] = (hi[a] - lo[a]) / nb[a];
    }
    // "the kind has a VdW rule": with the plan's probe, or with any probe of a multi-probe plan
    const std::vector<int32_t>& off = p->nprobes > 0 ? p->h_offset_union : p->h_offset;
    auto kind_has_rule = [&](int32_t k) { return k >= 0 && k + 1 < (int32_t)off.size() && off[k + 1] > off[k]; };
    // ---- cache look-up: everything the list below is a function of
    Hash128 hk;
    std::shared_ptr<ImageSet> hit;
    const int cap = image_cache_capacity();
    if (cap > 0) {
        hk.pod(p->device); hk.pod(p->natoms);
        hk.bytes(g.mat, sizeof g.mat); hk.bytes(g.invmat, sizeof g.invmat);
        hk.bytes(lo, sizeof lo); hk.bytes(hi, sizeof hi); hk.bytes(target, sizeof target);
        hk.bytes(p->h_pos.data(), p->h_pos.size() * sizeof(double));
        const int32_t mode = (p->has_rules ? 1 : 0) | (p->has_charge ? 2 : 0);
        hk.pod(mode);
        if (p->has_charge) hk.bytes(p->h_charge.data(), p->h_charge.size() * sizeof(double));
        if (p->has_rules)
            for (int64_t a = 0; a < p->natoms; ++a) {
                const int32_t k = p->h_kind[a];
                hk.pod((int32_t)(k < 0 ? -1 : (k | (kind_has_rule(k) ? (1 << 25) : 0))));
            }
        std::lock_guard<std::mutex> lock(g_image_mutex);
        for (size_t t = 0; t < g_image_cache.size(); ++t) {
            const std::shared_ptr<ImageSet> c = g_image_cache[t];
            if (c->device == p->device && c->key[0] == hk.a && c->key[1] == hk.b && c->natoms == p->natoms && c->cutoff2 == g.cutoff2 &&
                memcmp(c->lo, lo, sizeof lo) == 0 && memcmp(c->hi, hi, sizeof hi) == 0) {
                hit = c;
                std::rotate(g_image_cache.begin() + t, g_image_cache.begin() + t + 1, g_image_cache.end());      // most recently used last
                break;
            }
        }
    }
    if (hit) {
        g_image_hits.fetch_add(1);
        p->images = hit;
        const ImageSet& c = *hit;
        p->d_images = c.d_images; p->d_imgkind = c.d_imgkind; p->d_imgatom = c.d_imgatom; p->d_binstart = c.d_binstart;
        p->ib = c.ib;
        p->ib.kind = p->has_rules ? p->d_imgkind : nullptr;
        p->ib.atoms = p->d_atoms;
        p->images_built = true;
        p->images_from_cache = true;
        return CEG_OK;
    }
    if (cap > 0) g_image_misses.fetch_add(1);
    if (!std::getenv("CEG_HIP_IMAGES_ON_HOST")) {
        // round 4: the list is built on the device from the atom table that is already there (ceg_images.hip) -- byte-identical to
        // the host loop below, 1.0-1.4 ms -> ~0.2 ms for the 11 664-atom framework; CEG_HIP_IMAGES_ON_HOST=1 keeps the host build
        ceg::ImgBox B{};
        memcpy(B.mat, g.mat, sizeof B.mat);
        memcpy(B.invmat, g.invmat, sizeof B.invmat);
        for (int a = 0; a < 3; ++a) { B.lo[a] = lo[a]; B.hi[a] = hi[a]; B.bin[a] = bin[a]; B.nb[a] = nb[a]; }
        B.has_rules = p->has_rules ? 1 : 0;
        B.has_charge = p->has_charge ? 1 : 0;
        B.vdw_only = (p->has_rules && !p->has_charge) ? 1 : 0;
        B.nkinds = p->nkinds;
        int32_t* d_has = nullptr;
        if (p->has_rules && p->nkinds <= 1024) {
            for (int32_t k = 0; k < p->nkinds; ++k)
                if (kind_has_rule(k)) B.hasbits[k >> 6] |= 1ull << (k & 63);
        } else if (p->has_rules) {
            std::vector<int32_t> has((size_t)std::max(p->nkinds, 1), 0);
            for (int32_t k = 0; k < p->nkinds; ++k) has[k] = kind_has_rule(k) ? 1 : 0;
            if (int rc = upload(&d_has, has.data(), has.size())) return rc;
        }
        auto set = std::make_shared<ImageSet>();
        set->device = p->device;
        set->key[0] = hk.a; set->key[1] = hk.b;
        set->natoms = p->natoms; set->cutoff2 = g.cutoff2;
        memcpy(set->lo, lo, sizeof lo); memcpy(set->hi, hi, sizeof hi);
        int64_t n = 0;
        const hipError_t e = ceg::build_images_device(B, p->d_atoms, p->has_rules ? p->d_kind : nullptr, d_has, p->natoms, &set->d_images, &set->d_imgkind,
                                                      &set->d_imgatom, &set->d_binstart, &n);
        if (d_has) cached_free(d_has);          // (build_images_device has synchronised)
        if (e != hipSuccess) return fail(CEG_ERR_HIP, "building the lattice-image list on the device failed: %s", hipGetErrorString(e));
        set->bytes = (size_t)n * (sizeof(double4) + 2 * sizeof(int32_t)) + (size_t)(nb[0] * nb[1] * nb[2] + 1) * sizeof(int32_t);
        p->images = set;
        p->d_images = set->d_images; p->d_imgkind = set->d_imgkind; p->d_imgatom = set->d_imgatom; p->d_binstart = set->d_binstart;
        ImageBins& ib = p->ib;
        ib.xyzq = p->d_images;
        ib.kind = p->has_rules ? p->d_imgkind : nullptr;
        ib.atom = p->d_imgatom;
        ib.atoms = p->d_atoms;
        ib.bin_start = p->d_binstart;
        for (int a = 0; a < 3; ++a) {
            ib.lo[a] = lo[a];
            ib.bin[a] = bin[a];
            ib.inv_bin[a] = 1.0 / bin[a];
            ib.nb[a] = nb[a];
        }
        ib.nimages = (int32_t)n;
        p->images_built = true;
        set->ib = ib;
        if (cap > 0) {
            std::lock_guard<std::mutex> lock(g_image_mutex);
            g_image_cache.push_back(set);
            while ((int)g_image_cache.size() > cap) g_image_cache.erase(g_image_cache.begin());
        }
        return CEG_OK;
    }
    struct Img { double x, y, z, q; int32_t kind; int32_t bin; int32_t atom; };
    std::vector<Img> imgs;
    imgs.reserve((size_t)p->natoms * 8);
    const double* M = g.mat;
    const double* I = g.invmat;
    for (int64_t a = 0; a < p->natoms; ++a) {
        if (p->has_rules && !p->has_charge) {
            // a VdW-only plan (create_grid_vdw: one probe atom against the framework): atoms whose kind has no rule for the probe
            // contribute exact zeros (an empty rule run, src/interactions.jl:599-610) -- they need not be staged at all
            // (Si / Al against Ar or Na in the fixture force field: a third of the framework)
            if (!kind_has_rule(p->h_kind[a])) continue;
        }
        const double pa[3] = {p->h_pos[3 * a], p->h_pos[3 * a + 1], p->h_pos[3 * a + 2]};
        double fmin[3] = {1e300, 1e300, 1e300}, fmax[3] = {-1e300, -1e300, -1e300};
        for (int c = 0; c < 8; ++c) {
            const double d[3] = {((c & 1) ? hi[0] : lo[0]) - pa[0], ((c & 2) ? hi[1] : lo[1]) - pa[1],
                                 ((c & 4) ? hi[2] : lo[2]) - pa[2]};
            double f[3];
            matvec(I, d, f);
            for (int q = 0; q < 3; ++q) {
                fmin[q] = std::min(fmin[q], f[q]);
                fmax[q] = std::max(fmax[q], f[q]);
            }
        }
        int n0[3], n1[3];
        // an image pa + M n lies in the box only if n = I (P - pa) is inside the fractional hull of
        // the box corners, i.e. fmin <= n <= fmax componentwise (1e-9 guards the rounding of I*d)
        for (int q = 0; q < 3; ++q) {
            n0[q] = (int)std::ceil(fmin[q] - 1e-9);
            n1[q] = (int)std::floor(fmax[q] + 1e-9);
        }
        for (int nx = n0[0]; nx <= n1[0]; ++nx)
            for (int ny = n0[1]; ny <= n1[1]; ++ny)
                for (int nz = n0[2]; nz <= n1[2]; ++nz) {
                    const double P[3] = {pa[0] + (nx * M[0] + ny * M[3] + nz * M[6]),
                                         pa[1] + (nx * M[1] + ny * M[4] + nz * M[7]),
                                         pa[2] + (nx * M[2] + ny * M[5] + nz * M[8])};
                    if (P[0] < lo[0] || P[0] > hi[0] || P[1] < lo[1] || P[1] > hi[1] || P[2] < lo[2] ||
                        P[2] > hi[2])
                        continue;
                    int b[3];
                    for (int q = 0; q < 3; ++q) {
                        b[q] = (int)std::floor((P[q] - lo[q]) / bin[q]);
                        b[q] = std::min(std::max(b[q], 0), nb[q] - 1);
                    }
                    Img im;
                    im.x = P[0]; im.y = P[1]; im.z = P[2];
                    im.q = p->has_charge ? p->h_charge[a] : 0.0;
                    // kind (low 24 bits) + bit 25: the kind has at least one VdW rule (= META_HASVDW of the kernel), so that the
                    // staging loop does not have to look the rule offsets up per image
                    im.kind = -1;
                    if (p->has_rules) {
                        const int32_t k = p->h_kind[a];
                        im.kind = k < 0 ? -1 : (k | (kind_has_rule(k) ? (1 << 25) : 0));
                    }
                    im.bin = (b[0] * nb[1] + b[1]) * nb[2] + b[2];
                    im.atom = (int32_t)a;
                    imgs.push_back(im);
                }
    }
    if (imgs.size() > 0x7ffffff0ull) return fail(CEG_ERR_UNSUPPORTED, "too many lattice images");
    const size_t nbins = (size_t)nb[0] * nb[1] * nb[2];
    std::vector<int32_t> start(nbins + 1, 0);
    for (const Img& im : imgs) start[im.bin + 1]++;
    for (size_t b = 0; b < nbins; ++b) start[b + 1] += start[b];
    std::vector<int32_t> cursor(start.begin(), start.end() - 1);
    std::vector<double4> xyzq(imgs.size());
    std::vector<int32_t> kind(imgs.size());
    std::vector<int32_t> atom(imgs.size());
    for (const Img& im : imgs) {       // stable: atom order, then lattice order, inside each bin
        const int32_t s = cursor[im.bin]++;
        xyzq[s] = make_double4(im.x, im.y, im.z, im.q);
        kind[s] = im.kind;
        atom[s] = im.atom;
    }
    auto set = std::make_shared<ImageSet>();
    set->device = p->device;
    set->key[0] = hk.a; set->key[1] = hk.b;
    set->natoms = p->natoms; set->cutoff2 = g.cutoff2;
    memcpy(set->lo, lo, sizeof lo); memcpy(set->hi, hi, sizeof hi);
    if (int rc = upload(&set->d_images, xyzq.data(), xyzq.size())) return rc;
    if (int rc = upload(&set->d_imgkind, kind.data(), kind.size())) return rc;
    if (int rc = upload(&set->d_imgatom, atom.data(), atom.size())) return rc;
    if (int rc = upload(&set->d_binstart, start.data(), start.size())) return rc;
    set->bytes = xyzq.size() * sizeof(double4) + (kind.size() + atom.size() + start.size()) * sizeof(int32_t);
    p->images = set;
    p->d_images = set->d_images; p->d_imgkind = set->d_imgkind; p->d_imgatom = set->d_imgatom; p->d_binstart = set->d_binstart;
    ImageBins& ib = p->ib;
    ib.xyzq = p->d_images;
    ib.kind = p->has_rules ? p->d_imgkind : nullptr;
    ib.atom = p->d_imgatom;
    ib.atoms = p->d_atoms;
    ib.bin_start = p->d_binstart;
    for (int a = 0; a < 3; ++a) {
        ib.lo[a] = lo[a];
        ib.bin[a] = bin[a];
        ib.inv_bin[a] = 1.0 / bin[a];
        ib.nb[a] = nb[a];
    }
    ib.nimages = (int32_t)imgs.size();
    p->images_built = true;
    set->ib = ib;
    if (cap > 0) {
        std::lock_guard<std::mutex> lock(g_image_mutex);
        g_image_cache.push_back(set);
        while ((int)g_image_cache.size() > cap) g_image_cache.erase(g_image_cache.begin());     // (freed when the last plan lets go)
    }
    return CEG_OK;
}

// Function tables of the fast real-space Ewald term (csrc/ceg_math.h): erfcx on
// [alpha*R_EXACT, alpha*cutoff] as ERFCX_TAB_N degree-5 pieces (Chebyshev-node interpolation in
// long double), and 2^(j/64).  Returns false (fast path disabled, libm-grade erfc used instead)
// if the fit does not reach 1e-14.
bool build_ewald_tables_uncached(double alpha, double cutoff2, std::vector<double>& tab, std::vector<double>& exp2_tab,
                                 double* inv_h_out, double* mx0_inv_h_out);

// the long-double fit takes ~2 ms; every Coulomb grid of a run uses the same (alpha, cutoff)
bool build_ewald_tables(double alpha, double cutoff2, std::vector<double>& tab, std::vector<double>& exp2_tab,
                        double* inv_h_out, double* mx0_inv_h_out)
{
    struct Memo { bool valid = false, ok = false; double alpha = 0, cutoff2 = 0, inv_h = 0, mx0 = 0; std::vector<double> tab, e2; };
    static std::mutex m;
    static Memo memo;
    std::lock_guard<std::mutex> lock(m);
    if (!(memo.valid && memo.alpha == alpha && memo.cutoff2 == cutoff2)) {
        memo.tab.clear(); memo.e2.clear();
        memo.ok = build_ewald_tables_uncached(alpha, cutoff2, memo.tab, memo.e2, &memo.inv_h, &memo.mx0);
        memo.alpha = alpha; memo.cutoff2 = cutoff2; memo.valid = true;
    }
    tab = memo.tab; exp2_tab = memo.e2; *inv_h_out = memo.inv_h; *mx0_inv_h_out = memo.mx0;
    return memo.ok;
}

bool build_ewald_tables_uncached(double alpha, double cutoff2, std::vector<double>& tab, std::vector<double>& exp2_tab,
                                 double* inv_h_out, double* mx0_inv_h_out)
{
    const int N = CEG_ERFCX_TAB_N;
    const long double x0 = (long double)alpha * std::sqrt((long double)CEG_R_EXACT2) * (1.0L - 1e-6L);
    const long double x1 = (long double)alpha * std::sqrt((long double)cutoff2 * (1.0L + 2e-9L)) * (1.0L + 1e-6L);
    if (!(x1 > x0) || x1 > 5.0L * (1.0L - 1e-9L)) return false;   // 5 = ERFCX_XMAX of the slow path's polynomial
    const long double h = (x1 - x0) / N;
    auto erfcx = [](long double x) { return expl(x * x) * erfcl(x); };
    tab.assign((size_t)N * 6, 0.0);
    const long double PI = 3.14159265358979323846264338327950288L;
    long double node[6];
    for (int k = 0; k < 6; ++k) node[k] = 0.5L + 0.5L * cosl(PI * (k + 0.5L) / 6.0L);   // s in [0, 1]
    double worst = 0.0;
    for (int i = 0; i < N; ++i) {
        const long double lo = x0 + i * h;
        long double V[6][7];
        for (int r = 0; r < 6; ++r) {
            long double pw = 1.0L;
            for (int c = 0; c < 6; ++c) { V[r][c] = pw; pw *= node[r]; }
            V[r][6] = erfcx(lo + node[r] * h);
        }
        for (int c = 0; c < 6; ++c) {                       // Gauss-Jordan with partial pivoting
            int piv = c;
            for (int r = c + 1; r < 6; ++r) if (fabsl(V[r][c]) > fabsl(V[piv][c])) piv = r;
            for (int q = 0; q < 7; ++q) std::swap(V[c][q], V[piv][q]);
            const long double d = V[c][c];
            for (int q = 0; q < 7; ++q) V[c][q] /= d;
            for (int r = 0; r < 6; ++r) if (r != c) {
                const long double f = V[r][c];
                for (int q = 0; q < 7; ++q) V[r][q] -= f * V[c][q];
            }
        }
        for (int c = 0; c < 6; ++c) tab[(size_t)i * 6 + c] = (double)V[c][6];
        for (int t = 0; t <= 8; ++t) {                      // accuracy check with a double Horner
            const double sl = t / 8.0;
            double pv = tab[(size_t)i * 6 + 5];
            for (int c = 4; c >= 0; --c) pv = pv * sl + tab[(size_t)i * 6 + c];
            const long double ref = erfcx(lo + (long double)sl * h);
            worst = std::max(worst, (double)fabsl(((long double)pv - ref) / ref));
        }
    }
    if (!(worst < 1e-14)) return false;
    exp2_tab.resize(64);
    for (int j = 0; j < 64; ++j) exp2_tab[j] = (double)exp2l((long double)j / 64.0L);
    *inv_h_out = (double)(1.0L / h);
    *mx0_inv_h_out = (double)(-x0 / h);
    return true;
}

// r^2-indexed tables of B0(s) = erfc(alpha sqrt(s))/sqrt(s) and C(s) = 2 alpha/sqrt(pi) exp(-alpha^2 s) (ceg_internal.h,
// CEG_EW2_*): per interval a degree-5 interpolant at the Chebyshev nodes of the interval, in long double, expressed in
// t = s - s_lo (the interval's lower end: clearing the low bits of s gives it).  Returns false if the range needs more than CEG_EW2_NI_MAX intervals or a polynomial misses `tol`
// (relative) anywhere -- the caller then keeps the erfcx / libm variants.
// `which` 0: the two Ewald functions (record = CEG_EW2_STRIDE doubles, tolerance relative);  1: G0(s) = A exp(-B sqrt(s)) of a
// Buckingham class with alpha := B, record = CEG_BK2_STRIDE doubles, tolerance relative to G0 at the start of the table + the
// dispersion term C/s^3 it is added to (exp(-B r) spans 17 decades up to the cutoff).
bool build_ew2_table_uncached(double alpha, double r_exact2, double cutoff2, std::vector<double>& tab, int32_t* base_out,
                              int32_t* ni_out, double* worst_out, int which = 0, double bkA = 0.0, double bkC = 0.0)
{
    const int SHIFT = which == 0 ? CEG_EW2_SHIFT : CEG_BK2_SHIFT, ni_max = which == 0 ? CEG_EW2_NI_MAX : CEG_BK2_NI_MAX;
    auto key_of = [SHIFT](double s) { uint64_t b; memcpy(&b, &s, 8); return (int32_t)((uint32_t)(b >> 32) >> SHIFT); };
    if (!(alpha > 0.0) || !(r_exact2 >= 1.0) || !(cutoff2 > r_exact2)) return false;
    const int32_t base = key_of(r_exact2), last = key_of(cutoff2 * (1.0 + 4e-9) + 4e-9);
    const int32_t ni = last - base + 1;
    if (ni < 1 || ni > ni_max) return false;
    const long double a = alpha, ka = 2.0L * a / sqrtl(3.14159265358979323846264338327950288L);
    auto B0 = [&](long double s) { const long double r = sqrtl(s); return which == 0 ? erfcl(a * r) / r : (long double)bkA * expl(-a * r); };
    auto Cf = [&](long double s) { return ka * expl(-a * a * s); };
    const int nf = which == 0 ? 2 : 1, stride = which == 0 ? CEG_EW2_STRIDE : CEG_BK2_STRIDE;
    const int ND = which == 0 ? 7 : CEG_BK2_ND;            // coefficients per polynomial: degree 6 (Ewald pair, 14 doubles) / degree 7
    const long double PI = 3.14159265358979323846264338327950288L;
    long double node[8];
    for (int k = 0; k < ND; ++k) node[k] = cosl(PI * (k + 0.5L) / (long double)ND);       // u in [-1, 1]
    tab.assign((size_t)ni * stride, 0.0);
    double worst = 0.0;
    for (int32_t i = 0; i < ni; ++i) {
        const uint64_t lo_bits = (uint64_t)(uint32_t)((base + i) << SHIFT) << 32;
        const uint64_t mid_bits = lo_bits | ((uint64_t)1 << (32 + SHIFT - 1));
        const uint64_t hi_bits = (uint64_t)(uint32_t)((base + i + 1) << SHIFT) << 32;
        double s_lo, s_mid, s_hi;
        memcpy(&s_lo, &lo_bits, 8); memcpy(&s_mid, &mid_bits, 8); memcpy(&s_hi, &hi_bits, 8);
        const long double hh = 0.5L * ((long double)s_hi - (long double)s_lo);           // half width; s_mid is the exact centre
        for (int f = 0; f < nf; ++f) {
            long double V[8][9];
            for (int r = 0; r < ND; ++r) {
                long double pw = 1.0L;
                for (int c = 0; c < ND; ++c) { V[r][c] = pw; pw *= node[r]; }
                const long double s = (long double)s_mid + node[r] * hh;
                V[r][ND] = f == 0 ? B0(s) : Cf(s);
            }
            for (int c = 0; c < ND; ++c) {                      // Gauss-Jordan with partial pivoting
                int piv = c;
                for (int r = c + 1; r < ND; ++r) if (fabsl(V[r][c]) > fabsl(V[piv][c])) piv = r;
                for (int q = 0; q <= ND; ++q) std::swap(V[c][q], V[piv][q]);
                const long double d = V[c][c];
                for (int q = 0; q <= ND; ++q) V[c][q] /= d;
                for (int r = 0; r < ND; ++r) if (r != c) {
                    const long double g = V[r][c];
                    for (int q = 0; q <= ND; ++q) V[r][q] -= g * V[c][q];
                }
            }
            // P(u), u = (t - hh)/hh with t = s - s_lo  ->  coefficients in t (binomial expansion in long double)
            long double cu[8], ct[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            long double sc = 1.0L;
            for (int c = 0; c < ND; ++c) { cu[c] = V[c][ND] * sc; sc /= hh; }             // in (t - hh)
            for (int c = 0; c < ND; ++c) {                                                 // (t - hh)^c = sum_k C(c,k) t^k (-hh)^(c-k)
                long double binom = 1.0L;
                for (int k = 0; k <= c; ++k) {
                    ct[k] += cu[c] * binom * powl(-hh, c - k);
                    binom = binom * (c - k) / (k + 1);
                }
            }
            double* co = &tab[(size_t)i * stride + ND * f];
            for (int c = 0; c < ND; ++c) co[c] = (double)ct[c];
            for (int q = 0; q <= 16; ++q) {                      // accuracy check with the kernel's double Horner
                const double t = (double)(((long double)q / 16.0L) * 2.0L * hh * (1.0L - 1e-12L));
                double pv = co[ND - 1];
                for (int c = ND - 2; c >= 0; --c) pv = __builtin_fma(pv, t, co[c]);
                const long double s = (long double)s_lo + (long double)t;
                const long double ref = f == 0 ? B0(s) : Cf(s);
                const long double scale = which == 0 ? fabsl(ref) : fabsl(ref) + fabsl((long double)bkC) / (s * s * s);
                worst = std::max(worst, (double)fabsl(((long double)pv - ref) / scale));
            }
        }
    }
    *base_out = base; *ni_out = ni; *worst_out = worst;
    return worst < 5e-11;
}

bool build_ew2_table(double alpha, double r_exact2, double cutoff2, std::vector<double>& tab, int32_t* base_out, int32_t* ni_out)
{
    struct Memo { bool valid = false, ok = false; double alpha = 0, r_exact2 = 0, cutoff2 = 0; int32_t base = 0, ni = 0; std::vector<double> tab; };
    static std::mutex m;
    static Memo memo;
    std::lock_guard<std::mutex> lock(m);
    if (!(memo.valid && memo.alpha == alpha && memo.r_exact2 == r_exact2 && memo.cutoff2 == cutoff2)) {
        memo.tab.clear();
        double worst = 0.0;
        memo.ok = build_ew2_table_uncached(alpha, r_exact2, cutoff2, memo.tab, &memo.base, &memo.ni, &worst);
        memo.alpha = alpha; memo.r_exact2 = r_exact2; memo.cutoff2 = cutoff2; memo.valid = true;
        if (std::getenv("CEG_HIP_TRACE"))
            fprintf(stderr, "[ceg_hip] r^2-indexed Ewald table: %d intervals, worst relative error %.2e -> %s\n", memo.ni, worst,
                    memo.ok ? "used" : "not used");
    }
    tab = memo.tab; *base_out = memo.base; *ni_out = memo.ni;
    return memo.ok;
}

bool build_bk2_table(double A, double B, double C, double r_exact2, double cutoff2, std::vector<double>& tab, int32_t* base_out, int32_t* ni_out)
{
    struct Memo { bool valid = false, ok = false; double A = 0, B = 0, C = 0, r_exact2 = 0, cutoff2 = 0; int32_t base = 0, ni = 0; std::vector<double> tab; };
    static std::mutex m;
    static Memo memo;
    std::lock_guard<std::mutex> lock(m);
    if (!(memo.valid && memo.A == A && memo.B == B && memo.C == C && memo.r_exact2 == r_exact2 && memo.cutoff2 == cutoff2)) {
        memo.tab.clear();
        double worst = 0.0;
        memo.ok = build_ew2_table_uncached(B, r_exact2, cutoff2, memo.tab, &memo.base, &memo.ni, &worst, 1, A, C) && worst < CEG_BK2_TOL;
        memo.A = A; memo.B = B; memo.C = C; memo.r_exact2 = r_exact2; memo.cutoff2 = cutoff2; memo.valid = true;
        if (std::getenv("CEG_HIP_TRACE"))
            fprintf(stderr, "[ceg_hip] r^2-indexed Buckingham table: %d intervals, worst error %.2e of the pair energy -> %s\n", memo.ni, worst,
                    memo.ok ? "used" : "not used");
    }
    tab = memo.tab; *base_out = memo.base; *ni_out = memo.ni;
    return memo.ok;
}

int check_common(const double* pos, int64_t natoms, const double* mat, const double* invmat,
                 const int32_t* dims, const double* size, const double* shift, const double* delta)
{
    if (!pos && natoms > 0) return fail(CEG_ERR_INVALID, "pos is NULL");
    if (natoms < 0) return fail(CEG_ERR_INVALID, "natoms < 0");
    if (!mat || !invmat || !dims || !size || !shift || !delta)
        return fail(CEG_ERR_INVALID, "NULL geometry argument");
    for (int a = 0; a < 3; ++a)
        if (dims[a] < 1) return fail(CEG_ERR_INVALID, "dims[%d] = %d < 1", a, dims[a]);
    return CEG_OK;
}

}  // namespace

// out[i] = kind of image i | META_HASVDW iff has[kind] (the flag bit is bit 25: what build_images sets for the union of the probes)
__global__ void k_probe_image_flags(const int32_t* __restrict__ kind_union, const int32_t* __restrict__ has, int32_t nkinds, int32_t* __restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t kw = kind_union[i];
    if (kw < 0) { out[i] = kw; return; }
    const int32_t k = kw & ((1 << 24) - 1);
    out[i] = k | ((k < nkinds && has[k]) ? (1 << 25) : 0);
}

static int probe_image_flags(const int32_t* d_union, const int32_t* d_has, int32_t nkinds, int32_t* d_out, int64_t n)
{
    if (n <= 0) return CEG_OK;
    hipLaunchKernelGGL(k_probe_image_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d_union, d_has, nkinds, d_out, n);
    return hipGetLastError() == hipSuccess ? CEG_OK : fail(CEG_ERR_HIP, "image-flag kernel launch failed");
}

// nprobes == 0: the ordinary plan of ceg_plan_create (rules / rule_offset may be NULL: Coulomb only);
// nprobes >= 1: a multi-probe plan, mrules / moffset [nprobes], probes of any rule class ceg_plan_create takes
static int create_impl(ceg_plan_t** plan, int32_t device,
                       const double* pos, const int64_t* atomkind, const double* charge,
                       int64_t natoms,
                       const double mat[9], const double invmat[9],
                       int32_t ortho, double safemin2, double cutoff2,
                       const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds,
                       int32_t nprobes, const ceg_rule_t* const* mrules, const int32_t* const* moffset,
                       double alpha,
                       const int32_t dims[3], const double size[3], const double shift[3],
                       const double delta[3])
{
    if (!plan) return fail(CEG_ERR_INVALID, "plan is NULL");
    *plan = nullptr;
    if (int rc = check_common(pos, natoms, mat, invmat, dims, size, shift, delta)) return rc;
    if (nprobes < 0 || nprobes > CEG_MAX_PROBES) return fail(CEG_ERR_INVALID, "nprobes = %d outside 1..%d", nprobes, CEG_MAX_PROBES);
    if (nprobes > 0) {
        if (!mrules || !moffset || !atomkind || nkinds <= 0) return fail(CEG_ERR_INVALID, "rule tables / atomkind missing");
        for (int q = 0; q < nprobes; ++q)
            if (!mrules[q] || !moffset[q]) return fail(CEG_ERR_INVALID, "rule table of probe %d is NULL", q);
        rules = mrules[0];
        rule_offset = moffset[0];
    }
    const bool trace = std::getenv("CEG_HIP_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (trace)
            fprintf(stderr, "[ceg plan] %-28s %8.3f ms\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    const int ndev = ceg_device_count();
    if (ndev <= 0) return fail(CEG_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(CEG_ERR_NO_DEVICE, "device %d not present (%d devices)", device, ndev);
    const bool has_rules = rules && rule_offset && atomkind && nkinds > 0;
    if (!has_rules && !charge) return fail(CEG_ERR_INVALID, "plan needs rules+atomkind and/or charges");

    ceg_plan* p = new ceg_plan();
    p->device = device;
    p->natoms = natoms;
    p->has_rules = has_rules;
    p->has_charge = charge != nullptr;
    Geom& g = p->g;
    memcpy(g.mat, mat, sizeof g.mat);
    memcpy(g.invmat, invmat, sizeof g.invmat);
    for (int a = 0; a < 3; ++a) {
        g.size[a] = size[a];
        g.shift[a] = shift[a];
        g.delta[a] = delta[a];
        g.dims[a] = dims[a];
    }
    g.ortho = ortho ? 1 : 0;
    g.safemin2 = safemin2;
    g.cutoff2 = cutoff2;
    g.alpha = alpha;
    g.diag = (mat[1] == 0 && mat[2] == 0 && mat[3] == 0 && mat[5] == 0 && mat[6] == 0 && mat[7] == 0) ? 1 : 0;

    p->h_pos.assign(pos, pos + 3 * natoms);
    if (charge) p->h_charge.assign(charge, charge + natoms);
    p->h_kind.assign(natoms, -1);
    if (has_rules) {
        for (int64_t a = 0; a < natoms; ++a) {
            const int64_t k = atomkind[a];
            if (k < 1 || k > nkinds) {
                delete p;
                return fail(CEG_ERR_INVALID, "atomkind[%lld] = %lld outside 1..%d", (long long)a, (long long)k, nkinds);
            }
            p->h_kind[a] = (int32_t)(k - 1);
        }
        if (int rc = convert_rules(p, rules, rule_offset, nkinds)) {
            delete p;
            return rc;
        }
        if (nprobes > 0) {
            // every probe through the same classification; the plan keeps probe 0 in its single-probe fields and the union of the
            // probes' "kind has a rule" sets in h_offset (what build_images turns into the per-image flag)
            p->nprobes = nprobes;
            p->probes.resize(nprobes);
            std::vector<int32_t> any(nkinds, 0);
            double r_exact2 = CEG_R_EXACT2;
            for (int q = 0; q < nprobes; ++q) {
                if (q > 0)
                    if (int rc = convert_rules(p, mrules[q], moffset[q], nkinds)) { delete p; return rc; }
                // Probes of any rule class share the plan (round 4: the grids of Na + CO2, src/raspa.jl:497-520, from one image list).
                // The Lennard-Jones-only ones share accumulating loops (ceg_plan_build_multi); a probe of another class -- a
                // Buckingham cation -- runs the single-probe kernels of its class from the same list, alone or fused with the Coulomb grid.
                ceg_plan::ProbeTab& t = p->probes[q];
                t.rules = p->h_rules; t.offset = p->h_offset; t.fast = p->h_fast;
                t.vdwk = p->vdwk; t.single_buck = p->single_buck;
                for (int c = 0; c < 4; ++c) t.bk[c] = p->bk[c];
                r_exact2 = std::max(r_exact2, p->r_exact2);          // ONE exact-path radius for the plan: the largest any probe asks for
                for (int32_t k = 0; k < nkinds; ++k) any[k] |= (t.offset[k + 1] > t.offset[k]) ? 1 : 0;
            }
            p->r_exact2 = r_exact2;
            if (!(r_exact2 < p->g.cutoff2)) {
                delete p;
                return fail(CEG_ERR_UNSUPPORTED, "a hard sphere of one of the probes reaches the cutoff: build that grid with a plan of its own");
            }
            p->h_rules = p->probes[0].rules; p->h_fast = p->probes[0].fast;
            p->vdwk = p->probes[0].vdwk; p->single_buck = p->probes[0].single_buck;
            for (int c = 0; c < 4; ++c) p->bk[c] = p->probes[0].bk[c];
            p->h_offset_union.assign(nkinds + 1, 0);
            for (int32_t k = 0; k < nkinds; ++k) p->h_offset_union[k + 1] = p->h_offset_union[k] + any[k];
            p->h_offset = p->probes[0].offset;
        }
    }
    // culling needs: finite cutoff, every perpendicular width >= 2*cutoff (two images of one atom
    // are then never both inside the cutoff), which ProbeSystem guarantees (src/probes.jl:24)
    double w[3];
    perpendicular_widths(g.mat, w);
    const double cutoff = std::sqrt(cutoff2);
    p->can_cull = std::isfinite(cutoff2) && cutoff2 > 0 &&
                  std::min(w[0], std::min(w[1], w[2])) >= 2.0 * cutoff * (1.0 - 1e-12);

    stamp("host: rules, geometry");
    DeviceGuard guard(device);
    if (!guard.ok) {
        delete p;
        return fail(CEG_ERR_HIP, "hipSetDevice(%d) failed", device);
    }
    StagingLease lease;                 // the page-locked area the uploads below go through
    std::vector<double4> xyzq((size_t)natoms);
    for (int64_t a = 0; a < natoms; ++a)
        xyzq[a] = make_double4(pos[3 * a], pos[3 * a + 1], pos[3 * a + 2], charge ? charge[a] : 0.0);
    int rc = upload(&p->d_atoms, xyzq.data(), xyzq.size());
    if (!rc) rc = upload(&p->d_kind, p->h_kind.data(), p->h_kind.size());
    if (!rc) rc = upload(&p->d_rules, p->h_rules.data(), p->h_rules.size());
    if (!rc) {
        if (p->h_offset.empty()) p->h_offset.assign(1, 0);
        rc = upload(&p->d_offset, p->h_offset.data(), p->h_offset.size());
    }
    stamp("atom / rule tables uploaded");
    if (!rc && p->can_cull) rc = build_images(p);
    stamp(p->images_from_cache ? "images: cache hit" : "images built + uploaded");
    if (!rc && p->can_cull) {
        PlanConst hc{};
        hc.g = p->g;
        hc.ib = p->ib;
        hc.rt = RuleTable{p->d_rules, p->d_offset, p->nkinds};
        hc.alpha2 = alpha * alpha;
        hc.r_exact2 = p->r_exact2;
        {   // |f_k(centre - image)| <= |row_k(invmat)| (reach of a kept image from the tile centre) for every image a 4x4x4 tile can
            // keep; if that plus the fractional half-extent of the tile stays below 1/2 on all three axes the kernel skips the test
            // (the tile's extent follows from size / dims, which is what grid_coord steps by -- NOT from the caller's `delta`, which
            // only scales the derivative channels, grids.jl:126-133, and may be anything)
            const double hx = 1.5 * size[0] / dims[0], hy = 1.5 * size[1] / dims[1], hz = 1.5 * size[2] / dims[2];
            const double reach = std::sqrt(cutoff2 * (1.0 + 1e-9) + 1e-9) + std::sqrt(hx * hx + hy * hy + hz * hz);
            bool all = true;
            for (int k = 0; k < 3; ++k) {
                const double rown = std::sqrt(invmat[k] * invmat[k] + invmat[k + 3] * invmat[k + 3] + invmat[k + 6] * invmat[k + 6]);
                const double ek = std::fabs(invmat[k]) * hx + std::fabs(invmat[k + 3]) * hy + std::fabs(invmat[k + 6]) * hz + 1e-9;
                if (!(rown * reach + ek < 0.5 - 1e-6)) all = false;
            }
            hc.all_simple = all ? 1 : 0;
        }
        if (p->h_fast.empty()) p->h_fast.assign(1, FastVdw{});
        rc = upload(&p->d_fast, p->h_fast.data(), p->h_fast.size());
        hc.fastvdw = p->d_fast;
        p->fast_ewald = false;
        bool any_vdwk2 = p->vdwk == 2;
        for (const auto& t : p->probes) any_vdwk2 = any_vdwk2 || t.vdwk == 2;
        if (!rc && ((p->has_charge && std::isfinite(alpha) && alpha > 0) || any_vdwk2)) {
            std::vector<double> tab, e2;
            double inv_h = 0, mx0 = 0;
            const bool want_ewald = p->has_charge && std::isfinite(alpha) && alpha > 0;
            const bool ok = build_ewald_tables(want_ewald ? alpha : 0.25, cutoff2, tab, e2, &inv_h, &mx0);
            if (!ok && p->vdwk == 2) p->vdwk = 0;       // no exp table: Buckingham stays on the generic path
            if (!ok)
                for (auto& t : p->probes)
                    if (t.vdwk == 2) t.vdwk = 0;
            if (ok) {
                rc = upload(&p->d_erfcx, tab.data(), tab.size());
                if (!rc) rc = upload(&p->d_exp2, e2.data(), e2.size());
                hc.erfcx_tab = p->d_erfcx;
                hc.exp2_tab = p->d_exp2;
                hc.erfcx_inv_h = inv_h;
                hc.erfcx_mx0_inv_h = mx0;
                p->fast_ewald = want_ewald;
            }
            hc.ew_k3 = 2.0 * alpha * alpha / 3.0;
            hc.ew_k15 = 4.0 * (alpha * alpha) * (alpha * alpha) / 15.0;
            if (ok && want_ewald && !rc && !std::getenv("CEG_HIP_NO_EW2")) {
                std::vector<double> t2;
                int32_t base = 0, ni = 0;
                if (build_ew2_table(alpha, p->r_exact2, cutoff2, t2, &base, &ni)) {
                    rc = upload(&p->d_ew2, t2.data(), t2.size());
                    hc.ew2_tab = p->d_ew2;
                    hc.ew2_ni = ni;
                    hc.ew2_base = base;
                    p->ew2 = !rc;
                }
            }
        }
        // the r^2-indexed table of ONE Buckingham class (VDWK = 3): G0/C, the hot loop accumulates the channels divided by C, 6C, -48C,
        // 480C (ceg_kernels.hip); `target` receives the table and its scalar constants
        auto setup_bk2 = [&](int& vdwk, bool single_buck, const double bk[4], double** d_tab, PlanConst& target) {
            if (rc || vdwk != 2 || !single_buck || std::getenv("CEG_HIP_NO_BK2")) return;
            std::vector<double> tb;
            int32_t base = 0, ni = 0;
            if (!build_bk2_table(bk[0] / bk[2], bk[1], 1.0, p->r_exact2, cutoff2, tb, &base, &ni)) return;
            rc = upload(d_tab, tb.data(), tb.size());
            target.bk2_tab = *d_tab;
            target.bk2_ni = ni;
            target.bk2_base = base;
            const double B = bk[1], C = bk[2];
            target.bk_B = B; target.bk_C = C; target.bk_invC = 1.0 / C; target.bk_nshift = -bk[3] / C;
            target.bk_c1 = -B / 6.0; target.bk_c2 = -B / 48.0; target.bk_c3 = B * B / 3.0; target.bk_c4 = -B / 160.0;
            target.bk_s1 = 1.0 / (6.0 * C); target.bk_s2 = -1.0 / (48.0 * C); target.bk_s3 = 1.0 / (480.0 * C);
            if (!rc) vdwk = 3;
        };
        if (p->nprobes == 0) setup_bk2(p->vdwk, p->single_buck, p->bk, &p->d_bk2, hc);
        if (!rc && p->nprobes > 0) {
            // (without the r^2-indexed Ewald tables -- alpha * cutoff > 5, or a fit that misses its tolerance -- the Coulomb grid is
            //  built by its own launch with the erfcx / libm-grade arithmetic and only the VdW grids share a pass: ceg_plan_build_multi)
            hc.nprobes = p->nprobes;
            for (int q = 0; q < p->nprobes && !rc; ++q) {
                ceg_plan::ProbeTab& t = p->probes[q];
                rc = upload(&t.d_rules, t.rules.data(), t.rules.size());
                if (!rc) rc = upload(&t.d_offset, t.offset.data(), t.offset.size());
                if (!rc) rc = upload(&t.d_fast, t.fast.data(), t.fast.size());
                hc.rtm[q] = RuleTable{t.d_rules, t.d_offset, p->nkinds};
                hc.fastm[q] = t.d_fast;
            }
            for (int q = 0; q < p->nprobes && !rc; ++q) {        // the block with probe q in the single-probe slots
                ceg_plan::ProbeTab& t = p->probes[q];
                PlanConst hq = hc;
                hq.rt = hc.rtm[q];
                hq.fastvdw = hc.fastm[q];
                setup_bk2(t.vdwk, t.single_buck, t.bk, &t.d_bk2, hq);
                // per image: does ITS kind have a rule with this probe?  The shared list carries the union of the probes (what the
                // launches of several Lennard-Jones probes stage by); a single-probe launch drops / declasses the images that are
                // inactive for its probe -- the tabulated Buckingham class carries no per-candidate parameters that could be zero.
                if (!rc && p->nprobes > 1) {
                    std::vector<int32_t> has((size_t)p->nkinds);
                    for (int32_t k = 0; k < p->nkinds; ++k) has[k] = t.offset[k + 1] > t.offset[k] ? 1 : 0;
                    int32_t* d_has = nullptr;
                    rc = upload(&d_has, has.data(), has.size());
                    if (!rc) {
                        void* raw = nullptr;
                        const int64_t n = p->ib.nimages;
                        if (cached_malloc(&raw, sizeof(int32_t) * (size_t)std::max<int64_t>(n, 1)) != hipSuccess) rc = fail(CEG_ERR_HIP, "hipMalloc failed");
                        else {
                            t.d_imgkind = (int32_t*)raw;
                            rc = probe_image_flags(p->d_imgkind, d_has, p->nkinds, t.d_imgkind, n);
                        }
                    }
                    if (d_has) { (void)hipDeviceSynchronize(); cached_free(d_has); }
                    hq.ib.kind = t.d_imgkind;
                }
                if (!rc) rc = upload(&t.d_pc, &hq, 1);
                if (q == 0) { p->vdwk = t.vdwk; }
            }
            hc.rt = hc.rtm[0];
            hc.fastvdw = hc.fastm[0];
        }
        if (!rc) rc = upload(&p->d_pc, &hc, 1);
    } else if (!rc && p->nprobes > 0) {
        rc = fail(CEG_ERR_UNSUPPORTED, "multi-probe plans need every perpendicular cell width >= 2*cutoff (what a ProbeSystem guarantees)");
    }
    if (flush_uploads() != hipSuccess && !rc) rc = fail(CEG_ERR_HIP, "uploading the plan's tables failed");       // (also before a failed plan's blocks go back to the pool)
    stamp("function tables, constants");
    if (rc) {
        ceg_plan_destroy(p);
        return rc;
    }
    *plan = p;
    return CEG_OK;
}

extern "C" int ceg_plan_create(ceg_plan_t** plan, int32_t device,
                               const double* pos, const int64_t* atomkind, const double* charge,
                               int64_t natoms,
                               const double mat[9], const double invmat[9],
                               int32_t ortho, double safemin2, double cutoff2,
                               const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds,
                               double alpha,
                               const int32_t dims[3], const double size[3], const double shift[3],
                               const double delta[3])
{
    return create_impl(plan, device, pos, atomkind, charge, natoms, mat, invmat, ortho, safemin2, cutoff2, rules, rule_offset, nkinds, 0,
                       nullptr, nullptr, alpha, dims, size, shift, delta);
}

extern "C" int ceg_plan_create_multi(ceg_plan_t** plan, int32_t device, const double* pos, const int64_t* atomkind, const double* charge,
                                     int64_t natoms, const double mat[9], const double invmat[9], int32_t ortho, double safemin2,
                                     double cutoff2, int32_t nprobes, const ceg_rule_t* const* rules,
                                     const int32_t* const* rule_offset, int32_t nkinds, double alpha, const int32_t dims[3],
                                     const double size[3], const double shift[3], const double delta[3])
{
    if (nprobes < 1) return fail(CEG_ERR_INVALID, "nprobes = %d < 1", nprobes);
    return create_impl(plan, device, pos, atomkind, charge, natoms, mat, invmat, ortho, safemin2, cutoff2, nullptr, nullptr, nkinds, nprobes,
                       rules, rule_offset, alpha, dims, size, shift, delta);
}

extern "C" int ceg_plan_num_probes(const ceg_plan_t* p) { return p ? p->nprobes : 0; }

extern "C" int ceg_plan_destroy(ceg_plan_t* p)
{
    if (!p) return CEG_OK;
    DeviceGuard guard(p->device);
    (void)hipDeviceSynchronize();          // what hipFree would do: no kernel of this plan is still running
    p->images.reset();                     // the image arrays belong to the (possibly cached, possibly shared) ImageSet
    for (void* ptr : {(void*)p->d_atoms, (void*)p->d_kind, (void*)p->d_rules, (void*)p->d_offset, (void*)p->d_pc, (void*)p->d_erfcx,
                      (void*)p->d_exp2, (void*)p->d_fast, (void*)p->d_ew2, (void*)p->d_bk2})
        cached_free(ptr);
    for (auto& t : p->probes)
        for (void* ptr : {(void*)t.d_rules, (void*)t.d_offset, (void*)t.d_fast, (void*)t.d_pc, (void*)t.d_bk2, (void*)t.d_imgkind}) cached_free(ptr);
    delete p;
    return CEG_OK;
}

extern "C" int ceg_image_cache_stats(int64_t* hits, int64_t* misses, int64_t* entries)
{
    std::lock_guard<std::mutex> lock(g_image_mutex);
    if (hits) *hits = g_image_hits.load();
    if (misses) *misses = g_image_misses.load();
    if (entries) *entries = (int64_t)g_image_cache.size();
    return CEG_OK;
}

extern "C" int ceg_plan_can_cull(const ceg_plan_t* p) { return (p && p->can_cull) ? 1 : 0; }

extern "C" int64_t ceg_plan_num_images(const ceg_plan_t* p) { return (p && p->images_built) ? p->ib.nimages : 0; }

// the lattice-image list of a plan copied to the host (tests: the device build against the host build): xyzq [4 n], kind / atom [n],
// bin_start [nbins + 1] with nbins = nb[0] nb[1] nb[2]; any output may be NULL; nb receives the bin counts
extern "C" int ceg_plan_copy_images(const ceg_plan_t* p, double* xyzq, int32_t* kind, int32_t* atom, int32_t* bin_start, int32_t nb[3])
{
    if (!p || !p->images_built) return fail(CEG_ERR_INVALID, "the plan has no image list");
    DeviceGuard guard(p->device);
    if (!guard.ok) return fail(CEG_ERR_HIP, "hipSetDevice(%d) failed", p->device);
    const size_t n = (size_t)p->ib.nimages;
    const size_t nbins = (size_t)p->ib.nb[0] * p->ib.nb[1] * p->ib.nb[2];
    if (nb) for (int a = 0; a < 3; ++a) nb[a] = p->ib.nb[a];
    bool ok = hipDeviceSynchronize() == hipSuccess;
    if (ok && xyzq && n) ok = hipMemcpy(xyzq, p->d_images, n * sizeof(double4), hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && kind && n && p->d_imgkind) ok = hipMemcpy(kind, p->d_imgkind, n * sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && atom && n) ok = hipMemcpy(atom, p->d_imgatom, n * sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && bin_start) ok = hipMemcpy(bin_start, p->d_binstart, (nbins + 1) * sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess;
    return ok ? CEG_OK : fail(CEG_ERR_HIP, "copying the image list failed");
}

namespace {

int resolve_algo(const ceg_plan* p, int32_t algo, bool* culled)
{
    switch (algo) {
    case CEG_ALGO_AUTO: *culled = p->can_cull; return CEG_OK;
    case CEG_ALGO_BRUTEFORCE: *culled = false; return CEG_OK;
    case CEG_ALGO_CULLED:
        if (!p->can_cull)
            return fail(CEG_ERR_UNSUPPORTED,
                        "culled algorithm needs every perpendicular cell width >= 2*cutoff");
        *culled = true;
        return CEG_OK;
    default: return fail(CEG_ERR_INVALID, "unknown algo %d", algo);
    }
}

int run(ceg_plan* p, int mode, const Output& out, const Points& pts, bool culled, hipStream_t stream)
{
    if (mode != MODE_COULOMB && !p->has_rules) return fail(CEG_ERR_INVALID, "plan was created without rules");
    if (mode != MODE_VDW && !p->has_charge) return fail(CEG_ERR_INVALID, "plan was created without charges");
    RuleTable rt{p->d_rules, p->d_offset, p->nkinds};
    hipError_t e;
    if (culled) {
        e = launch_culled(mode, p->d_pc, p->g, p->vdwk, p->ew2 ? 2 : (p->fast_ewald ? 1 : 0), out, pts, stream);
    } else {
        AtomTable at{p->d_atoms, p->has_rules ? p->d_kind : nullptr, p->natoms};
        e = launch_bruteforce(mode, p->g, at, rt, out, pts, stream);
    }
    if (e != hipSuccess) return fail(CEG_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return CEG_OK;
}

int build_common(ceg_plan* p, int mode, double lv, double tv, double lc, double tc, int32_t i_begin,
                 int32_t i_end, float* d_v, float* d_c, int64_t channel_stride, int32_t i_origin,
                 int32_t algo, void* stream)
{
    if (!p) return fail(CEG_ERR_INVALID, "plan is NULL");
    if (i_begin < 0 || i_end > p->g.dims[0] + 1 || i_begin > i_end)
        return fail(CEG_ERR_INVALID, "bad x-plane range [%d,%d) for dims[0]+1 = %d", i_begin, i_end, p->g.dims[0] + 1);
    if (i_origin > i_begin) return fail(CEG_ERR_INVALID, "i_origin %d > i_begin %d", i_origin, i_begin);
    if ((mode != MODE_COULOMB && !d_v) || (mode != MODE_VDW && !d_c)) return fail(CEG_ERR_INVALID, "output pointer is NULL");
    const int64_t plane = (int64_t)(p->g.dims[1] + 1) * (p->g.dims[2] + 1);
    if (channel_stride < (int64_t)(i_end - i_origin) * plane)
        return fail(CEG_ERR_INVALID, "channel_stride %lld too small", (long long)channel_stride);
    bool culled;
    if (int rc = resolve_algo(p, algo, &culled)) return rc;
    Output out{};
    out.vdw = d_v;
    out.coulomb = d_c;
    out.channel_stride = channel_stride;
    out.i_origin = i_origin;
    out.i_begin = i_begin;
    out.i_end = i_end;
    out.lambda_vdw = lv;
    out.thr_vdw = tv;
    out.lambda_coulomb = lc;
    out.thr_coulomb = tc;
    DeviceGuard guard(p->device);
    if (!guard.ok) return fail(CEG_ERR_HIP, "hipSetDevice(%d) failed", p->device);
    return run(p, mode, out, Points{nullptr, 0}, culled, (hipStream_t)stream);
}

}  // namespace

extern "C" int ceg_plan_build_vdw(ceg_plan_t* plan, double lambda, double threshold, int32_t i_begin,
                                  int32_t i_end, float* d_out, int64_t channel_stride, int32_t i_origin,
                                  int32_t algo, void* stream)
{
    return build_common(plan, MODE_VDW, lambda, threshold, 0, 0, i_begin, i_end, d_out, nullptr, channel_stride,
                        i_origin, algo, stream);
}

extern "C" int ceg_plan_build_coulomb(ceg_plan_t* plan, double lambda, double threshold, int32_t i_begin,
                                      int32_t i_end, float* d_out, int64_t channel_stride, int32_t i_origin,
                                      int32_t algo, void* stream)
{
    return build_common(plan, MODE_COULOMB, 0, 0, lambda, threshold, i_begin, i_end, nullptr, d_out,
                        channel_stride, i_origin, algo, stream);
}

extern "C" int ceg_plan_build_fused(ceg_plan_t* plan, double lambda_vdw, double threshold_vdw,
                                    double lambda_coulomb, double threshold_coulomb, int32_t i_begin,
                                    int32_t i_end, float* d_out_vdw, float* d_out_coulomb,
                                    int64_t channel_stride, int32_t i_origin, int32_t algo, void* stream)
{
    return build_common(plan, MODE_FUSED, lambda_vdw, threshold_vdw, lambda_coulomb, threshold_coulomb, i_begin,
                        i_end, d_out_vdw, d_out_coulomb, channel_stride, i_origin, algo, stream);
}

// K VdW grids + the Coulomb grid of one framework from one image list (the grids setup_RASPA asks for one after the other,
// src/raspa.jl:497-520).  The requested set is cut into launches of the multi-probe kernels: with the Coulomb grid the first two
// probes ride in the fused launch (48 accumulator registers: 3 waves per SIMD), the others in VdW launches of up to four probes.
// Whatever the cut, every grid is bit-identical to the one a launch of that grid alone produces (same per-pair arithmetic, same
// order of the sums; ceg_kernels.hip).
extern "C" int ceg_plan_build_multi(ceg_plan_t* p, double lambda_vdw, double threshold_vdw, double lambda_coulomb, double threshold_coulomb,
                                    int32_t i_begin, int32_t i_end, float* const* d_out_vdw, float* d_out_coulomb,
                                    int64_t channel_stride, int32_t i_origin, void* stream)
{
    if (!p) return fail(CEG_ERR_INVALID, "plan is NULL");
    if (p->nprobes < 1) return fail(CEG_ERR_INVALID, "not a multi-probe plan (ceg_plan_create_multi)");
    if (i_begin < 0 || i_end > p->g.dims[0] + 1 || i_begin > i_end)
        return fail(CEG_ERR_INVALID, "bad x-plane range [%d,%d) for dims[0]+1 = %d", i_begin, i_end, p->g.dims[0] + 1);
    if (i_origin > i_begin) return fail(CEG_ERR_INVALID, "i_origin %d > i_begin %d", i_origin, i_begin);
    const int64_t plane = (int64_t)(p->g.dims[1] + 1) * (p->g.dims[2] + 1);
    if (channel_stride < (int64_t)(i_end - i_origin) * plane) return fail(CEG_ERR_INVALID, "channel_stride %lld too small", (long long)channel_stride);
    if (d_out_coulomb && !p->has_charge) return fail(CEG_ERR_INVALID, "plan was created without charges");
    std::vector<int> req;
    for (int q = 0; q < p->nprobes; ++q)
        if (d_out_vdw && d_out_vdw[q]) req.push_back(q);
    if (req.empty() && !d_out_coulomb) return CEG_OK;
    Output base{};
    base.channel_stride = channel_stride;
    base.i_origin = i_origin;
    base.i_begin = i_begin;
    base.i_end = i_end;
    base.lambda_vdw = lambda_vdw;
    base.thr_vdw = threshold_vdw;
    base.lambda_coulomb = lambda_coulomb;
    base.thr_coulomb = threshold_coulomb;
    DeviceGuard guard(p->device);
    if (!guard.ok) return fail(CEG_ERR_HIP, "hipSetDevice(%d) failed", p->device);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    const int ewk = p->ew2 ? 2 : (p->fast_ewald ? 1 : 0);        // real-space Ewald arithmetic available to this plan
    auto single = [&](int mode, int q) {          // one probe (q >= 0) and / or the Coulomb grid: the single-probe kernels of the probe's class
        Output o = base;
        o.vdw = q >= 0 ? d_out_vdw[q] : nullptr;
        o.coulomb = mode != MODE_VDW ? d_out_coulomb : nullptr;
        return launch_culled(mode, q >= 0 ? p->probes[q].d_pc : p->d_pc, p->g, q >= 0 ? p->probes[q].vdwk : 1, ewk, o, Points{nullptr, 0}, st);
    };
    auto multi = [&](int mode, const int* idx, int np) {
        Output o = base;
        for (int t = 0; t < np; ++t) { o.vdwm[t] = d_out_vdw[idx[t]]; o.probe_idx[t] = idx[t]; }
        o.coulomb = mode == MODE_FUSED ? d_out_coulomb : nullptr;
        return launch_culled_multi(mode, np, p->d_pc, p->g, o, st);
    };
    // the Lennard-Jones-only probes can share accumulating loops; a probe of another class launches alone (or fused with the Coulomb grid)
    std::vector<int> lj, other;
    for (int q : req) (p->probes[q].vdwk == 1 ? lj : other).push_back(q);
    if (d_out_coulomb) {
        int fused_np = CEG_MAX_PROBES_FUSED;          // probes that share the Coulomb launch (CEG_HIP_MULTI_FUSED_NP = 0 | 1 | 2: measurement aid)
        if (const char* env = std::getenv("CEG_HIP_MULTI_FUSED_NP")) fused_np = std::max(0, std::min(CEG_MAX_PROBES_FUSED, atoi(env)));
        if (!p->ew2 && p->nprobes > 1) fused_np = 0;  // the fused multi-probe variants are built on the r^2-indexed tables
        // what rides with the Coulomb grid, by what it saves on the roofline workload (profiles/r04_multi_probe_timing.txt): two
        // Lennard-Jones probes (25.3 -> 18.1 ms), else one probe of another class (18.2 -> 15.5 ms), else one Lennard-Jones probe
        if (fused_np >= 2 && lj.size() >= 2) {
            e = multi(MODE_FUSED, &lj[0], 2);
            lj.erase(lj.begin(), lj.begin() + 2);
        } else if (fused_np >= 1 && !other.empty()) {
            e = single(MODE_FUSED, other[0]);
            other.erase(other.begin());
        } else if (fused_np >= 1 && !lj.empty()) {
            e = single(MODE_FUSED, lj[0]);
            lj.erase(lj.begin());
        } else {
            e = single(MODE_COULOMB, -1);
        }
    }
    size_t at = 0;
    while (e == hipSuccess && at < lj.size()) {
        const int n = (int)std::min<size_t>(CEG_MAX_PROBES, lj.size() - at);
        e = n == 1 ? single(MODE_VDW, lj[at]) : multi(MODE_VDW, &lj[at], n);
        at += (size_t)n;
    }
    for (size_t t = 0; e == hipSuccess && t < other.size(); ++t) e = single(MODE_VDW, other[t]);
    if (e != hipSuccess) return fail(CEG_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return CEG_OK;
}

extern "C" int ceg_plan_eval_points(ceg_plan_t* p, int32_t which, int32_t algo, const double* points,
                                    int64_t npoints, double* out)
{
    if (!p) return fail(CEG_ERR_INVALID, "plan is NULL");
    if (npoints < 0 || (npoints > 0 && (!points || !out))) return fail(CEG_ERR_INVALID, "bad points/out");
    if (which != 0 && which != 1) return fail(CEG_ERR_INVALID, "which must be 0 (vdw) or 1 (coulomb)");
    if (npoints == 0) return CEG_OK;
    bool culled;
    if (int rc = resolve_algo(p, algo, &culled)) return rc;
    if (culled) {
        // the image list only covers the grid's bounding box grown by the cutoff
        bool inside = true;
        for (int64_t q = 0; q < npoints && inside; ++q)
            for (int a = 0; a < 3; ++a) {
                const double x = points[3 * q + a];
                if (!(x >= p->g.shift[a] - 1e-9 && x <= p->g.shift[a] + p->g.size[a] + 1e-9)) inside = false;
            }
        if (!inside) {
            if (algo == CEG_ALGO_CULLED)
                return fail(CEG_ERR_UNSUPPORTED, "culled evaluation needs points inside the grid bounding box");
            culled = false;
        }
    }
    DeviceGuard guard(p->device);
    if (!guard.ok) return fail(CEG_ERR_HIP, "hipSetDevice(%d) failed", p->device);
    double* d_pts = nullptr;
    double* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_pts, sizeof(double) * 3 * npoints));
    if (hipMalloc((void**)&d_out, sizeof(double) * 8 * npoints) != hipSuccess) {
        (void)hipFree(d_pts);
        return fail(CEG_ERR_HIP, "hipMalloc failed");
    }
    int rc = CEG_OK;
    if (hipMemcpy(d_pts, points, sizeof(double) * 3 * npoints, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(CEG_ERR_HIP, "hipMemcpy H2D failed");
    if (!rc) {
        Output o{};
        if (which == 0) o.raw_vdw = d_out; else o.raw_coulomb = d_out;
        rc = run(p, which == 0 ? MODE_VDW : MODE_COULOMB, o, Points{d_pts, npoints}, culled, nullptr);
    }
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(CEG_ERR_HIP, "kernel execution failed: %s", hipGetErrorString(hipGetLastError()));
    if (!rc && hipMemcpy(out, d_out, sizeof(double) * 8 * npoints, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(CEG_ERR_HIP, "hipMemcpy D2H failed");
    (void)hipFree(d_pts);
    (void)hipFree(d_out);
    return rc;
}

// ------------------------------------------------------------------ one-shot host API
namespace {

// x-planes [begin,end) of device `d` out of `n`
inline void slab(int nx, int n, int d, int* begin, int* end)
{
    const int base = nx / n, rem = nx % n;
    *begin = d * base + std::min(d, rem);
    *end = *begin + base + (d < rem ? 1 : 0);
}

// ---- pinned staging buffers, kept across calls (page-locking costs more than the copy it serves)
struct PinnedBuf { void* ptr; size_t bytes; bool busy; bool user = false; };     // user: handed to a caller by ceg_host_grid_alloc
std::mutex g_pinned_mutex;
std::vector<PinnedBuf> g_pinned;

// ---- device output slabs, kept across calls as well (hipMalloc + hipFree of 0.5 GB cost ~8 ms, as
// much as the VdW kernel itself); one idle buffer per device is retained, see ceg_release_cached_buffers
struct DeviceBuf { int device; void* ptr; size_t bytes; bool busy; };
std::vector<DeviceBuf> g_devbufs;      // guarded by g_pinned_mutex

// streams are expensive to create on ROCm (an HSA queue each, ~2 ms): keep a pair per device
struct StreamPair { int device; hipStream_t comp, copy; bool busy; };
std::vector<StreamPair> g_streams;     // guarded by g_pinned_mutex

bool streams_acquire(int device, hipStream_t* comp, hipStream_t* copy)
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (auto& p : g_streams)
        if (p.device == device && !p.busy) { p.busy = true; *comp = p.comp; *copy = p.copy; return true; }
    StreamPair sp{device, nullptr, nullptr, true};
    if (hipStreamCreateWithFlags(&sp.comp, hipStreamNonBlocking) != hipSuccess) return false;
    if (hipStreamCreateWithFlags(&sp.copy, hipStreamNonBlocking) != hipSuccess) { (void)hipStreamDestroy(sp.comp); return false; }
    g_streams.push_back(sp);
    *comp = sp.comp; *copy = sp.copy;
    return true;
}

void streams_release(hipStream_t comp)
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (auto& p : g_streams)
        if (p.comp == comp) p.busy = false;
}

void* device_acquire(int device, size_t bytes)     // current device must be `device`
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (auto& p : g_devbufs)
        if (p.device == device && !p.busy && p.bytes >= bytes) { p.busy = true; return p.ptr; }
    for (size_t t = 0; t < g_devbufs.size(); ++t)
        if (g_devbufs[t].device == device && !g_devbufs[t].busy) {
            (void)hipFree(g_devbufs[t].ptr);
            g_devbufs.erase(g_devbufs.begin() + t);
            break;
        }
    void* ptr = nullptr;
    if (hipMalloc(&ptr, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    g_devbufs.push_back({device, ptr, bytes, true});
    return ptr;
}

void device_release(void* ptr)
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (auto& p : g_devbufs)
        if (p.ptr == ptr) p.busy = false;
}

void* pinned_acquire(size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (auto& p : g_pinned)
        if (!p.busy && p.bytes >= bytes) { p.busy = true; return p.ptr; }
    for (size_t t = 0; t < g_pinned.size(); ++t)          // replace an idle, too small buffer
        if (!g_pinned[t].busy) {
            (void)hipHostFree(g_pinned[t].ptr);
            g_pinned.erase(g_pinned.begin() + t);
            break;
        }
    void* ptr = nullptr;
    if (hipHostMalloc(&ptr, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    g_pinned.push_back({ptr, bytes, true});
    return ptr;
}

// true when [ptr, ptr + bytes) lies inside a page-locked buffer this library handed out (ceg_host_grid_alloc): the one-shot
// pipelines then copy D2H straight to where the data belongs instead of through the ring + host threads
bool pinned_owns(const void* ptr, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    const char* q = static_cast<const char*>(ptr);
    for (const auto& p : g_pinned) {
        const char* base = static_cast<const char*>(p.ptr);
        if (p.busy && q >= base && q + bytes <= base + p.bytes) return true;
    }
    return false;
}

void pinned_release(void* ptr)
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (auto& p : g_pinned)
        if (p.ptr == ptr) { p.busy = false; p.user = false; }
}

// copy `nseg` segments in parallel; first touch of a fresh destination is page-fault bound, and the
// faults of different threads proceed in parallel
struct Segment { float* dst; const float* src; size_t n; int64_t file_offset; };
// fd >= 0: every piece is also written to the file at its byte offset (pwrite is thread-safe); dst may be null.
// Returns false if a write failed.
bool parallel_copy(const std::vector<Segment>& segs, int nthreads, int fd = -1)
{
    // cut every segment into pieces of <= 1 MiB so that all threads have work
    std::vector<Segment> pieces;
    const size_t piece = 256 * 1024;
    for (const auto& sg : segs)
        for (size_t o = 0; o < sg.n; o += piece)
            pieces.push_back({sg.dst ? sg.dst + o : nullptr, sg.src + o, std::min(piece, sg.n - o), sg.file_offset + (int64_t)(o * sizeof(float))});
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&]() {
        for (;;) {
            const size_t t = next.fetch_add(1);
            if (t >= pieces.size()) return;
            const Segment& pc = pieces[t];
            if (pc.dst) std::memcpy(pc.dst, pc.src, pc.n * sizeof(float));
            if (fd >= 0) {
                const char* ptr = reinterpret_cast<const char*>(pc.src);
                size_t left = pc.n * sizeof(float);
                int64_t off = pc.file_offset;
                while (left > 0) {
                    const ssize_t w = pwrite(fd, ptr, left, (off_t)off);
                    if (w <= 0) { ok.store(false); break; }
                    ptr += w; left -= (size_t)w; off += w;
                }
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    return ok.load();
}

// One device's share of a one-shot build: x-planes [b, e) of the grid, computed in chunks of `cx`
// planes; chunk j is copied D2H into a pinned ring slot (copy stream) and from there into the
// caller's array (host threads) while chunk j+1.. are being computed.
int device_pipeline(int mode, int d, int b, int e, int nx, int64_t plane, const double* pos, const int64_t* atomkind,
                    const double* charge, int64_t natoms, const double* mat, const double* invmat, int32_t ortho,
                    double safemin2, double cutoff2, const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds,
                    double alpha, const int32_t* dims, const double* size, const double* shift, const double* delta,
                    double lambda, double threshold, float* grid, int copy_threads, std::string* err, int fd = -1,
                    int64_t payload_offset = 0)
{
    auto bad = [&](int code, const char* what) {
        *err = std::string(what) + " (device " + std::to_string(d) + "): " + hipGetErrorString(hipGetLastError());
        return code;
    };
    const int64_t npts = plane * nx;
    const int64_t slab_pts = (int64_t)(e - b) * plane;
    if (slab_pts <= 0) return CEG_OK;
    const bool trace = std::getenv("CEG_HIP_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (trace)
            fprintf(stderr, "[ceg one-shot dev %d] %-28s %8.3f ms\n", d, what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    if (hipSetDevice(d) != hipSuccess) return bad(CEG_ERR_HIP, "hipSetDevice failed");
    ceg_plan* plan = nullptr;
    int rc = ceg_plan_create(&plan, d, pos, atomkind, charge, natoms, mat, invmat, ortho, safemin2, cutoff2, rules, rule_offset,
                             nkinds, alpha, dims, size, shift, delta);
    if (rc) { *err = g_err; return rc; }
    stamp("plan created");
    // chunk = a multiple of 4 planes (the kernel's tile edge) of about 32 MB over the 8 channels
    int cx = (int)std::max<int64_t>(4, ((32ll << 20) / (plane * 8 * (int64_t)sizeof(float))) / 4 * 4);
    cx = std::min(cx, (e - b + 3) / 4 * 4);
    const int nchunks = (e - b + cx - 1) / cx;
    const int R = std::min(3, nchunks);
    const size_t slot_floats = (size_t)8 * cx * plane;
    // The caller's array is page-locked memory of this library (ceg_host_grid_alloc) and no file is written on the way: every chunk
    // is copied D2H straight to its place -- no ring, no host threads, no second pass over the 537 MB.
    const bool direct = grid != nullptr && fd < 0 && pinned_owns(grid, sizeof(float) * 8 * (size_t)npts);
    float* d_out = nullptr;
    float* h_ring = nullptr;
    hipStream_t s_comp = nullptr, s_copy = nullptr;
    std::vector<hipEvent_t> ev_comp(nchunks, nullptr), ev_copy(nchunks, nullptr);
    std::thread drain;
    std::atomic<int> drained{0};
    std::atomic<int> drain_rc{CEG_OK};
    bool ok = streams_acquire(d, &s_comp, &s_copy) &&
              (d_out = static_cast<float*>(device_acquire(d, sizeof(float) * 8 * slab_pts))) != nullptr;
    for (int j = 0; j < nchunks && ok; ++j)
        ok = hipEventCreateWithFlags(&ev_comp[j], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ev_copy[j], hipEventDisableTiming) == hipSuccess;
    if (ok && !direct) {
        h_ring = static_cast<float*>(pinned_acquire(sizeof(float) * slot_floats * R));
        ok = h_ring != nullptr;
    }
    if (!ok) rc = bad(CEG_ERR_HIP, "allocation of streams / buffers failed");
    stamp("streams, buffers, pinned ring");
    // all kernels up front: the compute stream runs them back to back
    for (int j = 0; j < nchunks && !rc; ++j) {
        const int cb = b + j * cx, ce = std::min(e, cb + cx);
        rc = (mode == MODE_VDW) ? ceg_plan_build_vdw(plan, lambda, threshold, cb, ce, d_out, slab_pts, b, CEG_ALGO_AUTO, s_comp)
                                : ceg_plan_build_coulomb(plan, lambda, threshold, cb, ce, d_out, slab_pts, b, CEG_ALGO_AUTO, s_comp);
        if (rc) { *err = g_err; break; }
        if (hipEventRecord(ev_comp[j], s_comp) != hipSuccess) rc = bad(CEG_ERR_HIP, "hipEventRecord failed");
    }
    std::atomic<int> enqueued{0};
    stamp("kernels enqueued");
    if (!rc && direct) {
        for (int j = 0; j < nchunks && !rc; ++j) {
            const int cb = b + j * cx, ce = std::min(e, cb + cx);
            const size_t cpts = (size_t)(ce - cb) * plane;
            if (hipStreamWaitEvent(s_copy, ev_comp[j], 0) != hipSuccess) rc = bad(CEG_ERR_HIP, "hipStreamWaitEvent failed");
            // the 8 channel segments of the chunk as ONE strided copy (rows of cpts floats, pitches = channel strides)
            if (!rc && hipMemcpy2DAsync(grid + (size_t)cb * plane, sizeof(float) * (size_t)npts, d_out + (size_t)(cb - b) * plane,
                                        sizeof(float) * (size_t)slab_pts, sizeof(float) * cpts, 8, hipMemcpyDeviceToHost, s_copy) != hipSuccess)
                rc = bad(CEG_ERR_HIP, "hipMemcpy2DAsync D2H failed");
        }
        stamp("copies enqueued (direct)");
        if (hipStreamSynchronize(s_copy) != hipSuccess && !rc) rc = bad(CEG_ERR_HIP, "kernel execution or D2H copy failed");
        stamp("grid in the caller's page-locked array");
    } else if (!rc) {
        drain = std::thread([&]() {
            (void)hipSetDevice(d);
            for (int j = 0; j < nchunks; ++j) {
                while (j >= enqueued.load()) std::this_thread::yield();
                if (drain_rc.load() != CEG_OK) { drained.store(j + 1); continue; }
                if (hipEventSynchronize(ev_copy[j]) != hipSuccess) { drain_rc.store(CEG_ERR_HIP); drained.store(j + 1); continue; }
                const int cb = b + j * cx, ce = std::min(e, cb + cx);
                const size_t cpts = (size_t)(ce - cb) * plane;
                const float* slot = h_ring + (size_t)(j % R) * slot_floats;
                std::vector<Segment> segs;
                for (int c = 0; c < 8; ++c) {
                    const size_t at = (size_t)c * npts + (size_t)cb * plane;           // float offset inside the payload
                    segs.push_back({grid ? grid + at : nullptr, slot + (size_t)c * cpts, cpts, payload_offset + (int64_t)(at * sizeof(float))});
                }
                if (!parallel_copy(segs, copy_threads, fd)) drain_rc.store(CEG_ERR_INVALID);
                drained.store(j + 1);
            }
        });
        for (int j = 0; j < nchunks && !rc; ++j) {
            while (j - drained.load() >= R) std::this_thread::yield();        // ring slot still being emptied
            const int cb = b + j * cx, ce = std::min(e, cb + cx);
            const size_t cpts = (size_t)(ce - cb) * plane;
            float* slot = h_ring + (size_t)(j % R) * slot_floats;
            if (hipStreamWaitEvent(s_copy, ev_comp[j], 0) != hipSuccess) rc = bad(CEG_ERR_HIP, "hipStreamWaitEvent failed");
            // the chunk's 8 channel segments as one strided copy into the slot ([8][cpts])
            if (!rc && hipMemcpy2DAsync(slot, sizeof(float) * cpts, d_out + (size_t)(cb - b) * plane, sizeof(float) * (size_t)slab_pts,
                                        sizeof(float) * cpts, 8, hipMemcpyDeviceToHost, s_copy) != hipSuccess)
                rc = bad(CEG_ERR_HIP, "hipMemcpy2DAsync D2H failed");
            if (!rc && hipEventRecord(ev_copy[j], s_copy) != hipSuccess) rc = bad(CEG_ERR_HIP, "hipEventRecord failed");
            if (rc) drain_rc.store(rc);
            enqueued.store(j + 1);
        }
        if (rc) enqueued.store(nchunks);       // let the drain thread run to its end
        stamp("copies enqueued");
        drain.join();
        stamp("drained into caller array");
        if (!rc && drain_rc.load() == CEG_ERR_INVALID) { *err = "writing the grid file failed"; rc = CEG_ERR_INVALID; }
        if (!rc && drain_rc.load() != CEG_OK) rc = bad(CEG_ERR_HIP, "kernel execution or D2H copy failed");
    }
    (void)hipStreamSynchronize(s_comp);
    (void)hipStreamSynchronize(s_copy);
    for (int j = 0; j < nchunks; ++j) {
        if (ev_comp[j]) (void)hipEventDestroy(ev_comp[j]);
        if (ev_copy[j]) (void)hipEventDestroy(ev_copy[j]);
    }
    if (h_ring) pinned_release(h_ring);
    if (d_out) device_release(d_out);
    if (s_comp) streams_release(s_comp);
    ceg_plan_destroy(plan);
    stamp("cleaned up");
    return rc;
}

int oneshot(int mode, const double* pos, const int64_t* atomkind, const double* charge, int64_t natoms,
            const double* mat, const double* invmat, int32_t ortho, double safemin2, double cutoff2,
            const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds, double alpha,
            const int32_t* dims, const double* size, const double* shift, const double* delta,
            double lambda, double threshold, float* grid, int32_t ngpus, const char* path = nullptr,
            const void* header = nullptr, int64_t header_bytes = 0, const void* trailer = nullptr, int64_t trailer_bytes = 0)
{
    if (!grid && !path) return fail(CEG_ERR_INVALID, "grid is NULL");
    if (path && (header_bytes < 0 || trailer_bytes < 0 || (header_bytes > 0 && !header) || (trailer_bytes > 0 && !trailer)))
        return fail(CEG_ERR_INVALID, "bad header / trailer");
    if (int rc = check_common(pos, natoms, mat, invmat, dims, size, shift, delta)) return rc;
    const int ndev = ceg_device_count();
    if (ndev <= 0) return fail(CEG_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    // CEG_HIP_OVERSUBSCRIBE=1 (a rehearsal aid for one-GPU boxes): slab d is built on device d % ndev, so the
    // multi-device scheduling -- one host thread, one plan, one slab per "device" -- can run on a single card
    const bool oversubscribe = std::getenv("CEG_HIP_OVERSUBSCRIBE") != nullptr;
    if (ngpus < 1 || (ngpus > ndev && !oversubscribe))
        return fail(CEG_ERR_NO_DEVICE, "ngpus = %d but %d HIP devices are present", ngpus, ndev);
    const int nx = dims[0] + 1;
    const int64_t plane = (int64_t)(dims[1] + 1) * (dims[2] + 1);
    ngpus = std::min(ngpus, nx);
    int prev = -1;
    (void)hipGetDevice(&prev);
    // host threads that move finished chunks from the pinned ring into the caller's array
    int copy_threads = 8;
    if (const char* env = std::getenv("CEG_HIP_COPY_THREADS")) copy_threads = std::max(1, atoi(env));
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw > 0) copy_threads = std::min<int>(copy_threads, (int)hw);
    copy_threads = std::max(1, copy_threads / ngpus);

    // optional file: header, payload streamed chunk by chunk at its offsets while the build runs, trailer
    // The reference opens the file only once the whole grid exists (grids.jl:151,178), and its cache looks no
    // further than isfile(path) (raspa.jl:426): a file must never be visible at `path` unless it is complete.
    // Everything goes into `<path>.tmp.<pid>.<n>`, renamed onto `path` after every device pipeline returned
    // CEG_OK and close() succeeded; any failure unlinks the temporary.
    int fd = -1;
    std::string tmp_path;
    const int64_t payload_bytes = (int64_t)sizeof(float) * 8 * plane * nx;
    if (path) {
        static std::atomic<unsigned> tmp_serial{0};
        tmp_path = std::string(path) + ".tmp." + std::to_string((long long)getpid()) + "." + std::to_string(tmp_serial.fetch_add(1));
        fd = open(tmp_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) return fail(CEG_ERR_INVALID, "cannot open %s for writing", tmp_path.c_str());
        bool ok = ftruncate(fd, (off_t)(header_bytes + payload_bytes + trailer_bytes)) == 0;
        ok = ok && (header_bytes == 0 || pwrite(fd, header, (size_t)header_bytes, 0) == (ssize_t)header_bytes);
        ok = ok && (trailer_bytes == 0 ||
                    pwrite(fd, trailer, (size_t)trailer_bytes, (off_t)(header_bytes + payload_bytes)) == (ssize_t)trailer_bytes);
        if (!ok) {
            close(fd);
            (void)unlink(tmp_path.c_str());
            return fail(CEG_ERR_INVALID, "cannot write the header of %s", tmp_path.c_str());
        }
    }
    std::vector<int> rcs(ngpus, CEG_OK);
    std::vector<std::string> errs(ngpus);
    auto run = [&](int d) {
        int b, e;
        slab(nx, ngpus, d, &b, &e);
        rcs[d] = device_pipeline(mode, d % ndev, b, e, nx, plane, pos, atomkind, charge, natoms, mat, invmat, ortho, safemin2, cutoff2, rules,
                                 rule_offset, nkinds, alpha, dims, size, shift, delta, lambda, threshold, grid, copy_threads, &errs[d], fd,
                                 header_bytes);
    };
    std::vector<std::thread> workers;
    for (int d = 1; d < ngpus; ++d) workers.emplace_back(run, d);      // one host thread per extra device
    run(0);
    for (auto& w : workers) w.join();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (fd >= 0 && close(fd) != 0 && !rcs[0]) { rcs[0] = CEG_ERR_INVALID; errs[0] = "closing the grid file failed"; }
    for (int d = 0; d < ngpus; ++d)
        if (rcs[d]) {
            if (path) (void)unlink(tmp_path.c_str());
            return fail(rcs[d], "%s", errs[d].c_str());
        }
    if (path && rename(tmp_path.c_str(), path) != 0) {
        (void)unlink(tmp_path.c_str());
        return fail(CEG_ERR_INVALID, "cannot rename %s onto %s", tmp_path.c_str(), path);
    }
    return CEG_OK;
}

// One device's share of a multi-probe one-shot build (ceg_grids_multi): like device_pipeline, with NG = (requested VdW grids) +
// (Coulomb grid) outputs per chunk -- one multi-probe plan, one ceg_plan_build_multi per chunk, NG x 8 channel segments copied
// D2H into a slot of the pinned ring and moved into the callers' arrays by the host threads while the next chunks are computed.
int multi_device_pipeline(int d, int b, int e, int nx, int64_t plane, const double* pos, const int64_t* atomkind,
                          const double* charge, int64_t natoms, const double* mat, const double* invmat, int32_t ortho,
                          double safemin2, double cutoff2, int32_t nprobes, const ceg_rule_t* const* rules,
                          const int32_t* const* rule_offset, int32_t nkinds, double alpha, const int32_t* dims, const double* size,
                          const double* shift, const double* delta, double lambda_vdw, double threshold_vdw, double lambda_coulomb,
                          double threshold_coulomb, float* const* grids_vdw, float* grid_coulomb, int copy_threads, std::string* err)
{
    auto bad = [&](int code, const char* what) {
        *err = std::string(what) + " (device " + std::to_string(d) + "): " + hipGetErrorString(hipGetLastError());
        return code;
    };
    const int64_t npts = plane * nx;
    const int64_t slab_pts = (int64_t)(e - b) * plane;
    if (slab_pts <= 0) return CEG_OK;
    const bool trace = std::getenv("CEG_HIP_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (trace)
            fprintf(stderr, "[ceg multi one-shot dev %d] %-28s %8.3f ms\n", d, what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    // outputs of this call: host array + (device slab assigned below)
    std::vector<float*> host;               // NG host arrays, VdW grids in probe order, then the Coulomb grid
    std::vector<int> probe_of;              // probe index, -1 for the Coulomb grid
    for (int q = 0; q < nprobes; ++q)
        if (grids_vdw && grids_vdw[q]) { host.push_back(grids_vdw[q]); probe_of.push_back(q); }
    if (grid_coulomb) { host.push_back(grid_coulomb); probe_of.push_back(-1); }
    const int NG = (int)host.size();
    if (NG == 0) return CEG_OK;
    if (hipSetDevice(d) != hipSuccess) return bad(CEG_ERR_HIP, "hipSetDevice failed");
    ceg_plan* plan = nullptr;
    int rc = ceg_plan_create_multi(&plan, d, pos, atomkind, grid_coulomb ? charge : nullptr, natoms, mat, invmat, ortho, safemin2, cutoff2,
                                   nprobes, rules, rule_offset, nkinds, alpha, dims, size, shift, delta);
    if (rc) { *err = g_err; return rc; }
    stamp("plan created");
    // chunk = a multiple of 4 planes (the kernel's tile edge) of about 32 MB over the NG x 8 channels
    int cx = (int)std::max<int64_t>(4, ((32ll << 20) / (plane * 8 * NG * (int64_t)sizeof(float))) / 4 * 4);
    cx = std::min(cx, (e - b + 3) / 4 * 4);
    const int nchunks = (e - b + cx - 1) / cx;
    const int R = std::min(3, nchunks);
    const size_t grid_slot = (size_t)8 * cx * plane;           // floats of one grid in a ring slot
    const size_t slot_floats = grid_slot * NG;
    bool direct = true;                                         // every output is page-locked memory of this library: copy straight into it
    for (int gidx = 0; gidx < NG; ++gidx) direct = direct && pinned_owns(host[gidx], sizeof(float) * 8 * (size_t)npts);
    float* d_all = nullptr;                                     // NG slabs of 8 * slab_pts floats
    float* h_ring = nullptr;
    hipStream_t s_comp = nullptr, s_copy = nullptr;
    std::vector<hipEvent_t> ev_comp(nchunks, nullptr), ev_copy(nchunks, nullptr);
    std::thread drain;
    std::atomic<int> drained{0}, enqueued{0};
    std::atomic<int> drain_rc{CEG_OK};
    bool ok = streams_acquire(d, &s_comp, &s_copy) &&
              (d_all = static_cast<float*>(device_acquire(d, sizeof(float) * 8 * slab_pts * NG))) != nullptr;
    for (int j = 0; j < nchunks && ok; ++j)
        ok = hipEventCreateWithFlags(&ev_comp[j], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ev_copy[j], hipEventDisableTiming) == hipSuccess;
    if (ok && !direct) {
        h_ring = static_cast<float*>(pinned_acquire(sizeof(float) * slot_floats * R));
        ok = h_ring != nullptr;
    }
    if (!ok) rc = bad(CEG_ERR_HIP, "allocation of streams / buffers failed");
    std::vector<float*> d_vdw(nprobes, nullptr);
    float* d_coulomb = nullptr;
    for (int gidx = 0; gidx < NG && !rc; ++gidx) {
        float* slab = d_all + (size_t)gidx * 8 * slab_pts;
        if (probe_of[gidx] >= 0) d_vdw[probe_of[gidx]] = slab; else d_coulomb = slab;
    }
    stamp("streams, buffers, pinned ring");
    for (int j = 0; j < nchunks && !rc; ++j) {
        const int cb = b + j * cx, ce = std::min(e, cb + cx);
        rc = ceg_plan_build_multi(plan, lambda_vdw, threshold_vdw, lambda_coulomb, threshold_coulomb, cb, ce, d_vdw.data(), d_coulomb,
                                  slab_pts, b, s_comp);
        if (rc) { *err = g_err; break; }
        if (hipEventRecord(ev_comp[j], s_comp) != hipSuccess) rc = bad(CEG_ERR_HIP, "hipEventRecord failed");
    }
    stamp("kernels enqueued");
    if (!rc && direct) {
        for (int j = 0; j < nchunks && !rc; ++j) {
            const int cb = b + j * cx, ce = std::min(e, cb + cx);
            const size_t cpts = (size_t)(ce - cb) * plane;
            if (hipStreamWaitEvent(s_copy, ev_comp[j], 0) != hipSuccess) rc = bad(CEG_ERR_HIP, "hipStreamWaitEvent failed");
            for (int gidx = 0; gidx < NG && !rc; ++gidx)          // per grid ONE strided copy: 8 rows of cpts floats, pitches = channel strides
                if (hipMemcpy2DAsync(host[gidx] + (size_t)cb * plane, sizeof(float) * (size_t)npts,
                                     d_all + (size_t)gidx * 8 * slab_pts + (size_t)(cb - b) * plane, sizeof(float) * (size_t)slab_pts,
                                     sizeof(float) * cpts, 8, hipMemcpyDeviceToHost, s_copy) != hipSuccess)
                    rc = bad(CEG_ERR_HIP, "hipMemcpy2DAsync D2H failed");
        }
        if (hipStreamSynchronize(s_copy) != hipSuccess && !rc) rc = bad(CEG_ERR_HIP, "kernel execution or D2H copy failed");
        stamp("grids in the caller's page-locked arrays");
    } else if (!rc) {
        drain = std::thread([&]() {
            (void)hipSetDevice(d);
            for (int j = 0; j < nchunks; ++j) {
                while (j >= enqueued.load()) std::this_thread::yield();
                if (drain_rc.load() != CEG_OK) { drained.store(j + 1); continue; }
                if (hipEventSynchronize(ev_copy[j]) != hipSuccess) { drain_rc.store(CEG_ERR_HIP); drained.store(j + 1); continue; }
                const int cb = b + j * cx, ce = std::min(e, cb + cx);
                const size_t cpts = (size_t)(ce - cb) * plane;
                const float* slot = h_ring + (size_t)(j % R) * slot_floats;
                std::vector<Segment> segs;
                for (int gidx = 0; gidx < NG; ++gidx)
                    for (int c = 0; c < 8; ++c)
                        segs.push_back({host[gidx] + (size_t)c * npts + (size_t)cb * plane, slot + (size_t)gidx * grid_slot + (size_t)c * cpts, cpts, 0});
                (void)parallel_copy(segs, copy_threads);
                drained.store(j + 1);
            }
        });
        for (int j = 0; j < nchunks && !rc; ++j) {
            while (j - drained.load() >= R) std::this_thread::yield();        // ring slot still being emptied
            const int cb = b + j * cx, ce = std::min(e, cb + cx);
            const size_t cpts = (size_t)(ce - cb) * plane;
            float* slot = h_ring + (size_t)(j % R) * slot_floats;
            if (hipStreamWaitEvent(s_copy, ev_comp[j], 0) != hipSuccess) rc = bad(CEG_ERR_HIP, "hipStreamWaitEvent failed");
            for (int gidx = 0; gidx < NG && !rc; ++gidx)
                if (hipMemcpy2DAsync(slot + (size_t)gidx * grid_slot, sizeof(float) * cpts,
                                     d_all + (size_t)gidx * 8 * slab_pts + (size_t)(cb - b) * plane, sizeof(float) * (size_t)slab_pts,
                                     sizeof(float) * cpts, 8, hipMemcpyDeviceToHost, s_copy) != hipSuccess)
                    rc = bad(CEG_ERR_HIP, "hipMemcpy2DAsync D2H failed");
            if (!rc && hipEventRecord(ev_copy[j], s_copy) != hipSuccess) rc = bad(CEG_ERR_HIP, "hipEventRecord failed");
            if (rc) drain_rc.store(rc);
            enqueued.store(j + 1);
        }
        if (rc) enqueued.store(nchunks);       // let the drain thread run to its end
        stamp("copies enqueued");
        drain.join();
        stamp("drained into caller arrays");
        if (!rc && drain_rc.load() != CEG_OK) rc = bad(CEG_ERR_HIP, "kernel execution or D2H copy failed");
    }
    (void)hipStreamSynchronize(s_comp);
    (void)hipStreamSynchronize(s_copy);
    for (int j = 0; j < nchunks; ++j) {
        if (ev_comp[j]) (void)hipEventDestroy(ev_comp[j]);
        if (ev_copy[j]) (void)hipEventDestroy(ev_copy[j]);
    }
    if (h_ring) pinned_release(h_ring);
    if (d_all) device_release(d_all);
    if (s_comp) streams_release(s_comp);
    ceg_plan_destroy(plan);
    stamp("cleaned up");
    return rc;
}

// One slab of the device-resident one-shot build: slab 0 is built straight into the assembled grid on the target device; the other
// slabs are built on their own devices and travel to the target chunk by chunk with hipMemcpyPeerAsync (xGMI between the GPUs of a
// node) while the later chunks are still being computed.
int resident_pipeline(int mode, int slab_index, int d, int target, int b, int e, int nx, int64_t plane, const double* pos,
                      const int64_t* atomkind, const double* charge, int64_t natoms, const double* mat, const double* invmat,
                      int32_t ortho, double safemin2, double cutoff2, const ceg_rule_t* rules, const int32_t* rule_offset,
                      int32_t nkinds, double alpha, const int32_t* dims, const double* size, const double* shift,
                      const double* delta, double lambda, double threshold, float* d_grid, std::string* err)
{
    auto bad = [&](int code, const char* what) {
        *err = std::string(what) + " (device " + std::to_string(d) + "): " + hipGetErrorString(hipGetLastError());
        return code;
    };
    const int64_t npts = plane * nx;
    const int64_t slab_pts = (int64_t)(e - b) * plane;
    if (slab_pts <= 0) return CEG_OK;
    if (hipSetDevice(d) != hipSuccess) return bad(CEG_ERR_HIP, "hipSetDevice failed");
    ceg_plan* plan = nullptr;
    int rc = ceg_plan_create(&plan, d, pos, atomkind, charge, natoms, mat, invmat, ortho, safemin2, cutoff2, rules, rule_offset,
                             nkinds, alpha, dims, size, shift, delta);
    if (rc) { *err = g_err; return rc; }
    hipStream_t s_comp = nullptr, s_copy = nullptr;
    if (!streams_acquire(d, &s_comp, &s_copy)) { ceg_plan_destroy(plan); return bad(CEG_ERR_HIP, "stream creation failed"); }
    const bool direct = slab_index == 0;
    int cx = (int)std::max<int64_t>(4, ((32ll << 20) / (plane * 8 * (int64_t)sizeof(float))) / 4 * 4);
    cx = std::min(cx, (e - b + 3) / 4 * 4);
    const int nchunks = (e - b + cx - 1) / cx;
    float* d_out = nullptr;
    hipEvent_t ev = nullptr;
    // Can this device write into the target's memory?  (xGMI peers of one node: yes.)  If not -- or with CEG_HIP_NO_PEER=1, a
    // rehearsal aid -- the slab is not pushed chunk by chunk with hipMemcpyPeerAsync but travels through a pinned host buffer once
    // it is complete (D2H on this device, H2D on the target), and a note says so.
    bool peer = true;
    if (!direct) {
        if (d != target) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, d, target) != hipSuccess) can = 0;
            (void)hipGetLastError();
            if (can) {
                (void)hipDeviceEnablePeerAccess(target, 0);                 // "already enabled" is fine
                (void)hipGetLastError();
            }
            peer = can != 0;
        }
        if (std::getenv("CEG_HIP_NO_PEER")) peer = false;
        if (!peer) {
            static std::atomic<bool> said{false};
            if (!said.exchange(true))
                fprintf(stderr, "[ceg_hip] note: no peer access from device %d to device %d%s: slabs of the device-resident build travel "
                                "through pinned host memory\n", d, target, std::getenv("CEG_HIP_NO_PEER") ? " (CEG_HIP_NO_PEER)" : "");
        }
        d_out = static_cast<float*>(device_acquire(d, sizeof(float) * 8 * slab_pts));
        if (!d_out || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) rc = bad(CEG_ERR_HIP, "allocation of the slab buffer failed");
    }
    for (int j = 0; j < nchunks && !rc; ++j) {
        const int cb = b + j * cx, ce = std::min(e, cb + cx);
        float* out = direct ? d_grid : d_out;
        const int64_t stride = direct ? npts : slab_pts;
        const int origin = direct ? 0 : b;
        rc = (mode == MODE_VDW) ? ceg_plan_build_vdw(plan, lambda, threshold, cb, ce, out, stride, origin, CEG_ALGO_AUTO, s_comp)
                                : ceg_plan_build_coulomb(plan, lambda, threshold, cb, ce, out, stride, origin, CEG_ALGO_AUTO, s_comp);
        if (rc) { *err = g_err; break; }
        if (direct || !peer) continue;
        if (hipEventRecord(ev, s_comp) != hipSuccess || hipStreamWaitEvent(s_copy, ev, 0) != hipSuccess) { rc = bad(CEG_ERR_HIP, "event failed"); break; }
        const size_t cpts = (size_t)(ce - cb) * plane;
        for (int c = 0; c < 8 && !rc; ++c)
            if (hipMemcpyPeerAsync(d_grid + (size_t)c * npts + (size_t)cb * plane, target, d_out + (size_t)c * slab_pts + (size_t)(cb - b) * plane, d,
                                   sizeof(float) * cpts, s_copy) != hipSuccess)
                rc = bad(CEG_ERR_HIP, "hipMemcpyPeerAsync failed");
    }
    if (hipStreamSynchronize(s_comp) != hipSuccess && !rc) rc = bad(CEG_ERR_HIP, "kernel execution failed");
    if (hipStreamSynchronize(s_copy) != hipSuccess && !rc) rc = bad(CEG_ERR_HIP, "peer copy failed");
    if (!rc && !direct && !peer) {                         // host-staged hand-over of the finished slab
        float* h_buf = static_cast<float*>(pinned_acquire(sizeof(float) * 8 * (size_t)slab_pts));
        if (!h_buf) rc = bad(CEG_ERR_HIP, "pinned buffer allocation failed");
        if (!rc && hipMemcpy(h_buf, d_out, sizeof(float) * 8 * (size_t)slab_pts, hipMemcpyDeviceToHost) != hipSuccess) rc = bad(CEG_ERR_HIP, "D2H of the slab failed");
        if (!rc && hipSetDevice(target) != hipSuccess) rc = bad(CEG_ERR_HIP, "hipSetDevice(target) failed");
        for (int c = 0; c < 8 && !rc; ++c)
            if (hipMemcpy(d_grid + (size_t)c * npts + (size_t)b * plane, h_buf + (size_t)c * slab_pts, sizeof(float) * (size_t)slab_pts,
                          hipMemcpyHostToDevice) != hipSuccess)
                rc = bad(CEG_ERR_HIP, "H2D of the slab failed");
        (void)hipSetDevice(d);
        if (h_buf) pinned_release(h_buf);
    }
    if (ev) (void)hipEventDestroy(ev);
    if (d_out) device_release(d_out);
    streams_release(s_comp);
    ceg_plan_destroy(plan);
    return rc;
}

int oneshot_resident(int mode, const double* pos, const int64_t* atomkind, const double* charge, int64_t natoms, const double* mat,
                     const double* invmat, int32_t ortho, double safemin2, double cutoff2, const ceg_rule_t* rules,
                     const int32_t* rule_offset, int32_t nkinds, double alpha, const int32_t* dims, const double* size,
                     const double* shift, const double* delta, double lambda, double threshold, float* d_grid, int32_t target,
                     int32_t ngpus)
{
    if (!d_grid) return fail(CEG_ERR_INVALID, "d_grid is NULL");
    if (int rc = check_common(pos, natoms, mat, invmat, dims, size, shift, delta)) return rc;
    const int ndev = ceg_device_count();
    if (ndev <= 0) return fail(CEG_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (target < 0 || target >= ndev) return fail(CEG_ERR_NO_DEVICE, "target device %d not present (%d devices)", target, ndev);
    const bool oversubscribe = std::getenv("CEG_HIP_OVERSUBSCRIBE") != nullptr;      // rehearsal on one card, as in oneshot()
    if (ngpus < 1 || (ngpus > ndev && !oversubscribe))
        return fail(CEG_ERR_NO_DEVICE, "ngpus = %d but %d HIP devices are present", ngpus, ndev);
    const int nx = dims[0] + 1;
    const int64_t plane = (int64_t)(dims[1] + 1) * (dims[2] + 1);
    ngpus = std::min(ngpus, nx);
    int prev = -1;
    (void)hipGetDevice(&prev);
    std::vector<int> rcs(ngpus, CEG_OK);
    std::vector<std::string> errs(ngpus);
    auto run = [&](int slab_index) {
        int b, e;
        slab(nx, ngpus, slab_index, &b, &e);
        rcs[slab_index] = resident_pipeline(mode, slab_index, (target + slab_index) % ndev, target, b, e, nx, plane, pos, atomkind, charge, natoms, mat,
                                            invmat, ortho, safemin2, cutoff2, rules, rule_offset, nkinds, alpha, dims, size, shift, delta, lambda,
                                            threshold, d_grid, &errs[slab_index]);
    };
    std::vector<std::thread> workers;
    for (int t = 1; t < ngpus; ++t) workers.emplace_back(run, t);
    run(0);
    for (auto& w : workers) w.join();
    if (prev >= 0) (void)hipSetDevice(prev);
    for (int t = 0; t < ngpus; ++t)
        if (rcs[t]) return fail(rcs[t], "%s", errs[t].c_str());
    return CEG_OK;
}

}  // namespace

extern "C" int ceg_grid_vdw_device(const double* pos, const int64_t* atomkind, int64_t natoms, const double mat[9], const double invmat[9],
                                   int32_t ortho, double safemin2, double cutoff2, const ceg_rule_t* rules, const int32_t* rule_offset,
                                   int32_t nkinds, const int32_t dims[3], const double size[3], const double shift[3],
                                   const double delta[3], double lambda, double threshold, float* d_grid, int32_t target_device,
                                   int32_t ngpus)
{
    if (!rules || !rule_offset || !atomkind || nkinds <= 0) return fail(CEG_ERR_INVALID, "rule table / atomkind missing");
    return oneshot_resident(MODE_VDW, pos, atomkind, nullptr, natoms, mat, invmat, ortho, safemin2, cutoff2, rules, rule_offset, nkinds, 0.0,
                            dims, size, shift, delta, lambda, threshold, d_grid, target_device, ngpus);
}

extern "C" int ceg_grid_coulomb_device(const double* pos, const double* charge, int64_t natoms, const double mat[9], const double invmat[9],
                                       int32_t ortho, double safemin2, double cutoff2, double alpha, const int32_t dims[3],
                                       const double size[3], const double shift[3], const double delta[3], double lambda,
                                       double threshold, float* d_grid, int32_t target_device, int32_t ngpus)
{
    if (!charge) return fail(CEG_ERR_INVALID, "charge is NULL");
    return oneshot_resident(MODE_COULOMB, pos, nullptr, charge, natoms, mat, invmat, ortho, safemin2, cutoff2, nullptr, nullptr, 0, alpha, dims,
                            size, shift, delta, lambda, threshold, d_grid, target_device, ngpus);
}

extern "C" int ceg_grids_multi(const double* pos, const int64_t* atomkind, const double* charge, int64_t natoms, const double mat[9],
                               const double invmat[9], int32_t ortho, double safemin2, double cutoff2, int32_t nprobes,
                               const ceg_rule_t* const* rules, const int32_t* const* rule_offset, int32_t nkinds, double alpha,
                               const int32_t dims[3], const double size[3], const double shift[3], const double delta[3],
                               double lambda_vdw, double threshold_vdw, double lambda_coulomb, double threshold_coulomb,
                               float* const* grids_vdw, float* grid_coulomb, int32_t ngpus)
{
    if (nprobes < 1 || nprobes > CEG_MAX_PROBES) return fail(CEG_ERR_INVALID, "nprobes = %d outside 1..%d", nprobes, CEG_MAX_PROBES);
    if (!rules || !rule_offset || !atomkind || nkinds <= 0) return fail(CEG_ERR_INVALID, "rule tables / atomkind missing");
    if (grid_coulomb && !charge) return fail(CEG_ERR_INVALID, "charge is NULL");
    if (int rc = check_common(pos, natoms, mat, invmat, dims, size, shift, delta)) return rc;
    const int ndev = ceg_device_count();
    if (ndev <= 0) return fail(CEG_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    const bool oversubscribe = std::getenv("CEG_HIP_OVERSUBSCRIBE") != nullptr;
    if (ngpus < 1 || (ngpus > ndev && !oversubscribe)) return fail(CEG_ERR_NO_DEVICE, "ngpus = %d but %d HIP devices are present", ngpus, ndev);
    const int nx = dims[0] + 1;
    const int64_t plane = (int64_t)(dims[1] + 1) * (dims[2] + 1);
    ngpus = std::min(ngpus, nx);
    int prev = -1;
    (void)hipGetDevice(&prev);
    int copy_threads = 8;
    if (const char* env = std::getenv("CEG_HIP_COPY_THREADS")) copy_threads = std::max(1, atoi(env));
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw > 0) copy_threads = std::min<int>(copy_threads, (int)hw);
    copy_threads = std::max(1, copy_threads / ngpus);
    std::vector<int> rcs(ngpus, CEG_OK);
    std::vector<std::string> errs(ngpus);
    auto run = [&](int d) {
        int b, e;
        slab(nx, ngpus, d, &b, &e);
        rcs[d] = multi_device_pipeline(d % ndev, b, e, nx, plane, pos, atomkind, charge, natoms, mat, invmat, ortho, safemin2, cutoff2, nprobes,
                                       rules, rule_offset, nkinds, alpha, dims, size, shift, delta, lambda_vdw, threshold_vdw, lambda_coulomb,
                                       threshold_coulomb, grids_vdw, grid_coulomb, copy_threads, &errs[d]);
    };
    std::vector<std::thread> workers;
    for (int d = 1; d < ngpus; ++d) workers.emplace_back(run, d);
    run(0);
    for (auto& w : workers) w.join();
    if (prev >= 0) (void)hipSetDevice(prev);
    for (int d = 0; d < ngpus; ++d)
        if (rcs[d]) return fail(rcs[d], "%s", errs[d].c_str());
    return CEG_OK;
}

// ------------------------------------------------------------------ page-locked result arrays
// What the one-shot entry points hand back is 537 MB (256^3) per grid; into an ordinary host array that is D2H into a pinned ring
// plus a second pass by host threads (first touch of fresh pages).  A caller that lets the LIBRARY allocate the result gets
// page-locked memory (kept in the per-process cache, so the page-locking is paid once) and the pipelines copy every chunk straight
// to its place: the call is then bounded by the D2H alone.
// Page-locked result arrays in callers' hands are bounded (CEG_HIP_PINNED_LIMIT_MB, default 4096): a garbage-collected caller (the
// Julia shim wraps them as Arrays and returns them with a finalizer) cannot pin host memory without limit while its collector has not
// run yet -- beyond the limit the call fails with CEG_ERR_UNSUPPORTED and the shim falls back to an ordinary array.
extern "C" float* ceg_host_grid_alloc(const int32_t dims[3])
{
    if (!dims || dims[0] < 1 || dims[1] < 1 || dims[2] < 1) { (void)fail(CEG_ERR_INVALID, "bad dims"); return nullptr; }
    if (ceg_device_count() <= 0) { (void)fail(CEG_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)"); return nullptr; }
    const size_t bytes = sizeof(float) * 8 * (size_t)(dims[0] + 1) * (size_t)(dims[1] + 1) * (size_t)(dims[2] + 1);
    size_t limit = (size_t)4096 << 20;
    if (const char* e = std::getenv("CEG_HIP_PINNED_LIMIT_MB")) limit = (size_t)std::max(0ll, atoll(e)) << 20;
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        size_t out = 0;
        for (const auto& p : g_pinned)
            if (p.busy && p.user) out += p.bytes;
        if (out + bytes > limit) {
            (void)fail(CEG_ERR_UNSUPPORTED, "page-locked result arrays in use (%zu MB) + this one (%zu MB) exceed CEG_HIP_PINNED_LIMIT_MB = %zu: "
                                            "free some (ceg_host_grid_free) or use an ordinary array", out >> 20, bytes >> 20, limit >> 20);
            return nullptr;
        }
    }
    void* p = pinned_acquire(bytes);
    if (!p) { (void)fail(CEG_ERR_HIP, "page-locked allocation of %zu bytes failed", bytes); return nullptr; }
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        for (auto& q : g_pinned)
            if (q.ptr == p) q.user = true;
    }
    return static_cast<float*>(p);
}

extern "C" int ceg_host_grid_free(float* grid)
{
    if (!grid) return CEG_OK;
    if (!pinned_owns(grid, 1)) return fail(CEG_ERR_INVALID, "not an array of ceg_host_grid_alloc");
    pinned_release(grid);          // back to the cache; ceg_release_cached_buffers unpins it
    return CEG_OK;
}

// ------------------------------------------------------------------ cached .grid file -> interpolation handle
// The reference's common case is not "create" but "Retrieved ... grid" (src/raspa.jl:426-438 -> parse_grid, src/grids.jl:61-94):
// the file is read into a host array, multiplied by GRID_TO_KELVIN, and only then would a GPU consumer upload it and make its
// node-major copy.  Here the payload goes file -> pinned ring (parallel pread) -> device while the next chunk is being read,
// is scaled on the device exactly like grids.jl:78 (Float32(Float64(x) * scale)) and handed to ceg_interp_create in place.
extern "C" int ceg_interp_create(ceg_interp_t** handle, int32_t device, const float* grid, int32_t grid_on_device, const int32_t dims[3],
                                 const double size[3], const double shift[3], const double mat[9], const double invmat[9], int32_t is_vdw);
extern "C" int ceg_scale_grid_device(float* d_grid, int64_t nfloats, double scale, int32_t device, void* stream);

extern "C" int ceg_interp_create_from_file(ceg_interp_t** handle, int32_t device, const char* path, int32_t iscoulomb, double scale,
                                           const double* mat, const double* invmat, ceg_grid_header_t* header_out)
{
    if (!handle || !path) return fail(CEG_ERR_INVALID, "NULL argument");
    *handle = nullptr;
    if ((mat == nullptr) != (invmat == nullptr)) return fail(CEG_ERR_INVALID, "mat and invmat go together");
    const int ndev = ceg_device_count();
    if (ndev <= 0) return fail(CEG_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(CEG_ERR_NO_DEVICE, "device %d not present (%d devices)", device, ndev);
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(CEG_ERR_INVALID, "cannot open %s", path);
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    // header (src/grids.jl:62-75, written by :108-116): f64 spacing, 3 x i32 dims, 3 x f64 size, shift, delta, unitcell lengths,
    // 3 x i32 num_unitcell [, f64 Ewald precision]
    unsigned char hb[136];
    const size_t hbytes = iscoulomb ? 136 : 128;
    if (pread(fd, hb, hbytes, 0) != (ssize_t)hbytes) return fail(CEG_ERR_INVALID, "%s: truncated header", path);
    ceg_grid_header_t H{};
    size_t o = 0;
    auto rd = [&](void* dst, size_t n) { memcpy(dst, hb + o, n); o += n; };
    rd(&H.spacing, 8); rd(H.dims, 12); rd(H.size, 24); rd(H.shift, 24); rd(H.delta, 24); rd(H.unitcell, 24); rd(H.num_unitcell, 12);
    H.ewald_precision = std::numeric_limits<double>::infinity();          // EnergyGrid(..., Inf, ...) for a VdW grid (grids.jl:92)
    if (iscoulomb) rd(&H.ewald_precision, 8);
    for (int a = 0; a < 3; ++a)
        if (H.dims[a] < 1 || H.dims[a] > (1 << 20) || !(H.size[a] > 0.0)) return fail(CEG_ERR_INVALID, "%s: not a .grid header (dims / size)", path);
    // the node count is bounded by the FILE SIZE before anything is multiplied: with dims up to 2^20 per axis the product
    // reaches 2^60 and nodes * 32 wraps around int64, so that a crafted header could satisfy the size check below with a small file
    const off_t fsize = lseek(fd, 0, SEEK_END);
    const int64_t max_nodes = fsize > (off_t)hbytes ? (int64_t)(fsize - (off_t)hbytes) / 32 : 0;
    int64_t nodes = 1;
    for (int a = 0; a < 3; ++a) {
        const int64_t ext = (int64_t)H.dims[a] + 1;
        if (nodes > max_nodes / ext) return fail(CEG_ERR_INVALID, "%s: file shorter than its header says", path);
        nodes *= ext;
    }
    const int64_t nfl = 8 * nodes;
    const int64_t payload = nfl * (int64_t)sizeof(float);
    if (fsize < (off_t)(hbytes + payload)) return fail(CEG_ERR_INVALID, "%s: file shorter than its header says", path);
    // header + payload [+ the 72-byte cell matrix] and nothing else: a VdW file opened as a Coulomb one (or the reverse) is off by the
    // 8 bytes of the Ewald precision and is refused here instead of being read 8 bytes out of step
    if (fsize != (off_t)(hbytes + payload) && fsize != (off_t)(hbytes + payload + 72))
        return fail(CEG_ERR_INVALID, "%s: %lld bytes do not make a %s grid of %d x %d x %d points (wrong iscoulomb?)", path, (long long)fsize,
                    iscoulomb ? "Coulomb" : "VdW", H.dims[0] + 1, H.dims[1] + 1, H.dims[2] + 1);
    H.has_mat = 0;
    if (fsize == (off_t)(hbytes + payload + 72) && pread(fd, H.mat, 72, (off_t)(hbytes + payload)) == 72) H.has_mat = 1;   // :154 / :182
    double M[9], I[9];
    if (mat) { memcpy(M, mat, sizeof M); memcpy(I, invmat, sizeof I); }
    else {
        if (!H.has_mat) return fail(CEG_ERR_INVALID, "%s carries no cell matrix: pass mat / invmat (parse_grid's `mat` argument, grids.jl:80-90)", path);
        memcpy(M, H.mat, sizeof M);
        const double* m = M;                         // column-major: m[i + 3 j]
        const double c00 = m[4] * m[8] - m[7] * m[5], c01 = m[7] * m[2] - m[1] * m[8], c02 = m[1] * m[5] - m[4] * m[2];
        const double det = m[0] * c00 + m[3] * c01 + m[6] * c02;
        if (!(std::fabs(det) > 0.0)) return fail(CEG_ERR_INVALID, "%s: singular cell matrix", path);
        const double id = 1.0 / det;
        I[0] = c00 * id; I[1] = c01 * id; I[2] = c02 * id;
        I[3] = (m[6] * m[5] - m[3] * m[8]) * id; I[4] = (m[0] * m[8] - m[6] * m[2]) * id; I[5] = (m[3] * m[2] - m[0] * m[5]) * id;
        I[6] = (m[3] * m[7] - m[6] * m[4]) * id; I[7] = (m[6] * m[1] - m[0] * m[7]) * id; I[8] = (m[0] * m[4] - m[3] * m[1]) * id;
    }
    if (header_out) *header_out = H;
    DeviceGuard guard(device);
    if (!guard.ok) return fail(CEG_ERR_HIP, "hipSetDevice(%d) failed", device);
    float* d_raw = nullptr;
    HIP_TRY(hipMalloc((void**)&d_raw, (size_t)payload));
    hipStream_t s_comp = nullptr, s_copy = nullptr;
    if (!streams_acquire(device, &s_comp, &s_copy)) { (void)hipFree(d_raw); return fail(CEG_ERR_HIP, "stream creation failed"); }
    const size_t slot = 32ull << 20;
    const int nchunks = (int)((payload + (int64_t)slot - 1) / (int64_t)slot);
    const int R = std::min(3, nchunks);
    char* ring = static_cast<char*>(pinned_acquire(slot * R));
    std::vector<hipEvent_t> ev(R, nullptr);
    int rc = ring ? CEG_OK : fail(CEG_ERR_HIP, "pinned buffer allocation failed");
    for (int t = 0; t < R && !rc; ++t)
        if (hipEventCreateWithFlags(&ev[t], hipEventDisableTiming) != hipSuccess) rc = fail(CEG_ERR_HIP, "event creation failed");
    int nthreads = 8;
    if (const char* env = std::getenv("CEG_HIP_COPY_THREADS")) nthreads = std::max(1, atoi(env));
    for (int j = 0; j < nchunks && !rc; ++j) {
        char* buf = ring + (size_t)(j % R) * slot;
        if (j >= R && hipEventSynchronize(ev[j % R]) != hipSuccess) { rc = fail(CEG_ERR_HIP, "H2D copy failed"); break; }
        const int64_t off = (int64_t)j * (int64_t)slot, len = std::min<int64_t>((int64_t)slot, payload - off);
        std::atomic<bool> ok{true};
        std::atomic<int64_t> next{0};
        const int64_t piece = 1 << 20;
        auto work = [&]() {
            for (;;) {
                const int64_t b = next.fetch_add(piece);
                if (b >= len) return;
                int64_t left = std::min(piece, len - b), at = b;
                while (left > 0) {
                    const ssize_t got = pread(fd, buf + at, (size_t)left, (off_t)(hbytes + off + at));
                    if (got <= 0) { ok.store(false); return; }
                    left -= got; at += got;
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(work);
        work();
        for (auto& x : th) x.join();
        if (!ok.load()) { rc = fail(CEG_ERR_INVALID, "%s: read error", path); break; }
        if (hipMemcpyAsync(reinterpret_cast<char*>(d_raw) + off, buf, (size_t)len, hipMemcpyHostToDevice, s_copy) != hipSuccess ||
            hipEventRecord(ev[j % R], s_copy) != hipSuccess)
            rc = fail(CEG_ERR_HIP, "hipMemcpyAsync H2D failed");
    }
    if (hipStreamSynchronize(s_copy) != hipSuccess && !rc) rc = fail(CEG_ERR_HIP, "H2D copy failed");
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    if (ring) pinned_release(ring);
    if (!rc && scale != 1.0) {                       // grid .*= GRID_TO_KELVIN (grids.jl:78)
        rc = ceg_scale_grid_device(d_raw, nfl, scale, device, s_copy);
        if (!rc && hipStreamSynchronize(s_copy) != hipSuccess) rc = fail(CEG_ERR_HIP, "scaling kernel failed");
    }
    streams_release(s_comp);
    if (!rc) rc = ceg_interp_create(handle, device, d_raw, 1, H.dims, H.size, H.shift, M, I, iscoulomb ? 0 : 1);
    (void)hipFree(d_raw);
    return rc;
}

extern "C" int ceg_release_cached_buffers(void)
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (size_t t = g_devbufs.size(); t-- > 0;)
        if (!g_devbufs[t].busy) {
            if (hipSetDevice(g_devbufs[t].device) == hipSuccess) (void)hipFree(g_devbufs[t].ptr);
            g_devbufs.erase(g_devbufs.begin() + t);
        }
    for (size_t t = g_pinned.size(); t-- > 0;)
        if (!g_pinned[t].busy) {
            (void)hipHostFree(g_pinned[t].ptr);
            g_pinned.erase(g_pinned.begin() + t);
        }
    for (size_t t = g_streams.size(); t-- > 0;)
        if (!g_streams[t].busy) {
            if (hipSetDevice(g_streams[t].device) == hipSuccess) {
                (void)hipStreamDestroy(g_streams[t].comp);
                (void)hipStreamDestroy(g_streams[t].copy);
            }
            g_streams.erase(g_streams.begin() + t);
        }
    if (prev >= 0) (void)hipSetDevice(prev);
    image_cache_release();
    block_cache_release();
    return CEG_OK;
}

extern "C" int ceg_grid_vdw(const double* pos, const int64_t* atomkind, int64_t natoms,
                            const double mat[9], const double invmat[9],
                            int32_t ortho, double safemin2, double cutoff2,
                            const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds,
                            const int32_t dims[3], const double size[3], const double shift[3],
                            const double delta[3], double lambda, double threshold,
                            float* grid, int32_t ngpus)
{
    if (!rules || !rule_offset || !atomkind || nkinds <= 0) return fail(CEG_ERR_INVALID, "rule table / atomkind missing");
    return oneshot(MODE_VDW, pos, atomkind, nullptr, natoms, mat, invmat, ortho, safemin2, cutoff2, rules,
                   rule_offset, nkinds, 0.0, dims, size, shift, delta, lambda, threshold, grid, ngpus);
}

extern "C" int ceg_grid_vdw_file(const double* pos, const int64_t* atomkind, int64_t natoms, const double mat[9], const double invmat[9],
                                 int32_t ortho, double safemin2, double cutoff2, const ceg_rule_t* rules, const int32_t* rule_offset,
                                 int32_t nkinds, const int32_t dims[3], const double size[3], const double shift[3],
                                 const double delta[3], double lambda, double threshold, float* grid, int32_t ngpus,
                                 const char* path, const void* header, int64_t header_bytes, const void* trailer, int64_t trailer_bytes)
{
    if (!rules || !rule_offset || !atomkind || nkinds <= 0) return fail(CEG_ERR_INVALID, "rule table / atomkind missing");
    if (!path) return fail(CEG_ERR_INVALID, "path is NULL");
    return oneshot(MODE_VDW, pos, atomkind, nullptr, natoms, mat, invmat, ortho, safemin2, cutoff2, rules, rule_offset, nkinds, 0.0, dims,
                   size, shift, delta, lambda, threshold, grid, ngpus, path, header, header_bytes, trailer, trailer_bytes);
}

extern "C" int ceg_grid_coulomb_file(const double* pos, const double* charge, int64_t natoms, const double mat[9], const double invmat[9],
                                     int32_t ortho, double safemin2, double cutoff2, double alpha, const int32_t dims[3],
                                     const double size[3], const double shift[3], const double delta[3], double lambda, double threshold,
                                     float* grid, int32_t ngpus, const char* path, const void* header, int64_t header_bytes,
                                     const void* trailer, int64_t trailer_bytes)
{
    if (!charge) return fail(CEG_ERR_INVALID, "charge is NULL");
    if (!path) return fail(CEG_ERR_INVALID, "path is NULL");
    return oneshot(MODE_COULOMB, pos, nullptr, charge, natoms, mat, invmat, ortho, safemin2, cutoff2, nullptr, nullptr, 0, alpha, dims, size,
                   shift, delta, lambda, threshold, grid, ngpus, path, header, header_bytes, trailer, trailer_bytes);
}

extern "C" int ceg_grid_coulomb(const double* pos, const double* charge, int64_t natoms,
                                const double mat[9], const double invmat[9],
                                int32_t ortho, double safemin2, double cutoff2, double alpha,
                                const int32_t dims[3], const double size[3], const double shift[3],
                                const double delta[3], double lambda, double threshold,
                                float* grid, int32_t ngpus)
{
    if (!charge) return fail(CEG_ERR_INVALID, "charge is NULL");
    return oneshot(MODE_COULOMB, pos, nullptr, charge, natoms, mat, invmat, ortho, safemin2, cutoff2, nullptr,
                   nullptr, 0, alpha, dims, size, shift, delta, lambda, threshold, grid, ngpus);
}
