/*
 * ceg_hip.h -- C ABI of libceg_hip.so, the MI355X (gfx950) energy-grid builder.
 *
 * Drop-in boundary for the grid-build hot path of CrystalEnergyGrids.jl.  The
 * reference has no FFI of its own (it is 100 % Julia); the boundary is placed
 * at the two loop nests that fill the grid array:
 *
 *     create_grid_vdw      src/grids.jl:144-150   (calls src/probes.jl:71-92)
 *     create_grid_coulomb  src/grids.jl:171-177   (calls src/probes.jl:94-117)
 *
 * Everything above those loops (CIF / force-field parsing, ProbeSystem,
 * GridCoordinatesSetup, initialize_ewald, unit constants, the .grid writer)
 * stays in the host language and hands this library plain arrays.
 *
 * Conventions
 *   - all pointers are borrowed for the duration of the call; nothing is retained
 *     except inside a ceg_plan_t, which copies what it needs to the device;
 *   - 3x3 matrices are column-major (Julia SMatrix order);
 *   - return value 0 = ok, <0 = error, message via ceg_last_error() (thread-local);
 *   - no physical constants live in the library: lambda / threshold / alpha are
 *     arguments computed by the host (src/grids.jl:141-143,168-170, src/ewald.jl:198-204);
 *   - there is NO CPU fallback: without a HIP device every compute entry point
 *     returns CEG_ERR_NO_DEVICE.
 */
#ifndef CEG_HIP_H
#define CEG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CEG_ABI_VERSION 1

#if defined(__GNUC__)
#define CEG_API __attribute__((visibility("default")))
#else
#define CEG_API
#endif

/* error codes */
#define CEG_OK                 0
#define CEG_ERR_INVALID       -1   /* bad argument (null pointer, negative size, ...)        */
#define CEG_ERR_NO_DEVICE     -2   /* no HIP device / requested device not present           */
#define CEG_ERR_HIP           -3   /* a HIP runtime call failed                              */
#define CEG_ERR_RULE          -4   /* rule kind not valid in a VdW grid (mirrors the Julia
                                      error()/throw sites of src/interactions.jl:442-467)    */
#define CEG_ERR_UNSUPPORTED   -5

/* Interaction kinds.  Numeric values = order of `@enum InteractionKind`,
 * src/interactions.jl:23-33. */
enum ceg_kind {
    CEG_HARDSPHERE            = 0,
    CEG_COULOMB_EWALD_DIRECT  = 1,
    CEG_COULOMB               = 2,
    CEG_LENNARDJONES          = 3,
    CEG_BUCKINGHAM            = 4,
    CEG_MONOMIAL              = 5,
    CEG_EXPONENTIAL           = 6,
    CEG_UNDEFINED_INTERACTION = 7,
    CEG_NOINTERACTION         = 8
};

/* One InteractionRule (src/interactions.jl:232-237): kind, params (<=3), shift.
 * An InteractionRuleSum (src/interactions.jl:557-583) is a run of these. */
typedef struct ceg_rule {
    int32_t kind;
    int32_t _pad;
    double  p[3];
    double  shift;
} ceg_rule_t;

/* Algorithm selector for the build entry points. */
enum ceg_algo {
    CEG_ALGO_AUTO       = 0,  /* culled when the supercell allows it, else brute force   */
    CEG_ALGO_BRUTEFORCE = 1,  /* every atom tested for every point, literal min-image
                                 routine (src/utils.jl:210-246); reference loop shape    */
    CEG_ALGO_CULLED     = 2   /* lattice-image list + spatial bins; same selection rule
                                 as the reference, evaluated per image                   */
};

/* ---- library / device info ------------------------------------------------ */
CEG_API int         ceg_abi_version(void);
CEG_API int         ceg_device_count(void);         /* number of HIP devices, 0 if none   */
CEG_API const char* ceg_last_error(void);           /* thread-local, never NULL           */

/* ---- one-shot host API ---------------------------------------------------- */
/*
 * Fills `grid` exactly like the loop nest of create_grid_vdw (src/grids.jl:144-150):
 * for every (i,j,k) in 0:dims[0] x 0:dims[1] x 0:dims[2],
 *   pos   = abc_to_xyz(i,j,k)                       src/coordinates.jl:72-76
 *   deriv = compute_derivatives_vdw(probe, pos)     src/probes.jl:71-92
 *   _set_gridpoint!(grid,i,j,k,delta,lambda,threshold,deriv)  src/grids.jl:118-135
 *
 *  pos       [3*natoms]  cartesian A, supercell-tiled ProbeSystem.positions (src/probes.jl:29-55)
 *  atomkind  [natoms]    1-based force-field index   (src/probes.jl:23,54)
 *  mat,invmat            supercell matrix / inverse  (src/probes.jl:27-28), column-major
 *  ortho,safemin2        from prepare_periodic_distance_computations (src/utils.jl:146-155)
 *  cutoff2               forcefield.cutoff^2         (src/probes.jl:75)
 *  rules,rule_offset     rules[rule_offset[k-1] .. rule_offset[k]) = flattened
 *                        forcefield.interactions[k, probe] (src/forcefields.jl:302-304);
 *                        rule_offset has nkinds+1 entries
 *  dims,size,shift,delta GridCoordinatesSetup fields (src/coordinates.jl:32-41), A
 *  lambda,threshold      1/GRID_TO_KELVIN, GRID_TO_KELVIN*1e7 (src/grids.jl:141-143,148)
 *  grid      [(dims[2]+1)*(dims[1]+1)*(dims[0]+1)*8] float, host memory,
 *            column-major [z,y,x,channel] (src/grids.jl:126-133)
 *  ngpus     1..ceg_device_count(): x-slabs are spread over that many devices
 */
CEG_API int ceg_grid_vdw(const double* pos, const int64_t* atomkind, int64_t natoms,
                 const double mat[9], const double invmat[9],
                 int32_t ortho, double safemin2, double cutoff2,
                 const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds,
                 const int32_t dims[3], const double size[3], const double shift[3],
                 const double delta[3],
                 double lambda, double threshold,
                 float* grid, int32_t ngpus);

/*
 * Same for create_grid_coulomb (src/grids.jl:171-177) with
 * compute_derivatives_ewald (src/probes.jl:94-117) / derivatives_ewald
 * (src/ewald.jl:299-312).
 *  charge [natoms] e, alpha = ewald.alpha in 1/A,
 *  lambda = COULOMBIC_CONVERSION_FACTOR/GRID_TO_KELVIN, threshold = 1e7/lambda
 *  (src/grids.jl:169-170).
 */
CEG_API int ceg_grid_coulomb(const double* pos, const double* charge, int64_t natoms,
                     const double mat[9], const double invmat[9],
                     int32_t ortho, double safemin2, double cutoff2, double alpha,
                     const int32_t dims[3], const double size[3], const double shift[3],
                     const double delta[3],
                     double lambda, double threshold,
                     float* grid, int32_t ngpus);

/*
 * The same two builds with the .grid file written on the way ("device -> file"):
 * `header` (the bytes of _create_grid_common, src/grids.jl:108-116, + the Ewald precision for a Coulomb
 * grid, :180) is written first, every finished chunk of the payload is written at its offset while later
 * chunks are still being computed, `trailer` (the cell matrix, :154/:182) goes after the payload.  The
 * library does not interpret header or trailer -- the host produces them with the reference's own writer.
 * `grid` may be NULL (file only) or a host array that is filled as by ceg_grid_vdw / ceg_grid_coulomb.
 */
CEG_API int ceg_grid_vdw_file(const double* pos, const int64_t* atomkind, int64_t natoms,
                 const double mat[9], const double invmat[9],
                 int32_t ortho, double safemin2, double cutoff2,
                 const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds,
                 const int32_t dims[3], const double size[3], const double shift[3],
                 const double delta[3], double lambda, double threshold,
                 float* grid, int32_t ngpus,
                 const char* path, const void* header, int64_t header_bytes,
                 const void* trailer, int64_t trailer_bytes);
CEG_API int ceg_grid_coulomb_file(const double* pos, const double* charge, int64_t natoms,
                     const double mat[9], const double invmat[9],
                     int32_t ortho, double safemin2, double cutoff2, double alpha,
                     const int32_t dims[3], const double size[3], const double shift[3],
                     const double delta[3], double lambda, double threshold,
                     float* grid, int32_t ngpus,
                     const char* path, const void* header, int64_t header_bytes,
                     const void* trailer, int64_t trailer_bytes);

/* Page-locked result arrays.  ceg_grid_vdw / ceg_grid_coulomb / ceg_grids_multi into an ordinary host array go through a pinned
 * ring and a second pass by host threads; when `grid` was allocated HERE, every chunk is copied D2H straight to its place and the
 * call is bounded by the PCIe transfer alone (256^3: 10.3 instead of 13.4 ms for a VdW grid).  The memory is page-locked host memory,
 * [8*(dims[0]+1)*(dims[1]+1)*(dims[2]+1)] floats in the layout above, owned by the library: hand it back with ceg_host_grid_free
 * (it returns to the per-process cache, so the page-locking is paid once; ceg_release_cached_buffers unpins idle arrays).  In Julia:
 * `unsafe_wrap(Array, ptr, (dz+1, dy+1, dx+1, 8))`.  NULL on failure (ceg_last_error). */
CEG_API float* ceg_host_grid_alloc(const int32_t dims[3]);
CEG_API int    ceg_host_grid_free(float* grid);

/* The one-shot entry points keep, per process, one idle device output slab per GPU and one pinned
 * staging ring (page-locking / hipMalloc of 0.5 GB cost as much as the build itself).  This frees
 * whatever is idle; safe to call at any time, never required. */
CEG_API int ceg_release_cached_buffers(void);

/* The same builds with the assembled grid left in DEVICE memory: d_grid [8*(dims[0]+1)*(dims[1]+1)*(dims[2]+1)] floats on
 * `target_device`, layout as above -- for callers that feed the grid consumers (ceg_scale_grid_device + ceg_interp_create with
 * grid_on_device = 1, then ceg_mc_create) without a trip through host memory.  The x-slabs are spread over `ngpus` devices starting
 * at the target; the target's slab is built in place, the others travel chunk by chunk with hipMemcpyPeerAsync (xGMI inside a
 * node) while later chunks are still being computed: the single-process counterpart of the all-gather that bench.py / the
 * torch.distributed ranks do with RCCL.  Synchronous: d_grid is complete on return. */
CEG_API int ceg_grid_vdw_device(const double* pos, const int64_t* atomkind, int64_t natoms,
                                const double mat[9], const double invmat[9], int32_t ortho, double safemin2, double cutoff2,
                                const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds,
                                const int32_t dims[3], const double size[3], const double shift[3], const double delta[3],
                                double lambda, double threshold, float* d_grid, int32_t target_device, int32_t ngpus);
CEG_API int ceg_grid_coulomb_device(const double* pos, const double* charge, int64_t natoms,
                                    const double mat[9], const double invmat[9], int32_t ortho, double safemin2, double cutoff2,
                                    double alpha,
                                    const int32_t dims[3], const double size[3], const double shift[3], const double delta[3],
                                    double lambda, double threshold, float* d_grid, int32_t target_device, int32_t ngpus);

/* ---- resident-plan API (device buffers, caller-owned stream) ---------------- */
/*
 * A plan is one ProbeSystem + one GridCoordinatesSetup made resident on one
 * device: atom table, rule table, lattice-image list and spatial bins.  It is
 * what a multi-process driver (one rank per GPU) uses: each rank builds its
 * own x-slab [i_begin, i_end) into device memory on its own stream and the
 * slabs are then exchanged by the caller (RCCL all-gather).
 *
 * rules/rule_offset/nkinds/atomkind may be NULL/0 for a Coulomb-only plan;
 * charge may be NULL for a VdW-only plan.
 */
typedef struct ceg_plan ceg_plan_t;

CEG_API int ceg_plan_create(ceg_plan_t** plan, int32_t device,
                    const double* pos, const int64_t* atomkind, const double* charge,
                    int64_t natoms,
                    const double mat[9], const double invmat[9],
                    int32_t ortho, double safemin2, double cutoff2,
                    const ceg_rule_t* rules, const int32_t* rule_offset, int32_t nkinds,
                    double alpha,
                    const int32_t dims[3], const double size[3], const double shift[3],
                    const double delta[3]);
CEG_API int ceg_plan_destroy(ceg_plan_t* plan);
/* The lattice-image list + bins of a plan (the ~1 ms host part of plan creation for a 10 k-atom framework) are kept on the device
 * and shared between plans that need exactly the same list -- same framework, cell, cutoff, grid box, per-atom kind flags and
 * charges --, which is what the K + 1 one-shot calls of one setup_RASPA and every later call on the same framework are; the
 * last CEG_HIP_IMAGE_CACHE (default 6, 0 = off) lists are kept, ceg_release_cached_buffers drops them.  Counters since load: */
CEG_API int ceg_image_cache_stats(int64_t* hits, int64_t* misses, int64_t* entries);

/* 1 if the culled algorithm is valid for this plan (every perpendicular width of
 * `mat` is >= 2*cutoff, which ProbeSystem guarantees, src/probes.jl:24), else 0. */
CEG_API int ceg_plan_can_cull(const ceg_plan_t* plan);

/* number of lattice images kept by the culled algorithm (0 before first use) */
CEG_API int64_t ceg_plan_num_images(const ceg_plan_t* plan);
/* The plan's lattice-image list copied to the host (diagnostics / tests: the list is built on the device since round 4 and must be
 * byte-identical to the host build, CEG_HIP_IMAGES_ON_HOST=1): xyzq [4 n] (position + charge), kind [n] (kind | 1 << 25 when the kind
 * has a VdW rule; -1 without rules), atom [n] (index of the framework atom), bin_start [nb0 nb1 nb2 + 1]; any output may be NULL. */
CEG_API int ceg_plan_copy_images(const ceg_plan_t* plan, double* xyzq, int32_t* kind, int32_t* atom, int32_t* bin_start, int32_t nb[3]);

/*
 * Build x-planes i in [i_begin, i_end) (0 <= i_begin <= i_end <= dims[0]+1).
 * Element (k,j,i,c) is written at
 *     d_out[c*channel_stride + ((i - i_origin)*(dims[1]+1) + j)*(dims[2]+1) + k]
 * so the same call serves a full grid buffer (i_origin = 0, channel_stride =
 * full grid points) or a compact slab (i_origin = i_begin, channel_stride =
 * slab points).  d_out is DEVICE memory on the plan's device; `stream` is a
 * hipStream_t (NULL = default stream).  The call is asynchronous.
 */
CEG_API int ceg_plan_build_vdw(ceg_plan_t* plan, double lambda, double threshold,
                       int32_t i_begin, int32_t i_end,
                       float* d_out, int64_t channel_stride, int32_t i_origin,
                       int32_t algo, void* stream);
CEG_API int ceg_plan_build_coulomb(ceg_plan_t* plan, double lambda, double threshold,
                           int32_t i_begin, int32_t i_end,
                           float* d_out, int64_t channel_stride, int32_t i_origin,
                           int32_t algo, void* stream);
/* VdW and Coulomb grids of the same slab in one pass over the atoms (shared
 * geometry work).  Same semantics as the two calls above. */
CEG_API int ceg_plan_build_fused(ceg_plan_t* plan,
                         double lambda_vdw, double threshold_vdw,
                         double lambda_coulomb, double threshold_coulomb,
                         int32_t i_begin, int32_t i_end,
                         float* d_out_vdw, float* d_out_coulomb,
                         int64_t channel_stride, int32_t i_origin,
                         int32_t algo, void* stream);

/*
 * Multi-probe plans: ALL the grids of one setup from one pass.  setup_RASPA (src/raspa.jl:497-520) asks for one
 * create_grid_vdw per distinct guest atom plus one create_grid_coulomb, all on the same framework, cutoff and grid geometry;
 * here the K probes' rule tables go into ONE plan (one lattice-image list, one set of bins and function tables) and
 * ceg_plan_build_multi computes the requested grids together: per candidate image one staging, one distance, one 1/r^2, K
 * Lennard-Jones evaluations with K accumulator sets, one real-space Ewald evaluation.
 *
 *   nprobes       1 .. 4 (CEG_MAX_PROBES)
 *   rules[q], rule_offset[q]   the flattened column ff.interactions[:, probe_q] as for ceg_plan_create (same nkinds for all).
 *                 Every probe may be of any rule class ceg_plan_create takes (round 4: Na + the C and O of CO2 are one plan).  The
 *                 probes that are Lennard-Jones-only against the framework kinds that are present (at most one LJ rule per kind;
 *                 NoInteraction / CoulombEwaldDirect count as none) share accumulating loops -- two of them with the Coulomb grid,
 *                 up to four in a VdW launch --; a probe of another class (a Buckingham / hard-sphere cation) is launched alone with
 *                 the kernel of its class, or fused with the Coulomb grid when no Lennard-Jones pair takes that place: all from the
 *                 plan's one image list, bins and function tables.  The exact-path radius of the plan is the largest any probe
 *                 asks for (hard spheres); CEG_ERR_UNSUPPORTED only if that reaches the cutoff.  charge may be NULL (VdW grids only).
 *   d_out_vdw     [nprobes] device pointers, NULL entries are skipped; d_out_coulomb may be NULL.  Layout, channel_stride,
 *                 i_begin / i_end / i_origin, lambda / threshold and the asynchronous stream semantics as ceg_plan_build_*.
 *
 * Every grid is bit-identical to the one the same call produces when it is asked for that grid alone (identical per-pair
 * arithmetic whatever the grouping into launches; candidates that contribute exact zeros may be staged in one launch and
 * dropped in another, which leaves the FP64 sums unchanged); against a single-probe plan of ceg_plan_create the
 * values agree to the last bits of the FP64 sums (a VdW-only single-probe plan lists fewer images, which reorders the sums).
 * The ordinary ceg_plan_build_vdw / _coulomb / _fused calls work on a multi-probe plan too and use probe 0.
 */
#define CEG_MAX_PROBES 4
CEG_API int ceg_plan_create_multi(ceg_plan_t** plan, int32_t device,
                          const double* pos, const int64_t* atomkind, const double* charge, int64_t natoms,
                          const double mat[9], const double invmat[9],
                          int32_t ortho, double safemin2, double cutoff2,
                          int32_t nprobes, const ceg_rule_t* const* rules, const int32_t* const* rule_offset, int32_t nkinds,
                          double alpha,
                          const int32_t dims[3], const double size[3], const double shift[3], const double delta[3]);
CEG_API int ceg_plan_num_probes(const ceg_plan_t* plan);    /* 0 for an ordinary plan */
/* One-shot form (what the Julia binding calls once per setup_RASPA): host arrays out, grids_vdw [nprobes] (NULL entries skipped),
 * grid_coulomb may be NULL; the x-slabs are spread over `ngpus` devices and every device pipelines compute / D2H / host copy as
 * ceg_grid_vdw does.  Probes of any rule class, as for ceg_plan_create_multi. */
CEG_API int ceg_grids_multi(const double* pos, const int64_t* atomkind, const double* charge, int64_t natoms,
                    const double mat[9], const double invmat[9], int32_t ortho, double safemin2, double cutoff2,
                    int32_t nprobes, const ceg_rule_t* const* rules, const int32_t* const* rule_offset, int32_t nkinds, double alpha,
                    const int32_t dims[3], const double size[3], const double shift[3], const double delta[3],
                    double lambda_vdw, double threshold_vdw, double lambda_coulomb, double threshold_coulomb,
                    float* const* grids_vdw, float* grid_coulomb, int32_t ngpus);
CEG_API int ceg_plan_build_multi(ceg_plan_t* plan,
                         double lambda_vdw, double threshold_vdw,
                         double lambda_coulomb, double threshold_coulomb,
                         int32_t i_begin, int32_t i_end,
                         float* const* d_out_vdw, float* d_out_coulomb,
                         int64_t channel_stride, int32_t i_origin, void* stream);

/*
 * Raw FP64 results of compute_derivatives_vdw / compute_derivatives_ewald
 * (src/probes.jl:71-117) at arbitrary cartesian points, before
 * _set_gridpoint!: out[8*p + 0..7] = value, d1x, d1y, d1z, d2xy, d2xz, d2yz, d3.
 * points/out are HOST memory; synchronous.  which: 0 = vdw, 1 = coulomb.
 */
CEG_API int ceg_plan_eval_points(ceg_plan_t* plan, int32_t which, int32_t algo,
                         const double* points, int64_t npoints, double* out);

/* ---- grid consumer: batched tricubic interpolation --------------- */
/*
 * interpolate_grid (src/grids.jl:212-273) for many points at once on a device-resident
 * EnergyGrid: offsetpoint/wrap_atom (src/coordinates.jl:58-66), 8 corners x 8 channels gather
 * (grids.jl:227-244), the VdW "any corner value > 5e6 -> 1e100 K" rule (:245-248) and the
 * tricubic polynomial (:252-258, evaluated as the equivalent tensor product of cubic Hermite
 * bases instead of the 64x64 COEFF product).  This is what framework_interactions
 * (src/montecarlo.jl:490-504) and energy_grid (src/grids.jl:394-419) call per atom.
 *
 *  grid        [8*(dims[0]+1)*(dims[1]+1)*(dims[2]+1)] float, layout as above, ALREADY in K
 *              (i.e. after parse_grid's `grid .*= GRID_TO_KELVIN`, grids.jl:78);
 *              host memory if grid_on_device == 0, else a device pointer; either way the handle
 *              keeps its own node-major copy ([x][y][z][8]), the input is not referenced later
 *  mat,invmat  UNIT-cell matrix of csetup.cell (not the supercell), column-major
 *  is_vdw      1 for a VdW grid (ewald_precision == Inf): enables the 5e6 rule
 */
typedef struct ceg_interp ceg_interp_t;

/* What parse_grid (src/grids.jl:61-94) reads from a .grid file besides the payload. */
typedef struct ceg_grid_header {
    double  spacing;
    int32_t dims[3];
    int32_t has_mat;            /* 1 if the file ends with the 9 x f64 cell matrix (grids.jl:154,182), then in `mat` */
    double  size[3], shift[3], delta[3], unitcell[3];
    int32_t num_unitcell[3];
    int32_t _pad;
    double  ewald_precision;    /* Inf for a VdW grid (grids.jl:92) */
    double  mat[9];             /* column-major */
} ceg_grid_header_t;

CEG_API int ceg_interp_create(ceg_interp_t** handle, int32_t device,
                              const float* grid, int32_t grid_on_device,
                              const int32_t dims[3], const double size[3], const double shift[3],
                              const double mat[9], const double invmat[9], int32_t is_vdw);
/* A cached grid straight from its file ("Retrieved ... grid", src/raspa.jl:426-438 -> parse_grid, src/grids.jl:61-94): header
 * parsed, payload streamed file -> pinned ring -> device, multiplied by `scale` (GRID_TO_KELVIN, grids.jl:78: each product formed
 * in Float64 and rounded to Float32) on the device, node-major copy made -- no host array, no second upload.
 *  iscoulomb   the file carries the Ewald precision after the header (136 bytes instead of 128); VdW grids get the 5e6 rule
 *  mat,invmat  both NULL: the cell matrix stored at the end of the file is used (its inverse is formed here); else the unit-cell
 *              matrix and its inverse as for ceg_interp_create (parse_grid's `mat` argument)
 *  header_out  may be NULL */
CEG_API int ceg_interp_create_from_file(ceg_interp_t** handle, int32_t device, const char* path, int32_t iscoulomb, double scale,
                                const double* mat, const double* invmat, ceg_grid_header_t* header_out);
/* EnergyGrid.higherorder (grids.jl:21-29): 1 (default) the tricubic branch; 0 the "no derivatives" branch of interpolate_grid
 * (:259-269) -- channel 1 at the 8 corners, trilinear weights, no blocking rule, with the reference's index order for that branch
 * (it addresses the [z, y, x, channel] array as [x, y, z, 1]); an index beyond its axis (a BoundsError in Julia) gives NaN. */
CEG_API int ceg_interp_set_higherorder(ceg_interp_t* handle, int32_t higherorder);
CEG_API int ceg_interp_destroy(ceg_interp_t* handle);
/* points [3*n] cartesian A, out [n] K; host memory, synchronous */
CEG_API int ceg_interp_points(ceg_interp_t* handle, const double* points, int64_t npoints, double* out);
/* device memory on the handle's device, asynchronous on `stream` */
CEG_API int ceg_interp_points_device(ceg_interp_t* handle, const double* d_points, int64_t npoints,
                                     double* d_out, void* stream);
/* in-place `grid .*= scale` in Float32 on device memory (parse_grid, grids.jl:78), so a grid
 * that was just built by ceg_plan_build_* can be interpolated without leaving the GPU */
CEG_API int ceg_scale_grid_device(float* d_grid, int64_t nfloats, double scale, int32_t device, void* stream);

/* ---- grid consumer: batched reciprocal-space Ewald energy --------- */
/*
 * The `coulomb_reciprocal` term of energy_point (src/grids.jl:319-325): compute_ewald(ctx)
 * (src/ewald.jl:555-577) for a context holding ONE rigid molecule, evaluated for many placements of
 * that molecule at once:
 *   E = 2*(sum_k kf_k Re(conj(S_f(k)) S_a(k)) + energy_net_charges) + sum_k kf_k |S_a(k)|^2
 *       + static_contribution,      S_a(k) = sum_atoms q exp(2 pi i k.f),  f = invmat * position
 * With ceg_interp_* this completes energy_point / energy_grid on the device.
 *
 *  kvec_ijk  [3*nk] integer k-vectors in the order of kspace.kindices (src/ewald.jl:213-236)
 *  kfactors  [nk]   (src/ewald.jl:247-259);  sf_re, sf_im [nk] = StoreRigidChargeFramework (:267-271)
 *  ks        (kx, ky, kz);  invmat: inverse of the SUPERCELL matrix (eframework.invmat), column-major
 */
typedef struct ceg_recip ceg_recip_t;

CEG_API int ceg_recip_create(ceg_recip_t** handle, int32_t device, const int32_t* kvec_ijk,
                             const double* kfactors, const double* sf_re, const double* sf_im, int64_t nk,
                             const int32_t ks[3], const double invmat[9]);
CEG_API int ceg_recip_destroy(ceg_recip_t* handle);
/* Host side only (works without a device): the row / segment layout ceg_recip_create gives these k-vectors.  The kernel walks
 * them as rows (j, k) x i = i0..i1 -- the structure of the reference's kspace.kindices (src/ewald.jl:213-236) --, cut into segments
 * dealt to the 64 lanes in `nrounds` rounds; round r runs to its longest segment, `nslots` = the sum of those lengths.
 *  slot_of [nk]            (may be NULL) slot * 64 + lane of every k-vector
 *  desc    [nrounds * 64]  (may be NULL; size it from a first call) i0 | (j + ky) << 9 | (k + kz) << 18 | round length << 27 */
CEG_API int ceg_recip_layout(const int32_t* kvec_ijk, int64_t nk, const int32_t ks[3], int32_t* nrounds, int32_t* nslots,
                             int64_t* slot_of, int32_t* desc);
/* positions [n][natoms][3] A, charges [natoms] e, out [n] K -- host memory, synchronous.
 * energy_net_charges / static_contribution: the two EwaldContext constants (src/ewald.jl:497-544). */
CEG_API int ceg_recip_energy(ceg_recip_t* handle, const double* positions, const double* charges,
                             int32_t natoms, int64_t n, double energy_net_charges,
                             double static_contribution, double* out);
/* positions / out in device memory (charges on the host), asynchronous on `stream` */
CEG_API int ceg_recip_energy_device(ceg_recip_t* handle, const double* d_positions, const double* charges,
                                    int32_t natoms, int64_t n, double energy_net_charges,
                                    double static_contribution, double* d_out, void* stream);

/* replace the structure factor the placements are summed against (same nk).  With
 * sf = framework + all other guests ("rest" of single_contribution_ewald, src/ewald.jl:718-737) and
 * energy_net_charges = static_contribution = 0, ceg_recip_energy* returns single_contribution_ewald
 * of the moved molecule for every trial placement. */
CEG_API int ceg_recip_set_structure_factor(ceg_recip_t* handle, const double* sf_re, const double* sf_im);

/* ---- guest-guest pair energies for trial placements ---------------- */
/*
 * single_contribution_vdw (src/energy.jl:397-427, the exhaustive variant :407-427) of a rigid
 * molecule against all other guest atoms of the system, for many trial placements at once:
 *   E = sum_{atom k2 of the molecule} sum_{guest atom l1 not of the excluded molecule, d2 < cutoff2}
 *           ff[kind(l1), kind(k2)](d2)
 * d2 by unsafe_periodic_distance2! (src/utils.jl:294-302: wrap to the nearest image of the MC cell, no
 * image search); rule energies as src/interactions.jl:367-406 (sum rules: :589-595), which includes the
 * CoulombEwaldDirect pair term q_i q_j erfc(alpha r)/r that carries the real-space guest-guest Ewald sum.
 *
 *  mat, invmat   MC cell (= supercell) matrix and inverse, column-major
 *  rules, rule_offset[nkinds*nkinds + 1]   rule run of the pair (a, b), 0-based kinds, at index
 *                a*nkinds + b (the table is symmetric); kinds rejected as for the grids -> CEG_ERR_RULE
 *                only for UndefinedInteraction (every other kind has an energy form)
 *  coulombic     COULOMBIC_CONVERSION_FACTOR in K A / e^2 (src/constants.jl), an argument like lambda
 */
typedef struct ceg_pairs ceg_pairs_t;

CEG_API int ceg_pairs_create(ceg_pairs_t** handle, int32_t device, const double mat[9], const double invmat[9],
                             double cutoff2, const ceg_rule_t* rules, const int32_t* rule_offset,
                             int32_t nkinds, double coulombic);
CEG_API int ceg_pairs_destroy(ceg_pairs_t* handle);
/* the guest atoms currently in the system: positions [3*natoms] A, kinds [natoms] 0-based ff index,
 * molecule [natoms] id of the molecule each atom belongs to (any non-negative labelling) */
CEG_API int ceg_pairs_set_atoms(ceg_pairs_t* handle, const double* positions, const int32_t* kinds,
                                const int32_t* molecule, int64_t natoms);
/* trial [n][m][3] A, trial_kinds [m]; atoms with molecule id == exclude_molecule are skipped (-1: none);
 * out [n] K.  Host memory, synchronous. */
CEG_API int ceg_pairs_energy(ceg_pairs_t* handle, const double* trial, const int32_t* trial_kinds, int32_t m,
                             int64_t n, int32_t exclude_molecule, double* out);
/* trial / out in device memory, asynchronous on `stream` */
CEG_API int ceg_pairs_energy_device(ceg_pairs_t* handle, const double* d_trial, const int32_t* trial_kinds,
                                    int32_t m, int64_t n, int32_t exclude_molecule, double* d_out, void* stream);
/* 1 when the atoms are kept sorted by neighbour cell (MC cells much larger than the cutoff sphere: the reference's
 * CellListMap branch, energy.jl:399-404; CEG_HIP_MC_CELLS=1|0 forces the choice at create time), with the bins per
 * fractional axis; 0 for the exhaustive loop.  Same sums either way. */
CEG_API int ceg_pairs_neighbour_cells(ceg_pairs_t* handle, int32_t nb[3]);

/* ---- device-resident Monte-Carlo energy state (BASELINE config 5: f1 + f2 + f3 in one launch) ---- */
/*
 * movement_energy (src/montecarlo.jl:563-579) of one rigid molecule of a MonteCarloSetup for a batch of trial
 * placements, from state that lives on the device: guest atoms, pair table, k-space tables, the framework
 * structure factor, the per-molecule structure factors sums[:, ij+1] and their total sums[:, 1] of the reference's
 * IncrementalEwaldContext (src/ewald.jl:584-652).  One kernel launch evaluates, per placement,
 *   framework_interactions    (montecarlo.jl:490-504)  sum_atoms interpolate_grid(vdw grid of the atom's kind) and
 *                                                      sum_atoms charge * interpolate_grid(coulomb grid) (1e100 kept)
 *   single_contribution_vdw   (energy.jl:407-427)      against every guest atom of the OTHER molecules
 *   single_contribution_ewald (ewald.jl:704-738)       2 sum kf Re(conj(rest) S) + sum kf |S|^2,
 *                                                      rest = framework + sums[:,1] - sums[:,ij+1]
 * and ceg_mc_accept applies update_mc! / update_ewald_context! (montecarlo.jl:615-628, ewald.jl:757-773) on the
 * device: no host-built structure factor is uploaded between moves.  Molecules are rigid, <= 16 atoms.
 *
 *  vdw_grids    [nkinds] interpolation handles by 0-based force-field index, NULL where the kind has no grid / a zero grid;
 *               coulomb_grid NULL when the framework carries no charges.  The handles must outlive this object.
 *  kind_charge  [nkinds] e;  mat, invmat: MC cell (= supercell), column-major;  rules / rule_offset / coulombic as ceg_pairs_create
 *  kvec_ijk, kfactors, sf_re, sf_im, nk, ks, ewald_invmat as ceg_recip_create (nk = 0: no Ewald summation)
 *
 * Threading: a handle is NOT thread-safe -- one Markov chain, one caller at a time (the reference's update_mc! is not either);
 * different handles may be driven from different threads.
 * Errors: accept / insert / remove keep a host mirror (molecule table, free atom slots, neighbour-cell lists) in step with the
 * device state.  If one of them fails after it has started to change either side (kernel launch failure, allocation failure
 * while growing the arrays or rebuilding the cells) the handle is marked inconsistent: every later call except
 * ceg_mc_set_guests and ceg_mc_destroy returns CEG_ERR_HIP, and ceg_mc_set_guests rebuilds both sides from scratch.
 */
typedef struct ceg_mc ceg_mc_t;

CEG_API int ceg_mc_create(ceg_mc_t** handle, int32_t device, ceg_interp_t* const* vdw_grids, ceg_interp_t* coulomb_grid,
                          const double* kind_charge, int32_t nkinds, const double mat[9], const double invmat[9],
                          double cutoff2, const ceg_rule_t* rules, const int32_t* rule_offset, double coulombic,
                          const int32_t* kvec_ijk, const double* kfactors, const double* sf_re, const double* sf_im,
                          int64_t nk, const int32_t ks[3], const double ewald_invmat[9]);
CEG_API int ceg_mc_destroy(ceg_mc_t* handle);
/* the guests currently in the system, molecule j = atoms [mol_first[j], mol_first[j+1]) (mol_first[0] = 0):
 * positions [3*natoms] A, kinds [natoms] 0-based ff index.  Computes every sums[:, ij+1] and sums[:, 1] on the device
 * (compute_ewald(::IncrementalEwaldContext), ewald.jl:630-652).  Synchronous. */
CEG_API int ceg_mc_set_guests(ceg_mc_t* handle, const double* positions, const int32_t* kinds,
                              const int32_t* mol_first, int32_t nmol);
/* trial [n][m][3] A placements of molecule `molecule` (m = its atom count); out [(n+1)][4] K:
 * row 0 = movement_energy where the molecule is now, row 1+t = at trial t; columns framework vdw, framework direct,
 * guest-guest, reciprocal.  Host memory; one launch + one stream synchronisation (small batches travel through pinned,
 * device-mapped buffers). */
CEG_API int ceg_mc_trial(ceg_mc_t* handle, int32_t molecule, const double* trial, int64_t n, double* out);
/* The same with the trial placements and the result rows in DEVICE memory (like ceg_recip_energy_device / ceg_pairs_energy_device):
 * for callers that generate trials on the device or evaluate large batches repeatedly -- a 65 536-placement call through the host
 * entry point spends a third of its time moving 4.7 MB in and 2.1 MB out of pageable memory.  Enqueued on `stream` (a hipStream_t,
 * NULL = the null stream), ordered behind everything this handle has enqueued so far; later accept / insert / remove calls are
 * ordered behind it; no synchronisation: the rows are valid when `stream` has reached this point.  Always the wave-per-placement
 * kernels.  ceg_mc_trial_insert_device: the GCMC counterpart (see ceg_mc_trial_insert). */
CEG_API int ceg_mc_trial_device(ceg_mc_t* handle, int32_t molecule, const double* d_trial, int64_t n, double* d_out, void* stream);
CEG_API int ceg_mc_trial_insert_device(ceg_mc_t* handle, const int32_t* kinds, int32_t m, const double* d_trial, int64_t n, double* d_out, void* stream);
/* the molecule now sits at positions [m][3]: update_mc! on the device.  Asynchronous; later calls on this handle are
 * ordered behind it. */
CEG_API int ceg_mc_accept(ceg_mc_t* handle, int32_t molecule, const double* positions);
/* GCMC swaps (gcmc.jl / mcmoves.jl evaluate them with the same movement_energy):
 * trial_insert: movement_energy of a molecule that is NOT in the system (kinds [m] 0-based ff indices) at each of n trial
 *   placements trial [n][m][3]: nothing excluded from the pair sum, rest = framework + sums[:, 1]
 *   (single_contribution_ewald with ij < 0, ewald.jl:704-728); out [n][4], no current-position row.  Synchronous.
 * insert: add_one_system! (ewald.jl:775-792): the molecule becomes index nmol (returned in *molecule_out); asynchronous.
 * remove: remove_one_system! (ewald.jl:794-810, :404-413): sums[:, 1] -= sums[:, ij+1]; the LAST molecule takes index
 *   `molecule` (*moved_out = its old index, = `molecule` when it was the last one); asynchronous. */
CEG_API int ceg_mc_trial_insert(ceg_mc_t* handle, const int32_t* kinds, int32_t m, const double* trial, int64_t n, double* out);
CEG_API int ceg_mc_insert(ceg_mc_t* handle, const int32_t* kinds, int32_t m, const double* positions, int32_t* molecule_out);
CEG_API int ceg_mc_remove(ceg_mc_t* handle, int32_t molecule, int32_t* moved_out);
/* read back (any pointer may be NULL): positions [3*natoms] in molecule order, total guest structure factor sums[:, 1] as re / im [nk] */
CEG_API int ceg_mc_get_state(ceg_mc_t* handle, double* positions, double* sf_total_re, double* sf_total_im);
/* The guest-guest sum runs over neighbour cells (the reference's CellListMap branch, energy.jl:341-349,399-404) when the MC
 * cell is large enough for that to pay: fractional bins of the cell kept current on the device by accept / insert / remove.
 * Returns 1 with the bin counts and the per-cell capacity when the cells are in use, 0 (and zeros) for the exhaustive loop;
 * the energies are the same sums either way.  Environment: CEG_HIP_MC_CELLS=1|0 forces the choice, CEG_HIP_MC_BIN = bin width, A. */
CEG_API int ceg_mc_neighbour_cells(ceg_mc_t* handle, int32_t nb[3], int32_t* capacity);

/* ---- blocking masks on the grid lattice ----------------------------- */
/*
 * BlockFile(g::EnergyGrid), src/grids.jl:188-204: a lattice cell (i, j, k), i < dims[0] etc., whose
 * value g.grid[k,j,i,1] exceeds `threshold` (5e6 K there) blocks its 8 corners.
 *  value   [(dims[0]+1)*(dims[1]+1)*(dims[2]+1)] float = channel 0 of the grid (host or device memory)
 *  block   [same count] uint8 out, host memory, [x][y][z] with z fastest, 1 = blocked
 */
CEG_API int ceg_block_from_grid(int32_t device, const float* value, int32_t value_on_device,
                                const int32_t dims[3], double threshold, uint8_t* block);
/*
 * The scan of parse_blockfile, src/coordinates.jl:139-152: lattice point (i, j, k) (0-based here) at
 * inverse_offsetpoint = (i, j, k) .* delta .+ shift (src/coordinates.jl:68-70) is blocked iff its
 * minimum-image distance (periodic_distance2_fromcartesian!, src/utils.jl:210-246, UNIT cell `mat`)
 * to the centre of one of the spheres is < radius.  centers [3*nspheres] are the snapped centres the
 * reference computes on the host (:128-131); radius2 [nspheres] = radius^2.
 */
CEG_API int ceg_block_spheres(int32_t device, const int32_t dims[3], const double delta[3], const double shift[3],
                              const double mat[9], const double invmat[9], int32_t ortho, double safemin2,
                              const double* centers, const double* radius2, int32_t nspheres, uint8_t* block);

#ifdef __cplusplus
}
#endif
#endif /* CEG_HIP_H */
