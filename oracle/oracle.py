"""ctypes front-end of the CPU oracle ``libceg_oracle.so`` (see ``ceg_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py -- never by the product package.  Parity status:
pinned to the reference through the literals of test/runtests.jl
(tests/test_reference_pins.py; reference tolerance 1e-3, reproduced here to 1e-8 .. 1e-16);
the HIP kernels are compared with this restatement at 1e-6.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "libceg_oracle.so"

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    newest = max((HERE / f).stat().st_mtime for f in ("ceg_oracle.c", "ceg_oracle_mc.c", "Makefile"))
    if force or not LIB.exists() or LIB.stat().st_mtime < newest:
        subprocess.check_call(["make", "-C", str(HERE), "-B", "libceg_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        l = C.CDLL(str(LIB))
        l.oracle_periodic_distance2_fromcartesian.restype = C.c_double
        l.oracle_periodic_distance2_fromcartesian.argtypes = [_dp, _dp, _dp, C.c_int, C.c_double]
        l.oracle_derivatives_grid.restype = C.c_int
        l.oracle_derivatives_grid.argtypes = [C.c_void_p, C.c_int, C.c_double, _dp]
        l.oracle_derivatives_ewald.restype = None
        l.oracle_derivatives_ewald.argtypes = [C.c_double, C.c_double, C.c_double, _dp]
        l.oracle_set_gridpoint.restype = None
        l.oracle_set_gridpoint.argtypes = [_fp, C.c_int64, C.c_int64, _dp, C.c_double, C.c_double, _dp]
        l.oracle_abc_to_xyz.restype = None
        l.oracle_abc_to_xyz.argtypes = [_i32p, _dp, _dp, C.c_int32, C.c_int32, C.c_int32, _dp]
        l.oracle_grid_vdw.restype = C.c_int
        l.oracle_grid_vdw.argtypes = [_dp, _i64p, C.c_int64, _dp, _dp, C.c_int, C.c_double, C.c_double,
                                      C.c_void_p, _i32p, C.c_int32, _i32p, _dp, _dp, _dp,
                                      C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp, _dp, C.c_int32]
        l.oracle_grid_coulomb.restype = C.c_int
        l.oracle_grid_coulomb.argtypes = [_dp, _dp, C.c_int64, _dp, _dp, C.c_int, C.c_double, C.c_double,
                                          C.c_double, _i32p, _dp, _dp, _dp,
                                          C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp, _dp, C.c_int32]
        l.oracle_points_vdw.restype = C.c_int
        l.oracle_points_vdw.argtypes = [_dp, _i64p, C.c_int64, _dp, _dp, C.c_int, C.c_double, C.c_double,
                                        C.c_void_p, _i32p, _dp, C.c_int64, _dp, C.c_int32]
        l.oracle_points_coulomb.restype = C.c_int
        l.oracle_points_coulomb.argtypes = [_dp, _dp, C.c_int64, _dp, _dp, C.c_int, C.c_double, C.c_double,
                                            C.c_double, _dp, C.c_int64, _dp, C.c_int32]
        l.oracle_interpolate_points_noderiv.restype = None
        l.oracle_interpolate_points_noderiv.argtypes = [_fp, _i32p, _dp, _dp, _dp, _dp, _dp, C.c_int64, _dp, C.c_int32]
        l.oracle_interpolate_points.restype = None
        l.oracle_interpolate_points.argtypes = [_fp, _i32p, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, C.c_int64, _dp, C.c_int32]
        l.oracle_reciprocal_energies.restype = None
        l.oracle_reciprocal_energies.argtypes = [_i32p, C.c_int64, _i32p, _dp, _dp, _dp, C.c_int64, _dp, _dp, _dp, C.c_int32,
                                                 C.c_int64, C.c_double, C.c_double, _dp, C.c_int32]
        l.oracle_single_contribution_vdw.restype = None
        l.oracle_single_contribution_vdw.argtypes = [_dp, _dp, C.c_double, C.c_void_p, _i32p, C.c_int32, C.c_double, _dp, _i32p, _i32p,
                                                     C.c_int64, _dp, _i32p, C.c_int32, C.c_int64, C.c_int32, _dp, C.c_int32]
        l.oracle_block_from_grid.restype = None
        l.oracle_block_from_grid.argtypes = [C.c_void_p, _i32p, C.c_double, C.c_void_p]
        l.oracle_block_spheres.restype = None
        l.oracle_block_spheres.argtypes = [_i32p, _dp, _dp, _dp, _dp, C.c_int32, C.c_double, _dp, _dp, C.c_int32, C.c_void_p, C.c_int32]
        l.oracle_max_threads.restype = C.c_int
        l.oracle_max_threads.argtypes = []
        _lib = l
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _cm(m) -> np.ndarray:
    """3x3 -> column-major 9 vector"""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float64).T.reshape(9))


class _Probe:
    """INPUT ADAPTER: flat view of a ProbeSystem-like object (positions, atomkinds, charges, mat, invmat, the probe's rule
    table) for the C calls.  ortho / safemin come from the oracle's own restatement of prepare_periodic_distance_computations
    (probes.jl:72-73 -> utils.jl:146-155), not from the object."""

    def __init__(self, probe):
        from .hostlogic import prepare_periodic_distance_computations
        self.pos = np.ascontiguousarray(probe.positions, dtype=np.float64)
        self.kinds = np.ascontiguousarray(probe.atomkinds, dtype=np.int64)
        self.q = np.ascontiguousarray(probe.charges, dtype=np.float64)
        self.n = len(self.pos)
        self.mat = _cm(probe.mat)
        self.invmat = _cm(probe.invmat)
        self.ortho, safemin = prepare_periodic_distance_computations(probe.mat)
        self.safemin2 = safemin * safemin
        self.cutoff2 = probe.cutoff2
        if probe.probe:
            self.rules, self.offsets = probe.forcefield.rule_table(probe.probe)
            self.nkinds = probe.forcefield.nkinds


def _geom(cset):
    return (np.ascontiguousarray(cset.dims, dtype=np.int32), np.ascontiguousarray(cset.size, dtype=np.float64),
            np.ascontiguousarray(cset.shift, dtype=np.float64), np.ascontiguousarray(cset.delta, dtype=np.float64))


def grid_vdw(probe, cset, lam, thr, i_begin=0, i_end=None, want_raw=False, nthreads=0, j_begin=0, j_end=None, out=None):
    """Loop nest of create_grid_vdw (grids.jl:144-150) for x-planes [i_begin, i_end).
    Returns (grid float32[8,nx,ny,nz], raw float64[nx,ny,nz,8] or None); planes outside the
    range are left NaN.  ``out``: a caller-owned float32[8,nx,ny,nz] array written in place (not NaN-filled:
    for timing the loop nest without the allocation)."""
    p = _Probe(probe)
    dims, size, shift, delta = _geom(cset)
    nx, ny, nz = (int(d) + 1 for d in dims)
    i_end = nx if i_end is None else i_end
    j_end = ny if j_end is None else j_end
    if out is None:
        grid = np.full((8, nx, ny, nz), np.nan, dtype=np.float32)
    else:
        assert out.shape == (8, nx, ny, nz) and out.dtype == np.float32 and out.flags.c_contiguous
        grid = out
    raw = np.full((nx, ny, nz, 8), np.nan, dtype=np.float64) if want_raw else None
    rc = lib().oracle_grid_vdw(_d(p.pos), p.kinds.ctypes.data_as(_i64p), p.n, _d(p.mat), _d(p.invmat),
                               int(p.ortho), p.safemin2, p.cutoff2, p.rules.ctypes.data,
                               p.offsets.ctypes.data_as(_i32p), p.nkinds,
                               dims.ctypes.data_as(_i32p), _d(size), _d(shift), _d(delta),
                               lam, thr, i_begin, i_end, j_begin, j_end, grid.ctypes.data_as(_fp),
                               _d(raw) if want_raw else None, nthreads)
    if rc:
        raise RuntimeError(f"oracle_grid_vdw: rule kind rejected by derivativesGrid (code {rc})")
    return grid, raw


def grid_coulomb(probe, alpha, cset, lam, thr, i_begin=0, i_end=None, want_raw=False, nthreads=0, j_begin=0, j_end=None, out=None):
    """Loop nest of create_grid_coulomb (grids.jl:171-177).  ``out`` as for grid_vdw."""
    p = _Probe(probe)
    dims, size, shift, delta = _geom(cset)
    nx, ny, nz = (int(d) + 1 for d in dims)
    i_end = nx if i_end is None else i_end
    j_end = ny if j_end is None else j_end
    if out is None:
        grid = np.full((8, nx, ny, nz), np.nan, dtype=np.float32)
    else:
        assert out.shape == (8, nx, ny, nz) and out.dtype == np.float32 and out.flags.c_contiguous
        grid = out
    raw = np.full((nx, ny, nz, 8), np.nan, dtype=np.float64) if want_raw else None
    lib().oracle_grid_coulomb(_d(p.pos), _d(p.q), p.n, _d(p.mat), _d(p.invmat), int(p.ortho), p.safemin2,
                              p.cutoff2, alpha, dims.ctypes.data_as(_i32p), _d(size), _d(shift), _d(delta),
                              lam, thr, i_begin, i_end, j_begin, j_end, grid.ctypes.data_as(_fp),
                              _d(raw) if want_raw else None, nthreads)
    return grid, raw


def points_vdw(probe, points, nthreads=0) -> np.ndarray:
    """compute_derivatives_vdw (probes.jl:71-92) at arbitrary points -> float64[n,8]"""
    p = _Probe(probe)
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    out = np.empty((len(pts), 8), dtype=np.float64)
    rc = lib().oracle_points_vdw(_d(p.pos), p.kinds.ctypes.data_as(_i64p), p.n, _d(p.mat), _d(p.invmat),
                                 int(p.ortho), p.safemin2, p.cutoff2, p.rules.ctypes.data,
                                 p.offsets.ctypes.data_as(_i32p), _d(pts), len(pts), _d(out), nthreads)
    if rc:
        raise RuntimeError(f"oracle_points_vdw: rule kind rejected (code {rc})")
    return out


def points_coulomb(probe, alpha, points, nthreads=0) -> np.ndarray:
    """compute_derivatives_ewald (probes.jl:94-117) at arbitrary points -> float64[n,8]"""
    p = _Probe(probe)
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    out = np.empty((len(pts), 8), dtype=np.float64)
    lib().oracle_points_coulomb(_d(p.pos), _d(p.q), p.n, _d(p.mat), _d(p.invmat), int(p.ortho), p.safemin2,
                                p.cutoff2, alpha, _d(pts), len(pts), _d(out), nthreads)
    return out


def set_gridpoints(raw8: np.ndarray, delta, lam, thr) -> np.ndarray:
    """_set_gridpoint! (grids.jl:118-135) applied to float64[n,8] -> float32[n,8]"""
    raw8 = np.ascontiguousarray(raw8, dtype=np.float64).reshape(-1, 8)
    n = len(raw8)
    out = np.empty((8, n), dtype=np.float32)
    dl = np.ascontiguousarray(delta, dtype=np.float64)
    l = lib()
    for t in range(n):
        l.oracle_set_gridpoint(out.ctypes.data_as(_fp), t, n, _d(dl), lam, thr, _d(raw8[t]))
    return out.T.copy()


def periodic_distance2(d, mat, invmat, ortho, safemin2):
    """periodic_distance2_fromcartesian! (utils.jl:210-246) -> (d2, image vector)"""
    buf = np.array(d, dtype=np.float64)
    d2 = lib().oracle_periodic_distance2_fromcartesian(_d(buf), _d(_cm(mat)), _d(_cm(invmat)), int(ortho), safemin2)
    return d2, buf


def derivatives_grid(rules: np.ndarray, r2: float) -> np.ndarray:
    out = np.empty(4)
    rc = lib().oracle_derivatives_grid(rules.ctypes.data, len(rules), r2, _d(out))
    if rc:
        raise RuntimeError("rule kind rejected by derivativesGrid")
    return out


def derivatives_ewald(alpha: float, charge: float, r2: float) -> np.ndarray:
    out = np.empty(4)
    lib().oracle_derivatives_ewald(alpha, charge, r2, _d(out))
    return out


def interpolate_points(g, points, nthreads=0) -> np.ndarray:
    """interpolate_grid (grids.jl:212-273) of an EnergyGrid (values in K) at many points, literal
    COEFF*X evaluation; the COEFF matrix is the oracle's own (oracle/hostlogic.py, pinned to the reference's literal
    src/constants.jl:24-89 through tests/golden/coeff.json).  ``g``: any object with the EnergyGrid fields (input adapter)."""
    import math
    from .hostlogic import tricubic_coeff
    cs = g.csetup
    grid = np.ascontiguousarray(g.grid, dtype=np.float32)
    dims = np.ascontiguousarray(cs.dims, dtype=np.int32)
    size = np.ascontiguousarray(cs.size, dtype=np.float64)
    shift = np.ascontiguousarray(cs.shift, dtype=np.float64)
    coeff = np.ascontiguousarray(tricubic_coeff(), dtype=np.float64)
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    out = np.empty(len(pts), dtype=np.float64)
    if not g.higherorder:          # the "no derivatives" branch (grids.jl:259-269): channel 1, trilinear, the reference's index order
        lib().oracle_interpolate_points_noderiv(grid.ctypes.data_as(_fp), dims.ctypes.data_as(_i32p), _d(size), _d(shift),
                                                _d(_cm(cs.cell.mat)), _d(_cm(cs.cell.invmat)), _d(pts), len(pts), _d(out), nthreads)
        return out
    lib().oracle_interpolate_points(grid.ctypes.data_as(_fp), dims.ctypes.data_as(_i32p), _d(size), _d(shift),
                                    _d(_cm(cs.cell.mat)), _d(_cm(cs.cell.invmat)),
                                    1 if g.ewald_precision == math.inf else 0, _d(coeff), _d(pts), len(pts), _d(out), nthreads)
    return out


def reciprocal_energies(ef, molecule, positions, nthreads=0) -> np.ndarray:
    """compute_ewald (ewald.jl:555-577) of one rigid molecule placed at positions[n, natoms, 3]: literal
    restatement with the reference's power tables and summation order.  ``ef``: an EwaldFramework-like object (input adapter:
    attribute access only); the two context constants come from the oracle's own restatement (oracle/hostlogic.py)."""
    from .hostlogic import adapt_ewald_framework, ewald_context_constants
    kind = np.ascontiguousarray(np.array(ef.kspace.kindices, dtype=np.int32).reshape(-1, 5))
    ks = np.asarray(ef.kspace.ks, dtype=np.int32)
    kf = np.ascontiguousarray(ef.kfactors, dtype=np.float64)
    re = np.ascontiguousarray(ef.StoreRigidChargeFramework.real, dtype=np.float64)
    im = np.ascontiguousarray(ef.StoreRigidChargeFramework.imag, dtype=np.float64)
    q = np.ascontiguousarray(molecule.atomic_charge, dtype=np.float64)
    pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, len(q), 3)
    enc, static = ewald_context_constants(adapt_ewald_framework(ef),
                                          [(q, np.asarray(molecule.position, dtype=np.float64).reshape(-1, 3), 1)])
    out = np.empty(len(pos), dtype=np.float64)
    lib().oracle_reciprocal_energies(kind.ctypes.data_as(_i32p), len(kind), ks.ctypes.data_as(_i32p), _d(kf), _d(re), _d(im),
                                     len(kf), _d(_cm(ef.invmat)), _d(pos.reshape(-1)), _d(q), len(q), len(pos), enc, static,
                                     _d(out), nthreads)
    return out


def single_contribution_vdw(mc, idx, trial, nthreads=0) -> np.ndarray:
    """single_contribution_vdw_noneighbour (energy.jl:407-427) of molecule ``idx`` = (kind, molecule),
    0-based, of a MonteCarloSetup-like object (input adapter) at trial[n, natoms, 3]."""
    from .hostlogic import COULOMBIC_CONVERSION_FACTOR
    rules, offsets = mc.ff.pair_table()
    pos, kinds, mol = [], [], []
    for m, (i, j, ids, p) in enumerate(mc.molecules()):
        pos.append(p)
        kinds += [k - 1 for k in ids]
        mol += [m] * len(ids)
    pos = np.concatenate(pos) if pos else np.empty((0, 3))
    tk = [k - 1 for k in mc.ffidx[idx[0]]]
    return single_contribution_vdw_raw(mc.mat, mc.invmat, mc.ff.cutoff ** 2, rules, offsets, mc.ff.nkinds, COULOMBIC_CONVERSION_FACTOR,
                                       pos, kinds, mol, trial, tk, mc.flat_index(*idx), nthreads)

def single_contribution_vdw_raw(mat, invmat, cutoff2, rules, offsets, nkinds, coulombic, positions, kinds, molecule, trial,
                                trial_kinds, exclude, nthreads=0) -> np.ndarray:
    """oracle_single_contribution_vdw on an explicit pair table (0-based kinds; the same arrays
    ``ceg_pairs_create`` / ``ceg_pairs_set_atoms`` take): energy.jl:407-427 without a MonteCarloSetup."""
    pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
    kinds = np.ascontiguousarray(kinds, dtype=np.int32)
    mol = np.ascontiguousarray(molecule, dtype=np.int32)
    tk = np.ascontiguousarray(trial_kinds, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    t = np.ascontiguousarray(trial, dtype=np.float64).reshape(-1, len(tk), 3)
    out = np.empty(len(t), dtype=np.float64)
    lib().oracle_single_contribution_vdw(_d(_cm(mat)), _d(_cm(invmat)), float(cutoff2), rules.ctypes.data,
                                         offsets.ctypes.data_as(_i32p), int(nkinds), float(coulombic), _d(pos.reshape(-1)),
                                         kinds.ctypes.data_as(_i32p), mol.ctypes.data_as(_i32p), len(pos), _d(t.reshape(-1)),
                                         tk.ctypes.data_as(_i32p), len(tk), len(t), int(exclude), _d(out), nthreads)
    return out


def block_from_grid(g, threshold=5e6) -> np.ndarray:
    """BlockFile(g::EnergyGrid) (grids.jl:188-204) -> bool[nx, ny, nz]."""
    dims = np.ascontiguousarray(g.csetup.dims, dtype=np.int32)
    value = np.ascontiguousarray(g.grid[0], dtype=np.float32)
    out = np.empty(value.shape, dtype=np.uint8)
    lib().oracle_block_from_grid(value.ctypes.data, dims.ctypes.data_as(_i32p), float(threshold), out.ctypes.data)
    return out.astype(bool)


def block_spheres(csetup, centers, radius2, nthreads=0) -> np.ndarray:
    """The scan of parse_blockfile (coordinates.jl:139-152) -> bool[nx, ny, nz]."""
    from .hostlogic import prepare_periodic_distance_computations
    dims = np.ascontiguousarray(csetup.dims, dtype=np.int32)
    ortho, safemin = prepare_periodic_distance_computations(csetup.cell.mat)
    c = np.ascontiguousarray(centers, dtype=np.float64).reshape(-1, 3)
    r2 = np.ascontiguousarray(radius2, dtype=np.float64)
    out = np.empty(tuple(int(d) + 1 for d in csetup.dims), dtype=np.uint8)
    lib().oracle_block_spheres(dims.ctypes.data_as(_i32p), _d(np.ascontiguousarray(csetup.delta, dtype=np.float64)),
                               _d(np.ascontiguousarray(csetup.shift, dtype=np.float64)), _d(_cm(csetup.cell.mat)),
                               _d(_cm(csetup.cell.invmat)), int(ortho), safemin ** 2, _d(c.reshape(-1)), _d(r2), len(r2),
                               out.ctypes.data, nthreads)
    return out.astype(bool)


def max_threads() -> int:
    return lib().oracle_max_threads()


def usable_cpus() -> int:
    """Host threads this process may really use: the scheduler affinity mask, capped by the cgroup CPU quota
    (v2 ``cpu.max``, v1 ``cpu.cfs_quota_us``) -- omp_get_max_threads() reports the machine's logical CPUs even when
    the container owns a fraction of them."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, math.floor(quota + 1e-9)))
    return max(1, min(n, os.cpu_count() or n))       # (not omp_get_max_threads(): a previous call with nthreads = 1 lowers it)
