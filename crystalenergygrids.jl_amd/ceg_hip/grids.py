"""Energy grids: build, file format, interpolation, ``energy_point`` (host-side
mirror of ``src/grids.jl``).

The two loop nests that fill the grid -- ``create_grid_vdw`` (grids.jl:144-150) and
``create_grid_coulomb`` (grids.jl:171-177) -- are replaced by one call each into
``libceg_hip.so``; everything around them (geometry, ProbeSystem, lambda / threshold,
the byte-exact ``.grid`` writer) follows the reference line by line.  The product
path has no CPU fallback: without the HIP library these functions raise.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from .hostmirror.constants import COULOMBIC_CONVERSION_FACTOR, GRID_TO_KELVIN, tricubic_coeff
from .hostmirror.coordinates import CellMatrix, GridCoordinatesSetup, offsetpoint
from .hostmirror.ewald import EwaldFramework, compute_ewald, initialize_ewald
from .hostmirror.forcefields import ForceField
from .hostmirror.probes import ProbeSystem
from .hostmirror.utils import find_supercell


# ------------------------------------------------------------------ EnergyGrid
@dataclass
class EnergyGrid:
    """grids.jl:10-16.  ``grid`` has numpy shape ``(8, nx, ny, nz)`` C-order, which is the
    same memory as Julia's column-major ``Array{Cfloat,4}(nz, ny, nx, 8)``."""
    csetup: Optional[GridCoordinatesSetup]
    num_unitcell: Tuple[int, int, int]
    ewald_precision: float       # -inf zero grid; +inf VdW grid; nan invalid grid
    higherorder: bool
    grid: np.ndarray

    @classmethod
    def trivial(cls, zero: bool) -> "EnergyGrid":
        """``EnergyGrid(zero::Bool)`` grids.jl:17-19"""
        return cls(None, (0, 0, 0), -math.inf if zero else math.nan, False,
                   np.empty((0, 0, 0, 0), dtype=np.float32))


def _setup_grid_common(framework, spacing: float, cutoff: float):
    """grids.jl:102-106"""
    csetup = GridCoordinatesSetup.from_cell(framework.mat, spacing)
    num_unitcell = find_supercell(framework.mat, cutoff)
    return csetup, num_unitcell


def _create_grid_common(io, csetup: GridCoordinatesSetup, num_unitcell) -> None:
    """grids.jl:108-116 -- 128-byte little-endian unpadded header."""
    io.write(struct.pack("<d", float(csetup.spacing)))
    io.write(struct.pack("<3i", *(int(x) for x in csetup.dims)))
    io.write(struct.pack("<3d", *csetup.size))
    io.write(struct.pack("<3d", *csetup.shift))
    io.write(struct.pack("<3d", *csetup.delta))
    io.write(struct.pack("<3d", *csetup.unitcell))
    io.write(struct.pack("<3i", *(int(x) for x in num_unitcell)))


def vdw_scaling():
    """``λ = inv(GRID_TO_KELVIN)``, threshold ``GRID_TO_KELVIN*1e7`` (grids.jl:141-143,148)"""
    lam_inv = GRID_TO_KELVIN
    return 1.0 / lam_inv, lam_inv * 1e7


def coulomb_scaling():
    """``λ = COULOMBIC_CONVERSION_FACTOR/GRID_TO_KELVIN``, threshold ``inv(λ)*1e7``
    (grids.jl:169-170)"""
    lam = COULOMBIC_CONVERSION_FACTOR / GRID_TO_KELVIN
    return lam, (1.0 / lam) * 1e7


def _grid_args(cset: GridCoordinatesSetup):
    dims = np.ascontiguousarray(cset.dims, dtype=np.int32)
    size = np.ascontiguousarray(cset.size, dtype=np.float64)
    shift = np.ascontiguousarray(cset.shift, dtype=np.float64)
    delta = np.ascontiguousarray(cset.delta, dtype=np.float64)
    return dims, size, shift, delta


def _matT(m: np.ndarray) -> np.ndarray:
    """column-major flattening of a 3x3 (what the C ABI expects)"""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float64).T.reshape(9))


def _file_frame(cset: GridCoordinatesSetup, num_unitcell, ewald_precision: Optional[float]):
    """(header, trailer) bytes of a .grid file: what _create_grid_common (grids.jl:108-116) [+ the Ewald
    precision, :180] writes before the payload and the cell matrix written after it (:154, :182)."""
    import io
    buf = io.BytesIO()
    _create_grid_common(buf, cset, num_unitcell)
    if ewald_precision is not None:
        buf.write(struct.pack("<d", float(ewald_precision)))
    return buf.getvalue(), np.asarray(cset.cell.mat, dtype="<f8").T.tobytes()


def alloc_host_grid(cset: GridCoordinatesSetup) -> np.ndarray:
    """A float32[8, nx, ny, nz] result array in page-locked memory of the library (``ceg_host_grid_alloc``): passed as ``out=`` to
    ``build_vdw_array`` / ``build_coulomb_array`` / ``build_multi_arrays`` it lets every chunk be copied D2H straight to its place (no
    pinned ring, no second pass by host threads).  The memory goes back to the library's cache when the array is collected."""
    import weakref
    lib = _abi.load_library()
    dims = np.ascontiguousarray(cset.dims, dtype=np.int32)
    ptr = lib.ceg_host_grid_alloc(_abi.i32ptr(dims))
    if not ptr:
        raise _abi.CegError(-3, (lib.ceg_last_error() or b"").decode())
    nx, ny, nz = cset.npoints
    arr = np.ctypeslib.as_array(ptr, shape=(8, nx, ny, nz))
    addr = C.cast(ptr, C.c_void_p).value
    weakref.finalize(arr, lambda a=addr: lib.ceg_host_grid_free(C.cast(a, _abi.c_float_p)))
    return arr


def _result_array(cset: GridCoordinatesSetup, out) -> np.ndarray:
    nx, ny, nz = cset.npoints
    if out is None:
        return np.empty((8, nx, ny, nz), dtype=np.float32)
    if out.shape != (8, nx, ny, nz) or out.dtype != np.float32 or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous float32[8, nx, ny, nz] array")
    return out


def build_vdw_array(probe: ProbeSystem, cset: GridCoordinatesSetup, ngpus: int = 1, file=None, num_unitcell=None, out=None) -> np.ndarray:
    """The loop nest of grids.jl:144-150 on the GPU (``ceg_grid_vdw``).  With ``file`` the .grid file is
    written by the library while the grid is being built (``ceg_grid_vdw_file``).  ``out``: result array to fill
    (``alloc_host_grid`` gives a page-locked one: the call is then bounded by the PCIe transfer alone)."""
    lib = _abi.load_library()
    ff = probe.forcefield
    ff.check_vdw_grid(probe.probe, np.unique(probe.atomkinds))
    rules, offsets = ff.rule_table(probe.probe)
    ortho, safemin2 = probe.periodic_setup()
    lam, thr = vdw_scaling()
    dims, size, shift, delta = _grid_args(cset)
    grid = _result_array(cset, out)
    pos = np.ascontiguousarray(probe.positions, dtype=np.float64)
    kinds = np.ascontiguousarray(probe.atomkinds, dtype=np.int64)
    mat, invmat = _matT(probe.mat), _matT(probe.invmat)
    args = (_abi.dptr(pos), _abi.i64ptr(kinds), len(kinds), _abi.dptr(mat), _abi.dptr(invmat),
            int(ortho), safemin2, probe.cutoff2,
            rules.ctypes.data, _abi.i32ptr(offsets), ff.nkinds,
            _abi.i32ptr(dims), _abi.dptr(size), _abi.dptr(shift), _abi.dptr(delta),
            lam, thr, _abi.fptr(grid), ngpus)
    if file is None:
        rc = lib.ceg_grid_vdw(*args)
    else:
        header, trailer = _file_frame(cset, num_unitcell, None)
        rc = lib.ceg_grid_vdw_file(*args, os.fsencode(str(file)), header, len(header), trailer, len(trailer))
    _abi.check(lib, rc)
    return grid


def build_coulomb_array(probe: ProbeSystem, alpha: float, cset: GridCoordinatesSetup, ngpus: int = 1, file=None,
                        num_unitcell=None, ewald_precision: float = 1e-6, out=None) -> np.ndarray:
    """The loop nest of grids.jl:171-177 on the GPU (``ceg_grid_coulomb``; with ``file``: ``ceg_grid_coulomb_file``)."""
    lib = _abi.load_library()
    ortho, safemin2 = probe.periodic_setup()
    lam, thr = coulomb_scaling()
    dims, size, shift, delta = _grid_args(cset)
    grid = _result_array(cset, out)
    pos = np.ascontiguousarray(probe.positions, dtype=np.float64)
    q = np.ascontiguousarray(probe.charges, dtype=np.float64)
    mat, invmat = _matT(probe.mat), _matT(probe.invmat)
    args = (_abi.dptr(pos), _abi.dptr(q), len(q), _abi.dptr(mat), _abi.dptr(invmat),
            int(ortho), safemin2, probe.cutoff2, alpha,
            _abi.i32ptr(dims), _abi.dptr(size), _abi.dptr(shift), _abi.dptr(delta),
            lam, thr, _abi.fptr(grid), ngpus)
    if file is None:
        rc = lib.ceg_grid_coulomb(*args)
    else:
        header, trailer = _file_frame(cset, num_unitcell, ewald_precision)
        rc = lib.ceg_grid_coulomb_file(*args, os.fsencode(str(file)), header, len(header), trailer, len(trailer))
    _abi.check(lib, rc)
    return grid


def build_vdw_device(probe: ProbeSystem, cset: GridCoordinatesSetup, ngpus: int = 1, device: int = 0):
    """``ceg_grid_vdw_device``: the same build with the assembled grid left on the GPU -> ``torch.float32[8, nx, ny, nz]`` on
    ``cuda:device`` (x-slabs built on ``ngpus`` devices of this process, gathered by peer copies)."""
    import torch
    lib = _abi.load_library()
    ff = probe.forcefield
    ff.check_vdw_grid(probe.probe, np.unique(probe.atomkinds))
    rules, offsets = ff.rule_table(probe.probe)
    ortho, safemin2 = probe.periodic_setup()
    lam, thr = vdw_scaling()
    dims, size, shift, delta = _grid_args(cset)
    nx, ny, nz = cset.npoints
    grid = torch.empty((8, nx, ny, nz), dtype=torch.float32, device=f"cuda:{device}")
    pos = np.ascontiguousarray(probe.positions, dtype=np.float64)
    kinds = np.ascontiguousarray(probe.atomkinds, dtype=np.int64)
    mat, invmat = _matT(probe.mat), _matT(probe.invmat)
    torch.cuda.synchronize(device)
    _abi.check(lib, lib.ceg_grid_vdw_device(_abi.dptr(pos), _abi.i64ptr(kinds), len(kinds), _abi.dptr(mat), _abi.dptr(invmat), int(ortho), safemin2,
                                            probe.cutoff2, rules.ctypes.data, _abi.i32ptr(offsets), ff.nkinds, _abi.i32ptr(dims), _abi.dptr(size),
                                            _abi.dptr(shift), _abi.dptr(delta), lam, thr, grid.data_ptr(), device, ngpus))
    return grid


def build_coulomb_device(probe: ProbeSystem, alpha: float, cset: GridCoordinatesSetup, ngpus: int = 1, device: int = 0):
    """``ceg_grid_coulomb_device`` -> ``torch.float32[8, nx, ny, nz]`` on ``cuda:device``."""
    import torch
    lib = _abi.load_library()
    ortho, safemin2 = probe.periodic_setup()
    lam, thr = coulomb_scaling()
    dims, size, shift, delta = _grid_args(cset)
    nx, ny, nz = cset.npoints
    grid = torch.empty((8, nx, ny, nz), dtype=torch.float32, device=f"cuda:{device}")
    pos = np.ascontiguousarray(probe.positions, dtype=np.float64)
    q = np.ascontiguousarray(probe.charges, dtype=np.float64)
    mat, invmat = _matT(probe.mat), _matT(probe.invmat)
    torch.cuda.synchronize(device)
    _abi.check(lib, lib.ceg_grid_coulomb_device(_abi.dptr(pos), _abi.dptr(q), len(q), _abi.dptr(mat), _abi.dptr(invmat), int(ortho), safemin2,
                                                probe.cutoff2, alpha, _abi.i32ptr(dims), _abi.dptr(size), _abi.dptr(shift), _abi.dptr(delta),
                                                lam, thr, grid.data_ptr(), device, ngpus))
    return grid


def write_grid_file(file, cset: GridCoordinatesSetup, num_unitcell, grid: np.ndarray,
                    ewald_precision: Optional[float] = None) -> None:
    """File body of grids.jl:151-155 / :178-183."""
    with open(file, "wb") as f:
        _create_grid_common(f, cset, num_unitcell)
        if ewald_precision is not None:
            f.write(struct.pack("<d", float(ewald_precision)))
        payload = np.ascontiguousarray(grid, dtype="<f4")
        if payload.size:
            f.write(memoryview(payload.reshape(-1)).cast("B"))                       # no intermediate copy of the payload
        f.write(np.asarray(cset.cell.mat, dtype="<f8").T.tobytes())   # column-major 3x3, not part of RASPA grids


def create_grid_vdw(file, framework, forcefield: ForceField, spacing: float, atom: str, ngpus: int = 1) -> np.ndarray:
    """grids.jl:137-157"""
    cset, num_unitcell = _setup_grid_common(framework, spacing, forcefield.cutoff)
    probe_vdw = ProbeSystem.build(framework, forcefield, atom)
    # header, payload (chunk by chunk while the build is running) and trailer are written by the library
    return build_vdw_array(probe_vdw, cset, ngpus, file=file, num_unitcell=num_unitcell)


def create_grid_coulomb(file, framework, forcefield: ForceField, spacing: float,
                        _ewald: Optional[EwaldFramework] = None, ngpus: int = 1) -> np.ndarray:
    """grids.jl:159-185"""
    cset, num_unitcell = _setup_grid_common(framework, spacing, 12.0)
    ewald = _ewald if isinstance(_ewald, EwaldFramework) else initialize_ewald(framework, num_unitcell)
    probe_coulomb = ProbeSystem.build(framework, forcefield)
    return build_coulomb_array(probe_coulomb, ewald.alpha, cset, ngpus, file=file, num_unitcell=num_unitcell,
                               ewald_precision=ewald.precision)


def build_multi_arrays(probes, coulomb_probe: Optional[ProbeSystem], alpha: float, cset: GridCoordinatesSetup, ngpus: int = 1,
                       pinned: bool = False):
    """All the grids of one setup from one pass over one lattice-image list (``ceg_grids_multi``): the VdW grid of every probe
    in ``probes`` (1..4 ProbeSystems of the same framework, of any rule class ``create_grid_vdw`` takes: the library lets the
    Lennard-Jones-only ones share accumulating loops and launches a Buckingham cation alone or fused with the Coulomb grid) and, with
    ``coulomb_probe``, the Coulomb grid.  -> (list of float32[8, nx, ny, nz], float32[8, nx, ny, nz] or None)."""
    lib = _abi.load_library()
    probes = list(probes)
    ref = probes[0]
    for pr in probes[1:] + ([coulomb_probe] if coulomb_probe is not None else []):
        # one framework, one supercell, one cutoff: the probes differ in their rule tables only
        if (pr.positions.shape != ref.positions.shape or not np.array_equal(pr.positions, ref.positions) or not np.array_equal(pr.mat, ref.mat)
                or not np.array_equal(pr.atomkinds, ref.atomkinds) or pr.cutoff2 != ref.cutoff2):
            raise ValueError("the probes of a multi-probe build must share framework positions, atom kinds, cell and cutoff")
    tables = []
    for pr in probes:
        pr.forcefield.check_vdw_grid(pr.probe, np.unique(pr.atomkinds))
        tables.append(pr.forcefield.rule_table(pr.probe))
    K = len(probes)
    rules_pp = (C.c_void_p * K)(*[t[0].ctypes.data for t in tables])
    offs_pp = (C.c_void_p * K)(*[t[1].ctypes.data for t in tables])
    ortho, safemin2 = ref.periodic_setup()
    lv, tv = vdw_scaling()
    lc, tc = coulomb_scaling()
    dims, size, shift, delta = _grid_args(cset)
    nx, ny, nz = cset.npoints
    new = (lambda: alloc_host_grid(cset)) if pinned else (lambda: np.empty((8, nx, ny, nz), dtype=np.float32))
    vgrids = [new() for _ in range(K)]
    cgrid = new() if coulomb_probe is not None else None
    pos = np.ascontiguousarray(ref.positions, dtype=np.float64)
    kinds = np.ascontiguousarray(ref.atomkinds, dtype=np.int64)
    q = np.ascontiguousarray(coulomb_probe.charges, dtype=np.float64) if coulomb_probe is not None else None
    mat, invmat = _matT(ref.mat), _matT(ref.invmat)
    out_pp = (C.c_void_p * K)(*[g.ctypes.data for g in vgrids])
    rc = lib.ceg_grids_multi(_abi.dptr(pos), _abi.i64ptr(kinds), _abi.dptr(q) if q is not None else None, len(kinds),
                             _abi.dptr(mat), _abi.dptr(invmat), int(ortho), safemin2, ref.cutoff2,
                             K, rules_pp, offs_pp, ref.forcefield.nkinds, float(alpha),
                             _abi.i32ptr(dims), _abi.dptr(size), _abi.dptr(shift), _abi.dptr(delta),
                             lv, tv, lc, tc, out_pp, _abi.fptr(cgrid) if cgrid is not None else None, ngpus)
    _abi.check(lib, rc)
    return vgrids, cgrid


def create_grids_multi(vdw_files, coulomb_file, framework, forcefield: ForceField, spacing: float, atoms,
                       _ewald: Optional[EwaldFramework] = None, ngpus: int = 1):
    """``create_grid_vdw`` (grids.jl:137-157) for every atom of ``atoms`` and -- with ``coulomb_file`` -- ``create_grid_coulomb``
    (grids.jl:159-185) of one framework in ONE pass: the files are byte-identical in format to the ones the two functions write
    (same header / payload / trailer writer), the payloads come from ``ceg_grids_multi``.  This is the call pattern of
    setup_RASPA (raspa.jl:497-520) collapsed into one call.  The atoms may be of any rule class (Na + the C and O of CO2: round 4).

    Every file is written to a temporary name in its directory and renamed onto the target only after ALL of them are complete:
    the reference's cache looks no further than ``isfile`` (raspa.jl:426), so a truncated file at the cache path would be
    "retrieved" by every later setup_RASPA.  On any failure the temporaries are removed and no target is touched."""
    atoms = list(atoms)
    assert len(vdw_files) == len(atoms) and 1 <= len(atoms) <= 4
    cset, num_unitcell = _setup_grid_common(framework, spacing, forcefield.cutoff)
    probes = [ProbeSystem.build(framework, forcefield, a) for a in atoms]
    pc, alpha, prec, num_unitcell_c = None, 0.0, None, num_unitcell
    if coulomb_file is not None:
        _cset_c, num_unitcell_c = _setup_grid_common(framework, spacing, 12.0)        # create_grid_coulomb: 12 A whatever the force field says (grids.jl:160)
        ewald = _ewald if isinstance(_ewald, EwaldFramework) else initialize_ewald(framework, num_unitcell_c)
        pc, alpha, prec = ProbeSystem.build(framework, forcefield), ewald.alpha, ewald.precision
    vgrids, cgrid = build_multi_arrays(probes, pc, alpha, cset, ngpus)
    jobs = [(f, num_unitcell, g, None) for f, g in zip(vdw_files, vgrids)]
    if coulomb_file is not None:
        jobs.append((coulomb_file, num_unitcell_c, cgrid, prec))
    tmps = []
    try:
        for n, (f, nuc, g, pr) in enumerate(jobs):
            tmp = f"{os.fspath(f)}.tmp.{os.getpid()}.{n}"
            tmps.append(tmp)
            if os.environ.get("CEG_HIP_INJECT_WRITE_FAILURE") == str(n):             # test hook: the n-th file fails half way
                with open(tmp, "wb") as fh:
                    fh.write(b"truncated")
                raise OSError(f"injected failure while writing {tmp}")
            write_grid_file(tmp, cset, nuc, g, ewald_precision=pr)
        for tmp, (f, *_rest) in zip(tmps, jobs):
            os.replace(tmp, f)
    except BaseException:
        for tmp in tmps:
            try:
                os.unlink(tmp)
            except OSError:
                pass
        raise
    return vgrids, cgrid


def parse_grid(file, iscoulomb: bool, mat=None) -> EnergyGrid:
    """grids.jl:61-94.  The payload is multiplied by GRID_TO_KELVIN in Float32."""
    with open(file, "rb") as io:
        spacing, = struct.unpack("<d", io.read(8))
        dims = np.array(struct.unpack("<3i", io.read(12)), dtype=np.int32)
        size = np.array(struct.unpack("<3d", io.read(24)))
        shift = np.array(struct.unpack("<3d", io.read(24)))
        delta = np.array(struct.unpack("<3d", io.read(24)))
        unitcell = np.array(struct.unpack("<3d", io.read(24)))
        num_unitcell = struct.unpack("<3i", io.read(12))
        ewald_precision = struct.unpack("<d", io.read(8))[0] if iscoulomb else math.inf
        nx, ny, nz = (int(d) + 1 for d in dims)
        n = 8 * nx * ny * nz
        grid = np.fromfile(io, dtype="<f4", count=n)
        if len(grid) != n:
            raise ValueError("truncated grid file")
        grid = grid.reshape(8, nx, ny, nz)
        # grid .*= GRID_TO_KELVIN (grids.jl:78): Float32 array times Float64 scalar, each product formed in
        # Float64 and rounded to Float32 on store -- one buffered pass, no Float64 copy of the array
        np.multiply(grid, GRID_TO_KELVIN, out=grid, dtype=np.float64, casting="same_kind")
        if mat is not None:
            newmat = mat if isinstance(mat, CellMatrix) else CellMatrix.from_mat(mat)
        else:
            tail = io.read(72)
            if len(tail) != 72:
                raise ValueError("Missing `mat` argument to `parse_grid` not provided by the grid.")
            newmat = CellMatrix.from_mat(np.frombuffer(tail, dtype="<f8").reshape(3, 3).T)
    cs = GridCoordinatesSetup(newmat, spacing, dims, size, shift, unitcell, delta)
    return EnergyGrid(cs, tuple(int(x) for x in num_unitcell), ewald_precision, True, grid)


# ------------------------------------------------------------------ interpolation
def interpolation_stencil(csetup: GridCoordinatesSetup, gridshape, point):
    """grids.jl:215-221 -> (p0, p1, r), 1-based indices like the reference.
    ``gridshape`` = (nx, ny, nz)."""
    shifted = offsetpoint(point, csetup)
    p0 = np.floor(shifted).astype(np.int64)
    p1 = p0 + np.array([p0[0] != gridshape[0], p0[1] != gridshape[1], p0[2] != gridshape[2]], dtype=np.int64)
    r = shifted - p0
    return p0, p1, r


def gather_corners(grid: np.ndarray, p0, p1) -> np.ndarray:
    """grids.jl:227-244 -- X[8*c + corner], corner = x + 2y + 4z (x fastest)."""
    x0, y0, z0 = (int(v) - 1 for v in p0)
    x1, y1, z1 = (int(v) - 1 for v in p1)
    X = np.empty(64, dtype=np.float64)
    t = 0
    for c in range(8):
        for (z, y, x) in ((z0, y0, x0), (z0, y0, x1), (z0, y1, x0), (z0, y1, x1),
                          (z1, y0, x0), (z1, y0, x1), (z1, y1, x0), (z1, y1, x1)):
            X[t] = grid[c, x, y, z]
            t += 1
    return X


def interpolate_from_corners(X: np.ndarray, r, is_vdw: bool) -> float:
    """grids.jl:245-258"""
    if is_vdw and np.any(X[:8] > 5e6):
        return 1e100
    a = tricubic_coeff() @ X
    rx, ry, rz = (float(v) for v in r)
    rxs = (1.0, rx, rx * rx, rx * rx * rx)
    rys = (1.0, ry, ry * ry, ry * ry * ry)
    rzs = (1.0, rz, rz * rz, rz * rz * rz)
    ret = 0.0
    for k in range(4):
        for j in range(4):
            for i in range(4):
                ret += a[i + 4 * j + 16 * k] * rxs[i] * rys[j] * rzs[k]
    return ret


def interpolate_grid(g: EnergyGrid, point) -> float:
    """grids.jl:212-273 (K)."""
    if g.ewald_precision == -math.inf:
        return 0.0
    if math.isnan(g.ewald_precision):
        raise ValueError("Invalid grid cannot be interpolated!")
    _, nx, ny, nz = g.grid.shape
    p0, p1, r = interpolation_stencil(g.csetup, (nx, ny, nz), point)
    if not g.higherorder:
        # "no derivatives" (grids.jl:259-269).  The reference indexes this branch g.grid[x, y, z, 1] although its array is
        # [z, y, x, channel]: the first array index (along z) receives the x cell index and the third (along x) the z cell index.
        # Kept as written; an index beyond its axis is Julia's BoundsError, IndexError here.  parse_grid never produces such a
        # grid (grids.jl:92), so only hand-made EnergyGrids get here.
        (x0, y0, z0), (x1, y1, z1) = p0, p1
        rx, ry, rz = r
        mrx, mry, mrz = 1 - rx, 1 - ry, 1 - rz

        def at(a, b, c):          # g.grid[a, b, c, 1] of the Julia array = self.grid[0, c-1, b-1, a-1] of this channel-first copy
            if not (1 <= a <= nz and 1 <= b <= ny and 1 <= c <= nx):
                raise IndexError(f"BoundsError: attempt to access {nz}x{ny}x{nx}x8 grid at index [{a}, {b}, {c}, 1]")
            return float(g.grid[0, c - 1, b - 1, a - 1])
        return (at(x0, y0, z0) * mrx * mry * mrz + at(x1, y0, z0) * rx * mry * mrz + at(x0, y1, z0) * mrx * ry * mrz +
                at(x0, y0, z1) * mrx * mry * rz + at(x1, y1, z0) * rx * ry * mrz + at(x1, y0, z1) * rx * mry * rz +
                at(x0, y1, z1) * mrx * ry * rz + at(x1, y1, z1) * rx * ry * rz)
    X = gather_corners(g.grid, p0, p1)
    return interpolate_from_corners(X, r, g.ewald_precision == math.inf)


# ------------------------------------------------------------------ block files
@dataclass
class BlockFile:
    """coordinates.jl:83-101"""
    csetup: GridCoordinatesSetup
    block: Optional[np.ndarray] = None     # bool[nx,ny,nz]

    @property
    def empty(self) -> bool:
        return self.block is None or not self.block.any()

    def __getitem__(self, pos) -> bool:
        if self.empty:
            return False
        a, b, c = np.round(offsetpoint(pos, self.csetup)).astype(np.int64)   # RoundNearest ties-to-even, like Julia
        return bool(self.block[a - 1, b - 1, c - 1])


def read_block_spheres(file, csetup: GridCoordinatesSetup):
    """The host part of parse_blockfile (coordinates.jl:113-133): sphere centres snapped through
    offsetpoint / inverse_offsetpoint, squared radii -> (centers[n, 3], radius2[n])."""
    from .hostmirror.coordinates import inverse_offsetpoint
    with open(file) as f:
        lines = f.read().splitlines()
    num = int(lines.pop(0))
    centers, radius2 = [], []
    for l in lines[:num]:
        sl = l.split()
        radius = float(sl.pop())
        _center = csetup.cell.mat @ np.array([float(x) for x in sl])
        centers.append(inverse_offsetpoint(offsetpoint(_center, csetup), csetup))
        radius2.append(radius * radius)          # the reference squares the parsed radius (:127, :132)
    return np.array(centers, dtype=np.float64).reshape(-1, 3), np.array(radius2, dtype=np.float64)


def parse_blockfile_gpu(file, csetup: GridCoordinatesSetup, device: int = 0) -> BlockFile:
    """parse_blockfile with the point x sphere scan (coordinates.jl:139-152) on the GPU (``ceg_block_spheres``)."""
    from .hostmirror.utils import prepare_periodic_distance_computations
    centers, radius2 = read_block_spheres(file, csetup)
    if len(radius2) == 0:
        return BlockFile(csetup)
    lib = _abi.load_library()
    dims = np.ascontiguousarray(csetup.dims, dtype=np.int32)
    ortho, safemin = prepare_periodic_distance_computations(csetup.cell.mat)
    out = np.empty(tuple(int(d) + 1 for d in csetup.dims), dtype=np.uint8)
    _abi.check(lib, lib.ceg_block_spheres(device, _abi.i32ptr(dims), _abi.dptr(np.ascontiguousarray(csetup.delta, dtype=np.float64)),
                                          _abi.dptr(np.ascontiguousarray(csetup.shift, dtype=np.float64)), _abi.dptr(_matT(csetup.cell.mat)),
                                          _abi.dptr(_matT(csetup.cell.invmat)), int(ortho), safemin ** 2,
                                          _abi.dptr(np.ascontiguousarray(centers.reshape(-1))), _abi.dptr(radius2), len(radius2),
                                          out.ctypes.data))
    return BlockFile(csetup, out.astype(bool))


def blockfile_from_grid_gpu(g: EnergyGrid, device: int = 0, threshold: float = 5e6) -> BlockFile:
    """``BlockFile(g::EnergyGrid)`` (grids.jl:188-204) on the GPU (``ceg_block_from_grid``)."""
    lib = _abi.load_library()
    dims = np.ascontiguousarray(g.csetup.dims, dtype=np.int32)
    value = np.ascontiguousarray(g.grid[0], dtype=np.float32)
    out = np.empty(value.shape, dtype=np.uint8)
    _abi.check(lib, lib.ceg_block_from_grid(device, value.ctypes.data, 0, _abi.i32ptr(dims), float(threshold), out.ctypes.data))
    return BlockFile(g.csetup, out.astype(bool))


def parse_blockfile(file, csetup: GridCoordinatesSetup) -> BlockFile:
    """coordinates.jl:112-167 (min-image sphere test on every grid point), host-side numpy mirror."""
    from .hostmirror.coordinates import inverse_offsetpoint
    from .hostmirror.utils import prepare_periodic_distance_computations
    with open(file) as f:
        lines = f.read().splitlines()
    num = int(lines.pop(0))
    if num == 0:
        return BlockFile(csetup)
    a, b, c = (int(d) + 1 for d in csetup.dims)
    mat, invmat = csetup.cell.mat, csetup.cell.invmat
    ortho, safemin = prepare_periodic_distance_computations(mat)
    safemin2 = safemin ** 2
    ii, jj, kk = np.meshgrid(np.arange(1, a + 1), np.arange(1, b + 1), np.arange(1, c + 1), indexing="ij")
    pts = (np.stack([ii, jj, kk], axis=-1).reshape(-1, 3) - 1) * csetup.delta + csetup.shift
    block = np.zeros(a * b * c, dtype=bool)
    for l in lines[:num]:
        sl = l.split()
        radius = float(sl.pop())
        _center = mat @ np.array([float(x) for x in sl])
        center = inverse_offsetpoint(offsetpoint(_center, csetup), csetup)
        d = center[None, :] - pts
        f = d @ invmat.T
        f = (f + 0.5) - np.floor(f + 0.5) - 0.5
        v = f @ mat.T
        d2 = (v ** 2).sum(axis=1)
        if not ortho:
            need = d2 > safemin2
            if need.any():
                fn = f[need]
                best = d2[need].copy()
                found = np.zeros(len(best), dtype=bool)
                for ax in range(3):
                    for s in (1.0, -1.0):
                        g = fn.copy()
                        g[:, ax] += s
                        w = g @ mat.T
                        n2 = (w ** 2).sum(axis=1)
                        take = (~found) & (n2 < d2[need])
                        best[take] = n2[take]
                        found |= take
                d2[need] = best
        block |= d2 < radius * radius
    return BlockFile(csetup, block.reshape(a, b, c))


# ------------------------------------------------------------------ CrystalEnergySetup
@dataclass
class CrystalEnergySetup:
    """grids.jl:284-294"""
    framework: object
    molecule: object
    coulomb: EnergyGrid
    charges: List[float]
    grids: List[EnergyGrid]
    atomsidx: List[int]          # 0-based index into ``grids``
    ewald: EwaldFramework
    forcefield: ForceField
    block: BlockFile


def energy_point(setup: CrystalEnergySetup, positions) -> Tuple[float, float]:
    """grids.jl:311-327 -> (vdw, coulomb) in K."""
    positions = [np.asarray(p, dtype=np.float64) for p in positions]
    for pos in positions:
        if setup.block[pos]:
            return 1e100, 0.0
    num_atoms = len(setup.atomsidx)
    vdw = sum(interpolate_grid(setup.grids[setup.atomsidx[i]], positions[i]) for i in range(num_atoms))
    if setup.coulomb.ewald_precision == -math.inf:
        return vdw, 0.0
    coulomb_direct = sum(setup.charges[i] * interpolate_grid(setup.coulomb, positions[i]) for i in range(num_atoms))
    newmolecule = setup.molecule.with_positions(positions)
    coulomb_reciprocal = compute_ewald(setup.ewald, ((newmolecule,),))
    return vdw, coulomb_direct + coulomb_reciprocal
