"""``GridPlan`` -- a ProbeSystem + GridCoordinatesSetup made resident on one GPU
(``ceg_plan_*`` entry points of ``include/ceg_hip.h``).

This is what a one-process-per-GPU driver uses: every rank builds its own x-slab of the
grid into device memory on its own HIP stream; slabs are exchanged by the caller (RCCL
all-gather in ``bench.py`` / ``ceg_hip.distributed``).  Device buffers are passed as raw
pointers, so any allocator works (torch tensors' ``data_ptr()`` in this repo).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _abi
from .hostmirror.coordinates import GridCoordinatesSetup
from .grids import _grid_args, _matT, coulomb_scaling, vdw_scaling
from .hostmirror.probes import ProbeSystem


class GridPlan:
    def __init__(self, cset: GridCoordinatesSetup, vdw: Optional[ProbeSystem] = None,
                 coulomb: Optional[ProbeSystem] = None, alpha: float = 0.0, device: int = 0):
        if vdw is None and coulomb is None:
            raise ValueError("GridPlan needs a VdW probe and/or a Coulomb probe")
        ref = vdw if vdw is not None else coulomb
        if vdw is not None and coulomb is not None:
            if vdw.positions.shape != coulomb.positions.shape or not np.array_equal(vdw.positions, coulomb.positions):
                raise ValueError("VdW and Coulomb probes must share the same framework")
        self._lib = _abi.load_library()
        self.cset = cset
        self.device = device
        self.has_vdw = vdw is not None
        self.has_coulomb = coulomb is not None
        pos = np.ascontiguousarray(ref.positions, dtype=np.float64)
        mat, invmat = _matT(ref.mat), _matT(ref.invmat)
        ortho, safemin2 = ref.periodic_setup()
        dims, size, shift, delta = _grid_args(cset)
        kinds_p = rules_p = offs_p = None
        nkinds = 0
        keep = [pos, mat, invmat, dims, size, shift, delta]
        if vdw is not None:
            ff = vdw.forcefield
            ff.check_vdw_grid(vdw.probe, np.unique(vdw.atomkinds))
            rules, offsets = ff.rule_table(vdw.probe)
            kinds = np.ascontiguousarray(vdw.atomkinds, dtype=np.int64)
            kinds_p, rules_p, offs_p, nkinds = _abi.i64ptr(kinds), rules.ctypes.data, _abi.i32ptr(offsets), ff.nkinds
            keep += [rules, offsets, kinds]
        q_p = None
        if coulomb is not None:
            q = np.ascontiguousarray(coulomb.charges, dtype=np.float64)
            q_p = _abi.dptr(q)
            keep.append(q)
        handle = C.c_void_p()
        rc = self._lib.ceg_plan_create(C.byref(handle), device, _abi.dptr(pos), kinds_p, q_p, len(pos),
                                       _abi.dptr(mat), _abi.dptr(invmat), int(ortho), safemin2, ref.cutoff2,
                                       rules_p, offs_p, nkinds, float(alpha),
                                       _abi.i32ptr(dims), _abi.dptr(size), _abi.dptr(shift), _abi.dptr(delta))
        _abi.check(self._lib, rc)
        self._h = handle
        self.npoints = cset.npoints
        self.plane = self.npoints[1] * self.npoints[2]

    # ------------------------------------------------------------------ info
    @property
    def can_cull(self) -> bool:
        return bool(self._lib.ceg_plan_can_cull(self._h))

    @property
    def num_images(self) -> int:
        return int(self._lib.ceg_plan_num_images(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ceg_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ builds (device pointers)
    def build_vdw(self, d_out: int, channel_stride: int, i_begin: int, i_end: int, i_origin: int = 0,
                  algo: int = _abi.ALGO_AUTO, stream: int = 0) -> None:
        lam, thr = vdw_scaling()
        _abi.check(self._lib, self._lib.ceg_plan_build_vdw(self._h, lam, thr, i_begin, i_end, d_out, channel_stride,
                                                           i_origin, algo, stream))

    def build_coulomb(self, d_out: int, channel_stride: int, i_begin: int, i_end: int, i_origin: int = 0,
                      algo: int = _abi.ALGO_AUTO, stream: int = 0) -> None:
        lam, thr = coulomb_scaling()
        _abi.check(self._lib, self._lib.ceg_plan_build_coulomb(self._h, lam, thr, i_begin, i_end, d_out,
                                                               channel_stride, i_origin, algo, stream))

    def build_fused(self, d_vdw: int, d_coulomb: int, channel_stride: int, i_begin: int, i_end: int,
                    i_origin: int = 0, algo: int = _abi.ALGO_AUTO, stream: int = 0) -> None:
        lv, tv = vdw_scaling()
        lc, tc = coulomb_scaling()
        _abi.check(self._lib, self._lib.ceg_plan_build_fused(self._h, lv, tv, lc, tc, i_begin, i_end, d_vdw,
                                                             d_coulomb, channel_stride, i_origin, algo, stream))

    # ------------------------------------------------------------------ raw FP64 sums at points
    def eval_points(self, which: str, points, algo: int = _abi.ALGO_AUTO) -> np.ndarray:
        """compute_derivatives_vdw / _ewald (probes.jl:71-117) at cartesian points ->
        float64[n, 8] (value, d1[3], d2[3], d3)."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        out = np.empty((len(pts), 8), dtype=np.float64)
        w = {"vdw": 0, "coulomb": 1}[which]
        _abi.check(self._lib, self._lib.ceg_plan_eval_points(self._h, w, algo, _abi.dptr(pts), len(pts), _abi.dptr(out)))
        return out


class MultiGridPlan:
    """The K VdW probes (one grid each) + optionally the charges of ONE framework made resident on one GPU
    (``ceg_plan_create_multi``): what ``setup_RASPA`` needs for a molecule -- one ``create_grid_vdw`` per distinct
    guest atom and one ``create_grid_coulomb`` (raspa.jl:497-520) -- from one lattice-image list in one pass.

    ``probes``: ProbeSystems of the same framework (same positions / kinds / cutoff), one per probe atom, of any rule class
    ``ceg_plan_create`` takes (round 4): the Lennard-Jones-only probes share accumulating loops, a probe of another class (a
    Buckingham cation) is launched alone or fused with the Coulomb grid -- all from the one image list."""
    MAX_PROBES = 4

    def __init__(self, cset: GridCoordinatesSetup, probes, coulomb: Optional[ProbeSystem] = None, alpha: float = 0.0,
                 device: int = 0):
        probes = list(probes)
        if not 1 <= len(probes) <= self.MAX_PROBES:
            raise ValueError(f"1..{self.MAX_PROBES} probes per multi-probe plan")
        ref = probes[0]
        for other in probes[1:] + ([coulomb] if coulomb is not None else []):
            if other.positions.shape != ref.positions.shape or not np.array_equal(other.positions, ref.positions):
                raise ValueError("all probes of a multi-probe plan must share the same framework")
        for other in probes[1:]:
            if not np.array_equal(other.atomkinds, ref.atomkinds) or other.cutoff2 != ref.cutoff2:
                raise ValueError("all probes of a multi-probe plan must share atom kinds and cutoff")
        self._lib = _abi.load_library()
        self.cset, self.device = cset, device
        self.nprobes = len(probes)
        self.has_coulomb = coulomb is not None
        pos = np.ascontiguousarray(ref.positions, dtype=np.float64)
        mat, invmat = _matT(ref.mat), _matT(ref.invmat)
        ortho, safemin2 = ref.periodic_setup()
        dims, size, shift, delta = _grid_args(cset)
        kinds = np.ascontiguousarray(ref.atomkinds, dtype=np.int64)
        tables = []
        for pr in probes:
            pr.forcefield.check_vdw_grid(pr.probe, np.unique(pr.atomkinds))
            tables.append(pr.forcefield.rule_table(pr.probe))
        rules_pp = (C.c_void_p * self.nprobes)(*[t[0].ctypes.data for t in tables])
        offs_pp = (C.c_void_p * self.nprobes)(*[t[1].ctypes.data for t in tables])
        q = np.ascontiguousarray(coulomb.charges, dtype=np.float64) if coulomb is not None else None
        handle = C.c_void_p()
        rc = self._lib.ceg_plan_create_multi(C.byref(handle), device, _abi.dptr(pos), _abi.i64ptr(kinds), _abi.dptr(q) if q is not None else None,
                                             len(pos), _abi.dptr(mat), _abi.dptr(invmat), int(ortho), safemin2, ref.cutoff2,
                                             self.nprobes, rules_pp, offs_pp, ref.forcefield.nkinds, float(alpha),
                                             _abi.i32ptr(dims), _abi.dptr(size), _abi.dptr(shift), _abi.dptr(delta))
        _abi.check(self._lib, rc)
        self._h = handle
        self.npoints = cset.npoints
        self.plane = self.npoints[1] * self.npoints[2]

    @property
    def num_images(self) -> int:
        return int(self._lib.ceg_plan_num_images(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ceg_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build(self, d_vdw, d_coulomb: int, channel_stride: int, i_begin: int, i_end: int, i_origin: int = 0, stream: int = 0) -> None:
        """``d_vdw``: device pointer per probe (0 / None = skip that grid), ``d_coulomb``: device pointer or 0.  Asynchronous."""
        d_vdw = list(d_vdw) + [0] * (self.nprobes - len(d_vdw))
        ptrs = (C.c_void_p * self.nprobes)(*[int(x) if x else None for x in d_vdw])
        lv, tv = vdw_scaling()
        lc, tc = coulomb_scaling()
        _abi.check(self._lib, self._lib.ceg_plan_build_multi(self._h, lv, tv, lc, tc, i_begin, i_end, ptrs, int(d_coulomb) if d_coulomb else None,
                                                             channel_stride, i_origin, stream))
