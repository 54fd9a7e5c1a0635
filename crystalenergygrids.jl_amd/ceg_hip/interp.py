"""Batched ``interpolate_grid`` on the GPU (reference: ``src/grids.jl:212-273``,
consumers ``framework_interactions`` ``src/montecarlo.jl:490-504`` and ``energy_grid``
``src/grids.jl:394-419``).  The grid stays resident on the device; many positions are
interpolated per call.  Scalar reference semantics are in :func:`ceg_hip.grids.interpolate_grid`.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np

from . import _abi
from .grids import EnergyGrid, _matT


class GridInterpolator:
    """Device-resident :class:`EnergyGrid` + ``ceg_interp_*`` handle."""

    def __init__(self, g: EnergyGrid, device: int = 0, device_ptr: Optional[int] = None):
        """``g.grid`` must already be in K (as returned by ``parse_grid``).  With ``device_ptr`` the
        values are read in place from that device buffer (same layout) instead of being uploaded."""
        if g.ewald_precision == -math.inf or math.isnan(g.ewald_precision):
            raise ValueError("zero / invalid grids are handled on the host (interpolate_grid returns 0 / raises)")
        self._lib = _abi.load_library()
        cs = g.csetup
        dims = np.ascontiguousarray(cs.dims, dtype=np.int32)
        size = np.ascontiguousarray(cs.size, dtype=np.float64)
        shift = np.ascontiguousarray(cs.shift, dtype=np.float64)
        mat, inv = _matT(cs.cell.mat), _matT(cs.cell.invmat)
        self._keep = (dims, size, shift, mat, inv)
        h = C.c_void_p()
        if device_ptr is None:
            grid = np.ascontiguousarray(g.grid, dtype=np.float32)
            ptr, on_dev = grid.ctypes.data, 0
        else:
            ptr, on_dev = int(device_ptr), 1
        rc = self._lib.ceg_interp_create(C.byref(h), device, ptr, on_dev, _abi.i32ptr(dims), _abi.dptr(size),
                                         _abi.dptr(shift), _abi.dptr(mat), _abi.dptr(inv),
                                         1 if g.ewald_precision == math.inf else 0)
        _abi.check(self._lib, rc)
        self._h = h
        if not g.higherorder:
            _abi.check(self._lib, self._lib.ceg_interp_set_higherorder(self._h, 0))

    @classmethod
    def from_file(cls, path, iscoulomb: bool, mat=None, device: int = 0, with_header: bool = False):
        """A cached ``.grid`` file straight onto the device (``ceg_interp_create_from_file``): what ``parse_grid`` (grids.jl:61-94)
        + ``GridInterpolator(g)`` do, without the host array -- payload streamed file -> pinned ring -> device, scaled by
        GRID_TO_KELVIN there.  ``mat``: the unit-cell matrix as for ``parse_grid``; None = the one stored in the file."""
        import os
        from .hostmirror.constants import GRID_TO_KELVIN
        from .hostmirror.coordinates import CellMatrix
        self = cls.__new__(cls)
        self._lib = _abi.load_library()
        m = i = None
        if mat is not None:
            cm = mat if isinstance(mat, CellMatrix) else CellMatrix.from_mat(mat)
            m, i = _matT(cm.mat), _matT(cm.invmat)
        self._keep = (m, i)
        h = C.c_void_p()
        hdr = _abi.GridHeader()
        rc = self._lib.ceg_interp_create_from_file(C.byref(h), device, os.fsencode(str(path)), 1 if iscoulomb else 0, GRID_TO_KELVIN,
                                                   _abi.dptr(m) if m is not None else None, _abi.dptr(i) if i is not None else None,
                                                   C.cast(C.byref(hdr), C.c_void_p))
        _abi.check(self._lib, rc)
        self._h = h
        self.header = hdr
        return (self, hdr) if with_header else self

    def __call__(self, points) -> np.ndarray:
        """interpolate_grid(g, p) for every row p of ``points`` (Å) -> K, float64[n]."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        out = np.empty(len(pts), dtype=np.float64)
        _abi.check(self._lib, self._lib.ceg_interp_points(self._h, _abi.dptr(pts), len(pts), _abi.dptr(out)))
        return out

    def on_device(self, d_points: int, n: int, d_out: int, stream: int = 0) -> None:
        _abi.check(self._lib, self._lib.ceg_interp_points_device(self._h, d_points, n, d_out, stream))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ceg_interp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def framework_interactions(interps, coulomb: Optional[GridInterpolator], charges, indices, positions):
    """Batched ``framework_interactions`` (montecarlo.jl:490-504): ``positions[k]`` is interpolated on
    VdW grid ``interps[indices[k]]`` and, if given, on the Coulomb grid times ``charges[indices[k]]``
    (a blocked Coulomb value, 1e100, is passed through unscaled).  Returns ``(vdw, direct)`` in K."""
    positions = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
    indices = np.asarray(indices)
    vdw = 0.0
    for ix in np.unique(indices):
        sel = indices == ix
        vdw += float(np.sum(interps[ix](positions[sel])))
    direct = 0.0
    if coulomb is not None:
        c = coulomb(positions)
        q = np.asarray(charges, dtype=np.float64)[indices]
        direct = float(np.sum(np.where(c == 1e100, c, q * c)))
    return vdw, direct
