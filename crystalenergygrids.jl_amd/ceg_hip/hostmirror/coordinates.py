"""Grid geometry (mirror of ``src/coordinates.jl:15-76``).  Lengths in Å, plain floats."""
from __future__ import annotations

from dataclasses import dataclass
import math
from typing import Tuple

import numpy as np


@dataclass
class CellMatrix:
    """utils.jl:282-292"""
    mat: np.ndarray
    invmat: np.ndarray

    @classmethod
    def from_mat(cls, mat) -> "CellMatrix":
        m = np.array(mat, dtype=np.float64)
        return cls(m, np.linalg.inv(m))


@dataclass
class GridCoordinatesSetup:
    """coordinates.jl:15-23.  ``dims`` = number of points minus one per axis (int32,
    always odd), ``size`` = extent of the cartesian bounding box of the unit cell,
    ``shift`` = its lower corner, ``delta`` = actual spacing."""
    cell: CellMatrix
    spacing: float
    dims: np.ndarray       # int32[3]
    size: np.ndarray       # float64[3]
    shift: np.ndarray
    unitcell: np.ndarray
    delta: np.ndarray

    @classmethod
    def from_cell(cls, cell, spacing: float) -> "GridCoordinatesSetup":
        """coordinates.jl:32-41"""
        if not isinstance(cell, CellMatrix):
            cell = CellMatrix.from_mat(cell)
        a, b, c = cell.mat[:, 0], cell.mat[:, 1], cell.mat[:, 2]
        size = np.abs(a) + np.abs(b) + np.abs(c)
        shift = np.minimum(a, 0.0) + np.minimum(b, 0.0) + np.minimum(c, 0.0)
        _dims = np.floor(size / spacing).astype(np.int32)
        dims = (_dims + (_dims % 2 == 0)).astype(np.int32)
        unitcell = np.array([np.linalg.norm(a), np.linalg.norm(b), np.linalg.norm(c)])
        delta = size / dims
        return cls(cell, float(spacing), dims, size, shift, unitcell, delta)

    @property
    def npoints(self) -> Tuple[int, int, int]:
        return tuple(int(d) + 1 for d in self.dims)  # type: ignore[return-value]


def wrap_atom(point, cell: CellMatrix) -> np.ndarray:
    """coordinates.jl:58-61"""
    abc = cell.invmat @ np.asarray(point, dtype=np.float64)
    return cell.mat @ (abc - np.floor(abc))


def offsetpoint(point, csetup: GridCoordinatesSetup) -> np.ndarray:
    """coordinates.jl:63-66 -- 1-based fractional grid index of a (wrapped) point."""
    newpoint = wrap_atom(point, csetup.cell)
    return (newpoint - csetup.shift) * csetup.dims / csetup.size + 1


def inverse_offsetpoint(ipoint, csetup: GridCoordinatesSetup) -> np.ndarray:
    """coordinates.jl:68-70"""
    return (np.asarray(ipoint, dtype=np.float64) - 1) * csetup.delta + csetup.shift


def abc_to_xyz(cset: GridCoordinatesSetup, i: int, j: int, k: int) -> np.ndarray:
    """coordinates.jl:72-76, evaluated as ``(i*size)/dims + shift``."""
    return np.array([i * cset.size[0] / cset.dims[0] + cset.shift[0],
                     j * cset.size[1] / cset.dims[1] + cset.shift[1],
                     k * cset.size[2] / cset.dims[2] + cset.shift[2]])
