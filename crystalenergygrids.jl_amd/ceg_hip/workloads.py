"""Named workloads of BASELINE.json / SURVEY §8d, built from the reference's own fixture data
(``tests/golden/raspa`` = data files of ``/root/reference/test/raspa``).  Deterministic,
no RNG."""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

from ._abi import REPO_DIR
from .coordinates import CellMatrix, GridCoordinatesSetup
from .ewald import ewald_alpha
from .forcefields import ForceField
from .probes import ProbeSystem
from .raspa import RASPASystem, load_framework_RASPA, parse_forcefield_RASPA, setdir_RASPA

FIXTURE_RASPA = REPO_DIR / "tests" / "golden" / "raspa"
FORCEFIELD = "BoulfelfelSholl2021"


def use_fixture_dir() -> None:
    setdir_RASPA(FIXTURE_RASPA)


def tile_framework(fw: RASPASystem, reps: Tuple[int, int, int]) -> RASPASystem:
    """Replicate a framework ``reps`` times along a, b, c into a bigger unit cell."""
    na, nb, nc = reps
    a, b, c = fw.mat[:, 0], fw.mat[:, 1], fw.mat[:, 2]
    pos, sym, mass, q = [], [], [], []
    for ia in range(na):
        for ib in range(nb):
            for ic in range(nc):
                pos.append(fw.position + (ia * a + ib * b + ic * c))
                sym += list(fw.atomic_symbol)
                mass.append(fw.atomic_mass)
                q.append(fw.atomic_charge)
    mat = np.column_stack((na * a, nb * b, nc * c))
    return RASPASystem(mat, np.concatenate(pos), sym, np.concatenate(mass), np.concatenate(q), False)


def grid_setup_with_dims(mat: np.ndarray, dims: Tuple[int, int, int]) -> GridCoordinatesSetup:
    """GridCoordinatesSetup (coordinates.jl:32-41) with ``dims`` imposed instead of derived from a
    spacing -- size/shift/delta follow the reference's rules."""
    cell = CellMatrix.from_mat(mat)
    a, b, c = cell.mat[:, 0], cell.mat[:, 1], cell.mat[:, 2]
    size = np.abs(a) + np.abs(b) + np.abs(c)
    shift = np.minimum(a, 0.0) + np.minimum(b, 0.0) + np.minimum(c, 0.0)
    d = np.asarray(dims, dtype=np.int32)
    assert np.all(d % 2 == 1), "dims must be odd (coordinates.jl:36-37)"
    unitcell = np.array([np.linalg.norm(a), np.linalg.norm(b), np.linalg.norm(c)])
    delta = size / d
    return GridCoordinatesSetup(cell, float(np.max(delta)), d, size, shift, unitcell, delta)


@dataclass
class Workload:
    name: str
    framework: RASPASystem
    forcefield: ForceField
    cset: GridCoordinatesSetup
    probe_vdw: Optional[ProbeSystem]
    probe_coulomb: Optional[ProbeSystem]
    alpha: float

    @property
    def npoints(self) -> int:
        nx, ny, nz = self.cset.npoints
        return nx * ny * nz

    @property
    def natoms(self) -> int:
        p = self.probe_vdw if self.probe_vdw is not None else self.probe_coulomb
        return len(p.positions)


def fixture_workload(framework: str, atom: Optional[str], spacing: float, coulomb: bool = True,
                     dims: Optional[Tuple[int, int, int]] = None, tile: Optional[Tuple[int, int, int]] = None,
                     name: Optional[str] = None) -> Workload:
    use_fixture_dir()
    ff = parse_forcefield_RASPA(FORCEFIELD)
    fw = load_framework_RASPA(framework, FORCEFIELD)
    if tile is not None:
        fw = tile_framework(fw, tile)
    cset = grid_setup_with_dims(fw.mat, dims) if dims is not None else GridCoordinatesSetup.from_cell(fw.mat, spacing)
    pv = ProbeSystem.build(fw, ff, atom) if atom else None
    pc = ProbeSystem.build(fw, ff) if coulomb else None
    alpha, _ = ewald_alpha()
    return Workload(name or f"{framework}/{atom}/{spacing}", fw, ff, cset, pv, pc, alpha)


def roofline_workload(atom: str = "Ar", n: int = 255) -> Workload:
    """SURVEY §8d run "R": CHA_1.4_3b4eeb96 tiled 2x2x3 (11 664 atoms), (n+1)^3 grid points over
    the cartesian bounding box of that cell (n = 255 -> 256^3)."""
    return fixture_workload("CHA_1.4_3b4eeb96", atom, 0.0, coulomb=True, dims=(n, n, n), tile=(2, 2, 3),
                            name=f"CHA_1.4_3b4eeb96 tiled 2x2x3 (11664 atoms) x {n + 1}^3 grid, {atom} probe LJ + real-space Ewald")
