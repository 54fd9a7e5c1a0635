"""``ProbeSystem`` -- framework replicated to the minimal supercell (mirror of
``src/probes.jl:12-63``).  The per-point sums ``compute_derivatives_vdw`` /
``compute_derivatives_ewald`` (probes.jl:71-117) are NOT here: they are the HIP
kernels behind ``libceg_hip.so``."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .forcefields import ForceField
from .utils import find_supercell, get_atom_name, prepare_periodic_distance_computations


@dataclass
class ProbeSystem:
    positions: np.ndarray      # float64[natoms, 3] cartesian Å
    mat: np.ndarray            # supercell matrix (columns = axes)
    invmat: np.ndarray
    forcefield: ForceField
    atomkinds: np.ndarray      # int64[natoms], 1-based force-field index
    charges: np.ndarray        # float64[natoms] (Coulomb probe) or empty
    probe: int                 # 1-based ff index of the probe atom, 0 for Coulomb
    num_supercell: tuple = (1, 1, 1)

    @classmethod
    def build(cls, framework, forcefield: ForceField, atom: Optional[str] = None) -> "ProbeSystem":
        """probes.jl:21-63.  ``framework`` is a :class:`ceg_hip.raspa.RASPASystem`."""
        n = len(framework)
        _atomkinds = np.array([forcefield.sdict[get_atom_name(s)] for s in framework.atomic_symbol],
                              dtype=np.int64)
        bbox = framework.mat                       # columns a, b, c
        nx, ny, nz = find_supercell(bbox, 12.0)
        numsupercell = nx * ny * nz
        mat = np.column_stack((nx * bbox[:, 0], ny * bbox[:, 1], nz * bbox[:, 2]))
        invmat = np.linalg.inv(mat)
        base = np.array(framework.position, dtype=np.float64).reshape(n, 3)
        if numsupercell == 1:
            atomkinds, positions = _atomkinds, base.copy()
        else:
            wx, wy, wz = bbox[:, 0], bbox[:, 1], bbox[:, 2]
            positions = np.empty((numsupercell * n, 3), dtype=np.float64)
            positions[:n] = base
            for iz in range(nz):                    # probes.jl:37-53, same index arithmetic
                izn = iz * n
                nnz = n * nz
                stepz = iz * wz
                for iy in range(ny):
                    nnynz = ny * nnz
                    iyz = iy * nnz + izn
                    stepyz = iy * wy + stepz
                    for ix in range(1 if (iz == 0 and iy == 0) else 0, nx):
                        ixyz = ix * nnynz + iyz
                        stepxyz = ix * wx + stepyz
                        positions[ixyz:ixyz + n] = base + stepxyz
            atomkinds = np.tile(_atomkinds, numsupercell)
        if atom is None or atom == "":
            charges = np.tile(np.asarray(framework.atomic_charge, dtype=np.float64), numsupercell)
            return cls(positions, mat, invmat, forcefield, atomkinds, charges, 0, (nx, ny, nz))
        probe = forcefield.sdict[get_atom_name(atom)]
        return cls(positions, mat, invmat, forcefield, atomkinds, np.empty(0), probe, (nx, ny, nz))

    def periodic_setup(self):
        """``prepare_periodic_distance_computations(s.mat)`` (probes.jl:72-73) ->
        ``(ortho, safemin2)``."""
        ortho, safemin = prepare_periodic_distance_computations(self.mat)
        return ortho, safemin * safemin

    @property
    def cutoff2(self) -> float:
        return self.forcefield.cutoff ** 2
