"""``energy_point`` / ``energy_grid`` for many placements at once, entirely on the GPU (SURVEY §8f
rows f1 + f2): Van der Waals and real-space Coulomb terms by batched interpolation of the
device-resident grids (:mod:`ceg_hip.interp`), reciprocal-space Ewald term by ``ceg_recip_*``.
Reference: ``energy_point`` ``src/grids.jl:311-327``, ``energy_grid`` ``src/grids.jl:346-424``.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np

from . import _abi
from .ewald import EwaldFramework, ewald_context_constants
from .grids import CrystalEnergySetup, _matT
from .interp import GridInterpolator


class ReciprocalEwald:
    """Device-resident k-space tables of an :class:`EwaldFramework` (``ceg_recip_*``)."""

    def __init__(self, ef: EwaldFramework, device: int = 0):
        if ef.alpha == 0.0:
            raise ValueError("Ewald summation is not defined for this framework (alpha == 0)")
        self._lib = _abi.load_library()
        self.ef = ef
        ijk = np.ascontiguousarray(ef.kvec_ijk, dtype=np.int32)
        kf = np.ascontiguousarray(ef.kfactors, dtype=np.float64)
        re = np.ascontiguousarray(ef.StoreRigidChargeFramework.real, dtype=np.float64)
        im = np.ascontiguousarray(ef.StoreRigidChargeFramework.imag, dtype=np.float64)
        ks = np.asarray(ef.kspace.ks, dtype=np.int32)
        inv = _matT(ef.invmat)
        h = C.c_void_p()
        rc = self._lib.ceg_recip_create(C.byref(h), device, _abi.i32ptr(ijk.reshape(-1)), _abi.dptr(kf), _abi.dptr(re),
                                        _abi.dptr(im), len(kf), _abi.i32ptr(ks), _abi.dptr(inv))
        _abi.check(self._lib, rc)
        self._h = h

    def energies(self, molecule, positions) -> np.ndarray:
        """compute_ewald for ``molecule`` (a RASPASystem; charges + internal geometry) placed at
        ``positions[n, natoms, 3]`` -> K, float64[n]."""
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, len(molecule), 3)
        q = np.ascontiguousarray(molecule.atomic_charge, dtype=np.float64)
        enc, static = ewald_context_constants(self.ef, ((molecule,),))
        out = np.empty(len(pos), dtype=np.float64)
        _abi.check(self._lib, self._lib.ceg_recip_energy(self._h, _abi.dptr(pos.reshape(-1)), _abi.dptr(q), len(q), len(pos),
                                                         enc, static, _abi.dptr(out)))
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ceg_recip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuEnergySetup:
    """A :class:`CrystalEnergySetup` whose grids and k-space tables live on the GPU."""

    def __init__(self, setup: CrystalEnergySetup, device: int = 0):
        self.setup = setup
        self.vdw = [GridInterpolator(g, device) if g.ewald_precision == math.inf else None for g in setup.grids]
        self.has_coulomb = setup.coulomb.ewald_precision != -math.inf
        self.coulomb = GridInterpolator(setup.coulomb, device) if self.has_coulomb else None
        self.recip = ReciprocalEwald(setup.ewald, device) if self.has_coulomb else None

    def energy_points(self, positions) -> np.ndarray:
        """``energy_point(setup, positions[p])`` for every placement p -> float64[n, 2] (vdw, coulomb).
        ``positions[n, natoms, 3]`` in Å.  Blocking spheres short-circuit to (1e100, 0) like the
        reference (grids.jl:312-314)."""
        s = self.setup
        natoms = len(s.atomsidx)
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, natoms, 3)
        n = len(pos)
        vdw = np.zeros(n)
        for a in range(natoms):
            it = self.vdw[s.atomsidx[a]]
            if it is not None:                       # zero grid -> 0 K (grids.jl:213)
                vdw += it(pos[:, a])
        out = np.zeros((n, 2))
        out[:, 0] = vdw
        if self.has_coulomb:
            direct = np.zeros(n)
            for a in range(natoms):
                direct += s.charges[a] * self.coulomb(pos[:, a])
            out[:, 1] = direct + self.recip.energies(s.molecule, pos)
        if not s.block.empty:
            blocked = np.array([any(s.block[p] for p in mol) for mol in pos])
            out[blocked] = (1e100, 0.0)
        return out

    def energy_grid(self, step: float) -> np.ndarray:
        """``energy_grid(setup, step)`` for a mono-atomic guest or ``num_rotate == 0`` (grids.jl:346-424):
        float64[numA, numB, numC] of ``sum(energy_point)`` on the fractional lattice of the unit cell."""
        s = self.setup
        a, b, c = s.framework.mat[:, 0], s.framework.mat[:, 1], s.framework.mat[:, 2]
        numA = int(math.floor(np.linalg.norm(a) / step)) + 1
        numB = int(math.floor(np.linalg.norm(b) / step)) + 1
        numC = int(math.floor(np.linalg.norm(c) / step)) + 1
        stepA, stepB, stepC = a / numA, b / numB, c / numC
        iA, iB, iC = np.meshgrid(np.arange(numA), np.arange(numB), np.arange(numC), indexing="ij")
        ofs = iA[..., None] * stepA + iB[..., None] * stepB + iC[..., None] * stepC          # grids.jl:396
        base = np.asarray(s.molecule.position, dtype=np.float64).reshape(-1, 3)
        pos = ofs.reshape(-1, 1, 3) + base[None, :, :]
        e = self.energy_points(pos)
        return (e[:, 0] + e[:, 1]).reshape(numA, numB, numC)

    def close(self) -> None:
        for it in self.vdw:
            if it is not None:
                it.close()
        if self.coulomb is not None:
            self.coulomb.close()
        if self.recip is not None:
            self.recip.close()
