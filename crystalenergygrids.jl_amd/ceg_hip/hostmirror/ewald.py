"""Ewald summation set-up and reciprocal-space energy (host-side mirror of
``src/ewald.jl:195-281`` and ``:475-577``).

Only ``alpha`` feeds the grid-build kernels (``derivatives_ewald``, ewald.jl:299-312,
is a HIP kernel).  The reciprocal part is kept on the host, as in the reference,
and is needed so ``energy_point`` (grids.jl:311-327) returns the same number as
the reference for charged guests -- which is how the oracle is pinned to the
literals of ``test/runtests.jl``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from .constants import COULOMBIC_CONVERSION_FACTOR, nint
from .utils import find_supercell, prepare_periodic_distance_computations


@dataclass
class EwaldKspace:
    """ewald.jl:27-31.  ``kindices`` rows: (j, k, i_first, i_last, rangeidx)."""
    ks: Tuple[int, int, int]
    num_kvecs: int
    kindices: List[Tuple[int, int, int, int, int]]


@dataclass
class EwaldFramework:
    """ewald.jl:39-49"""
    kspace: EwaldKspace
    alpha: float
    mat: np.ndarray
    invmat: np.ndarray
    kfactors: np.ndarray
    UIon: float
    StoreRigidChargeFramework: np.ndarray
    net_charges_framework: float
    precision: float
    kvec_ijk: np.ndarray = None      # int64[num_kvecs,3], derived from kindices

    @classmethod
    def empty(cls, mat) -> "EwaldFramework":
        """ewald.jl:50-54"""
        m = np.array(mat, dtype=np.float64)
        return cls(EwaldKspace((0, 0, 0), 0, []), 0.0, m, np.linalg.inv(m), np.empty(0), 0.0,
                   np.empty(0, dtype=np.complex128), 0.0, 0.0, np.empty((0, 3), dtype=np.int64))


def ewald_alpha(precision: float = 1e-6, cutoff: float = 12.0):
    """ewald.jl:198-204 -> (alpha, tol1)"""
    eps = cutoff * min(0.5, abs(precision))
    tol = math.sqrt(abs(math.log(eps)))
    alpha = math.sqrt(abs(math.log(eps * tol))) / cutoff
    tol1 = math.sqrt(-math.log(eps * 4.0 * (tol * alpha) ** 2))
    return alpha, tol1


def _structure_factor(kvec_ijk: np.ndarray, frac: np.ndarray, charges: np.ndarray) -> np.ndarray:
    """sum_sites q * exp(2*pi*i * (ijk . frac)) -- what ewald_main_loop! (ewald.jl:148-185)
    accumulates from the Eikx/Eiky/Eikz recurrences (ewald.jl:75-146), evaluated directly."""
    out = np.zeros(len(kvec_ijk), dtype=np.complex128)
    if len(charges) == 0 or len(kvec_ijk) == 0:
        return out
    chunk = max(1, int(4_000_000 // max(1, len(kvec_ijk))))
    kf = kvec_ijk.astype(np.float64)
    for s in range(0, len(charges), chunk):
        phase = 2.0 * np.pi * (frac[s:s + chunk] @ kf.T)          # [sites, nk]
        out += (charges[s:s + chunk, None] * np.exp(1j * phase)).sum(axis=0)
    return out


def initialize_ewald(syst, supercell=None, precision: float = 1e-6) -> EwaldFramework:
    """ewald.jl:195-281.  ``syst`` is a RASPASystem (or a bare 3x3 matrix for an
    empty framework, ewald.jl:291-296)."""
    if isinstance(syst, np.ndarray):
        from .raspa import RASPASystem
        syst = RASPASystem(np.array(syst, dtype=np.float64), np.empty((0, 3)), [], np.empty(0), np.empty(0))
    if supercell is None:
        supercell = find_supercell(syst.mat, 12.0)
    alpha, tol1 = ewald_alpha(precision)
    mat = syst.mat * np.asarray(supercell, dtype=np.float64)[None, :]
    len_a, len_b, len_c = (float(np.linalg.norm(mat[:, q])) for q in range(3))
    __a = alpha * tol1 / math.pi
    kx = nint(0.25 + __a * len_a)
    ky = nint(0.25 + __a * len_b)
    kz = nint(0.25 + __a * len_c)
    recip_cutoff2 = (1.05 * max(kx, ky, kz)) ** 2
    num_kvecs = 0
    kindices: List[Tuple[int, int, int, int, int]] = []
    nextidx = 0
    for j in range(-ky, ky + 1):
        for k in range(-kz, kz + 1):
            started = False
            first = 1 if (j == 0 and k == 0) else 0
            closed = False
            for i in range(0, kx + 1):
                r2_a = i * i + j * j + k * k
                if r2_a != 0 and r2_a < recip_cutoff2:
                    num_kvecs += 1
                    started = True
                elif started:
                    kindices.append((j, k, first, i - 1, nextidx))
                    nextidx += (i - 1) - first + 1
                    started = False
                    closed = True
                    break
            if started and not closed:
                kindices.append((j, k, first, kx, nextidx))
                nextidx += kx - first + 1
    assert nextidx == num_kvecs
    invmat = np.linalg.inv(mat)
    volume_factor = COULOMBIC_CONVERSION_FACTOR * 2 * math.pi / float(np.linalg.det(mat))
    assert volume_factor > 0
    alpha_factor = -0.25 / alpha ** 2
    kvec_ijk = np.empty((num_kvecs, 3), dtype=np.int64)
    for (j, k, i0, i1, ridx) in kindices:
        n = i1 - i0 + 1
        kvec_ijk[ridx:ridx + n, 0] = np.arange(i0, i1 + 1)
        kvec_ijk[ridx:ridx + n, 1] = j
        kvec_ijk[ridx:ridx + n, 2] = k
    rk = 2 * math.pi * (kvec_ijk.astype(np.float64) @ invmat)      # rows: invmat^T . (i,j,k)
    rksqr = (rk ** 2).sum(axis=1)
    kfactors = volume_factor * (1 + (kvec_ijk[:, 0] != 0)) * np.exp(alpha_factor * rksqr) / rksqr
    UIon = COULOMBIC_CONVERSION_FACTOR * alpha / math.sqrt(math.pi) - float(kfactors.sum())

    PA, PB, PC = supercell
    n = len(syst)
    base_frac = (np.asarray(syst.position, dtype=np.float64).reshape(n, 3) @ invmat.T) if n else np.empty((0, 3))
    # sites: atom-major, images inner in order (pa, pb, pc) with pc fastest (ewald.jl:127-131)
    img = np.array([[pa / PA, pb / PB, pc / PC] for pa in range(PA) for pb in range(PB) for pc in range(PC)])
    frac = (base_frac[:, None, :] + img[None, :, :]).reshape(-1, 3)
    charges = np.repeat(np.asarray(syst.atomic_charge, dtype=np.float64), PA * PB * PC)
    store = _structure_factor(kvec_ijk, frac, charges)
    net = float(charges.sum()) if len(charges) else 0.0
    return EwaldFramework(EwaldKspace((kx, ky, kz), num_kvecs, kindices), alpha, mat, invmat, kfactors,
                          UIon, store, net, precision, kvec_ijk)


def _periodic_distance(d: np.ndarray, mat, invmat, ortho, safemin2) -> float:
    """sqrt of periodic_distance2_fromcartesian! (utils.jl:210-246), host-side use only
    (intramolecular exclusion term, ewald.jl:527-540)."""
    f = invmat @ d
    f = (f + 0.5) - np.floor(f + 0.5) - 0.5
    v = mat @ f
    ref2 = float(v @ v)
    if ortho or ref2 <= safemin2:
        return math.sqrt(ref2)
    for i in range(3):
        for s in (1.0, -1.0):
            g = f.copy()
            g[i] += s
            v = mat @ g
            n2 = float(v @ v)
            if n2 < ref2:
                return math.sqrt(n2)
    return math.sqrt(ref2)


def ewald_context_constants(eframework: EwaldFramework, systems: Sequence[Sequence]):
    """The two constants an ``EwaldContext`` carries (ewald.jl:497-544):
    ``(energy_net_charges, static_contribution)`` in K.  ``systems`` is a sequence of kinds, each a
    sequence of molecules; the intramolecular exclusion term uses the first molecule of each kind
    (rigid molecules)."""
    ef = eframework
    allcharges = [np.asarray(kind[0].atomic_charge, dtype=np.float64) for kind in systems]
    numspecies = [len(kind) for kind in systems]
    chargefactor = COULOMBIC_CONVERSION_FACTOR / math.sqrt(math.pi) * ef.alpha
    energies = [float((c ** 2).sum()) * chargefactor for c in allcharges]
    energy_adsorbate_self = sum(e * num for num, e in zip(numspecies, energies))
    net_charges = [float(c.sum()) for c in allcharges]
    total_net_charges = sum(c * num for num, c in zip(numspecies, net_charges))
    ortho, safemin = prepare_periodic_distance_computations(ef.mat)
    safemin2 = safemin ** 2
    energy_adsorbate_excluded = 0.0
    for num, kind, charges in zip(numspecies, systems, allcharges):
        syst = kind[0]
        pos = np.asarray(syst.position, dtype=np.float64).reshape(len(syst), 3)
        this_energy = 0.0
        for A in range(len(syst)):
            for B in range(A + 1, len(syst)):
                r = _periodic_distance(pos[B] - pos[A], ef.mat, ef.invmat, ortho, safemin2)
                this_energy += math.erf(ef.alpha * r) * charges[A] * charges[B] / r
        energy_adsorbate_excluded += num * this_energy * COULOMBIC_CONVERSION_FACTOR
    static_contribution = ef.UIon * total_net_charges ** 2 - energy_adsorbate_self - energy_adsorbate_excluded
    energy_net_charges = ef.UIon * ef.net_charges_framework * total_net_charges
    return energy_net_charges, static_contribution


def kindices_array(eframework: EwaldFramework) -> np.ndarray:
    """``kspace.kindices`` as int32[nkind, 5] rows (j, k, i_first, i_last, rangeidx)."""
    return np.ascontiguousarray(np.array(eframework.kspace.kindices, dtype=np.int32).reshape(-1, 5))


def compute_ewald(eframework: EwaldFramework, systems: Sequence[Sequence], skipcontribution: int = 0) -> float:
    """``compute_ewald(eframework, systems)`` = ``compute_ewald(EwaldContext(eframework,
    systems))`` (ewald.jl:475-577), in K.  ``systems`` is a sequence of kinds, each a
    sequence of molecules (RASPASystem) of that kind."""
    if eframework.alpha == 0.0:
        return 0.0
    assert skipcontribution == 0
    ef = eframework
    energy_net_charges, static_contribution = ewald_context_constants(ef, systems)

    fr, ch = [], []
    for kind in systems:
        for syst in kind:
            pos = np.asarray(syst.position, dtype=np.float64).reshape(len(syst), 3)
            fr.append(pos @ ef.invmat.T)
            ch.append(np.asarray(syst.atomic_charge, dtype=np.float64))
    frac = np.concatenate(fr) if fr else np.empty((0, 3))
    charges = np.concatenate(ch) if ch else np.empty(0)
    new = _structure_factor(ef.kvec_ijk, frac, charges)
    f = ef.StoreRigidChargeFramework
    framework_adsorbate = float((ef.kfactors * (f.real * new.real + f.imag * new.imag)).sum())
    adsorbate_adsorbate = float((ef.kfactors * (new.real ** 2 + new.imag ** 2)).sum())
    return 2 * (framework_adsorbate + energy_net_charges) + adsorbate_adsorbate + static_contribution
