"""Host-side pieces of the reference the CHECKER needs, restated under oracle/ so that the oracle does not
import the product package's host mirror (a bug in ``ceg_hip.hostmirror.ewald`` / ``ceg_hip.hostmirror.utils`` /
``ceg_hip.hostmirror.constants`` used to be common-mode to the GPU path and to its checker).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` /
self-check legs -- never by the product package.  Nothing here imports ``ceg_hip``.

Restated (file:line under /root/reference):
  * ``COEFF``                                   src/constants.jl:24-89 (derived, and pinned to the literal through
                                                tests/golden/coeff.json by tests/test_oracle_hostlogic.py)
  * ``nint``                                    src/constants.jl:92
  * ``prepare_periodic_distance_computations``  src/utils.jl:129-138,146-155 (C: ceg_oracle_mc.c)
  * ``initialize_ewald``                        src/ewald.jl:195-281 (alpha, k-space box, kindices here; kfactors and
                                                StoreRigidChargeFramework in C with the literal power tables)
  * ``EwaldContext`` constants                  src/ewald.jl:475-544
The two unit constants are CODATA-2018 numbers (the reference takes them from Unitful / UnitfulAtomic, third party,
version unpinned: SURVEY 8c); they are written out here digit by digit rather than imported.
"""
from __future__ import annotations

import ctypes as C
import json
import math
from dataclasses import dataclass
from functools import lru_cache
from pathlib import Path
from typing import List, Sequence, Tuple

import numpy as np

_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)

# src/constants.jl:20-21 evaluated with CODATA 2018: u = 1.66053906660e-27 kg, k_B = 1.380649e-23 J/K,
# e = 1.602176634e-19 C, eps0 = 8.8541878128e-12 F/m
GRID_TO_KELVIN = 1.66053906660e-27 * 1e-20 / 1e-24 / 1.380649e-23
COULOMBIC_CONVERSION_FACTOR = (1.602176634e-19 ** 2) / (4.0 * math.pi * 8.8541878128e-12) / 1e-10 / 1.380649e-23


def _lib():
    from . import oracle as O
    l = O.lib()
    if not getattr(l, "_hostlogic_bound", False):
        l.oracle_prepare_periodic_distance_computations.restype = None
        l.oracle_prepare_periodic_distance_computations.argtypes = [_dp, _i32p, _dp]
        l.oracle_ewald_kfactors.restype = None
        l.oracle_ewald_kfactors.argtypes = [_i32p, C.c_int64, _dp, C.c_double, C.c_double, _dp]
        l.oracle_framework_structure_factor.restype = None
        l.oracle_framework_structure_factor.argtypes = [_i32p, C.c_int64, _i32p, C.c_int64, _dp, _i32p, _dp, _dp, C.c_int64, _dp, _dp]
        l.oracle_molecule_sums.restype = None
        l.oracle_molecule_sums.argtypes = [_i32p, C.c_int64, _i32p, C.c_int64, _dp, _dp, _dp, C.c_int32, _dp, _dp]
        l.oracle_single_contribution_ewald.restype = C.c_double
        l.oracle_single_contribution_ewald.argtypes = [_dp, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        l.oracle_compute_ewald_total.restype = C.c_double
        l.oracle_compute_ewald_total.argtypes = [_dp, C.c_int64, _dp, _dp, _dp, _dp, C.c_double, C.c_double]
        l._hostlogic_bound = True
    return l


def _d(a):
    return a.ctypes.data_as(_dp)


def _cm(m) -> np.ndarray:
    """3x3 (columns = cell vectors) -> column-major 9-vector"""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float64).T.reshape(9))


def nint(x: float) -> int:
    """src/constants.jl:92: floor(Int, ifelse(x >= 0.0, x + 0.5, x - 0.5))"""
    return int(math.floor(x + 0.5 if x >= 0.0 else x - 0.5))


# ------------------------------------------------------------------------------------------------ COEFF
@lru_cache(maxsize=1)
def tricubic_coeff() -> np.ndarray:
    """``COEFF`` (src/constants.jl:24-89) from the 1-D cubic Hermite matrix.

    In one dimension the cubic p(t) = c0 + c1 t + c2 t^2 + c3 t^3 on [0, 1] with data (p(0), p(1), p'(0), p'(1)) has
    c = H @ data with H below.  ``interpolate_grid`` (src/grids.jl:227-258) gathers X[8*ch + corner] with channels
    (value, dx, dy, dz, dxy, dxz, dyz, dxyz) and corners x fastest, and evaluates sum a[i + 4j + 16k] x^i y^j z^k: the 3-D
    matrix is the tensor product of three H, re-indexed to that layout."""
    H = np.array([[1, 0, 0, 0],          # columns: f(0), f(1), f'(0), f'(1)
                  [0, 0, 1, 0],
                  [-3, 3, -2, -1],
                  [2, -2, 1, 1]], dtype=np.int64)
    channels = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]   # derivative orders
    out = np.zeros((64, 64), dtype=np.int64)
    for ch, (ox, oy, oz) in enumerate(channels):
        for corner in range(8):
            cx, cy, cz = corner & 1, (corner >> 1) & 1, (corner >> 2) & 1
            col = 8 * ch + corner
            for k in range(4):
                for j in range(4):
                    for i in range(4):
                        out[i + 4 * j + 16 * k, col] = H[i, 2 * ox + cx] * H[j, 2 * oy + cy] * H[k, 2 * oz + cz]
    return out.astype(np.float64)


def reference_coeff_literal() -> np.ndarray:
    """The reference's own literal (src/constants.jl:24-89) as committed data: tests/golden/coeff.json
    (written by tests/golden/make_coeff.py from the reference tree)."""
    path = Path(__file__).resolve().parent.parent / "tests" / "golden" / "coeff.json"
    return np.array(json.loads(path.read_text())["rows"], dtype=np.float64)


# ------------------------------------------------------------------------------------------------ cell analysis
def prepare_periodic_distance_computations(mat) -> Tuple[bool, float]:
    """src/utils.jl:146-155 -> (ortho, safemin)"""
    ortho = C.c_int32(0)
    safemin = C.c_double(0.0)
    _lib().oracle_prepare_periodic_distance_computations(_d(_cm(mat)), C.byref(ortho), C.byref(safemin))
    return bool(ortho.value), float(safemin.value)


# ------------------------------------------------------------------------------------------------ EwaldFramework
@dataclass
class OracleEwaldFramework:
    """The fields of ``EwaldFramework`` (src/ewald.jl:39-49) the energies need."""
    ks: Tuple[int, int, int]
    num_kvecs: int
    kindices: np.ndarray            # int32[nrows, 5]: (j, k, i_first, i_last, rangeidx 0-based)
    alpha: float
    mat: np.ndarray                 # supercell matrix, columns = cell vectors
    invmat: np.ndarray
    kfactors: np.ndarray
    UIon: float
    sf_re: np.ndarray               # StoreRigidChargeFramework
    sf_im: np.ndarray
    net_charges_framework: float
    precision: float


def ewald_kindices(ks: Sequence[int]) -> Tuple[np.ndarray, int]:
    """src/ewald.jl:213-236: the rows (j, k, irange, rangeidx) of ``kspace.kindices`` and ``num_kvecs``.
    Written with the reference's own bookkeeping: each pushed row starts where the previous one ended
    (``lastrangeidx + length(lastrange)``), the range of the (0, 0) row starts at 1 (``(j==k==0):...``)."""
    kx, ky, kz = (int(v) for v in ks)
    recip_cutoff2 = (1.05 * max(kx, ky, kz)) ** 2
    num_kvecs = 0
    rows: List[Tuple[int, int, int, int, int]] = []

    def push(j, k, last_i):
        if rows:
            _, _, lo, hi, idx = rows[-1]
            nxt = idx + (hi - lo + 1)
        else:
            nxt = 0
        rows.append((j, k, 1 if (j == 0 and k == 0) else 0, last_i, nxt))

    for j in range(-ky, ky + 1):
        for k in range(-kz, kz + 1):
            started = False
            for i in range(0, kx + 1):
                r2_a = i * i + j * j + k * k
                if (r2_a != 0) and (r2_a < recip_cutoff2):
                    num_kvecs += 1
                    if not started:
                        started = True
                elif started:
                    push(j, k, i - 1)
                    started = False
                    break
            if started:
                push(j, k, kx)
    arr = np.array(rows, dtype=np.int32).reshape(-1, 5)
    return np.ascontiguousarray(arr), num_kvecs


def initialize_ewald(cell_mat, positions, charges, supercell: Sequence[int], precision: float = 1e-6) -> OracleEwaldFramework:
    """src/ewald.jl:195-281.  ``cell_mat``: unit-cell matrix (columns = a, b, c, in A); ``positions`` cartesian
    float64[n, 3] of the unit cell's atoms, ``charges`` float64[n] (both may be empty: :291-296)."""
    cutoff_coulomb = 12.0
    eps = cutoff_coulomb * min(0.5, abs(precision))                     # :201
    tol = math.sqrt(abs(math.log(eps)))
    alpha = math.sqrt(abs(math.log(eps * tol))) / cutoff_coulomb
    tol1 = math.sqrt(-math.log(eps * 4.0 * (tol * alpha) ** 2))

    sc = tuple(int(v) for v in supercell)
    mat = np.array(cell_mat, dtype=np.float64) * np.array(sc, dtype=np.float64)[None, :]   # bounding_box .* supercell: column q scaled
    lens = [math.sqrt(float(mat[0, q] ** 2 + mat[1, q] ** 2 + mat[2, q] ** 2)) for q in range(3)]
    a_ = alpha * tol1 / math.pi
    ks = (nint(0.25 + a_ * lens[0]), nint(0.25 + a_ * lens[1]), nint(0.25 + a_ * lens[2]))
    kind, num_kvecs = ewald_kindices(ks)

    invmat = np.linalg.inv(mat)
    volume_factor = COULOMBIC_CONVERSION_FACTOR * 2 * math.pi / float(np.linalg.det(mat))
    assert volume_factor > 0
    alpha_factor = -0.25 / alpha ** 2
    kfactors = np.zeros(num_kvecs, dtype=np.float64)
    l = _lib()
    inv_cm = _cm(invmat)
    l.oracle_ewald_kfactors(kind.ctypes.data_as(_i32p), len(kind), _d(inv_cm), volume_factor, alpha_factor, _d(kfactors))
    ksum = 0.0
    for v in kfactors:                                                    # sum(kfactors; init=0.0)
        ksum += float(v)
    UIon = COULOMBIC_CONVERSION_FACTOR * alpha / math.sqrt(math.pi) - ksum

    pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
    q = np.ascontiguousarray(charges, dtype=np.float64).reshape(-1)
    assert len(pos) == len(q)
    re = np.zeros(num_kvecs)
    im = np.zeros(num_kvecs)
    ks32 = np.array(ks, dtype=np.int32)
    sc32 = np.array(sc, dtype=np.int32)
    l.oracle_framework_structure_factor(kind.ctypes.data_as(_i32p), len(kind), ks32.ctypes.data_as(_i32p), num_kvecs, _d(inv_cm),
                                        sc32.ctypes.data_as(_i32p), _d(pos.reshape(-1)) if len(pos) else None,
                                        _d(q) if len(q) else None, len(pos), _d(re), _d(im))
    net = 0.0
    for c in np.repeat(q, sc[0] * sc[1] * sc[2]):                         # sum(charges; init=0.0), charges repeated inner = Pi
        net += float(c)
    return OracleEwaldFramework(ks, num_kvecs, kind, alpha, mat, invmat, kfactors, UIon, re, im, net, precision)


def ewald_context_constants(ef, kinds: Sequence[Tuple[Sequence[float], np.ndarray, int]]) -> Tuple[float, float]:
    """The two constants of an ``EwaldContext`` (src/ewald.jl:485-544) -> (energy_net_charges, static_contribution) in K.
    ``kinds``: one (charges, positions of the FIRST molecule [natoms, 3], number of molecules) per kind."""
    from . import oracle as O
    alpha = float(ef.alpha)
    chargefactor = COULOMBIC_CONVERSION_FACTOR / math.sqrt(math.pi) * alpha
    energies = []
    for charges, _pos, _num in kinds:
        s = 0.0
        for c in charges:
            s += float(c) * float(c)                                      # sum(abs2, charges; init=0.0)
        energies.append(s * chargefactor)
    energy_adsorbate_self = 0.0
    for (_c, _p, num), eas in zip(kinds, energies):
        energy_adsorbate_self += eas * num
    total_net_charges = 0.0
    for charges, _p, num in kinds:
        net = 0.0
        for c in charges:
            net += float(c)
        total_net_charges += net * num
    ortho, safemin = prepare_periodic_distance_computations(ef.mat)
    safemin2 = safemin ** 2
    energy_adsorbate_excluded = 0.0
    for charges, pos, num in kinds:
        pos = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
        this_energy = 0.0
        n = len(charges)
        for A in range(n):
            for B in range(A + 1, n):
                d2, _ = O.periodic_distance2(pos[B] - pos[A], ef.mat, ef.invmat, ortho, safemin2)
                r = math.sqrt(d2)
                this_energy += math.erf(alpha * r) * float(charges[A]) * float(charges[B]) / r
        energy_adsorbate_excluded += num * this_energy * COULOMBIC_CONVERSION_FACTOR
    static_contribution = ef.UIon * total_net_charges ** 2 - energy_adsorbate_self - energy_adsorbate_excluded
    energy_net_charges = ef.UIon * ef.net_charges_framework * total_net_charges
    return energy_net_charges, static_contribution


def molecule_sums(ef, positions, charges) -> Tuple[np.ndarray, np.ndarray]:
    """Structure factor of one molecule (move_one_system! + update_sums!, src/ewald.jl:352-366,660-684) -> (re, im)."""
    pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
    q = np.ascontiguousarray(charges, dtype=np.float64).reshape(-1)
    assert len(pos) == len(q)
    re = np.zeros(ef.num_kvecs)
    im = np.zeros(ef.num_kvecs)
    ks32 = np.array(ef.ks, dtype=np.int32)
    kind = np.ascontiguousarray(ef.kindices, dtype=np.int32)
    _lib().oracle_molecule_sums(kind.ctypes.data_as(_i32p), len(kind), ks32.ctypes.data_as(_i32p), ef.num_kvecs, _d(_cm(ef.invmat)),
                                _d(pos.reshape(-1)), _d(q), len(q), _d(re), _d(im))
    return re, im


def adapt_ewald_framework(ef) -> OracleEwaldFramework:
    """INPUT ADAPTER: read the plain arrays out of an EwaldFramework-like object built elsewhere (attribute access only)."""
    ksp = ef.kspace
    kind = np.ascontiguousarray(np.array(ksp.kindices, dtype=np.int32).reshape(-1, 5))
    sf = np.asarray(ef.StoreRigidChargeFramework)
    return OracleEwaldFramework(tuple(int(v) for v in ksp.ks), int(ksp.num_kvecs), kind, float(ef.alpha), np.array(ef.mat, dtype=np.float64),
                                np.array(ef.invmat, dtype=np.float64), np.ascontiguousarray(ef.kfactors, dtype=np.float64), float(ef.UIon),
                                np.ascontiguousarray(sf.real, dtype=np.float64), np.ascontiguousarray(sf.imag, dtype=np.float64),
                                float(ef.net_charges_framework), float(ef.precision))
