"""Unit constants and the tricubic coefficient matrix.

Mirrors ``src/constants.jl:20-92`` of the reference.  The reference obtains its two
conversion factors from Unitful/UnitfulAtomic (third-party, version unpinned); the
values below are the CODATA-2018 ones those packages ship since Unitful 1.0.  They
are host-side only: the HIP library is constant-free (lambda / threshold are
arguments, ``src/grids.jl:141-143,168-170``).
"""
from __future__ import annotations

from fractions import Fraction
from functools import lru_cache
import math

import numpy as np

# CODATA 2018 (exact SI values for k and e)
_K_BOLTZMANN = 1.380649e-23          # J/K      (k_au)
_E_CHARGE = 1.602176634e-19          # C        (e_au)
_AMU = 1.66053906660e-27             # kg       (u)
_EPS0 = 8.8541878128e-12             # F/m      (Unitful.ε0)

#: ``NoUnits(true*u"u * Å^2 / ps^2 / k_au / K")``  (constants.jl:20)
GRID_TO_KELVIN: float = _AMU * 1e-20 / 1e-24 / _K_BOLTZMANN
#: ``NoUnits(inv(4π*ε0)*u"e_au^2/Å/k_au/K")`` in K·Å/e²  (constants.jl:21)
COULOMBIC_CONVERSION_FACTOR: float = _E_CHARGE ** 2 / (4.0 * math.pi * _EPS0) / 1e-10 / _K_BOLTZMANN


def nint(x: float) -> int:
    """constants.jl:92"""
    return math.floor(x + 0.5 if x >= 0.0 else x - 0.5)


@lru_cache(maxsize=1)
def tricubic_coeff() -> np.ndarray:
    """The 64x64 integer matrix ``COEFF`` of constants.jl:24-89, *derived* here.

    ``a = COEFF @ X`` maps the 64 Hermite data X (8 channels x 8 corners, channel
    major; corners ordered x fastest, then y, then z; channels value, dx, dy, dz,
    dxy, dxz, dyz, dxyz -- the gather order of grids.jl:227-244) to the
    coefficients ``a[i + 4j + 16k]`` of ``sum a_ijk x^i y^j z^k`` (grids.jl:253-258).
    It is the inverse of the matrix that evaluates those 64 quantities from the
    polynomial coefficients, computed in exact rational arithmetic.
    """
    def dpow(e: int, order: int, x: int) -> int:
        # d^order/dx^order x^e evaluated at x in {0,1}
        if order > e:
            return 0
        c = 1
        for t in range(order):
            c *= (e - t)
        e2 = e - order
        return c if (x == 1 or e2 == 0) else 0

    chans = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1),
             (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
    B = [[Fraction(0)] * 64 for _ in range(64)]
    for c, (ox, oy, oz) in enumerate(chans):
        for corner in range(8):
            x, y, z = corner & 1, (corner >> 1) & 1, (corner >> 2) & 1
            row = 8 * c + corner
            for k in range(4):
                for j in range(4):
                    for i in range(4):
                        B[row][i + 4 * j + 16 * k] = Fraction(
                            dpow(i, ox, x) * dpow(j, oy, y) * dpow(k, oz, z))
    # Gauss-Jordan inverse over the rationals
    n = 64
    A = [B[r] + [Fraction(int(r == c)) for c in range(n)] for r in range(n)]
    for col in range(n):
        piv = next(r for r in range(col, n) if A[r][col] != 0)
        A[col], A[piv] = A[piv], A[col]
        pv = A[col][col]
        A[col] = [v / pv for v in A[col]]
        for r in range(n):
            if r != col and A[r][col] != 0:
                f = A[r][col]
                A[r] = [a - f * b for a, b in zip(A[r], A[col])]
    inv = np.array([[float(v) for v in row[n:]] for row in A], dtype=np.float64)
    assert np.all(inv == np.round(inv))
    return inv
