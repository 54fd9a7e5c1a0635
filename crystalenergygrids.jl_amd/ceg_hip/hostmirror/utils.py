"""Host-side geometry helpers (mirror of the parts of ``src/utils.jl`` the grid
build needs).  3x3 cell matrices are numpy arrays whose COLUMNS are the cell
vectors a, b, c, exactly like the Julia ``SMatrix`` (``mat[:, 0]`` is a).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np


def perpendicular_lengths(mat: np.ndarray) -> np.ndarray:
    """utils.jl:10-29 -- perpendicular widths (volume / face area)."""
    a, b, c = mat[:, 0], mat[:, 1], mat[:, 2]
    axb = np.cross(a, b)
    bxc = np.cross(b, c)
    cxa = np.cross(c, a)
    volume = abs(float(np.dot(a, bxc)))
    return np.array([volume / np.linalg.norm(bxc),
                     volume / np.linalg.norm(cxa),
                     volume / np.linalg.norm(axb)])


def find_supercell(mat_or_widths, cutoff: float) -> Tuple[int, int, int]:
    """utils.jl:43-67 -- ``ceil(2*cutoff / perpendicular width)`` per axis."""
    x = np.asarray(mat_or_widths, dtype=np.float64)
    widths = perpendicular_lengths(x) if x.shape == (3, 3) else x
    return tuple(int(math.ceil(2 * cutoff / w)) for w in widths)  # type: ignore[return-value]


def cell_parameters(mat: np.ndarray):
    """utils.jl:123-132"""
    _a, _b, _c = mat[:, 0], mat[:, 1], mat[:, 2]
    a = float(np.linalg.norm(_a))
    b = float(np.linalg.norm(_b))
    c = float(np.linalg.norm(_c))
    alpha = math.degrees(math.acos(float(np.dot(_b, _c)) / (b * c)))
    beta = math.degrees(math.acos(float(np.dot(_c, _a)) / (c * a)))
    gamma = math.degrees(math.acos(float(np.dot(_a, _b)) / (a * b)))
    return (a, b, c), (alpha, beta, gamma)


def mat_from_parameters(lengths, angles) -> np.ndarray:
    """utils.jl:134-138 (same convention as Chemfiles: a along x, b in the xy plane)."""
    a, b, c = lengths
    al, be, ga = (math.radians(x) for x in angles)
    cosa, cosb = math.cos(al), math.cos(be)
    sing, cosg = math.sin(ga), math.cos(ga)
    omega = math.sqrt(1 - cosa ** 2 - cosb ** 2 - cosg ** 2 + 2 * cosa * cosb * cosg)
    return np.array([[a, b * cosg, c * cosb],
                     [0.0, b * sing, c * (cosa - cosb * cosg) / sing],
                     [0.0, 0.0, c * omega / sing]], dtype=np.float64)


def prepare_periodic_distance_computations(mat: np.ndarray) -> Tuple[bool, float]:
    """utils.jl:146-155 -> ``(ortho, safemin)``.

    ``ortho`` is evaluated like ``isapprox(Float16(x), 90; rtol=0.02)``: the angle is
    rounded to half precision, the difference to 90 is taken in half precision and
    compared with ``0.02*max(|x|, 90)`` in double precision.
    """
    (a, b, c), angles = cell_parameters(mat)

    def approx90(x: float) -> bool:
        x16 = np.float16(x)
        diff = abs(np.float16(x16 - np.float16(90)))
        return bool(x16 == 90 or float(diff) <= 0.02 * max(abs(float(x16)), 90.0))

    ortho = all(approx90(x) for x in angles)
    _a, _b, _c = mat[:, 0], mat[:, 1], mat[:, 2]
    safemin = min(float(np.dot(np.cross(_b, _c), _a)) / (b * c),
                  float(np.dot(np.cross(_c, _a), _b)) / (a * c),
                  float(np.dot(np.cross(_a, _b), _c)) / (a * b)) / 2
    return ortho, safemin


def get_atom_name(atom) -> str:
    """utils.jl:521-538 -- strip a trailing ``_<digits>`` (or trailing digits)."""
    name = str(atom)
    assert name.isascii()
    if all(ch.isalpha() for ch in name):
        return name
    s = name.split('_')
    if len(s) > 1:
        if all(ch.isnumeric() for ch in s[-1]):     # all() of an empty string is True, as in Julia
            return '_'.join(s[:-1])
    else:
        i = len(name)
        while name[i - 1].isnumeric():
            i -= 1
        if i < len(name):
            return name[:i]
    return name
