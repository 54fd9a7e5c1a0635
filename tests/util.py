"""Helpers for the parity tests: synthetic ProbeSystems (no RASPA files) and comparisons."""
from __future__ import annotations

import numpy as np

from ceg_hip.hostmirror.coordinates import GridCoordinatesSetup
from ceg_hip.hostmirror.forcefields import ForceField
from ceg_hip.hostmirror.interactions import FF, InteractionRule, InteractionRuleSum, make_rule
from ceg_hip.hostmirror.probes import ProbeSystem
from ceg_hip.workloads import grid_setup_with_dims


def tiny_forcefield(cutoff: float = 12.0, hs_radius: float = 1.5, generic: bool = False) -> ForceField:
    """kinds: 1 LJ-shifted, 2 Buckingham+HardSphere(hs_radius), 3 none, 4 LJ (other params) -- or,
    with generic=True, a LJ+Buckingham sum that has no fast class in the kernel; probe = 5 (P).  Probes 6 (Q) and 7 (R) are
    Lennard-Jones-only against all four kinds (Q: none against C, LJ + CoulombEwaldDirect against D; R: none against B)."""
    lj = InteractionRule(FF.LennardJones, [107.69, 3.15], 0.0, False)
    lj = InteractionRule(FF.LennardJones, [107.69, 3.15], lj(cutoff), False)
    buck = InteractionRuleSum([InteractionRule(FF.HardSphere, [hs_radius, 0.0]), InteractionRule(FF.Buckingham, [5.581e7, 3.985, 9.167e5]),
                               InteractionRule(FF.CoulombEwaldDirect, [0.265, 0.9, -0.9], 0.0, False)])
    none = make_rule(FF.NoInteraction)
    lj2 = InteractionRule(FF.LennardJones, [262.0, 2.396])
    if generic:
        lj2 = InteractionRuleSum([InteractionRule(FF.LennardJones, [40.0, 2.9]), InteractionRule(FF.Buckingham, [3.0e6, 3.2, 2.0e4])])
    n = 7
    inter = [[none] * n for _ in range(n)]
    for k, r in enumerate((lj, buck, none, lj2)):
        inter[k][4] = inter[4][k] = r
    # two more probes, Lennard-Jones against every framework kind they meet (what a multi-probe plan takes): Q (6) and R (7)
    ljq = [InteractionRule(FF.LennardJones, [55.0, 3.4]), InteractionRule(FF.LennardJones, [81.5, 2.9]), none,
           InteractionRuleSum([InteractionRule(FF.LennardJones, [23.0, 3.05]), InteractionRule(FF.CoulombEwaldDirect, [0.265, 0.3, -0.6], 0.0, False)])]
    ljq[0] = InteractionRule(FF.LennardJones, [55.0, 3.4], ljq[0](cutoff), False)               # shifted, like kind A with P
    ljr = [InteractionRule(FF.LennardJones, [140.0, 3.3]), none, InteractionRule(FF.LennardJones, [12.0, 3.9]),
           InteractionRule(FF.LennardJones, [66.0, 2.75])]
    for k in range(4):
        inter[k][5] = inter[5][k] = ljq[k]
        inter[k][6] = inter[6][k] = ljr[k]
    sdict = {"A": 1, "B": 2, "C": 3, "D": 4, "P": 5, "Q": 6, "R": 7}
    return ForceField(inter, sdict, list(sdict), cutoff, "tiny")


def synthetic_probes(mat, positions, kinds, charges, cutoff: float = 12.0, probes=None, **ffkw):
    """(vdw probe, coulomb probe) over an explicit supercell `mat` -- bypasses find_supercell so
    that cells violating the 2*cutoff rule can be tested too."""
    ff = tiny_forcefield(cutoff, **ffkw)
    mat = np.array(mat, dtype=np.float64)
    pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
    kinds = np.asarray(kinds, dtype=np.int64)
    q = np.asarray(charges, dtype=np.float64)
    inv = np.linalg.inv(mat)
    pv = ProbeSystem(pos, mat, inv, ff, kinds, np.empty(0), 5)
    pc = ProbeSystem(pos, mat, inv, ff, kinds, q, 0)
    if probes is not None:         # one VdW ProbeSystem per requested probe index (5 = P, 6 = Q, 7 = R)
        return [ProbeSystem(pos, mat, inv, ff, kinds, np.empty(0), p) for p in probes], pc
    return pv, pc


def random_atoms(mat, n, rng, min_sep=1.6):
    """n positions uniformly in the cell with a minimum separation (brute force rejection)."""
    out = []
    inv = np.linalg.inv(mat)
    while len(out) < n:
        p = mat @ rng.uniform(0, 1, 3)
        ok = True
        for q in out:
            d = inv @ (p - q)
            d -= np.round(d)
            if np.linalg.norm(mat @ d) < min_sep:
                ok = False
                break
        if ok:
            out.append(p)
    return np.array(out)


def grid_points(cset: GridCoordinatesSetup) -> np.ndarray:
    nx, ny, nz = cset.npoints
    ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    return np.stack([ii * cset.size[0] / cset.dims[0] + cset.shift[0],
                     jj * cset.size[1] / cset.dims[1] + cset.shift[1],
                     kk * cset.size[2] / cset.dims[2] + cset.shift[2]], axis=-1).reshape(-1, 3)


def compare_raw(got: np.ndarray, ref: np.ndarray, what: str, rtol: float = 1e-9):
    """FP64 8-vectors before _set_gridpoint!.  Non-finite patterns must match exactly; finite
    values to `rtol` relative with an absolute floor of 1e-3*rtol x the column's upper-quartile magnitude
    (cancellation between +/- pair terms makes tiny sums order dependent)."""
    assert got.shape == ref.shape
    assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{what}: NaN pattern differs"
    inf = np.isinf(ref)
    assert np.array_equal(np.isinf(got), inf) and np.array_equal(got[inf], ref[inf]), f"{what}: Inf pattern differs"
    worst = 0.0
    for c in range(ref.shape[1]):
        m = np.isfinite(ref[:, c])
        if not m.any():
            continue
        g, r = got[m, c], ref[m, c]
        scale = float(np.percentile(np.abs(r), 75))
        tol = rtol * np.abs(r) + rtol * 1e-3 * scale
        bad = np.abs(g - r) > tol
        if bad.any():
            w = int(np.argmax(np.abs(g - r) - tol))
            raise AssertionError(f"{what}: column {c}: {int(bad.sum())} values off; worst got {g[w]!r} ref {r[w]!r} "
                                 f"(scale {scale:.3e}, tol {tol[w]:.3e})")
        nz = np.abs(r) > 1e-3 * scale
        if nz.any():
            worst = max(worst, float(np.max(np.abs(g - r)[nz] / np.abs(r)[nz])))
    return worst
