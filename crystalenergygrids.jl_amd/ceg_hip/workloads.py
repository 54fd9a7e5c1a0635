"""Named workloads of BASELINE.json / SURVEY §8d, built from the reference's own fixture data
(``tests/golden/raspa`` = data files of ``/root/reference/test/raspa``).  Deterministic,
no RNG."""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

from ._abi import REPO_DIR
from .hostmirror.coordinates import CellMatrix, GridCoordinatesSetup
from .hostmirror.ewald import ewald_alpha
from .hostmirror.forcefields import ForceField
from .hostmirror.probes import ProbeSystem
from .hostmirror.raspa import RASPASystem, load_framework_RASPA, parse_forcefield_RASPA, setdir_RASPA

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


def roofline_workload(atom: str = "Ar", n: int = 255, truncate: Optional[int] = None) -> Workload:
    """SURVEY §8d run "R": CHA_1.4_3b4eeb96 tiled 2x2x3 (11 664 atoms), (n+1)^3 grid points over
    the cartesian bounding box of that cell (n = 255 -> 256^3).  ``truncate=10000`` keeps the first
    10 000 atoms (the "exact 10 k" variant of BASELINE.json config 3)."""
    w = fixture_workload("CHA_1.4_3b4eeb96", atom, 0.0, coulomb=True, dims=(n, n, n), tile=(2, 2, 3),
                         name=f"CHA_1.4_3b4eeb96 tiled 2x2x3 (11664 atoms) x {n + 1}^3 grid, {atom} probe LJ + real-space Ewald")
    if truncate is not None:
        fw = w.framework
        fw = RASPASystem(fw.mat, fw.position[:truncate], list(fw.atomic_symbol[:truncate]), fw.atomic_mass[:truncate],
                         fw.atomic_charge[:truncate], False)
        w = Workload(w.name.replace("11664 atoms", f"first {truncate} of 11664 atoms"), fw, w.forcefield, w.cset,
                     ProbeSystem.build(fw, w.forcefield, atom), ProbeSystem.build(fw, w.forcefield), w.alpha)
    return w


def _random_atoms_min_sep(n: int, edge: float, min_sep: float, rng) -> np.ndarray:
    """n points uniform in [0, edge)^3, periodic minimum separation ``min_sep`` (sequential rejection
    against a cell hash, so the sequence is fully determined by ``rng``)."""
    nc = int(edge // min_sep)
    h = edge / nc
    cells = {}
    out = np.empty((n, 3))
    k = 0
    while k < n:
        p = rng.uniform(0.0, edge, 3)
        c = np.minimum((p / h).astype(int), nc - 1)
        ok = True
        for d in np.ndindex(3, 3, 3):
            key = tuple((c + np.array(d) - 1) % nc)
            for q in cells.get(key, ()):
                dd = p - out[q]
                dd -= edge * np.round(dd / edge)
                if dd @ dd < min_sep * min_sep:
                    ok = False
                    break
            if not ok:
                break
        if ok:
            out[k] = p
            cells.setdefault(tuple(c), []).append(k)
            k += 1
    return out


def synthetic_workload(natoms: int, n: int = 127, edge: float = 40.0, seed: int = 0) -> Workload:
    """SURVEY §8d fully synthetic sweep variant: orthorhombic ``edge`` A cube, ``natoms`` uniform-random
    atoms with 1.5 A minimum separation (``numpy.random.default_rng(seed)``), a single LJ kind
    (eps = 100 K, sigma = 3 A, shifted at the 12 A cutoff), charges +1/-1 alternating; (n+1)^3 grid."""
    from .hostmirror.interactions import FF, InteractionRule, make_rule
    rng = np.random.default_rng(seed)
    pos = _random_atoms_min_sep(natoms, edge, 1.5, rng)
    mat = np.diag([edge] * 3)
    lj = InteractionRule(FF.LennardJones, [100.0, 3.0], 0.0, False)
    lj = InteractionRule(FF.LennardJones, [100.0, 3.0], lj(12.0), False)
    none = make_rule(FF.NoInteraction)
    inter = [[none, lj], [lj, none]]
    sdict = {"X": 1, "P": 2}
    ff = ForceField(inter, sdict, list(sdict), 12.0, "synthetic")
    q = np.where(np.arange(natoms) % 2 == 0, 1.0, -1.0)
    fw = RASPASystem(mat, pos, ["X"] * natoms, np.ones(natoms), q, False)
    inv = np.linalg.inv(mat)
    kinds = np.ones(natoms, dtype=np.int64)
    pv = ProbeSystem(pos, mat, inv, ff, kinds, np.empty(0), 2)
    pc = ProbeSystem(pos, mat, inv, ff, kinds, q, 0)
    alpha, _ = ewald_alpha()
    cset = grid_setup_with_dims(mat, (n, n, n))
    return Workload(f"synthetic {edge:g} A cube, {natoms} random LJ atoms (+1/-1), {n + 1}^3 grid", fw, ff, cset, pv, pc, alpha)


def count_pair_work(w: Workload, planes: int = 8, stride: int = 4) -> dict:
    """Counted minimum work of a grid build (the flop count behind bench.py's ``roofline``): on a sample of the
    grid points -- ``planes`` x-planes spread over the grid, every ``stride``-th point along y and z -- the exact
    number of framework-atom images inside the cutoff (what compute_derivatives_* accumulates, probes.jl:83,107)
    and of those whose kind has a VdW rule for the probe, per rule class.  Host side, k-d tree over the lattice
    images; because every perpendicular width of a ProbeSystem is >= 2 cutoffs (probes.jl:24) an image inside the
    cutoff is the one the min-image routine selects."""
    from scipy.spatial import cKDTree
    p = w.probe_vdw if w.probe_vdw is not None else w.probe_coulomb
    pos = np.asarray(p.positions, dtype=np.float64)
    mat = np.asarray(p.mat, dtype=np.float64)
    cutoff = float(np.sqrt(p.cutoff2))
    nx, ny, nz = w.cset.npoints
    ii = np.unique(np.linspace(0, nx - 1, planes).round().astype(int))
    jj, kk = np.arange(0, ny, stride), np.arange(0, nz, stride)
    I, J, K = np.meshgrid(ii, jj, kk, indexing="ij")
    pts = np.stack([I * w.cset.size[0] / w.cset.dims[0] + w.cset.shift[0],
                    J * w.cset.size[1] / w.cset.dims[1] + w.cset.shift[1],
                    K * w.cset.size[2] / w.cset.dims[2] + w.cset.shift[2]], axis=-1).reshape(-1, 3)
    lo, hi = pts.min(axis=0) - cutoff, pts.max(axis=0) + cutoff
    shifts = np.array([[a, b, c] for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)], dtype=np.float64) @ mat.T
    img = (pos[None, :, :] + shifts[:, None, :]).reshape(-1, 3)
    idx = np.tile(np.arange(len(pos)), len(shifts))
    keep = np.all((img >= lo) & (img <= hi), axis=1)
    img, idx = img[keep], idx[keep]
    n_in = cKDTree(img).query_ball_point(pts, cutoff, return_length=True)
    out = {"sampled_points": int(len(pts)), "in_cutoff_per_point": float(np.mean(n_in)), "lj_per_point": 0.0,
           "buckingham_per_point": 0.0, "other_vdw_per_point": 0.0}
    if w.probe_vdw is not None:
        from .hostmirror.interactions import FF, rules_of
        ff, probe = w.forcefield, w.probe_vdw.probe
        kinds = np.asarray(w.probe_vdw.atomkinds)
        cls = np.zeros(ff.nkinds + 1, dtype=np.int8)           # 0 none, 1 LJ, 2 Buckingham (+ hard sphere), 3 anything else
        for k in range(1, ff.nkinds + 1):
            ks = [int(r.kind) for r in rules_of(ff.interactions[k - 1][probe - 1])
                  if int(r.kind) not in (int(FF.NoInteraction), int(FF.CoulombEwaldDirect))]
            if not ks:
                continue
            if ks == [int(FF.LennardJones)]:
                cls[k] = 1
            elif int(FF.Buckingham) in ks and set(ks) <= {int(FF.Buckingham), int(FF.HardSphere)}:
                cls[k] = 2
            else:
                cls[k] = 3
        c_img = cls[kinds[idx]]
        for c, name in ((1, "lj_per_point"), (2, "buckingham_per_point"), (3, "other_vdw_per_point")):
            sel = c_img == c
            if sel.any():
                out[name] = float(np.mean(cKDTree(img[sel]).query_ball_point(pts, cutoff, return_length=True)))
    return out
