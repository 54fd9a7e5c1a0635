"""Energy side of the reference's Monte-Carlo set-up (host-side mirror):
``setup_montecarlo`` (montecarlo.jl:70-323, rigid molecules, explicit positions),
``TailCorrection`` (tailcorrection.jl:23-83), ``compute_vdw`` / ``single_contribution_vdw``
(energy.jl:355-427, the no-neighbour-list variants), ``baseline_energy`` / ``movement_energy``
(montecarlo.jl:530-579) and ``single_contribution_ewald`` (ewald.jl:704-738).

The MC driver itself (moves, acceptance, GCMC swaps, outputs) is out of scope; these functions
are the consumers of the grids the HIP kernels build, and what ``test/runtests.jl:186-267`` pins.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .coordinates import GridCoordinatesSetup
from .ewald import EwaldFramework, _structure_factor, ewald_context_constants, initialize_ewald
from .forcefields import ForceField
from ..grids import EnergyGrid, interpolate_grid
from .raspa import RASPASystem, _ff, load_framework_RASPA
from .setup_raspa import decide_parse_block, default_system, grid_locations, retrieve_or_create_grid
from .utils import find_supercell, get_atom_name, perpendicular_lengths


def tail_correction(ff: ForceField, ffidx: List[List[int]], framework_atoms: Sequence[int], lam: float,
                    numspecies: Sequence[int]) -> Tuple[float, List[float], np.ndarray]:
    """``TailCorrection(ff, ffidx, framework_atoms, λ, numspecies)`` tailcorrection.jl:23-83 ->
    (value, framework[m], cross[m, m]) in K; ``lam`` = 2π/V."""
    n = len(framework_atoms)
    assert n == len(ff.sdict)
    allatoms = sorted(set([i + 1 for i, x in enumerate(framework_atoms) if x > 0] + [k for ids in ffidx for k in ids]))
    pairs = np.full((n + 1, n + 1), np.nan)
    for a, i in enumerate(allatoms):
        pairs[i, i] = ff[i, i].tail(ff.cutoff) * lam
        for j in allatoms[a + 1:]:
            pairs[i, j] = pairs[j, i] = ff[i, j].tail(ff.cutoff) * lam
    m = len(numspecies)
    cross = np.full((m, m), np.nan)
    framework = [0.0] * m
    value = 0.0
    for ki, xi in enumerate(framework_atoms):
        if xi == 0:
            continue
        for kj, xj in enumerate(framework_atoms):
            if xj == 0:
                continue
            value += xi * xj * pairs[ki + 1, kj + 1]
    for iA, idsA in enumerate(ffidx):
        f = 0.0
        for k2, x2 in enumerate(framework_atoms):
            if x2 == 0:
                continue
            for k1 in idsA:
                f += pairs[k1, k2 + 1] * x2
        framework[iA] = 2 * f
        numA = numspecies[iA]
        value += 2 * f * numA
        for iB in range(iA, m):
            c = 0.0
            for kA in idsA:
                for kB in ffidx[iB]:
                    c += pairs[kA, kB]
            cross[iA, iB] = cross[iB, iA] = c
            value += (2 - (iA == iB)) * c * numA * numspecies[iB]
    return value, framework, cross


def modify_species_dryrun(framework: Sequence[float], cross: np.ndarray, numspecies: Sequence[int], i: int, num: int) -> float:
    """tailcorrection.jl:89-99: change of the tail correction when ``num`` molecules of kind ``i`` (0-based)
    are added (negative: removed) to a system holding ``numspecies``."""
    diff = framework[i]
    for j, other in enumerate(numspecies):
        if j == i:
            diff += (num + 2 * other) * cross[i, i]
        else:
            diff += 2 * other * cross[j, i]
    return diff * num


@dataclass
class BaselineEnergyReport:
    """montecarlo.jl:447-456"""
    framework_vdw: float
    framework_direct: float
    inter: float
    reciprocal: float
    tailcorrection: float

    def __float__(self) -> float:
        return float(self.framework_vdw + self.framework_direct + self.inter + self.reciprocal + self.tailcorrection)


@dataclass
class MCEnergyReport:
    """energy.jl (MCEnergyReport): framework (vdw, direct) + inter + reciprocal"""
    framework_vdw: float
    framework_direct: float
    inter: float
    reciprocal: float

    def __float__(self) -> float:
        return float(self.framework_vdw + self.framework_direct + self.inter + self.reciprocal)

    def __sub__(self, o: "MCEnergyReport") -> "MCEnergyReport":
        return MCEnergyReport(self.framework_vdw - o.framework_vdw, self.framework_direct - o.framework_direct,
                              self.inter - o.inter, self.reciprocal - o.reciprocal)


@dataclass
class MonteCarloSetup:
    """The energy-relevant fields of the reference's ``MonteCarloSetup`` / ``SimulationStep``."""
    ff: ForceField
    mat: np.ndarray                       # MC cell = supercell matrix (columns)
    invmat: np.ndarray
    ffidx: List[List[int]]                # per kind: 1-based ff index of each atom
    charges: np.ndarray                   # per 1-based ff index (entry 0 unused), NaN when never met
    positions: List[List[np.ndarray]]     # [kind][molecule] -> float64[natoms, 3]
    ewald: EwaldFramework
    coulomb: EnergyGrid
    grids: List[Optional[EnergyGrid]]     # indexed by ff index - 1; empty list = no framework
    tailcorrection: float
    tail_framework: List[float] = field(default_factory=list)
    tail_cross: Optional[np.ndarray] = None
    sums: Optional[np.ndarray] = None     # complex[num_kvecs, 1 + nmolecules], set by baseline_energy
    models: List[np.ndarray] = field(default_factory=list)   # per kind: atom positions of the model molecule (mc.models)

    # ---- flat views
    def molecules(self):
        """(kind, index-in-kind, ff indices, positions) in the reference's flat order."""
        for i, kind in enumerate(self.positions):
            for j, pos in enumerate(kind):
                yield i, j, self.ffidx[i], pos

    def flat_index(self, i: int, j: int) -> int:
        return sum(len(k) for k in self.positions[:i]) + j


def _wrap_d2(d: np.ndarray, mat: np.ndarray, invmat: np.ndarray) -> float:
    """unsafe_periodic_distance2! (utils.jl:294-302): wrap to the nearest lattice image, no image search."""
    f = invmat @ d
    f = (f + 0.5) - np.floor(f + 0.5) - 0.5
    v = mat @ f
    return float(v @ v)


def framework_interactions(mc: MonteCarloSetup, indices: Sequence[int], positions) -> Tuple[float, float]:
    """montecarlo.jl:490-504 -> (vdw, direct) in K."""
    if not mc.grids:
        return 0.0, 0.0
    vdw = direct = 0.0
    hascoulomb = mc.coulomb.ewald_precision != -math.inf
    for k, pos in enumerate(positions):
        ix = indices[k]
        vdw += interpolate_grid(mc.grids[ix - 1], pos)
        if hascoulomb:
            c = interpolate_grid(mc.coulomb, pos)
            direct += c if c == 1e100 else float(mc.charges[ix]) * c
    return vdw, direct


def compute_vdw(mc: MonteCarloSetup) -> float:
    """compute_vdw_noneighbour (energy.jl:355-383): every pair of atoms of different molecules
    within the cutoff, wrapped to the nearest image of the MC cell."""
    cutoff2 = mc.ff.cutoff ** 2
    flat = [(i, j, ids[k], pos[k]) for i, j, ids, pos in mc.molecules() for k in range(len(ids))]
    energy = 0.0
    for l1, (i1, j1, ix1, p1) in enumerate(flat):
        for (i2, j2, ix2, p2) in flat[l1 + 1:]:
            if i1 == i2 and j1 == j2:
                continue
            d2 = _wrap_d2(p2 - p1, mc.mat, mc.invmat)
            if d2 < cutoff2:
                energy += mc.ff[ix1, ix2].at_r2(d2)
    return energy


def single_contribution_vdw(mc: MonteCarloSetup, idx: Tuple[int, int], poss2) -> float:
    """single_contribution_vdw_noneighbour (energy.jl:407-427) for a rigid molecule."""
    i2, j2 = idx
    cutoff2 = mc.ff.cutoff ** 2
    energy = 0.0
    for k2, pos2 in enumerate(poss2):
        ix2 = mc.ffidx[i2][k2]
        for i1, j1, ids, pos in mc.molecules():
            if i1 == i2 and j1 == j2:
                continue
            for k1 in range(len(ids)):
                d2 = _wrap_d2(np.asarray(pos2, dtype=np.float64) - pos[k1], mc.mat, mc.invmat)
                if d2 < cutoff2:
                    energy += mc.ff[ids[k1], ix2].at_r2(d2)
    return energy


def _molecule_sf(mc: MonteCarloSetup, ids: Sequence[int], pos) -> np.ndarray:
    ef = mc.ewald
    q = np.array([mc.charges[ix] for ix in ids], dtype=np.float64)
    frac = np.asarray(pos, dtype=np.float64).reshape(len(ids), 3) @ ef.invmat.T
    return _structure_factor(ef.kvec_ijk, frac, q)


def _ewald_systems(mc: MonteCarloSetup):
    out = []
    for i, kind in enumerate(mc.positions):
        q = np.array([mc.charges[ix] for ix in mc.ffidx[i]], dtype=np.float64)
        out.append([RASPASystem(mc.mat, p, [""] * len(q), np.zeros(len(q)), q, True) for p in kind])
    return out


def compute_ewald_mc(mc: MonteCarloSetup) -> float:
    """compute_ewald(::IncrementalEwaldContext) (ewald.jl:630-652); keeps the per-molecule structure
    factors (``sums``: column 0 the total, then one column per molecule) for single_contribution_ewald."""
    ef = mc.ewald
    if ef.alpha == 0.0:
        return 0.0
    mols = list(mc.molecules())
    sums = np.zeros((len(ef.kfactors), 1 + len(mols)), dtype=np.complex128)
    for m, (i, j, ids, pos) in enumerate(mols):
        sums[:, 1 + m] = _molecule_sf(mc, ids, pos)
    sums[:, 0] = sums[:, 1:].sum(axis=1)
    mc.sums = sums
    enc, static = ewald_context_constants(ef, [k for k in _ewald_systems(mc) if k])
    f, a = ef.StoreRigidChargeFramework, sums[:, 0]
    framework_adsorbate = float((ef.kfactors * (f.real * a.real + f.imag * a.imag)).sum())
    adsorbate_adsorbate = float((ef.kfactors * (a.real ** 2 + a.imag ** 2)).sum())
    return 2 * (framework_adsorbate + enc) + adsorbate_adsorbate + static


def ewald_rest(mc: MonteCarloSetup, idx: Optional[Tuple[int, int]]) -> np.ndarray:
    """Structure factor of everything but molecule ``idx``: framework + all guests - that guest
    (``rest`` of ewald.jl:722-728).  ``idx = None``: a molecule not in the system yet."""
    assert mc.sums is not None, "Please call baseline_energy(mc) before single_contribution_ewald"
    rest = mc.ewald.StoreRigidChargeFramework + mc.sums[:, 0]
    if idx is not None:
        rest = rest - mc.sums[:, 1 + mc.flat_index(*idx)]
    return rest


def single_contribution_ewald(mc: MonteCarloSetup, idx: Tuple[int, int], positions=None, new: bool = False) -> float:
    """ewald.jl:704-738: 2 Σ kf Re(conj(rest) S) + Σ kf |S|², S the structure factor of the molecule
    at ``positions`` (or where it currently is)."""
    ef = mc.ewald
    if ef.alpha == 0.0:
        return 0.0
    rest = ewald_rest(mc, None if new else idx)
    if positions is None:
        S = mc.sums[:, 1 + mc.flat_index(*idx)]
    else:
        S = _molecule_sf(mc, mc.ffidx[idx[0]], positions)
    rest_single = float((ef.kfactors * (rest.real * S.real + rest.imag * S.imag)).sum())
    single_single = float((ef.kfactors * (S.real ** 2 + S.imag ** 2)).sum())
    return 2 * rest_single + single_single


def baseline_energy(mc: MonteCarloSetup) -> BaselineEnergyReport:
    """montecarlo.jl:530-542"""
    reciprocal = compute_ewald_mc(mc)
    vdw = compute_vdw(mc)
    fv = fd = 0.0
    for i, j, ids, pos in mc.molecules():
        a, b = framework_interactions(mc, ids, pos)
        fv += a
        fd += b
    return BaselineEnergyReport(fv, fd, vdw, reciprocal, mc.tailcorrection)


def movement_energy(mc: MonteCarloSetup, idx: Tuple[int, int], positions=None) -> MCEnergyReport:
    """montecarlo.jl:563-579 (``idx`` 0-based (kind, molecule))."""
    i, j = idx
    poss = mc.positions[i][j] if positions is None else np.asarray(positions, dtype=np.float64).reshape(-1, 3)
    rec = single_contribution_ewald(mc, idx, None if positions is None else poss)
    fv, fd = framework_interactions(mc, mc.ffidx[i], poss)
    return MCEnergyReport(fv, fd, single_contribution_vdw(mc, idx, poss), rec)


def update_mc(mc: MonteCarloSetup, idx: Tuple[int, int], positions) -> None:
    """update_mc! for a displacement (montecarlo.jl:615-628) with update_ewald_context! (ewald.jl:757-773): molecule ``idx``
    now sits at ``positions``; sums[:, 1] += new - sums[:, ij+1]; sums[:, ij+1] = new."""
    i, j = idx
    pos = np.array(positions, dtype=np.float64).reshape(-1, 3)
    if mc.ewald.alpha != 0.0:
        assert mc.sums is not None, "Please call baseline_energy(mc) first"
        col = 1 + mc.flat_index(i, j)
        new = _molecule_sf(mc, mc.ffidx[i], pos)
        mc.sums[:, 0] += new - mc.sums[:, col]
        mc.sums[:, col] = new
    mc.positions[i][j] = pos


def insertion_energy(mc: MonteCarloSetup, i: int, positions) -> MCEnergyReport:
    """movement_energy(mc, (i, length + 1), positions) (montecarlo.jl:563-579 with ``ij = -i``): a molecule of kind ``i`` that is
    not in the system yet; no tail-correction change included."""
    poss = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
    idx = (i, len(mc.positions[i]))
    rec = single_contribution_ewald(mc, idx, poss, new=True)
    fv, fd = framework_interactions(mc, mc.ffidx[i], poss)
    return MCEnergyReport(fv, fd, single_contribution_vdw(mc, idx, poss), rec)


def add_molecule(mc: MonteCarloSetup, i: int, positions=None) -> int:
    """add_one_system! (montecarlo.jl:835-861, ewald.jl:775-792): append a molecule of kind ``i``; returns its index in the kind;
    the tail correction follows the new species count."""
    pos = np.array(mc.models[i] if positions is None else positions, dtype=np.float64).reshape(-1, 3)    # montecarlo.jl:844
    j = len(mc.positions[i])
    if mc.ewald.alpha != 0.0 and mc.sums is not None:         # an uninitialised Ewald state stays uninitialised
        col = 1 + mc.flat_index(i, j)
        new = _molecule_sf(mc, mc.ffidx[i], pos)
        mc.sums = np.insert(mc.sums, col, new, axis=1)
        mc.sums[:, 0] += new
    if mc.tail_cross is not None:                                  # modify_species!(mc.tailcorrection, i, 1), montecarlo.jl:858
        mc.tailcorrection += modify_species_dryrun(mc.tail_framework, mc.tail_cross, [len(k) for k in mc.positions], i, 1)
    mc.positions[i].append(pos)
    return j


def remove_molecule(mc: MonteCarloSetup, idx: Tuple[int, int]) -> int:
    """remove_one_system!(mc, i, j) (montecarlo.jl:756-817, ewald.jl:794-810): the molecule leaves and the LAST molecule of its
    kind takes index ``j`` (montecarlo.jl:798-808); returns that molecule's old index (== ``j`` when it was the last one),
    like the reference's ``lastj``."""
    i, j = idx
    last = len(mc.positions[i]) - 1
    if mc.ewald.alpha != 0.0 and mc.sums is not None:        # an uninitialised Ewald state stays uninitialised (ewald.jl:794-797)
        col, col_last = 1 + mc.flat_index(i, j), 1 + mc.flat_index(i, last)
        mc.sums[:, 0] -= mc.sums[:, col]
        mc.sums[:, col] = mc.sums[:, col_last]
        mc.sums = np.delete(mc.sums, col_last, axis=1)
    if mc.tail_cross is not None:                                  # modify_species!(mc.tailcorrection, i, -1), montecarlo.jl:814
        mc.tailcorrection += modify_species_dryrun(mc.tail_framework, mc.tail_cross, [len(k) for k in mc.positions], i, -1)
    mc.positions[i][j] = mc.positions[i][last]
    mc.positions[i].pop()
    return last


def setup_montecarlo(framework, pff, systems: Sequence[RASPASystem], *, blockfiles=None, gridstep: float = 0.15,
                     supercell=None, new: bool = False, cutoff: float = 12.0, ngpus: int = 1) -> MonteCarloSetup:
    """montecarlo.jl:266-323 + :70-216 for explicit rigid molecules (one entry of ``systems`` per
    molecule; entries with the same atom symbols form one kind).  ``framework`` is a RASPA name, or a
    3x3 matrix for an empty cell.  Blocking spheres do not enter any energy and are not parsed here."""
    is_void = isinstance(framework, np.ndarray)
    ff = _ff(pff, cutoff=cutoff)
    if is_void:
        syst_framework = RASPASystem(np.array(framework, dtype=np.float64), np.empty((0, 3)), [], np.empty(0), np.empty(0))
    else:
        syst_framework = load_framework_RASPA(framework, pff)
    mat = syst_framework.mat
    if supercell is None:
        supercell = find_supercell(mat, cutoff)
    cellmat = mat * np.asarray(supercell, dtype=np.float64)[None, :]
    if np.any(perpendicular_lengths(cellmat) <= 24.0):
        raise ValueError("The current cell has at least one perpendicular length lower than 24.0Å: please use a larger supercell")

    # a system is a molecule, or a pair (molecule, n): n copies at the model's positions (n = 0: the kind exists, empty;
    # montecarlo.jl:234-236)
    pairs = [(default_system(s[0] if isinstance(s, tuple) else s, pff), int(s[1]) if isinstance(s, tuple) else 1) for s in systems]
    mols = [s for s, _n in pairs]
    kinds: List[Tuple[str, ...]] = []
    positions: List[List[np.ndarray]] = []
    models: List[RASPASystem] = []
    for s, n in pairs:
        key = tuple(s.atomic_symbol)
        if key not in kinds:
            kinds.append(key)
            positions.append([])
            models.append(s)
        for _ in range(n):
            positions[kinds.index(key)].append(np.array(s.position, dtype=np.float64).reshape(-1, 3))
    ffidx = [[ff.sdict[get_atom_name(a)] if a not in ff.sdict else ff.sdict[a] for a in key] for key in kinds]
    charges = np.full(len(ff.sdict) + 1, np.nan)
    for ids, model in zip(ffidx, models):
        for k, ix in enumerate(ids):
            charges[ix] = model.atomic_charge[k]

    needcoulomb = any(q != 0 for s in mols for q in s.atomic_charge)
    atoms = sorted({(a, ff.sdict[a]) for key in kinds for a in key}, key=lambda t: t[1])
    if is_void:
        coulomb_grid_path, vdw_grid_paths = "", []
    else:
        coulomb_grid_path, vdw_grid_paths = grid_locations(framework, pff, ff, [a for a, _ in atoms], gridstep, supercell)
    if needcoulomb:
        ewald = initialize_ewald(syst_framework, supercell)
        coulomb = retrieve_or_create_grid(coulomb_grid_path, syst_framework, ff, gridstep, ewald, mat, new, cutoff, ngpus)
    else:
        coulomb, ewald = EnergyGrid.trivial(True), EwaldFramework.empty(mat)
    grids: List[Optional[EnergyGrid]] = []
    if vdw_grid_paths and len(syst_framework) > 0:
        grids = [None] * len(ff.sdict)
        for path, (atom, i) in zip(vdw_grid_paths, atoms):
            grids[i - 1] = retrieve_or_create_grid(path, syst_framework, ff, gridstep, atom, mat, new, cutoff, ngpus)

    num_framework_atoms = [0] * len(ff.sdict)
    PI = int(np.prod(supercell))
    for sym in syst_framework.atomic_symbol:
        num_framework_atoms[ff.sdict[get_atom_name(sym)] - 1] += PI
    lam = 2 * math.pi / float(np.linalg.det(cellmat))
    value, tframework, tcross = tail_correction(ff, ffidx, num_framework_atoms, lam, [len(p) for p in positions])
    return MonteCarloSetup(ff, cellmat, np.linalg.inv(cellmat), ffidx, charges, positions, ewald, coulomb, grids, value,
                           tframework, tcross, models=[np.array(m.position, dtype=np.float64).reshape(-1, 3) for m in models])
