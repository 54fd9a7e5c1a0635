"""``movement_energy`` / ``update_mc!`` of the reference (BASELINE config 5) as the CHECKER computes them.

TEST INFRASTRUCTURE ONLY.  Composed from the C restatements of oracle/ (``oracle_interpolate_grid``,
``oracle_single_contribution_vdw``, ``oracle_molecule_sums``, ``oracle_single_contribution_ewald``); nothing here imports
the product package.  The GPU tests of ``ceg_mc_*`` (tests/test_gpu_consumers.py) compare the device-resident state with
THIS state, move by move.

Restated (file:line under /root/reference):
  * ``movement_energy``            src/montecarlo.jl:563-579 (the three terms; ``ij = -i`` for a molecule not in the system)
  * ``framework_interactions``     src/montecarlo.jl:490-504
  * ``single_contribution_vdw``    src/energy.jl:407-427 (C)
  * ``IncrementalEwaldContext``    src/ewald.jl:584-652 (``sums[:, 1]`` = total, ``sums[:, ij+1]`` per molecule)
  * ``single_contribution_ewald``  src/ewald.jl:704-738 (C)
  * ``update_ewald_context!``      src/ewald.jl:757-773; ``add_one_system!`` :775-792; ``remove_one_system!`` :794-810
  * ``update_mc!``                 src/montecarlo.jl:615-628; removal renumbering src/montecarlo.jl:798-808
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import hostlogic as H
from . import oracle as O

_dp = C.POINTER(C.c_double)


def _d(a):
    return a.ctypes.data_as(_dp)


class OracleMonteCarlo:
    """Energy-relevant state of a ``MonteCarloSetup`` for rigid molecules.

    ``ffidx[i]``: 1-based force-field index of every atom of kind i; ``charges[ix]``: charge per 1-based ff index;
    ``positions[i][j]``: float64[natoms, 3]; ``grids[ix - 1]``: EnergyGrid-like objects (values in K) or an empty list
    (no framework); ``coulomb``: the Coulomb EnergyGrid; ``ef``: an OracleEwaldFramework; ``rules, offsets``: the
    nkinds^2 pair table (0-based kinds)."""

    def __init__(self, mat, cutoff: float, rules, offsets, nkinds: int, ffidx, charges, positions, grids, coulomb,
                 ef: H.OracleEwaldFramework):
        self.mat = np.array(mat, dtype=np.float64)
        self.invmat = np.linalg.inv(self.mat)
        self.cutoff2 = float(cutoff) ** 2
        self.rules, self.offsets, self.nkinds = rules, np.ascontiguousarray(offsets, dtype=np.int32), int(nkinds)
        self.ffidx = [list(ids) for ids in ffidx]
        self.charges = np.array(charges, dtype=np.float64)
        self.positions: List[List[np.ndarray]] = [[np.array(p, dtype=np.float64).reshape(-1, 3) for p in kind] for kind in positions]
        self.grids = list(grids)
        self.coulomb = coulomb
        self.ef = ef
        self.sums_re: Optional[np.ndarray] = None      # [num_kvecs, 1 + nmolecules]; column 0 = total (Julia's sums[:, 1])
        self.sums_im: Optional[np.ndarray] = None

    # ------------------------------------------------------------------ adapters
    @classmethod
    def from_setup(cls, mc) -> "OracleMonteCarlo":
        """INPUT ADAPTER: copy the plain data out of a MonteCarloSetup-like object (attribute access only; the parsed force
        field's pair table, the grids the library built and the EwaldFramework arrays are inputs of the energy functions)."""
        rules, offsets = mc.ff.pair_table()
        return cls(mc.mat, mc.ff.cutoff, rules, offsets, mc.ff.nkinds, mc.ffidx, mc.charges, mc.positions, mc.grids, mc.coulomb,
                   H.adapt_ewald_framework(mc.ewald))

    # ------------------------------------------------------------------ flat order
    def molecules(self):
        for i, kind in enumerate(self.positions):
            for j, pos in enumerate(kind):
                yield i, j, pos

    def flat_index(self, i: int, j: int) -> int:
        return sum(len(k) for k in self.positions[:i]) + j

    def _mol_charges(self, i: int) -> np.ndarray:
        return np.array([self.charges[ix] for ix in self.ffidx[i]], dtype=np.float64)

    @property
    def has_ewald(self) -> bool:
        return self.ef.alpha != 0.0

    # ------------------------------------------------------------------ Ewald state
    def compute_ewald(self) -> float:
        """compute_ewald(::IncrementalEwaldContext) (ewald.jl:630-652): fills the per-molecule sums; the total column is the
        left-to-right sum over the molecule columns (ewald_main_loop! :177-182)."""
        if not self.has_ewald:
            return 0.0
        mols = list(self.molecules())
        nk = self.ef.num_kvecs
        self.sums_re = np.zeros((nk, 1 + len(mols)))
        self.sums_im = np.zeros((nk, 1 + len(mols)))
        for m, (i, _j, pos) in enumerate(mols):
            re, im = H.molecule_sums(self.ef, pos, self._mol_charges(i))
            self.sums_re[:, 1 + m] = re
            self.sums_im[:, 1 + m] = im
        for m in range(len(mols)):                                   # sum(@view sums[_i, :]) in column order
            self.sums_re[:, 0] += self.sums_re[:, 1 + m]
            self.sums_im[:, 0] += self.sums_im[:, 1 + m]
        kinds = [(self._mol_charges(i), kind[0], len(kind)) for i, kind in enumerate(self.positions) if kind]
        enc, static = H.ewald_context_constants(self.ef, kinds)
        tr, ti = np.ascontiguousarray(self.sums_re[:, 0]), np.ascontiguousarray(self.sums_im[:, 0])
        return float(H._lib().oracle_compute_ewald_total(_d(self.ef.kfactors), nk, _d(self.ef.sf_re), _d(self.ef.sf_im), _d(tr), _d(ti), enc, static))

    def single_contribution_ewald(self, i: int, j: Optional[int], positions=None) -> float:
        """ewald.jl:704-738; ``j is None``: a molecule of kind i that is not in the system (``ij < 0``)."""
        if not self.has_ewald:
            return 0.0
        assert self.sums_re is not None, "Please call compute_ewald() before single_contribution_ewald"
        nk = self.ef.num_kvecs
        tr, ti = np.ascontiguousarray(self.sums_re[:, 0]), np.ascontiguousarray(self.sums_im[:, 0])
        if j is None:
            own_r = own_i = None
        else:
            col = 1 + self.flat_index(i, j)
            own_r, own_i = np.ascontiguousarray(self.sums_re[:, col]), np.ascontiguousarray(self.sums_im[:, col])
        if positions is None:
            assert j is not None
            sr, si = own_r, own_i
        else:
            sr, si = H.molecule_sums(self.ef, positions, self._mol_charges(i))
        return float(H._lib().oracle_single_contribution_ewald(_d(self.ef.kfactors), nk, _d(self.ef.sf_re), _d(self.ef.sf_im), _d(tr), _d(ti),
                                                               _d(own_r) if own_r is not None else None,
                                                               _d(own_i) if own_i is not None else None, _d(sr), _d(si)))

    # ------------------------------------------------------------------ framework and guest-guest terms
    def framework_interactions(self, i: int, positions) -> Tuple[float, float]:
        """montecarlo.jl:490-504 -> (vdw, direct): per atom, in order, interpolate_grid of its VdW grid and charge x the
        Coulomb grid (the 1e100 blocking value is added as is)."""
        if not self.grids:
            return 0.0, 0.0
        pos = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
        vdw = direct = 0.0
        hascoulomb = self.coulomb.ewald_precision != -math.inf
        for k, p in enumerate(pos):
            ix = self.ffidx[i][k]
            vdw += float(O.interpolate_points(self.grids[ix - 1], p[None, :], nthreads=1)[0])
            if hascoulomb:
                c = float(O.interpolate_points(self.coulomb, p[None, :], nthreads=1)[0])
                direct += c if c == 1e100 else float(self.charges[ix]) * c
        return vdw, direct

    def single_contribution_vdw(self, i: int, j: Optional[int], positions) -> float:
        """energy.jl:407-427 against the atoms of every OTHER molecule (``j is None``: nothing excluded)."""
        pos, kinds, mol = [], [], []
        for m, (ki, _kj, p) in enumerate(self.molecules()):
            pos.append(p)
            kinds += [ix - 1 for ix in self.ffidx[ki]]
            mol += [m] * len(p)
        if not pos:
            return 0.0
        allpos = np.concatenate(pos)
        tk = [ix - 1 for ix in self.ffidx[i]]
        exclude = -1 if j is None else self.flat_index(i, j)
        trial = np.asarray(positions, dtype=np.float64).reshape(1, len(tk), 3)
        return float(O.single_contribution_vdw_raw(self.mat, self.invmat, self.cutoff2, self.rules, self.offsets, self.nkinds,
                                                   H.COULOMBIC_CONVERSION_FACTOR, allpos, kinds, mol, trial, tk, exclude, nthreads=1)[0])

    # ------------------------------------------------------------------ movement_energy and the updates
    def movement_energy(self, idx: Tuple[int, int], positions=None) -> np.ndarray:
        """montecarlo.jl:563-579 -> [framework vdw, framework direct, guest-guest, reciprocal] in K; ``idx`` 0-based
        (kind, molecule)."""
        i, j = idx
        poss = self.positions[i][j] if positions is None else np.asarray(positions, dtype=np.float64).reshape(-1, 3)
        rec = self.single_contribution_ewald(i, j, None if positions is None else poss)
        fv, fd = self.framework_interactions(i, poss)
        return np.array([fv, fd, self.single_contribution_vdw(i, j, poss), rec])

    def insertion_energy(self, i: int, positions) -> np.ndarray:
        """movement_energy(mc, (i, length + 1), positions): ``ij = -i`` (montecarlo.jl:565, ewald.jl:722-724)."""
        poss = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
        rec = self.single_contribution_ewald(i, None, poss)
        fv, fd = self.framework_interactions(i, poss)
        return np.array([fv, fd, self.single_contribution_vdw(i, None, poss), rec])

    def update(self, idx: Tuple[int, int], positions) -> None:
        """update_mc! for a displacement (montecarlo.jl:615-628) + update_ewald_context! (ewald.jl:757-773):
        sums[:, 1] += tmpsums - sums[:, ij+1]; sums[:, ij+1] = tmpsums."""
        i, j = idx
        pos = np.array(positions, dtype=np.float64).reshape(-1, 3)
        if self.has_ewald and self.sums_re is not None:
            col = 1 + self.flat_index(i, j)
            re, im = H.molecule_sums(self.ef, pos, self._mol_charges(i))
            self.sums_re[:, 0] += re - self.sums_re[:, col]
            self.sums_im[:, 0] += im - self.sums_im[:, col]
            self.sums_re[:, col] = re
            self.sums_im[:, col] = im
        self.positions[i][j] = pos

    def add(self, i: int, positions) -> int:
        """add_one_system! (ewald.jl:775-792): the new molecule becomes the last of its kind; sums[:, 1] += its sums.
        (The reference appends the new column at the END of ``sums`` and keeps a flat index table; this state keeps the columns in
        (kind, molecule) order instead -- the columns are the same set, and every energy reads them by molecule.)"""
        pos = np.array(positions, dtype=np.float64).reshape(-1, 3)
        j = len(self.positions[i])
        if self.has_ewald and self.sums_re is not None:
            col = 1 + self.flat_index(i, j)
            re, im = H.molecule_sums(self.ef, pos, self._mol_charges(i))
            self.sums_re = np.insert(self.sums_re, col, re, axis=1)
            self.sums_im = np.insert(self.sums_im, col, im, axis=1)
            self.sums_re[:, 0] += re
            self.sums_im[:, 0] += im
        self.positions[i].append(pos)
        return j

    def remove(self, idx: Tuple[int, int]) -> int:
        """remove_one_system! (ewald.jl:794-810, montecarlo.jl:798-808): sums[:, 1] -= sums[:, ij+1]; the LAST molecule of
        the kind takes index j; returns that molecule's old index."""
        i, j = idx
        last = len(self.positions[i]) - 1
        if self.has_ewald and self.sums_re is not None:
            col, col_last = 1 + self.flat_index(i, j), 1 + self.flat_index(i, last)
            for a in (self.sums_re, self.sums_im):
                a[:, 0] -= a[:, col]
                a[:, col] = a[:, col_last]
            self.sums_re = np.delete(self.sums_re, col_last, axis=1)
            self.sums_im = np.delete(self.sums_im, col_last, axis=1)
        self.positions[i][j] = self.positions[i][last]
        self.positions[i].pop()
        return last

    def total_structure_factor(self) -> np.ndarray:
        return self.sums_re[:, 0] + 1j * self.sums_im[:, 0]

    def flat_positions(self) -> np.ndarray:
        return np.concatenate([p for _i, _j, p in self.molecules()])
