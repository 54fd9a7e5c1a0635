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
from .hostmirror.ewald import EwaldFramework, ewald_context_constants
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


class PairEnergies:
    """Device-resident guest atoms + pair table (``ceg_pairs_*``): batched single_contribution_vdw."""

    def __init__(self, ff, mat, invmat, device: int = 0):
        from .hostmirror.constants import COULOMBIC_CONVERSION_FACTOR
        self._lib = _abi.load_library()
        self.ff = ff
        rules, offsets = ff.pair_table()
        self._keep = (rules, offsets)
        h = C.c_void_p()
        rc = self._lib.ceg_pairs_create(C.byref(h), device, _abi.dptr(_matT(mat)), _abi.dptr(_matT(invmat)), ff.cutoff ** 2,
                                        rules.ctypes.data, _abi.i32ptr(offsets), ff.nkinds, COULOMBIC_CONVERSION_FACTOR)
        _abi.check(self._lib, rc)
        self._h = h

    def set_atoms(self, positions, kinds, molecule) -> None:
        """positions[N,3]; kinds 1-based ff indices; molecule ids (non-negative)."""
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
        k = np.ascontiguousarray(np.asarray(kinds, dtype=np.int32) - 1)
        mol = np.ascontiguousarray(molecule, dtype=np.int32)
        _abi.check(self._lib, self._lib.ceg_pairs_set_atoms(self._h, _abi.dptr(pos.reshape(-1)), _abi.i32ptr(k), _abi.i32ptr(mol), len(pos)))

    def energies(self, trial, trial_kinds, exclude_molecule: int = -1) -> np.ndarray:
        tk = np.ascontiguousarray(np.asarray(trial_kinds, dtype=np.int32) - 1)
        t = np.ascontiguousarray(trial, dtype=np.float64).reshape(-1, len(tk), 3)
        out = np.empty(len(t), dtype=np.float64)
        _abi.check(self._lib, self._lib.ceg_pairs_energy(self._h, _abi.dptr(t.reshape(-1)), _abi.i32ptr(tk), len(tk), len(t),
                                                         exclude_molecule, _abi.dptr(out)))
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ceg_pairs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuMonteCarloEnergy:
    """``movement_energy`` (montecarlo.jl:563-579) of a :class:`ceg_hip.hostmirror.montecarlo.MonteCarloSetup` for
    many trial placements at once: framework terms by batched grid interpolation, guest-guest terms by
    ``ceg_pairs_*``, reciprocal term by ``ceg_recip_*`` against the structure factor of everything else."""

    def __init__(self, mc, device: int = 0):
        from .hostmirror import montecarlo as M
        self.mc, self._M = mc, M
        self.interp = [GridInterpolator(g, device) if (g is not None and g.ewald_precision == math.inf) else None for g in mc.grids]
        self.has_coulomb = mc.coulomb.ewald_precision != -math.inf
        self.coulomb = GridInterpolator(mc.coulomb, device) if self.has_coulomb else None
        self.recip = ReciprocalEwald(mc.ewald, device) if mc.ewald.alpha != 0.0 else None
        self.pairs = PairEnergies(mc.ff, mc.mat, mc.invmat, device)
        self.refresh()

    def refresh(self) -> None:
        """Upload the current guest atoms (call after the host-side state changed)."""
        mc = self.mc
        pos, kinds, mol = [], [], []
        for m, (i, j, ids, p) in enumerate(mc.molecules()):
            pos.append(p)
            kinds += list(ids)
            mol += [m] * len(ids)
        self.pairs.set_atoms(np.concatenate(pos) if pos else np.empty((0, 3)), kinds, mol)

    def movement_energies(self, idx, positions) -> np.ndarray:
        """-> float64[n, 4]: (framework vdw, framework direct, inter, reciprocal) of molecule ``idx``
        (0-based (kind, molecule)) at each of ``positions[n, natoms, 3]``."""
        mc, M = self.mc, self._M
        i, j = idx
        ids = mc.ffidx[i]
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, len(ids), 3)
        out = np.zeros((len(pos), 4))
        if mc.grids:
            for a, ix in enumerate(ids):
                it = self.interp[ix - 1]
                if it is not None:
                    out[:, 0] += it(pos[:, a])
                if self.has_coulomb:
                    c = self.coulomb(pos[:, a])
                    out[:, 1] += np.where(c == 1e100, c, float(mc.charges[ix]) * c)
        out[:, 2] = self.pairs.energies(pos, ids, exclude_molecule=mc.flat_index(i, j))
        if self.recip is not None:
            rest = M.ewald_rest(mc, idx)
            lib = self.recip._lib
            _abi.check(lib, lib.ceg_recip_set_structure_factor(self.recip._h, _abi.dptr(np.ascontiguousarray(rest.real)),
                                                               _abi.dptr(np.ascontiguousarray(rest.imag))))
            q = np.ascontiguousarray([mc.charges[ix] for ix in ids], dtype=np.float64)
            rec = np.empty(len(pos))
            _abi.check(lib, lib.ceg_recip_energy(self.recip._h, _abi.dptr(pos.reshape(-1)), _abi.dptr(q), len(q), len(pos), 0.0, 0.0,
                                                 _abi.dptr(rec)))
            out[:, 3] = rec
        return out

    def baseline_energy(self):
        """baseline_energy (montecarlo.jl:530-542) with the framework and guest-guest sums on the GPU
        (each pair is seen from both molecules, hence the 1/2) and the many-molecule reciprocal sum on
        the host."""
        mc, M = self.mc, self._M
        reciprocal = M.compute_ewald_mc(mc)
        fv = fd = inter = 0.0
        for i, j, ids, p in mc.molecules():
            e = self.movement_energies((i, j), p[None])[0]
            fv += e[0]; fd += e[1]; inter += e[2]
        return M.BaselineEnergyReport(fv, fd, 0.5 * inter, reciprocal, mc.tailcorrection)

    def close(self) -> None:
        for it in self.interp:
            if it is not None:
                it.close()
        if self.coulomb is not None:
            self.coulomb.close()
        if self.recip is not None:
            self.recip.close()
        self.pairs.close()


class DeviceMonteCarlo:
    """Device-resident energy state of a :class:`ceg_hip.hostmirror.montecarlo.MonteCarloSetup` (``ceg_mc_*``, BASELINE config 5):
    ``movement_energy`` (montecarlo.jl:563-579) of a batch of trial placements in ONE launch, ``update_mc!``
    (montecarlo.jl:615-628) applied on the device.  The MC driver (proposals, acceptance) stays with the caller."""

    def __init__(self, mc, device: int = 0):
        from .hostmirror.constants import COULOMBIC_CONVERSION_FACTOR
        self._lib = _abi.load_library()
        self.mc = mc
        ff = mc.ff
        nk = ff.nkinds
        self.interp = [GridInterpolator(g, device) if (g is not None and g.ewald_precision == math.inf) else None
                       for g in (mc.grids if mc.grids else [None] * nk)]
        has_coulomb = bool(mc.grids) and mc.coulomb.ewald_precision != -math.inf
        self.coulomb = GridInterpolator(mc.coulomb, device) if has_coulomb else None
        handles = (C.c_void_p * nk)(*[it._h if it is not None else None for it in self.interp])
        charge = np.ascontiguousarray([0.0 if (k + 1 >= len(mc.charges) or np.isnan(mc.charges[k + 1])) else float(mc.charges[k + 1])
                                       for k in range(nk)], dtype=np.float64)
        rules, offsets = ff.pair_table()
        self._keep = (rules, offsets, handles, charge)
        ef = mc.ewald
        if ef.alpha != 0.0:
            ijk = np.ascontiguousarray(ef.kvec_ijk, dtype=np.int32).reshape(-1)
            kf = np.ascontiguousarray(ef.kfactors, dtype=np.float64)
            re = np.ascontiguousarray(ef.StoreRigidChargeFramework.real, dtype=np.float64)
            im = np.ascontiguousarray(ef.StoreRigidChargeFramework.imag, dtype=np.float64)
            ks = np.asarray(ef.kspace.ks, dtype=np.int32)
            einv = _matT(ef.invmat)
            kargs = (_abi.i32ptr(ijk), _abi.dptr(kf), _abi.dptr(re), _abi.dptr(im), len(kf), _abi.i32ptr(ks), _abi.dptr(einv))
        else:
            kargs = (None, None, None, None, 0, None, None)
        h = C.c_void_p()
        rc = self._lib.ceg_mc_create(C.byref(h), device, handles, self.coulomb._h if self.coulomb is not None else None, _abi.dptr(charge), nk,
                                     _abi.dptr(_matT(mc.mat)), _abi.dptr(_matT(mc.invmat)), ff.cutoff ** 2, rules.ctypes.data,
                                     _abi.i32ptr(offsets), COULOMBIC_CONVERSION_FACTOR, *kargs)
        _abi.check(self._lib, rc)
        self._h = h
        self.refresh()

    def neighbour_cells(self):
        """-> (bins per axis, capacity per cell) of the guest neighbour cells, or None when the guest-guest sum runs the exhaustive
        loop (MC cells of a few cutoffs, i.e. every fixture of the reference; energy.jl:340-349 is the reference's own switch)."""
        nb = np.zeros(3, dtype=np.int32)
        cap = C.c_int32(0)
        rc = self._lib.ceg_mc_neighbour_cells(self._h, _abi.i32ptr(nb), C.byref(cap))
        if rc < 0:
            _abi.check(self._lib, rc)
        return (tuple(int(x) for x in nb), int(cap.value)) if rc == 1 else None

    def refresh(self) -> None:
        """Upload the guests of ``self.mc`` (initial state, or to resynchronise with the host side)."""
        pos, kinds, first = [], [], [0]
        self._slot = [[] for _ in self.mc.positions]          # [kind][index in kind] -> molecule index on the device
        for i, j, ids, p in self.mc.molecules():
            self._slot[i].append(len(first) - 1)
            pos.append(np.asarray(p, dtype=np.float64).reshape(-1, 3))
            kinds += [k - 1 for k in ids]
            first.append(first[-1] + len(ids))
        pos = np.ascontiguousarray(np.concatenate(pos) if pos else np.empty((0, 3)), dtype=np.float64)
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        first = np.ascontiguousarray(first, dtype=np.int32)
        _abi.check(self._lib, self._lib.ceg_mc_set_guests(self._h, _abi.dptr(pos.reshape(-1)), _abi.i32ptr(kinds), _abi.i32ptr(first), len(first) - 1))

    def trial(self, idx, positions) -> np.ndarray:
        """-> float64[n + 1, 4]: row 0 movement_energy of molecule ``idx`` (0-based (kind, molecule)) where it is now, row 1 + t at
        ``positions[t]``; columns (framework vdw, framework direct, inter, reciprocal)."""
        mol = self._slot[idx[0]][idx[1]]
        m = len(self.mc.ffidx[idx[0]])
        t = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, m, 3)
        out = np.empty((len(t) + 1, 4), dtype=np.float64)
        _abi.check(self._lib, self._lib.ceg_mc_trial(self._h, mol, _abi.dptr(t.reshape(-1)) if len(t) else None, len(t), _abi.dptr(out.reshape(-1))))
        return out

    def trial_device(self, idx, d_trial: int, n: int, d_out: int, stream: int = 0) -> None:
        """``ceg_mc_trial_device``: ``n`` trial placements at device address ``d_trial`` (float64[n, m, 3]) -> rows at device address
        ``d_out`` (float64[n + 1, 4]), enqueued on ``stream``; nothing is copied or synchronised."""
        _abi.check(self._lib, self._lib.ceg_mc_trial_device(self._h, self._slot[idx[0]][idx[1]], C.c_void_p(d_trial), int(n), C.c_void_p(d_out),
                                                            C.c_void_p(stream) if stream else None))

    def trial_insert_device(self, i: int, d_trial: int, n: int, d_out: int, stream: int = 0) -> None:
        """``ceg_mc_trial_insert_device``: rows float64[n, 4] at ``d_out`` for a NEW molecule of kind ``i``."""
        k = np.ascontiguousarray([ix - 1 for ix in self.mc.ffidx[i]], dtype=np.int32)
        _abi.check(self._lib, self._lib.ceg_mc_trial_insert_device(self._h, _abi.i32ptr(k), len(k), C.c_void_p(d_trial), int(n), C.c_void_p(d_out),
                                                                   C.c_void_p(stream) if stream else None))

    def accept(self, idx, positions) -> None:
        """update_mc!(mc, idx, positions) on the device (asynchronous).  The host-side ``mc`` is NOT touched."""
        p = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1)
        _abi.check(self._lib, self._lib.ceg_mc_accept(self._h, self._slot[idx[0]][idx[1]], _abi.dptr(p)))

    def trial_insert(self, i: int, positions) -> np.ndarray:
        """movement_energy of a NEW molecule of kind ``i`` at each of ``positions[n, m, 3]`` -> float64[n, 4]."""
        k = np.ascontiguousarray([ix - 1 for ix in self.mc.ffidx[i]], dtype=np.int32)
        t = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, len(k), 3)
        out = np.empty((len(t), 4), dtype=np.float64)
        _abi.check(self._lib, self._lib.ceg_mc_trial_insert(self._h, _abi.i32ptr(k), len(k), _abi.dptr(t.reshape(-1)), len(t), _abi.dptr(out.reshape(-1))))
        return out

    def insert(self, i: int, positions) -> int:
        """add_one_system! on the device: a molecule of kind ``i`` joins (index in its kind returned, = append)."""
        k = np.ascontiguousarray([ix - 1 for ix in self.mc.ffidx[i]], dtype=np.int32)
        p = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1)
        mol = C.c_int32(-1)
        _abi.check(self._lib, self._lib.ceg_mc_insert(self._h, _abi.i32ptr(k), len(k), _abi.dptr(p), C.byref(mol)))
        self._slot[i].append(int(mol.value))
        return len(self._slot[i]) - 1

    def remove(self, idx) -> int:
        """remove_one_system!(mc, i, j) on the device; the host-side indices behave like the reference's (montecarlo.jl:798-808):
        the last molecule of kind ``i`` takes index ``j``; returns its old index."""
        i, j = idx
        d = self._slot[i][j]
        last = len(self._slot[i]) - 1
        self._slot[i][j] = self._slot[i][last]
        self._slot[i].pop()
        moved = C.c_int32(-1)
        _abi.check(self._lib, self._lib.ceg_mc_remove(self._h, d, C.byref(moved)))
        if moved.value != d:                         # the device moved its last molecule into the hole
            for kind in self._slot:
                for q, v in enumerate(kind):
                    if v == moved.value:
                        kind[q] = d
        return last

    def baseline_energy(self):
        """baseline_energy (montecarlo.jl:530-542) from the device-resident state: framework and guest-guest terms from row 0 of one
        trial launch per molecule (every pair is seen from both sides, hence the 1/2), the reciprocal term from the total guest
        structure factor kept on the device and the two EwaldContext constants (ewald.jl:497-544)."""
        from .hostmirror import montecarlo as M
        from .hostmirror.ewald import ewald_context_constants
        mc = self.mc
        fv = fd = inter = 0.0
        for i, kind in enumerate(self._slot):
            m = len(mc.ffidx[i])
            for j in range(len(kind)):
                row = self.trial((i, j), np.empty((0, m, 3)))[0]
                fv += row[0]; fd += row[1]; inter += row[2]
        reciprocal = 0.0
        ef = mc.ewald
        if ef.alpha != 0.0:
            _pos, a = self.state()
            enc, static = ewald_context_constants(ef, [k for k in M._ewald_systems(mc) if k])
            f = ef.StoreRigidChargeFramework
            reciprocal = (2 * (float((ef.kfactors * (f.real * a.real + f.imag * a.imag)).sum()) + enc)
                          + float((ef.kfactors * (a.real ** 2 + a.imag ** 2)).sum()) + static)
        return M.BaselineEnergyReport(fv, fd, 0.5 * inter, reciprocal, mc.tailcorrection)

    def state(self):
        """(positions[natoms, 3], total guest structure factor complex[nk]) read back from the device."""
        natoms = sum(len(ids) for _i, _j, ids, _p in self.mc.molecules())
        nk = len(self.mc.ewald.kfactors) if self.mc.ewald.alpha != 0.0 else 0
        pos = np.empty((natoms, 3)); re = np.empty(max(nk, 1)); im = np.empty(max(nk, 1))
        _abi.check(self._lib, self._lib.ceg_mc_get_state(self._h, _abi.dptr(pos.reshape(-1)) if natoms else None, _abi.dptr(re), _abi.dptr(im)))
        # device molecule order -> the host's (kind, index) order
        sizes = {}
        for i, kind in enumerate(self._slot):
            for d in kind:
                sizes[d] = len(self.mc.ffidx[i])
        start, o = {}, 0
        for d in sorted(sizes):
            start[d] = o
            o += sizes[d]
        host = [pos[start[d]:start[d] + sizes[d]] for kind in self._slot for d in kind]
        return (np.concatenate(host) if host else pos), (re[:nk] + 1j * im[:nk])

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ceg_mc_destroy(self._h)
            self._h = None
        for it in self.interp:
            if it is not None:
                it.close()
        if self.coulomb is not None:
            self.coulomb.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
