"""``setup_RASPA`` and the on-disk grid cache (mirror of ``src/raspa.jl:367-531``).
These are the *callers* of the hot path: they are kept signature-compatible so code
written against the reference's ``setup_RASPA`` / ``energy_point`` reads the same."""
from __future__ import annotations

import math
import os
from typing import List, Optional, Tuple

import numpy as np

from .ewald import EwaldFramework, initialize_ewald
from .forcefields import ForceField
from ..grids import (BlockFile, CrystalEnergySetup, EnergyGrid, create_grid_coulomb, create_grid_vdw, create_grids_multi,
                    parse_blockfile, parse_blockfile_gpu, parse_grid)
from .. import _abi
from .coordinates import GridCoordinatesSetup
from .raspa import (RASPASystem, _ff, _ffname, _ffpal, getdir_RASPA, load_framework_RASPA,
                    load_molecule_RASPA)
from .utils import find_supercell, get_atom_name


def default_system(s, pff) -> RASPASystem:
    """raspa.jl:367-381 (single atom defined in the framework force field)."""
    forcefield = _ff(pff)
    if isinstance(s, RASPASystem):
        return s
    symbol = get_atom_name(s)
    if symbol not in forcefield.sdict:
        return load_molecule_RASPA(s, "Default", forcefield)
    pseudo = _ffpal(pff)[symbol]
    return RASPASystem(np.diag([np.inf] * 3), np.zeros((1, 3)), [symbol], np.array([pseudo.mass]),
                       np.array([pseudo.charge]), True)


def decide_parse_block(blockfile, molecule: RASPASystem, framework_name) -> str:
    """raspa.jl:383-395"""
    if isinstance(framework_name, np.ndarray):
        return ""
    if isinstance(blockfile, (bool, str)):
        newblockfile = blockfile
    else:
        newblockfile = not (len(molecule) == 1 and molecule.atomic_charge[0] > np.finfo(float).eps)
    if isinstance(newblockfile, bool):
        if not newblockfile:
            return ""
        return str(framework_name).split('_')[0]
    return str(newblockfile)


def parse_block(blockfile, framework_name, framework: RASPASystem, molecule: RASPASystem, spacing: float,
                scan: str = "gpu") -> BlockFile:
    """raspa.jl:397-403.  The point x sphere scan of parse_blockfile runs on the GPU (``ceg_block_spheres``);
    ``scan="host"`` selects the numpy mirror of the reference loop (used by the CPU-only tests)."""
    blockpath = decide_parse_block(blockfile, molecule, framework_name)
    csetup = GridCoordinatesSetup.from_cell(framework.mat, spacing)
    if not blockpath:
        return BlockFile(csetup)
    path = os.path.join(getdir_RASPA(), "structures", "block", blockpath) + ".block"
    if scan == "host":
        return parse_blockfile(path, csetup)
    return parse_blockfile_gpu(path, csetup)


def grid_locations(framework, pff, forcefield: ForceField, atoms: List[str], gridstep: float, supercell):
    """raspa.jl:405-420 (the hashed name used for non-string force fields is replaced by
    the plain name: StableHashTraits is third-party and irrelevant to the hot path)."""
    if isinstance(framework, np.ndarray):
        return "", []
    rawname = _ffname(pff)
    raspa = getdir_RASPA()
    rootdir = os.path.join(raspa, "grids", rawname, framework)
    grid_dir = os.path.join(rootdir, "%.6f" % gridstep)
    supercell_name = "x".join(str(x) for x in supercell)
    coulomb_grid_path = os.path.join(grid_dir, supercell_name, framework + "_Electrostatics_Ewald.grid")
    with open(os.path.join(raspa, "forcefield", rawname, "force_field_mixing_rules.def")) as f:
        trunc_or_shift = next(l for l in f.read().splitlines() if not l.startswith('#')).strip()
    vdws = [os.path.join(grid_dir, f"{framework}_{atom}_{trunc_or_shift}.grid") for atom in atoms]
    return coulomb_grid_path, vdws


def retrieve_or_create_grid(grid_path, syst_framework, forcefield: ForceField, gridstep, atom_or_eframework,
                            mat, new: bool, cutoff: float, ngpus: int = 1) -> EnergyGrid:
    """raspa.jl:420-439"""
    if not grid_path or math.isinf(cutoff):
        return EnergyGrid.trivial(False)
    if cutoff != 12.0:
        raise ValueError("Cutoff other than 12 Å or infinity is not supported.")
    iscoulomb = isinstance(atom_or_eframework, EwaldFramework)
    if not iscoulomb and not forcefield.needsvdwgrid(atom_or_eframework):
        return EnergyGrid.trivial(True)
    if new or not os.path.isfile(grid_path):
        os.makedirs(os.path.dirname(grid_path), exist_ok=True)
        if iscoulomb:
            create_grid_coulomb(grid_path, syst_framework, forcefield, gridstep, atom_or_eframework, ngpus)
        else:
            create_grid_vdw(grid_path, syst_framework, forcefield, gridstep, atom_or_eframework, ngpus)
    return parse_grid(grid_path, iscoulomb, mat)


def setup_RASPA(framework, pff, molecule, ffname_molecule: Optional[str] = None, *, gridstep: float = 0.15,
                supercell=None, blockfile=None, new: bool = False, cutoff: float = 12.0,
                ngpus: int = 1, multi: bool = True) -> CrystalEnergySetup:
    """raspa.jl:472-531.  ``molecule`` may be a molecule name (with ``ffname_molecule``),
    a RASPASystem, or the name of a single atom of the framework force field.  ``multi=False`` builds missing grids
    one by one like the reference does."""
    syst_framework = load_framework_RASPA(framework, pff)
    if isinstance(molecule, RASPASystem):
        syst_mol = molecule
    elif ffname_molecule is not None:
        syst_mol = load_molecule_RASPA(molecule, ffname_molecule, pff, syst_framework)
    else:
        syst_mol = default_system(molecule, pff)
    if supercell is None:
        supercell = (1, 1, 1) if math.isinf(cutoff) else find_supercell(syst_framework.mat, cutoff)
    mat = syst_framework.mat
    forcefield = _ff(pff, cutoff=cutoff)
    block = parse_block(blockfile, framework, syst_framework, syst_mol, gridstep)

    atomdict = {}
    atoms = list(syst_mol.atomic_symbol)
    for atom in atoms:
        atomdict.setdefault(atom, len(atomdict))
    atomsidx = [atomdict[a] for a in atoms]
    rev_atomdict = [None] * len(atomdict)
    for at, i in atomdict.items():
        rev_atomdict[i] = at
    coulomb_grid_path, vdws = grid_locations(framework, pff, forcefield, rev_atomdict, gridstep, supercell)

    needcoulomb = any(q != 0 for q in syst_mol.atomic_charge)
    ewald = initialize_ewald(syst_framework, supercell) if needcoulomb else EwaldFramework.empty(mat)
    # The reference builds the missing grids one after the other (raspa.jl:497-520: one retrieve_or_create_grid for the
    # Coulomb grid, one per distinct atom of the molecule).  Here every grid that has to be CREATED is collected first and
    # built by one multi-probe call -- one lattice-image list, one pass over the framework (grids.create_grids_multi) --;
    # retrieve_or_create_grid then finds the files and only parses them.  The atoms go in groups of four whatever their rule
    # class (round 4: Na + the C and O of CO2 in one call; the library lets the Lennard-Jones-only probes share accumulating loops and
    # launches a Buckingham cation alone or fused with the Coulomb grid); anything the library refuses falls through to the one-by-one path.
    written = set()                                               # files the multi-probe calls below have just created
    if multi and not isinstance(framework, np.ndarray) and not math.isinf(cutoff) and cutoff == 12.0:
        todo = [i for i, atom in enumerate(rev_atomdict)
                if vdws[i] and forcefield.needsvdwgrid(atom) and (new or not os.path.isfile(vdws[i]))]
        want_c = bool(needcoulomb and coulomb_grid_path and (new or not os.path.isfile(coulomb_grid_path)))
        groups = [todo[lo:lo + 4] for lo in range(0, len(todo), 4)]
        for n, part in enumerate(groups):
            with_c = want_c and n == 0
            if len(part) + int(with_c) < 2:
                continue                                          # a single grid: the ordinary path does it
            for pth in [vdws[i] for i in part] + ([coulomb_grid_path] if with_c else []):
                os.makedirs(os.path.dirname(pth), exist_ok=True)
            try:
                create_grids_multi([vdws[i] for i in part], coulomb_grid_path if with_c else None, syst_framework, forcefield,
                                   gridstep, [rev_atomdict[i] for i in part], ewald if with_c else None, ngpus)
                written.update([vdws[i] for i in part] + ([coulomb_grid_path] if with_c else []))
            except _abi.CegError as exc:
                if exc.code != -5:                                # CEG_ERR_UNSUPPORTED: one by one instead
                    raise
    if needcoulomb:
        if isinstance(framework, np.ndarray):
            coulomb = EnergyGrid.trivial(True)
        else:
            coulomb = retrieve_or_create_grid(coulomb_grid_path, syst_framework, forcefield, gridstep, ewald,
                                              mat, new and coulomb_grid_path not in written, cutoff, ngpus)
    else:
        coulomb = EnergyGrid.trivial(True)

    grids: List[EnergyGrid] = [None] * len(atomdict)  # type: ignore[list-item]
    for atom, i in atomdict.items():
        if isinstance(framework, np.ndarray):
            grids[i] = EnergyGrid.trivial(True)
        else:
            grids[i] = retrieve_or_create_grid(vdws[i], syst_framework, forcefield, gridstep, atom, mat,
                                               new and vdws[i] not in written, cutoff, ngpus)
    charges = [float(q) for q in syst_mol.atomic_charge]
    return CrystalEnergySetup(syst_framework, syst_mol, coulomb, charges, grids, atomsidx, ewald, forcefield, block)
