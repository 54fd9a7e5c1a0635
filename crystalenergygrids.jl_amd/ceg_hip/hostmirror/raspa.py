"""RASPA2 directory interface (host-side mirror of ``src/raspa.jl``).

Keeps the reference's entry points -- ``setdir_RASPA``, ``parse_pseudoatoms_RASPA``,
``parse_forcefield_RASPA``, ``load_framework_RASPA``, ``load_molecule_RASPA``,
``retrieve_or_create_grid``, ``setup_RASPA`` -- with the same argument meaning.  The
reference reads CIFs through AtomsIO/Chemfiles (third-party); here a minimal P1 CIF
reader produces the same cartesian positions (cell matrix with a along x, b in the
xy plane -- identical to ``mat_from_parameters``, utils.jl:134-138).
"""
from __future__ import annotations

import math
import os
import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .constants import COULOMBIC_CONVERSION_FACTOR
from .forcefields import ForceField, build_forcefield, mix_rules
from .interactions import (FF, Mixing, InteractionRule, InteractionRuleSum, Rule, make_rule, map_rule,
                           rules_of, shifted_rule, sum_rules)
from .utils import find_supercell, get_atom_name, mat_from_parameters

_RASPADIR: List[str] = [os.path.join(os.path.expanduser("~"), "RASPA2", "simulations", "share", "raspa")]


def setdir_RASPA(path) -> None:
    """raspa.jl:14-22 (``setdir_RASPA!``)"""
    _RASPADIR[0] = str(path)


def getdir_RASPA() -> str:
    """raspa.jl:24-34"""
    ret = _RASPADIR[0]
    if not os.path.isdir(ret):
        raise FileNotFoundError(f"Could not find raspa directory at the given path {ret}. "
                                "Please set the correct path through the `setdir_RASPA` function.")
    return ret


# ---------------------------------------------------------------- pseudo atoms
@dataclass
class PseudoAtomInfo:
    """raspa.jl:36-55"""
    type: str
    printas: str
    symbol: str
    oxidation: float
    mass: float
    charge: float
    polarization: float
    Bfactor: float
    radius: float
    connectivity: float
    anisotropic: float
    anisotropy_absolute: bool
    tinker_type: int


class PseudoAtomListing:
    """raspa.jl:92-125.  Open-ended names (``Name_``) match any atom they prefix; the
    entry defined last in the file wins over earlier ones."""

    def __init__(self, open_names: Dict[str, int], exact: Dict[str, int], info: List[PseudoAtomInfo]):
        self.open_names = open_names      # name (without trailing '_') -> 1-based line
        self.exact = exact
        self.info = info

    def get_strict(self, atom) -> PseudoAtomInfo:
        name = str(atom)
        curr = self.exact.get(name, 0)
        for prefix, j in self.open_names.items():
            if name.startswith(prefix) and j > curr:
                curr = j
        if curr == 0:
            raise KeyError(f"Atom {atom} not found in pseudo_atoms.def")
        return self.info[curr - 1]

    def __getitem__(self, atom) -> PseudoAtomInfo:
        return self.get_strict(get_atom_name(atom))


def parse_pseudoatoms_RASPA(file) -> PseudoAtomListing:
    """raspa.jl:127-168"""
    with open(file) as f:
        lines = [l.rstrip("\n") for l in f]
    lines = [l for l in lines if l and l[0] != '#']
    n = int(lines.pop(0))
    if n != len(lines):
        raise ValueError(f"Found {len(lines)} non-empty lines but {n} declared")
    open_names: Dict[str, int] = {}
    exact: Dict[str, int] = {}
    info: List[PseudoAtomInfo] = []
    for i, _l in enumerate(lines, start=1):
        l = _l.split()
        if len(l) != 14:
            raise ValueError(f'malformed line "{l}" does not contain 14 fields.')
        is_open = l[0][-1] == '_'
        name = l[0][:-1] if is_open else l[0]
        if is_open:
            open_names[name] = i
        else:
            exact[name] = i
        info.append(PseudoAtomInfo(name, l[2] if l[1] == "yes" else "", l[3],
                                   float(l[4]), float(l[5]), float(l[6]), float(l[7]),
                                   float(l[8]), float(l[9]), float(l[10]), float(l[11]),
                                   l[12] == "absolute", int(l[13])))
    return PseudoAtomListing(open_names, exact, info)


def _ffname(pff) -> str:
    if isinstance(pff, str):
        return pff
    if isinstance(pff, ForceField):
        return pff.name
    return pff[0]


def _ffpal(pff) -> PseudoAtomListing:
    """raspa.jl:197-203"""
    if isinstance(pff, tuple) and isinstance(pff[1], PseudoAtomListing):
        return pff[1]
    return parse_pseudoatoms_RASPA(os.path.join(getdir_RASPA(), "forcefield", _ffname(pff), "pseudo_atoms.def"))


def _ff(pff, cutoff=None, ewald_precision=None) -> ForceField:
    """raspa.jl:171-186"""
    if isinstance(pff, ForceField):
        if cutoff is not None and pff.cutoff != cutoff:
            raise ValueError(f"Inconsistent cutoffs: choose between {pff.cutoff} from force field or given {cutoff}")
        return pff
    if not isinstance(pff, str) and isinstance(pff[1], ForceField):
        return pff[1]
    kwargs = {}
    if cutoff is not None:
        kwargs["cutoff"] = cutoff
    if ewald_precision is not None:
        kwargs["ewald_precision"] = ewald_precision
    return parse_forcefield_RASPA(pff, **kwargs)


def _ffcharge(pff, atom, pal: PseudoAtomListing, pseudo: PseudoAtomInfo) -> float:
    """raspa.jl:204-216"""
    if isinstance(pff, str) or (not isinstance(pff, ForceField) and isinstance(pff[1], PseudoAtomListing)):
        return pseudo.charge
    ff = _ff(pff)
    for inter in rules_of(ff[pseudo.type, pseudo.type]):
        if inter.kind == FF.CoulombEwaldDirect:
            return inter.params[1]
    return pseudo.charge


# ---------------------------------------------------------------- systems
@dataclass
class RASPASystem:
    """raspa.jl:248-256.  ``mat`` columns are the bounding-box vectors (Å)."""
    mat: np.ndarray
    position: np.ndarray                 # float64[n,3]
    atomic_symbol: List[str]
    atomic_mass: np.ndarray
    atomic_charge: np.ndarray
    ismolecule: bool = False

    def __len__(self) -> int:
        return len(self.atomic_symbol)

    def with_positions(self, positions) -> "RASPASystem":
        """``ChangePositionSystem(syst, positions)``"""
        return RASPASystem(self.mat, np.asarray(positions, dtype=np.float64).reshape(len(self), 3),
                           self.atomic_symbol, self.atomic_mass, self.atomic_charge, self.ismolecule)


_NUM = re.compile(r"^([-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?)(?:\(\d+\))?$")


def _cif_float(tok: str) -> float:
    m = _NUM.match(tok)
    if not m:
        raise ValueError(f"cannot parse CIF number {tok!r}")
    return float(m.group(1))


def read_cif_P1(path) -> Tuple[np.ndarray, List[str], np.ndarray]:
    """Minimal CIF reader: cell parameters + ``_atom_site_label`` /
    ``_atom_site_fract_{x,y,z}`` of a P1 structure -> (cell matrix, labels,
    cartesian positions).  Replaces ``load_system(AtomsIO.ChemfilesParser(), cif)``
    (raspa.jl:306); a file listing symmetry operations other than ``x,y,z`` is rejected."""
    cellp: Dict[str, float] = {}
    labels: List[str] = []
    fracs: List[List[float]] = []
    with open(path) as f:
        lines = [l.strip() for l in f]
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("_cell_length_") or l.startswith("_cell_angle_"):
            key, val = l.split()[:2]
            cellp[key] = _cif_float(val)
            i += 1
        elif l == "loop_":
            i += 1
            headers = []
            while i < len(lines) and lines[i].startswith("_"):
                headers.append(lines[i].split()[0])
                i += 1
            rows = []
            while i < len(lines) and lines[i] and not lines[i].startswith(("_", "loop_", "#", "data_")):
                rows.append(lines[i])
                i += 1
            if "_symmetry_equiv_pos_as_xyz" in headers or "_space_group_symop_operation_xyz" in headers:
                ops = [re.sub(r"^\d+\s+", "", r).replace("'", "").replace('"', "").replace(" ", "").lower()
                       for r in rows]
                if any(o != "x,y,z" for o in ops):
                    raise ValueError(f"{path}: only P1 CIF files are supported")
            if "_atom_site_fract_x" in headers:
                il = headers.index("_atom_site_label")
                ix, iy, iz = (headers.index(f"_atom_site_fract_{c}") for c in "xyz")
                for r in rows:
                    t = r.split()
                    labels.append(t[il])
                    fracs.append([_cif_float(t[ix]), _cif_float(t[iy]), _cif_float(t[iz])])
        else:
            i += 1
    mat = mat_from_parameters(
        (cellp["_cell_length_a"], cellp["_cell_length_b"], cellp["_cell_length_c"]),
        (cellp["_cell_angle_alpha"], cellp["_cell_angle_beta"], cellp["_cell_angle_gamma"]))
    frac = np.array(fracs, dtype=np.float64).reshape(len(labels), 3)
    return mat, labels, frac @ mat.T


def load_framework_RASPA(name, pff) -> RASPASystem:
    """raspa.jl:303-317 (and :319-325 for a bare cell matrix)."""
    if isinstance(name, np.ndarray):
        return RASPASystem(np.array(name, dtype=np.float64), np.empty((0, 3)), [], np.empty(0), np.empty(0), False)
    raspa = getdir_RASPA()
    cif = os.path.join(raspa, "structures", "cif", name if os.path.splitext(name)[1] == ".cif" else name + ".cif")
    mat, labels, pos = read_cif_P1(cif)
    pal = _ffpal(pff)
    mass = np.empty(len(labels))
    charges = np.empty(len(labels))
    for i, lab in enumerate(labels):
        symb = get_atom_name(lab)
        pseudo = pal[symb]
        mass[i] = pseudo.mass
        charges[i] = _ffcharge(pff, symb, pal, pseudo)
    return RASPASystem(mat, pos, labels, mass, charges, False)


def parse_molecule_RASPA(file):
    """raspa.jl:219-239"""
    with open(file) as f:
        lines = [l.rstrip("\n") for l in f]
    lines = [l for l in lines if l and l[0] != '#']
    num_atoms = int(lines[3])
    num_groups = int(lines[4])
    assert num_groups == 1
    assert num_atoms == int(lines[6])
    assert num_atoms == 1 or lines[5].strip() == "rigid"
    positions = np.zeros((num_atoms, 3))
    symbols = []
    for i in range(num_atoms):
        l = lines[7 + i].split()
        symbols.append(l[1])
        if num_atoms != 1:
            positions[i] = [float(l[2]), float(l[3]), float(l[4])]
    return symbols, positions


def load_molecule_RASPA(name: str, ffname_molecule: str, pff, framework_system: Optional[RASPASystem] = None) -> RASPASystem:
    """raspa.jl:343-364"""
    raspa = getdir_RASPA()
    symbols, positions = parse_molecule_RASPA(
        os.path.join(raspa, "molecules", ffname_molecule, name if os.path.splitext(name)[1] == ".def" else name + ".def"))
    pal = _ffpal(pff)
    mass = np.array([pal[a].mass for a in symbols])
    charges = np.array([pal[a].charge for a in symbols])
    bbox = framework_system.mat if framework_system is not None else np.diag([np.inf] * 3)
    return RASPASystem(bbox, positions, symbols, mass, charges, True)


# ---------------------------------------------------------------- force field
def parse_interaction_RASPA(l: str, mixingrules: bool, shift: bool, cutoff: float, tailcorrection: bool):
    """raspa.jl:533-556"""
    splits = l.split()
    for i, x in enumerate(splits):
        if x[0] == '#' or (len(x) > 1 and x[0] == '/' and x[1] == '/'):
            splits = splits[:i]
            break
    m = 1 if mixingrules else 0
    kind = splits[2 - m].lower()
    key = (splits[0], splits[1 - m])
    if kind == "lennard-jones":
        rule: Rule = shifted_rule(FF.LennardJones, [float(x) for x in splits[3 - m:]], shift, cutoff, tailcorrection)
    elif kind == "none":
        rule = make_rule(FF.NoInteraction)
    elif kind == "buckingham":
        rule = shifted_rule(FF.Buckingham, [float(x) for x in splits[3 - m:]], shift, cutoff, tailcorrection)
    elif kind == "buckingham2":
        rule = InteractionRuleSum([
            shifted_rule(FF.Buckingham, [float(x) for x in splits[3 - m:6 - m]], shift, cutoff, tailcorrection),
            shifted_rule(FF.HardSphere, [float(splits[6 - m]), 0.0], shift, cutoff, tailcorrection)])
    else:
        raise NotImplementedError(f"{kind} interaction potential not implemented")
    return key, rule


def parse_mixingrule_RASPA(l: str) -> Mixing:
    """raspa.jl:557-568"""
    x = l.lower()
    if x == "lorentz-berthelot":
        return Mixing.LorentzBerthelot
    if x in ("jorgensen", "good-hope", "geometric"):
        return Mixing.Geometric
    raise ValueError(f"Unknown mixing rule {l}")


def _parse_shift(s: str) -> bool:
    s2 = s.lower()
    if s2 == "shifted":
        return True
    assert s2 == "truncated"
    return False


def _parse_yesno(s: str) -> bool:
    s2 = s.lower()
    if s2 == "yes":
        return True
    assert s2 == "no"
    return False


def _next_noncomment_line(lines: List[str], i: int) -> int:
    """raspa.jl:587-593 (0-based here)"""
    j = i + 1
    while lines[j][0] == '#':
        j += 1
    return j


def parse_forcefield_RASPA(name, cutoff: float = 12.0, ewald_precision: Optional[float] = None) -> ForceField:
    """raspa.jl:611-702"""
    if ewald_precision is None:
        ewald_precision = 0.0 if math.isinf(cutoff) else 1e-6
    pseudoatoms = name[1] if (not isinstance(name, str) and isinstance(name[1], PseudoAtomListing)) else _ffpal(name)
    rawname = _ffname(name)
    input_: List[Tuple[Tuple[str, str], Rule]] = []
    general_mixingrule = Mixing.ErrorOnMix
    shift = True
    tailcorrection = False
    sdict: Dict[str, int] = {}
    ffdir = os.path.join(getdir_RASPA(), "forcefield", rawname)
    ffmr = os.path.join(ffdir, "force_field_mixing_rules.def")
    if os.path.isfile(ffmr):
        with open(ffmr) as f:
            lines = f.read().split("\n")
        idx = _next_noncomment_line(lines, 0)
        shift = _parse_shift(lines[idx])
        idx = _next_noncomment_line(lines, idx)
        tailcorrection = _parse_yesno(lines[idx])
        idx = _next_noncomment_line(lines, idx)
        num = int(lines[idx])
        for i in range(1, num + 1):
            idx = _next_noncomment_line(lines, idx)
            key, rule = parse_interaction_RASPA(lines[idx], True, shift, cutoff, tailcorrection)
            sdict[key[0]] = i
            input_.append((key, rule))
        idx = _next_noncomment_line(lines, idx)
        general_mixingrule = parse_mixingrule_RASPA(lines[idx])
    ff = build_forcefield(input_, general_mixingrule, cutoff, shift, tailcorrection, sdict=sdict, name=rawname)
    ffdef = os.path.join(ffdir, "force_field.def")
    if os.path.isfile(ffdef):
        with open(ffdef) as f:
            lines = f.read().split("\n")
        jdx = _next_noncomment_line(lines, 0)
        numnewrules = int(lines[jdx])
        startnewrules = jdx
        jdx = _next_noncomment_line(lines, jdx)
        for _ in range(numnewrules):
            jdx = _next_noncomment_line(lines, jdx)
        numnewinteractions = int(lines[jdx])
        for _ in range(numnewinteractions):
            jdx = _next_noncomment_line(lines, jdx)
            (x1, y1), newinter = parse_interaction_RASPA(lines[jdx], False, shift, cutoff, tailcorrection)
            i1, j1 = _ff_indices(x1, y1, sdict)
            ff.interactions[i1][j1] = ff.interactions[j1][i1] = newinter
        jdx = _next_noncomment_line(lines, jdx)
        numnewmixing = int(lines[jdx])
        for _ in range(numnewmixing):
            jdx = _next_noncomment_line(lines, jdx)
            x2, y2, _newmix = lines[jdx].split()
            i2, j2 = _ff_indices(x2, y2, sdict)
            newmix = parse_mixingrule_RASPA(_newmix)
            ff.interactions[i2][j2] = ff.interactions[j2][i2] = mix_rules(ff.interactions[i2][i2], ff.interactions[j2][j2], newmix)
        jdx = startnewrules
        for _ in range(numnewrules):
            jdx = _next_noncomment_line(lines, jdx)
            x3, y3, _newshift, _newtail = lines[jdx].split()
            newshift = _parse_shift(_newshift)
            newtail = _parse_yesno(_newtail)
            i3, j3 = _ff_indices(x3, y3, sdict)
            new = map_rule(lambda r: shifted_rule(r.kind, r.params, newshift, cutoff, newtail), ff.interactions[i3][j3])
            ff.interactions[i3][j3] = ff.interactions[j3][i3] = new
    if ewald_precision == 0:
        alpha = 0.0
    else:
        eps = math.log(float(ewald_precision * cutoff))
        alpha = math.sqrt(abs(eps + math.log(math.sqrt(abs(eps))))) / float(cutoff)
    for ati, i4 in sdict.items():
        chargei = pseudoatoms[ati].charge
        if chargei == 0.0:
            continue
        for atj, j4 in sdict.items():
            chargej = pseudoatoms[atj].charge
            if chargej == 0.0:
                continue
            if ewald_precision == 0:
                extra = InteractionRule(FF.Coulomb, [chargei, chargej], 0.0, False)
            else:
                extra = InteractionRule(FF.CoulombEwaldDirect, [alpha, chargei, chargej], 0.0, False)
            ff.interactions[i4 - 1][j4 - 1] = sum_rules(ff.interactions[i4 - 1][j4 - 1], extra)
    return ForceField(ff.interactions, ff.sdict, ff.symbols, ff.cutoff, rawname)


def _ff_indices(x: str, y: str, sdict: Dict[str, int]) -> Tuple[int, int]:
    """raspa.jl:581-585, returned 0-based"""
    if x not in sdict:
        raise KeyError(f"Atom {x} absent from force_field_mixing_rules.def")
    if y not in sdict:
        raise KeyError(f"Atom {y} absent from force_field_mixing_rules.def")
    return sdict[x] - 1, sdict[y] - 1


def setup_probe_RASPA(framework: str, pff, atom: str):
    """raspa.jl:704-708"""
    from .probes import ProbeSystem
    system = load_framework_RASPA(framework, pff)
    return ProbeSystem.build(system, _ff(pff), atom)
