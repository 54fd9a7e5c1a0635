"""ceg_hip -- host-side mirror of the CrystalEnergyGrids.jl grid-build interface on top of
``libceg_hip.so`` (hand-written HIP kernels for MI355X / gfx950).

Same names and argument meaning as the reference's public functions for this path
(``setup_RASPA``, ``create_grid_vdw``, ``create_grid_coulomb``, ``parse_grid``,
``interpolate_grid``, ``energy_point`` ...).  The compute path is the HIP library only;
importing this package does not load it, the first grid build does and fails loudly if it
is not built.

PRODUCT (bindings of the C ABI): ``_abi``, ``plan`` (resident-plan API, device buffers), ``grids`` (the reference's grid
functions on top of the one-shot entry points), ``interp`` / ``energy`` (batched GPU consumers of the grids: interpolation,
reciprocal Ewald, pair energies, ``GpuEnergySetup``, ``GpuMonteCarloEnergy``, ``DeviceMonteCarlo``), ``distributed`` (x-slab /
block-cyclic sharding over ranks).

HARNESS (``hostmirror/``, see its docstring): Python restatements of the reference's host code around the path -- RASPA parsers,
force fields, ProbeSystem, Ewald set-up, the Monte-Carlo energy functions, ``setup_RASPA`` -- which build the inputs of the C ABI
where Julia is absent; ``workloads`` assembles the BASELINE.json configurations from them.  The names re-exported below keep the
reference's public spelling (``ceg.setup_RASPA``, ``ceg.load_framework_RASPA`` ...).
"""
from .hostmirror.constants import GRID_TO_KELVIN, COULOMBIC_CONVERSION_FACTOR
from .hostmirror.interactions import FF, Mixing, InteractionRule, InteractionRuleSum, UndefinedInteractionError
from .hostmirror.forcefields import ForceField, build_forcefield
from .hostmirror.coordinates import CellMatrix, GridCoordinatesSetup, abc_to_xyz, offsetpoint
from .hostmirror.probes import ProbeSystem
from .hostmirror.ewald import EwaldFramework, initialize_ewald, compute_ewald
from .hostmirror.raspa import (setdir_RASPA, getdir_RASPA, parse_pseudoatoms_RASPA, parse_forcefield_RASPA,
                    load_framework_RASPA, load_molecule_RASPA, setup_probe_RASPA, RASPASystem)
from .grids import (EnergyGrid, CrystalEnergySetup, create_grid_vdw, create_grid_coulomb, parse_grid,
                    interpolate_grid, energy_point, build_vdw_array, build_coulomb_array)
from .hostmirror.setup_raspa import setup_RASPA, retrieve_or_create_grid
