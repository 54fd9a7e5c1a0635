"""ceg_hip -- host-side mirror of the CrystalEnergyGrids.jl grid-build interface on top of
``libceg_hip.so`` (hand-written HIP kernels for MI355X / gfx950).

Same names and argument meaning as the reference's public functions for this path
(``setup_RASPA``, ``create_grid_vdw``, ``create_grid_coulomb``, ``parse_grid``,
``interpolate_grid``, ``energy_point`` ...).  The compute path is the HIP library only;
importing this package does not load it, the first grid build does and fails loudly if it
is not built.

Modules beyond the re-exports below: ``plan`` (resident-plan API, device buffers), ``distributed``
(x-slab / block-cyclic sharding over ranks), ``interp`` / ``energy`` (batched GPU consumers of the
grids: interpolation, reciprocal Ewald, pair energies, ``GpuEnergySetup``, ``GpuMonteCarloEnergy``),
``montecarlo`` (host mirror of the reference's MC energy functions, used to pin them), ``workloads``
(the BASELINE.json configurations).
"""
from .constants import GRID_TO_KELVIN, COULOMBIC_CONVERSION_FACTOR
from .interactions import FF, Mixing, InteractionRule, InteractionRuleSum, UndefinedInteractionError
from .forcefields import ForceField, build_forcefield
from .coordinates import CellMatrix, GridCoordinatesSetup, abc_to_xyz, offsetpoint
from .probes import ProbeSystem
from .ewald import EwaldFramework, initialize_ewald, compute_ewald
from .raspa import (setdir_RASPA, getdir_RASPA, parse_pseudoatoms_RASPA, parse_forcefield_RASPA,
                    load_framework_RASPA, load_molecule_RASPA, setup_probe_RASPA, RASPASystem)
from .grids import (EnergyGrid, CrystalEnergySetup, create_grid_vdw, create_grid_coulomb, parse_grid,
                    interpolate_grid, energy_point, build_vdw_array, build_coulomb_array)
from .setup_raspa import setup_RASPA, retrieve_or_create_grid
