"""Host-side mirror of the parts of CrystalEnergyGrids.jl that surround the hot path: RASPA file parsers, force fields and
interaction rules, ProbeSystem, grid geometry, Ewald set-up, the Monte-Carlo energy functions and ``setup_RASPA``.

HARNESS, NOT PRODUCT: these modules are function-for-function Python restatements of the reference's Julia host code
(cited file:line in every docstring, bugs kept on purpose).  They exist because Julia is absent from the build image: they build the
INPUTS of the C ABI for tests, benchmarks and examples the way the reference's own callers would, and they are pinned to the literals of
``test/runtests.jl``.  The product is ``csrc/`` + ``include/ceg_hip.h`` + the thin bindings next to this package (``_abi``, ``plan``,
``grids``, ``interp``, ``energy``, ``distributed``) and ``julia/CEGHip.jl``.  Nothing here is imported by ``oracle/``.
"""
