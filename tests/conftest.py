"""pytest configuration: `gpu` marker, import paths, fixture data directory."""
import os
import sys
from pathlib import Path

import pytest

# torch ships its own HIP runtime: when it is first imported AFTER libceg_hip.so (linked against /opt/rocm) has initialised
# the device in the same process, its lazy CUDA initialisation reports "No HIP GPUs are available" (observed when
# tests/test_gpu_parity.py runs on its own).  Importing it up front keeps the suite independent of the collection order.
try:
    import torch  # noqa: F401
except ImportError:           # the CPU-only parts of the suite do not need it
    pass

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT / "crystalenergygrids.jl_amd"), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _raspa_dir():
    import ceg_hip
    ceg_hip.setdir_RASPA(GOLDEN / "raspa")
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def forcefield():
    import ceg_hip
    return ceg_hip.parse_forcefield_RASPA("BoulfelfelSholl2021")


@pytest.fixture(scope="session")
def hip_lib():
    """libceg_hip.so with a visible device; GPU tests fail (not skip) if it is missing."""
    from ceg_hip import _abi
    lib = _abi.load_library()
    assert lib.ceg_device_count() >= 1, "no HIP device visible: GPU parity tests cannot run"
    return lib
