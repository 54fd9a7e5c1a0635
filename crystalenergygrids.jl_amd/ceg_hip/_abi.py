"""ctypes view of ``include/ceg_hip.h`` (types + function prototypes).

This is the Python twin of the ``ccall`` stubs in ``julia/CEGHip.jl``: plain pointers
and sizes only.  Loading is strict: if ``libceg_hip.so`` is missing the import of the
product path fails loudly -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

RULE_DTYPE = np.dtype([('kind', '<i4'), ('_pad', '<i4'), ('p', '<f8', (3,)), ('shift', '<f8')],
                      align=True)
assert RULE_DTYPE.itemsize == 40

PKG_DIR = Path(__file__).resolve().parent.parent          # crystalenergygrids.jl_amd/
REPO_DIR = PKG_DIR.parent
LIB_PATH = PKG_DIR / "csrc" / "libceg_hip.so"

class GridHeader(C.Structure):
    """``ceg_grid_header_t``"""
    _fields_ = [("spacing", C.c_double), ("dims", C.c_int32 * 3), ("has_mat", C.c_int32),
                ("size", C.c_double * 3), ("shift", C.c_double * 3), ("delta", C.c_double * 3), ("unitcell", C.c_double * 3),
                ("num_unitcell", C.c_int32 * 3), ("_pad", C.c_int32), ("ewald_precision", C.c_double), ("mat", C.c_double * 9)]


c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)

# name -> (restype, argtypes); every symbol include/ceg_hip.h declares
PROTOTYPES = {
    "ceg_abi_version": (C.c_int, []),
    "ceg_device_count": (C.c_int, []),
    "ceg_last_error": (C.c_char_p, []),
    "ceg_grid_vdw": (C.c_int, [
        c_double_p, c_int64_p, C.c_int64, c_double_p, c_double_p,
        C.c_int32, C.c_double, C.c_double,
        C.c_void_p, c_int32_p, C.c_int32,
        c_int32_p, c_double_p, c_double_p, c_double_p,
        C.c_double, C.c_double, c_float_p, C.c_int32]),
    "ceg_grid_coulomb": (C.c_int, [
        c_double_p, c_double_p, C.c_int64, c_double_p, c_double_p,
        C.c_int32, C.c_double, C.c_double, C.c_double,
        c_int32_p, c_double_p, c_double_p, c_double_p,
        C.c_double, C.c_double, c_float_p, C.c_int32]),
    "ceg_grid_vdw_device": (C.c_int, [
        c_double_p, c_int64_p, C.c_int64, c_double_p, c_double_p,
        C.c_int32, C.c_double, C.c_double,
        C.c_void_p, c_int32_p, C.c_int32,
        c_int32_p, c_double_p, c_double_p, c_double_p,
        C.c_double, C.c_double, C.c_void_p, C.c_int32, C.c_int32]),
    "ceg_grid_coulomb_device": (C.c_int, [
        c_double_p, c_double_p, C.c_int64, c_double_p, c_double_p,
        C.c_int32, C.c_double, C.c_double, C.c_double,
        c_int32_p, c_double_p, c_double_p, c_double_p,
        C.c_double, C.c_double, C.c_void_p, C.c_int32, C.c_int32]),
    "ceg_grid_vdw_file": (C.c_int, [
        c_double_p, c_int64_p, C.c_int64, c_double_p, c_double_p,
        C.c_int32, C.c_double, C.c_double,
        C.c_void_p, c_int32_p, C.c_int32,
        c_int32_p, c_double_p, c_double_p, c_double_p,
        C.c_double, C.c_double, c_float_p, C.c_int32,
        C.c_char_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]),
    "ceg_grid_coulomb_file": (C.c_int, [
        c_double_p, c_double_p, C.c_int64, c_double_p, c_double_p,
        C.c_int32, C.c_double, C.c_double, C.c_double,
        c_int32_p, c_double_p, c_double_p, c_double_p,
        C.c_double, C.c_double, c_float_p, C.c_int32,
        C.c_char_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]),
    "ceg_release_cached_buffers": (C.c_int, []),
    "ceg_host_grid_alloc": (c_float_p, [c_int32_p]),
    "ceg_host_grid_free": (C.c_int, [c_float_p]),
    "ceg_plan_create": (C.c_int, [
        C.POINTER(C.c_void_p), C.c_int32,
        c_double_p, c_int64_p, c_double_p, C.c_int64,
        c_double_p, c_double_p, C.c_int32, C.c_double, C.c_double,
        C.c_void_p, c_int32_p, C.c_int32, C.c_double,
        c_int32_p, c_double_p, c_double_p, c_double_p]),
    "ceg_plan_destroy": (C.c_int, [C.c_void_p]),
    "ceg_image_cache_stats": (C.c_int, [c_int64_p, c_int64_p, c_int64_p]),
    "ceg_plan_can_cull": (C.c_int, [C.c_void_p]),
    "ceg_plan_num_images": (C.c_int64, [C.c_void_p]),
    "ceg_plan_copy_images": (C.c_int, [C.c_void_p, c_double_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    "ceg_plan_build_vdw": (C.c_int, [
        C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32,
        C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "ceg_plan_build_coulomb": (C.c_int, [
        C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32,
        C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "ceg_plan_build_fused": (C.c_int, [
        C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double,
        C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
        C.c_int32, C.c_void_p]),
    "ceg_plan_create_multi": (C.c_int, [
        C.POINTER(C.c_void_p), C.c_int32,
        c_double_p, c_int64_p, c_double_p, C.c_int64,
        c_double_p, c_double_p, C.c_int32, C.c_double, C.c_double,
        C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_double,
        c_int32_p, c_double_p, c_double_p, c_double_p]),
    "ceg_plan_num_probes": (C.c_int, [C.c_void_p]),
    "ceg_grids_multi": (C.c_int, [
        c_double_p, c_int64_p, c_double_p, C.c_int64, c_double_p, c_double_p, C.c_int32, C.c_double, C.c_double,
        C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_double,
        c_int32_p, c_double_p, c_double_p, c_double_p,
        C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_void_p), c_float_p, C.c_int32]),
    "ceg_plan_build_multi": (C.c_int, [
        C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double,
        C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "ceg_plan_eval_points": (C.c_int, [
        C.c_void_p, C.c_int32, C.c_int32, c_double_p, C.c_int64, c_double_p]),
    "ceg_interp_create": (C.c_int, [
        C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int32,
        c_int32_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int32]),
    "ceg_interp_create_from_file": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_char_p, C.c_int32, C.c_double, c_double_p, c_double_p, C.c_void_p]),
    "ceg_interp_set_higherorder": (C.c_int, [C.c_void_p, C.c_int32]),
    "ceg_interp_destroy": (C.c_int, [C.c_void_p]),
    "ceg_interp_points": (C.c_int, [C.c_void_p, c_double_p, C.c_int64, c_double_p]),
    "ceg_interp_points_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ceg_scale_grid_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_int32, C.c_void_p]),
    "ceg_recip_create": (C.c_int, [
        C.POINTER(C.c_void_p), C.c_int32, c_int32_p, c_double_p, c_double_p, c_double_p, C.c_int64,
        c_int32_p, c_double_p]),
    "ceg_recip_destroy": (C.c_int, [C.c_void_p]),
    "ceg_recip_layout": (C.c_int, [c_int32_p, C.c_int64, c_int32_p, c_int32_p, c_int32_p, C.c_void_p, C.c_void_p]),
    "ceg_recip_energy": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int32, C.c_int64, C.c_double, C.c_double, c_double_p]),
    "ceg_block_from_grid": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, c_int32_p, C.c_double, C.c_void_p]),
    "ceg_block_spheres": (C.c_int, [C.c_int32, c_int32_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int32, C.c_double,
                                    c_double_p, c_double_p, C.c_int32, C.c_void_p]),
    "ceg_recip_set_structure_factor": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "ceg_pairs_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, c_double_p, c_double_p, C.c_double, C.c_void_p, c_int32_p,
                                   C.c_int32, C.c_double]),
    "ceg_pairs_destroy": (C.c_int, [C.c_void_p]),
    "ceg_pairs_set_atoms": (C.c_int, [C.c_void_p, c_double_p, c_int32_p, c_int32_p, C.c_int64]),
    "ceg_pairs_energy": (C.c_int, [C.c_void_p, c_double_p, c_int32_p, C.c_int32, C.c_int64, C.c_int32, c_double_p]),
    "ceg_pairs_energy_device": (C.c_int, [C.c_void_p, C.c_void_p, c_int32_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "ceg_pairs_neighbour_cells": (C.c_int, [C.c_void_p, c_int32_p]),
    "ceg_mc_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.c_void_p, c_double_p, C.c_int32, c_double_p,
                                c_double_p, C.c_double, C.c_void_p, c_int32_p, C.c_double, c_int32_p, c_double_p, c_double_p, c_double_p,
                                C.c_int64, c_int32_p, c_double_p]),
    "ceg_mc_destroy": (C.c_int, [C.c_void_p]),
    "ceg_mc_set_guests": (C.c_int, [C.c_void_p, c_double_p, c_int32_p, c_int32_p, C.c_int32]),
    "ceg_mc_trial": (C.c_int, [C.c_void_p, C.c_int32, c_double_p, C.c_int64, c_double_p]),
    "ceg_mc_trial_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ceg_mc_trial_insert_device": (C.c_int, [C.c_void_p, c_int32_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ceg_mc_accept": (C.c_int, [C.c_void_p, C.c_int32, c_double_p]),
    "ceg_mc_get_state": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "ceg_mc_neighbour_cells": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p]),
    "ceg_mc_trial_insert": (C.c_int, [C.c_void_p, c_int32_p, C.c_int32, c_double_p, C.c_int64, c_double_p]),
    "ceg_mc_insert": (C.c_int, [C.c_void_p, c_int32_p, C.c_int32, c_double_p, c_int32_p]),
    "ceg_mc_remove": (C.c_int, [C.c_void_p, C.c_int32, c_int32_p]),
    "ceg_recip_energy_device": (C.c_int, [C.c_void_p, C.c_void_p, c_double_p, C.c_int32, C.c_int64, C.c_double, C.c_double,
                                          C.c_void_p, C.c_void_p]),
}

ALGO_AUTO, ALGO_BRUTEFORCE, ALGO_CULLED = 0, 1, 2


class CegError(RuntimeError):
    """Non-zero status from libceg_hip (message from ``ceg_last_error``)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"libceg_hip error {code}: {msg}")
        self.code = code


_lib = None


def load_library(path: os.PathLike | None = None) -> C.CDLL:
    """dlopen ``libceg_hip.so`` and bind every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path is not None else Path(os.environ.get("CEG_HIP_LIB", LIB_PATH))
    if not p.exists():
        raise ImportError(
            f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the grid build.")
    if "torch" not in sys.modules and not os.environ.get("CEG_HIP_NO_TORCH"):
        # PyTorch ships its own HIP runtime; when this package hands device memory to / takes it from torch (plan builds into
        # tensors, ceg_grid_*_device, torch.distributed ranks) that runtime must be the one the process loads first, or torch's
        # lazy initialisation later finds no device.  A pure C-ABI user without torch is unaffected.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(str(p))
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.ceg_abi_version() != 1:
        raise ImportError("libceg_hip.so ABI version mismatch")
    if path is None:
        _lib = lib
    return lib


def check(lib: C.CDLL, code: int) -> None:
    if code != 0:
        raise CegError(code, (lib.ceg_last_error() or b"").decode())


def dptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(c_double_p)


def i32ptr(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(c_int32_p)


def i64ptr(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags.c_contiguous
    return a.ctypes.data_as(c_int64_p)


def fptr(a: np.ndarray):
    assert a.dtype == np.float32
    return a.ctypes.data_as(c_float_p)
