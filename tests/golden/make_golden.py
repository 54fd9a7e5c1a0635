#!/usr/bin/env python3
"""Generate tests/golden/samples_*.npz: sampled outputs of the CPU oracle on the reference's
fixtures (the Julia reference cannot run here and ships no .grid file, so these vectors come
from oracle/ceg_oracle.c, which tests/test_reference_pins.py pins to the reference's literals).

Per case: grid indices (i,j,k), cartesian points, FP64 8-vectors of compute_derivatives_*
(probes.jl:71-117) and the Float32 8-vectors _set_gridpoint! stores (grids.jl:118-135).
Usage: python tests/golden/make_golden.py   (deterministic; rewrites the .npz files)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "crystalenergygrids.jl_amd"), str(ROOT)]

from ceg_hip import grids as G, workloads as W   # noqa: E402
from oracle import oracle as O                   # noqa: E402

CASES = {
    # name: (framework, spacing)   -- BASELINE.json configs 1/2 and the triclinic supercell case
    "cha_0.5": ("CHA_1.4_3b4eeb96", 0.5),
    "cha_0.1": ("CHA_1.4_3b4eeb96", 0.1),
    "cit7_0.15": ("CIT-7", 0.15),
}
NSAMPLE = 384


def sample_indices(w, rng):
    nx, ny, nz = w.cset.npoints
    idx = np.stack([rng.integers(0, nx, NSAMPLE), rng.integers(0, ny, NSAMPLE), rng.integers(0, nz, NSAMPLE)], axis=1)
    # corners / faces (partial tiles of the culled kernel) and the grid points nearest to some atoms
    extra = [(0, 0, 0), (nx - 1, ny - 1, nz - 1), (nx - 1, 0, nz - 2), (1, ny - 1, 0), (nx // 2, ny // 2, nz // 2)]
    for a in w.framework.position[:: max(1, len(w.framework) // 24)]:
        g = np.round((a - w.cset.shift) / w.cset.delta).astype(int)
        if np.all(g >= 0) and np.all(g < (nx, ny, nz)):
            extra.append(tuple(g))
    return np.unique(np.concatenate([idx, np.array(extra)]), axis=0).astype(np.int32)


def main():
    rng = np.random.default_rng(20241008)
    for name, (fwname, spacing) in CASES.items():
        out = {}
        for atom in ("Ar", "Na"):
            w = W.fixture_workload(fwname, atom, spacing)
            if "idx" not in out:
                out["idx"] = sample_indices(w, rng)
                i, j, k = out["idx"].T
                out["points"] = np.stack([i * w.cset.size[0] / w.cset.dims[0] + w.cset.shift[0],
                                          j * w.cset.size[1] / w.cset.dims[1] + w.cset.shift[1],
                                          k * w.cset.size[2] / w.cset.dims[2] + w.cset.shift[2]], axis=1)
                out["dims"] = w.cset.dims
            raw = O.points_vdw(w.probe_vdw, out["points"])
            lam, thr = G.vdw_scaling()
            out[f"raw_vdw_{atom}"] = raw
            out[f"f32_vdw_{atom}"] = O.set_gridpoints(raw, w.cset.delta, lam, thr)
        raw = O.points_coulomb(w.probe_coulomb, w.alpha, out["points"])
        lam, thr = G.coulomb_scaling()
        out["raw_coulomb"] = raw
        out["f32_coulomb"] = O.set_gridpoints(raw, w.cset.delta, lam, thr)
        path = Path(__file__).parent / f"samples_{name}.npz"
        np.savez_compressed(path, **out)
        nspecial = int((np.abs(out["f32_vdw_Na"][:, 0]) >= 1.9e7).sum())
        print(f"{path.name}: {len(out['idx'])} points, dims {tuple(out['dims'])}, {nspecial} clamped Na points")


def headers():
    """grid_headers.json: the bytes create_grid_vdw / create_grid_coulomb put around the payload
    (grids.jl:108-116 header, :154/:182 trailer, :180 Ewald precision) for the fixtures' default grids."""
    import io
    import json
    from ceg_hip.hostmirror.utils import find_supercell
    out = {}
    for fwname, spacing in (("CHA_1.4_3b4eeb96", 0.15), ("CHA_1.4_3b4eeb96", 0.5), ("CIT-7", 0.15)):
        w = W.fixture_workload(fwname, "Ar", spacing, coulomb=False)
        nuc = tuple(int(x) for x in find_supercell(w.framework.mat, 12.0))
        empty = np.empty((8, 0), dtype=np.float32)
        for kind, prec in (("vdw", None), ("coulomb", 1e-6)):
            buf = io.BytesIO()
            G._create_grid_common(buf, w.cset, nuc)
            head = buf.getvalue() + (np.float64(prec).tobytes() if prec is not None else b"")
            tail = np.asarray(w.cset.cell.mat, dtype="<f8").T.tobytes()
            out[f"{fwname}/{spacing}/{kind}"] = {"header_hex": head.hex(), "trailer_hex": tail.hex(), "dims": [int(d) for d in w.cset.dims],
                                                 "num_unitcell": list(nuc), "payload_bytes": int(32 * np.prod(np.asarray(w.cset.dims) + 1))}
    (Path(__file__).parent / "grid_headers.json").write_text(json.dumps(out, indent=1) + "\n")
    print("grid_headers.json:", ", ".join(out))


if __name__ == "__main__":
    main()
    headers()
