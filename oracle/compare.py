"""Grid comparison used by the parity tests, smoke() and bench.py's self-check (test
infrastructure).  Tolerance: north_star's 1e-6 relative on every stored value, NaN / Inf /
sentinel (|x| >= 1.9e7 file units, i.e. the 2e7 clamp of grids.jl:120-125) patterns must be
IDENTICAL.  A tiny absolute floor (FLOOR x the channel's median magnitude; 1e-11 since round 3,
1e-9 before) covers the rare point where positive and negative pair terms cancel to ~1e-10 of
their size, so that the FP64 summation order becomes visible.  Channel 0 (the energy north_star's
tolerance is quoted on) can be checked with NO floor: see `floor0`."""
from __future__ import annotations

import numpy as np

import os

RTOL = 1e-6
SENTINEL = 1.9e7
FLOOR = float(os.environ.get("CEG_COMPARE_FLOOR", "1e-11"))


def compare_grids(got: np.ndarray, ref: np.ndarray, what: str = "grid", rtol: float = RTOL, sentinel: float = SENTINEL,
                  floor: float = None, floor0: float = None):
    """got/ref: float arrays with the channel as FIRST axis.  Returns the max relative error over
    regular points; raises AssertionError with a diagnostic otherwise.  `floor` (default FLOOR) is the
    absolute allowance as a fraction of the channel's median magnitude, `floor0` the one for channel 0 (default: the
    same; the fixture-grid tests pass 0.0 -- every stored energy within rtol of the oracle's, no allowance at all.
    Synthetic systems with +q / -q atoms at equal distances from a grid point cannot: the oracle's wrap arithmetic
    leaves 1e-17 of a term there where the image-list arithmetic gives an exact 0)."""
    if floor is None:
        floor = FLOOR
    if floor0 is None:
        floor0 = floor
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    assert np.array_equal(nan_g, nan_r), f"{what}: NaN pattern differs at {int((nan_g != nan_r).sum())} values"
    special = (~np.isfinite(ref)) | (np.abs(ref) >= sentinel)
    spec_cmp = special & ~nan_r
    assert np.array_equal(got[spec_cmp], ref[spec_cmp]), \
        f"{what}: Inf/sentinel values differ at {int((got[spec_cmp] != ref[spec_cmp]).sum())} values"
    worst = 0.0
    for c in range(ref.shape[0]):
        m = ~special[c]
        if not m.any():
            continue
        g = got[c][m].astype(np.float64)
        r = ref[c][m].astype(np.float64)
        assert np.all(np.isfinite(g)), f"{what}: channel {c} has non-finite values where the oracle is finite"
        scale = float(np.median(np.abs(r)))
        diff = np.abs(g - r)
        tol = rtol * np.abs(r) + (floor0 if c == 0 else floor) * scale
        bad = diff > tol
        if bad.any():
            q = int(np.argmax(diff - tol))
            raise AssertionError(f"{what}: channel {c}: {int(bad.sum())} of {bad.size} values off; worst got {g[q]!r} "
                                 f"ref {r[q]!r} (rel {diff[q] / max(abs(r[q]), 1e-300):.3e})")
        nz = np.abs(r) > 1e-6 * scale
        if nz.any():
            worst = max(worst, float(np.max(diff[nz] / np.abs(r[nz]))))
    return worst
