#!/usr/bin/env python3
"""Writes tests/golden/coeff.json: the 64x64 integer matrix ``COEFF`` the reference holds as a literal in
src/constants.jl:24-89 (the tricubic-interpolation coefficient matrix ``interpolate_grid`` multiplies the 64 corner
values with, grids.jl:252).  The file is DATA read out of the reference's literal -- rows of integers -- and is the golden
vector the two derivations in this repository (oracle/hostlogic.py: tensor product of the 1-D Hermite matrix;
ceg_hip/constants.py: rational inverse of the evaluation matrix) are pinned to by tests/test_oracle_hostlogic.py.

Run in the build container only (the reference tree does not travel to the GPU box):
    python tests/golden/make_coeff.py [/root/reference]
"""
import json
import re
import sys
from pathlib import Path

ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
text = (ref / "src" / "constants.jl").read_text().splitlines()
start = next(i for i, l in enumerate(text) if l.startswith("const COEFF"))
rows = []
for n, line in enumerate(text[start + 1:], start + 2):
    body = line.strip()
    if body.startswith("]"):
        end = n
        break
    body = body.rstrip(";").strip()
    if body:
        rows.append([int(tok) for tok in body.split()])
assert len(rows) == 64 and all(len(r) == 64 for r in rows), (len(rows), {len(r) for r in rows})
out = {"source": f"src/constants.jl:{start + 1}-{end}", "rows": rows}
dst = Path(__file__).with_name("coeff.json")
dst.write_text(json.dumps(out, separators=(",", ":")).replace("],[", "],\n[") + "\n")
print(f"{dst}: 64 x 64, {sum(v != 0 for r in rows for v in r)} non-zero entries, source {out['source']}")
