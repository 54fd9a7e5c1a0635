#!/usr/bin/env python3
"""Registers / scratch / LDS / occupancy of every k_culled variant (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
    python scripts/resource_usage.py [extra -D flags]"""
import re, subprocess, sys
from pathlib import Path
csrc = Path(__file__).resolve().parent.parent / "crystalenergygrids.jl_amd" / "csrc"
flags = re.search(r"CXXFLAGS\s*=\s*(.*?)\nSRCS", (csrc / "Makefile").read_text(), re.S).group(1).replace("\\\n", " ").replace("$(ARCH)", "gfx950").split()
out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, *sys.argv[1:], "-c", "-o", "/dev/null", "ceg_kernels.hip", "-Rpass-analysis=kernel-resource-usage"],
                     cwd=csrc, capture_output=True, text=True).stderr
MODE = {0: "vdw", 1: "coulomb", 2: "fused"}
rec = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        rec = {"name": m.group(1)}
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m:
            rec[key] = int(m.group(1))
    if "lds" in rec and "name" in rec:
        m = re.match(r"_ZN3ceg8k_culledILi(\d)ELb(\d)ELi(\d)ELi(\d)ELi(\d)EE", rec["name"])
        if m:
            mode, pts, vdwk, ewk, npr = (int(x) for x in m.groups())
            print(f"k_culled<{MODE[mode]:7s} {'points' if pts else 'grid  '} VDWK={vdwk} EWK={ewk} NP={npr}>  VGPR {rec.get('vgpr'):3d}  SGPR {rec.get('sgpr'):3d}  "
                  f"scratch {rec.get('scratch'):3d} B/lane  spills S {rec.get('sspill'):3d} V {rec.get('vspill'):3d}  LDS {rec.get('lds'):6d} B  occupancy {rec.get('occ')}")
        rec = {}
