#!/usr/bin/env python3
"""Merge gpurun_out/pmc_<tag>/summary.json files (scripts/pmc.sh) into profiles/pmc_summary.json under their key
`mode/probe/n/world`, which is where bench.py looks up roofline.traffic / valu_issue_util / lane_util."""
import json
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
dst = root / "profiles" / "pmc_summary.json"
table = json.loads(dst.read_text()) if dst.exists() else {
    "_comment": "rocprofv3 PMC summaries per bench configuration `mode/probe/n/world` (separate --pmc passes, scripts/pmc.sh): "
                "raw counter means per launch of the dominant kernel, hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB "
                "(gfx950 correction of MI355X_MICROARCH.md), valu_issue_util = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), "
                "lane_util = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)"}
for f in sys.argv[1:]:
    rec = json.loads(Path(f).read_text())
    key = rec.pop("key")
    rec["source"] = f
    table[key] = rec
    print("merged", key, "from", f)
dst.write_text(json.dumps(table, indent=1) + "\n")
