#!/bin/bash
# usage: scripts/pmc.sh <tag> <pmc key: mode/probe/n/world> <bench args...>   -- PMC passes (separate runs, no tracing domains)
# Writes gpurun_out/pmc_<tag>/summary.json; merge it into profiles/pmc_summary.json under <pmc key> with scripts/pmc_merge.py.
tag=$1; shift
key=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$tag
i=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag/p$i -- python bench.py --steps 2 --warmup 1 --cpu-rows 0 --no-check "$@" > gpurun_out/pmc_$tag/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/pmc_$tag/p$i.log; }
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_$tag/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_culled" in r["Kernel_Name"] or "k_brute" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
import json
summ = {k: sum(v)/len(v) for k, v in acc.items()}
for k in sorted(summ):
    print(f"{k:28s} n={len(acc[k])} mean={summ[k]:.6g}")
if "FETCH_SIZE" in summ and "WRITE_SIZE" in summ:
    # rocprofv3 units: KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
    # (MI355X_MICROARCH.md, HBM section) -> x2
    summ["hbm_bytes_per_launch"] = (2.0 * summ["FETCH_SIZE"] + summ["WRITE_SIZE"]) * 1024.0
    print("hbm_bytes_per_launch", summ["hbm_bytes_per_launch"])
# convention-free utilisation figures (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs)
if "GRBM_GUI_ACTIVE" in summ and "SQ_ACTIVE_INST_VALU" in summ:
    cycles = summ["GRBM_GUI_ACTIVE"] / 8.0
    summ["valu_issue_util"] = 4.0 * summ["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles)          # 256 CU x 4 SIMD
    summ["valu_issue_util_by_count"] = 4.0 * summ.get("SQ_INSTS_VALU", 0.0) / (1024.0 * cycles)
    summ["lane_util"] = summ["SQ_THREAD_CYCLES_VALU"] / (64.0 * summ["SQ_ACTIVE_INST_VALU"])
    print("valu_issue_util", summ["valu_issue_util"], "lane_util", summ["lane_util"])
summ["key"] = "$key"
summ["command"] = "python bench.py --steps 2 --warmup 1 --cpu-rows 0 --no-check $*"
json.dump(summ, open("gpurun_out/pmc_$tag/summary.json", "w"), indent=1)
PY
