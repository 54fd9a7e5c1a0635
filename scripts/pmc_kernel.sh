#!/bin/bash
# PMC counters of ONE kernel of any python script: scripts/pmc_kernel.sh <tag> <kernel name substring> <script.py> [args...]
# (separate rocprofv3 --pmc passes, no tracing domains; the program itself follows `--`).  Prints per-launch averages and the
# issue / lane utilisation figures of scripts/pmc.sh; raw CSVs under gpurun_out/pmck_<tag>/.
tag=$1; shift
kern=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmck_$tag
i=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmck_$tag/p$i -- python "$@" > gpurun_out/pmck_$tag/p$i.log 2>&1 || { echo "pass $i ($ctrs) failed"; tail -5 gpurun_out/pmck_$tag/p$i.log; }
done
python - "$tag" "$kern" <<'PY'
import csv, glob, collections, sys
tag, kern = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
for f in glob.glob(f"gpurun_out/pmck_{tag}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
# launches of one script differ (e.g. molecules): report each dispatch position separately
names = sorted(per)
nd = max(len(v) for v in per.values())
for pos in range(nd):
    s = {}
    for n in names:
        d = sorted(per[n])
        if pos < len(d): s[n] = per[n][d[pos]]
    cyc = s.get("GRBM_GUI_ACTIVE", 0) / 8.0
    line = f"launch {pos}: " + "  ".join(f"{n}={s[n]:.4g}" for n in names if n in s)
    print(line)
    if cyc and "SQ_ACTIVE_INST_VALU" in s:
        d = sorted(dur)
        ms = dur[d[pos]] if pos < len(d) else float('nan')
        print(f"   kernel {ms:.3f} ms  clock {cyc / (ms * 1e-3) / 1e9:.3f} GHz  valu_issue_util {4 * s['SQ_ACTIVE_INST_VALU'] / (1024 * cyc):.3f}"
              f"  lane_util {s['SQ_THREAD_CYCLES_VALU'] / (64 * s['SQ_ACTIVE_INST_VALU']):.3f}  lds_pipe_util {4 * s.get('SQ_ACTIVE_INST_LDS', 0) / (256 * cyc):.3f}"
              f"  valu insts per wave {s.get('SQ_INSTS_VALU', 0) / max(1, s.get('SQ_WAVES', 1)):.1f}")
PY
