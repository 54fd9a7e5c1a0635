#!/bin/bash
# usage: scripts/pmc.sh <tag> <pmc key: mode/probe/n/world> <bench args...>   -- PMC passes (separate runs, no tracing domains)
# Writes gpurun_out/pmc_<tag>/summary.json; merge it into profiles/pmc_summary.json under <pmc key> with scripts/pmc_merge.py.
# Only the TIMED launches of bench.py count (its 3 set-up launches and the warm-up launch are dropped, like scripts/kernel_stats.sh
# does for the durations), and the summary records which library it was taken on (sha256 of the kernel sources, git commit, host).
tag=$1; shift
key=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$tag
steps=3; warm=1
i=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
            "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag/p$i -- python bench.py --steps $steps --warmup $warm --cpu-rows 0 --no-check --no-oneshot "$@" > gpurun_out/pmc_$tag/p$i.log 2>&1 || { echo "pass $i ($ctrs) failed"; tail -5 gpurun_out/pmc_$tag/p$i.log; }
done
python - "$tag" "$key" "$steps" "$warm" "$*" <<'PY'
import csv, glob, collections, hashlib, json, os, socket, subprocess, sys
tag, key, steps, warm, args = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
PREWARM = 3                                           # bench.py's PREWARM_STEPS
per_counter = collections.defaultdict(list)           # counter -> [(dispatch id, value)]
durations = {}
for f in glob.glob(f"gpurun_out/pmc_{tag}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_culled" in r["Kernel_Name"] or "k_brute" in r["Kernel_Name"]:
            per_counter[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":          # duration of the launch in the pass that counts the cycles
                durations[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
summ, used = {}, {}
for name, rows in per_counter.items():
    # one row per (dispatch, counter) -- or several (per XCD / SE) that must be summed per dispatch
    by_disp = collections.defaultdict(float)
    for d, v in rows:
        by_disp[d] += v
    disp = sorted(by_disp)
    per_step = max(1, len(disp) // (PREWARM + warm + steps))
    timed = disp[(PREWARM + warm) * per_step:]
    vals = [by_disp[d] for d in timed]
    summ[name] = sum(vals) / len(vals) * per_step     # per bench step (= per launch at N = 1)
    used[name] = len(vals)
for k in sorted(summ):
    print(f"{k:28s} timed launches {used[k]}  per step {summ[k]:.6g}")
if "FETCH_SIZE" in summ and "WRITE_SIZE" in summ:
    # rocprofv3 units: KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
    # (MI355X_MICROARCH.md, HBM section) -> x2
    summ["hbm_bytes_per_launch"] = (2.0 * summ["FETCH_SIZE"] + summ["WRITE_SIZE"]) * 1024.0
    print("hbm_bytes_per_launch", summ["hbm_bytes_per_launch"])
# convention-free utilisation figures (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs)
if "GRBM_GUI_ACTIVE" in summ and "SQ_ACTIVE_INST_VALU" in summ:
    cycles = summ["GRBM_GUI_ACTIVE"] / 8.0
    summ["gpu_cycles_per_launch"] = cycles
    if durations:
        disp = sorted(durations)
        per_step = max(1, len(disp) // (PREWARM + warm + steps))
        timed = disp[(PREWARM + warm) * per_step:]
        summ["kernel_ms_in_profile"] = sum(durations[d] for d in timed) / len(timed) * per_step
        summ["effective_clock_ghz"] = cycles / (summ["kernel_ms_in_profile"] * 1e-3) / 1e9
        print("kernel_ms (cycle-counting pass)", summ["kernel_ms_in_profile"], "effective clock GHz", summ["effective_clock_ghz"])
    summ["valu_issue_util"] = 4.0 * summ["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles)          # 256 CU x 4 SIMD
    summ["valu_issue_util_by_count"] = 4.0 * summ.get("SQ_INSTS_VALU", 0.0) / (1024.0 * cycles)
    summ["lane_util"] = summ["SQ_THREAD_CYCLES_VALU"] / (64.0 * summ["SQ_ACTIVE_INST_VALU"])
    summ["frac_issue"] = summ["valu_issue_util"] * summ["lane_util"]
    print("valu_issue_util", summ["valu_issue_util"], "lane_util", summ["lane_util"], "frac_issue", summ["frac_issue"])
f64 = [summ.get("SQ_INSTS_VALU_%s_F64" % c) for c in ("ADD", "MUL", "FMA", "TRANS")]
if all(v is not None for v in f64) and "lane_util" in summ:
    add, mul, fma, trans = f64
    # wave-level instruction counts x 64 lanes x the measured share of active lanes; an FMA is two flops
    summ["fp64_insts_per_launch"] = add + mul + fma + trans
    summ["fp64_flops_per_launch_all_lanes"] = (add + mul + trans + 2.0 * fma) * 64.0
    summ["fp64_flops_per_launch"] = summ["fp64_flops_per_launch_all_lanes"] * summ["lane_util"]
    summ["fma_share_of_fp64_insts"] = fma / max(1.0, add + mul + fma + trans)
    print("fp64 instructions per launch", summ["fp64_insts_per_launch"], " executed flops (x lane_util)", summ["fp64_flops_per_launch"])
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
import bench                                              # the recipe lives in one place: bench.csrc_sha256 (GRID_KERNEL_SOURCES)
summ["csrc_sha256"] = bench.csrc_sha256()
summ["host"] = socket.gethostname()
summ["timed_launches_per_pass"] = steps
summ["key"] = key
summ["command"] = f"python bench.py --steps {steps} --warmup {warm} --cpu-rows 0 --no-check --no-oneshot {args}"
json.dump(summ, open(f"gpurun_out/pmc_{tag}/summary.json", "w"), indent=1)
PY
