#!/bin/bash
# The large-batch path of ceg_mc_trial alone (65 536 placements of one CO2 among 64 in CHA: tests/perf/time_mc.py with
# CEG_TIME_MC_ONLY_BIG=1): rocprofv3 --kernel-trace durations of the three wave-per-placement kernels, then the PMC passes of
# scripts/pmc_kernel.sh (VALU issue / lane utilisation per kernel) -> gpurun_out/consumers/mc_big.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/consumers
mkdir -p $out
export CEG_TIME_MC_ONLY_BIG=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/mc_big_trace -- python tests/perf/time_mc.py > $out/mc_big_run.log 2>&1 || echo "trace run failed"
bash scripts/pmc_kernel.sh mcbig k_mcw tests/perf/time_mc.py > $out/mc_big_pmc.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys, re, collections
out = sys.argv[1]
spans = collections.defaultdict(list)
for f in glob.glob(f"{out}/mc_big_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mcw" in r["Kernel_Name"]:
            spans[re.search(r"k_mcw_\w+", r["Kernel_Name"]).group(0)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pmc = collections.defaultdict(list)
lines = open(f"{out}/mc_big_pmc.log").read().splitlines()
# pmc_kernel.sh prints the launches in dispatch order: frame, ewald, pairs, frame, ...
order = ["k_mcw_frame", "k_mcw_ewald", "k_mcw_pairs_frac" if "k_mcw_pairs_frac" in spans else "k_mcw_pairs"]
k = 0
for i, l in enumerate(lines):
    if l.startswith("launch "):
        m = re.search(r"valu_issue_util ([\d.]+)\s+lane_util ([\d.]+)\s+lds_pipe_util ([\d.]+)\s+valu insts per wave ([\d.]+)", lines[i + 1]) if i + 1 < len(lines) else None
        if m:
            pmc[order[k % 3]].append(tuple(float(x) for x in m.groups()))
        k += 1
fl = 65537 * (3 * 1368 * 16.0 + 1368 * 10.0 + 3 * 51 * 40.0 + 3 * 189 * 47.0 + 60 * 72.0 + 1500.0)
with open(f"{out}/mc_big.txt", "w") as fh:
    fh.write("# ceg_mc_trial at batch 65 536 alone (CEG_TIME_MC_ONLY_BIG=1 python tests/perf/time_mc.py): rocprofv3 --kernel-trace durations and, from separate\n")
    fh.write("# --pmc passes (scripts/pmc_kernel.sh), VALU issue utilisation = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), lane utilisation =\n")
    fh.write("# SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU), LDS pipe = 4 SQ_ACTIVE_INST_LDS / (256 x cycles)\n")
    fh.write("kernel | launches | avg us | min us | max us | VALU issue util | lane util | LDS pipe util | VALU insts per wave\n")
    tot = 0.0; wsum = 0.0
    for kname in order:
        d = spans.get(kname, [])
        if not d:
            continue
        avg = sum(d) / len(d)
        p = pmc.get(kname, [])
        iu = sum(x[0] for x in p) / len(p) if p else float("nan")
        lu = sum(x[1] for x in p) / len(p) if p else float("nan")
        ld = sum(x[2] for x in p) / len(p) if p else float("nan")
        vi = sum(x[3] for x in p) / len(p) if p else float("nan")
        fh.write(f"{kname} | {len(d)} | {avg:.1f} | {min(d):.1f} | {max(d):.1f} | {iu:.3f} | {lu:.3f} | {ld:.3f} | {vi:.0f}\n")
        tot += avg; wsum += avg * iu
    if tot:
        fh.write(f"sum of the three kernels: {tot:.1f} us per batch of 65 536 + 1 rows; time-weighted VALU issue utilisation {wsum / tot:.3f}\n")
        fh.write(f"roofline movement_energy, batch 65536, KERNELS: {fl / 1e9:.2f} Gflop (118 kflop per placement, the work model of tests/perf/time_mc.py) / {tot:.1f} us = "
                 f"{fl / (tot * 1e-6) / 1e12:.2f} TFLOP/s = {fl / (tot * 1e-6) / 78.6e12:.3f} of the FP64 vector peak\n")
    fh.write("\n# the timing script\n" + "".join(l + "\n" for l in open(f"{out}/mc_big_run.log").read().splitlines() if l.startswith(("GPU", "#"))))
print(open(f"{out}/mc_big.txt").read())
PY
