#!/bin/bash
# usage: scripts/kernel_stats.sh <tag> <bench args...>
# rocprofv3 --kernel-trace of `python bench.py <args>`; per-kernel statistics of the TIMED launches only (the warm-up
# launches of bench.py are dropped, so the average is comparable with ms_per_step) -> gpurun_out/stats_<tag>/summary.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/stats_$tag
mkdir -p $out
warm=2; steps=20
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python bench.py --steps $steps --warmup $warm --cpu-rows 0 --no-check --no-oneshot "$@" > $out/bench.json 2> $out/bench.err || { echo "rocprofv3 failed"; tail -5 $out/bench.err; }
python - "$out" "$warm" "$steps" "$*" <<'PY'
import csv, glob, sys, json, collections
out, warm, steps, args = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rows = collections.defaultdict(list)
for f in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
with open(out + "/summary.csv", "w") as fh:
    fh.write(f"# rocprofv3 --kernel-trace -- python bench.py --steps {steps} --warmup {warm} --cpu-rows 0 --no-check --no-oneshot {args}; warm-up launches excluded for the grid-build kernels\n")
    fh.write("kernel,calls,dropped_warmup,avg_ms,min_ms,max_ms,total_ms\n")
    for name, spans in sorted(rows.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        spans.sort()
        drop = 0
        if "k_culled" in name or "k_bruteforce" in name:
            prewarm = 3                                   # bench.py's PREWARM_STEPS (untimed set-up launches)
            per_step = max(1, len(spans) // (prewarm + warm + steps))
            drop = (prewarm + warm) * per_step
        d = [(e - s) / 1e6 for s, e in spans[drop:]]
        if not d:
            continue
        fh.write(f"\"{name}\",{len(d)},{drop},{sum(d)/len(d):.4f},{min(d):.4f},{max(d):.4f},{sum(d):.3f}\n")
print(open(out + "/summary.csv").read())
try:
    b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
    print("bench ms_per_step", b["ms_per_step"], "kernel_ms", b["roofline"]["kernel_ms"], "frac", b["roofline"]["frac"])
except Exception as e:
    print("no bench line:", e)
PY
