#!/bin/bash
# rocprofv3 kernel statistics + HBM byte counters (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the grid-consumer kernels
# (rows f1-f3 and the fused Monte-Carlo trial kernel), driven by the tests/perf timing scripts -> gpurun_out/consumers/<name>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/consumers
mkdir -p $out
names=${@:-interp recip pairs mc}
for name in $names; do
  script=tests/perf/time_$name.py
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${name}_trace -- python $script > $out/${name}_run.log 2>&1 || echo "$name: trace run failed"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $ctr --output-format csv -d $out/${name}_$ctr -- python $script > $out/${name}_$ctr.log 2>&1 || echo "$name: $ctr pass failed"
  done
  python - "$out" "$name" <<'PY'
import csv, glob, sys, collections
out, name = sys.argv[1], sys.argv[2]
spans = collections.defaultdict(list)
for f in glob.glob(f"{out}/{name}_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        spans[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ctr = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{name}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    ctr[c] = acc
with open(f"{out}/{name}.txt", "w") as fh:
    fh.write(f"# rocprofv3 --kernel-trace and --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python tests/perf/time_{name}.py`\n")
    fh.write("# bytes: WRITE_SIZE KB x 1024; FETCH_SIZE KB x 1024 x 2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md; gathers are uncalibrated)\n")
    fh.write("kernel | launches | avg us | min us | max us | FETCH_SIZE KB/launch (raw) | WRITE_SIZE KB/launch | GB/s at the average duration (2 x fetch + write)\n")
    for k, d in sorted(spans.items(), key=lambda kv: -sum(kv[1])):
        if sum(d) < 50 or k.startswith("__amd"):
            continue
        f = ctr["FETCH_SIZE"].get(k); w = ctr["WRITE_SIZE"].get(k)
        fm = sum(f) / len(f) if f else float("nan"); wm = sum(w) / len(w) if w else float("nan")
        avg = sum(d) / len(d)
        gbs = (2 * fm + wm) * 1024 / (avg * 1e-6) / 1e9
        fh.write(f"{k[:100]} | {len(d)} | {avg:.1f} | {min(d):.1f} | {max(d):.1f} | {fm:.1f} | {wm:.1f} | {gbs:.1f}\n")
    fh.write("\n# output of the timing script\n")
    fh.write("".join(l for l in open(f"{out}/{name}_run.log") if not l.startswith(("W2", "E2", "I2")) and "amdgpu.ids" not in l))
print(open(f"{out}/{name}.txt").read())
PY
done
