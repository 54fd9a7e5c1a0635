#!/bin/bash
# mean resident waves per SIMD of every kernel of a python script (rocprofv3 --pmc MeanOccupancyPerCU; on this stack the figure is
# waves per SIMD: k_pairs_frac read 1.95 with two 4-wave workgroups per CU resident, 2.8 with three):
#   scripts/occupancy.sh <tag> <script.py> [args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --pmc MeanOccupancyPerCU --output-format csv -d gpurun_out/occ_$tag -- python "$@" > gpurun_out/occ_$tag.log 2>&1 || { echo "rocprofv3 failed"; tail -3 gpurun_out/occ_$tag.log; }
python - "$tag" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/occ_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "MeanOccupancyPerCU":
            d[(r["Kernel_Name"][:70], r["LDS_Block_Size"], r["VGPR_Count"], r["Workgroup_Size"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for (k, lds, vg, wg), v in sorted(d.items(), key=lambda kv: -sum(t for _o, t in kv[1])):
    tot = sum(t for _o, t in v)
    if tot < 2e5: continue
    print(f"{k:70s} launches {len(v):5d}  total {tot / 1e6:8.2f} ms  occupancy {sum(o * t for o, t in v) / tot:5.2f}  LDS {lds:>6} VGPR field {vg:>4} workgroup {wg}")
PY
