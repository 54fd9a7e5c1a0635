#!/bin/bash
# A/B of library builds on one GPU box: kernel times of the roofline workload for the in-tree library and every
# build_variants/*.so (CEG_HIP_LIB selects the library).  usage: scripts/ab_variants.sh [reps]
reps=${1:-5}
shopt -s nullglob
[ -x build_variants/seed_accuracy ] && build_variants/seed_accuracy
for lib in crystalenergygrids.jl_amd/csrc/libceg_hip.so build_variants/*.so; do
  echo "=== $lib"
  CEG_HIP_LIB=$PWD/$lib python scripts/time_roofline.py $reps 2>&1 | grep -v amdgpu.ids | grep -v "erfcx"
done
