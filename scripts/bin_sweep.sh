#!/bin/bash
# kernel time of the roofline workload vs the z / xy edge of the image bins (CEG_HIP_BIN_Z, CEG_HIP_BIN_XY read at plan creation)
for z in 4.5 3.0 2.25 1.5 1.0 0.75; do
  echo "== CEG_HIP_BIN_Z=$z"
  CEG_HIP_BIN_Z=$z python scripts/time_roofline.py 3 2>&1 | grep -E "Ar fused|Na fused|Ar vdw" | grep -v erfcx
done
for xy in 3.5 5.5; do
  echo "== CEG_HIP_BIN_XY=$xy CEG_HIP_BIN_Z=1.5"
  CEG_HIP_BIN_XY=$xy CEG_HIP_BIN_Z=1.5 python scripts/time_roofline.py 3 2>&1 | grep -E "Ar fused|Na fused|Ar vdw" | grep -v erfcx
done
