#!/bin/bash
# time bench.py with every library variant in build_variants/ (plus the in-tree one)
mkdir -p gpurun_out
for lib in crystalenergygrids.jl_amd/csrc/libceg_hip.so build_variants/*.so; do
  for mode in "$@"; do
    echo "== $lib $mode"
    CEG_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-rows 0 --mode $mode 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ms/step %.2f  kernel_ms %.2f  pts/s %.3e  fp64 frac %.3f  check %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['roofline_fp64']['frac'], d['selfcheck']))
    elif 'Error' in l or 'error' in l: print(l.rstrip())
"
  done
done
