#!/bin/bash
# usage: scripts/time_variants.sh "<mode> <probe>" ...   -- bench.py with every library variant in build_variants/ (plus the in-tree one)
mkdir -p gpurun_out
shopt -s nullglob
for lib in crystalenergygrids.jl_amd/csrc/libceg_hip.so build_variants/*.so; do
  for mp in "$@"; do
    read -r mode probe <<< "$mp"; probe=${probe:-Ar}
    echo "== $lib $mode $probe"
    CEG_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-rows 0 --mode $mode --probe $probe 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ms/step %.3f  kernel_ms %.3f  pts/s %.3e  check %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['selfcheck']))
    elif 'Error' in l or 'error' in l: print(l.rstrip())
"
  done
done
